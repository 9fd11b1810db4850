"""Kernel time of every rank's band of the bench frame on ONE GPU (what each of N GPUs would run): strong-scaling estimate
without an N-GPU node.  usage: PYTHONPATH=. python tools/band_times.py [parts ...]"""
import ctypes as C
import sys

import torch

from raytracer_amd import abi, dist, presets

sc = presets.bouncing_balls_scene(1)
cam = presets.bouncing_balls_camera(1920, 64, 50)
lib = abi.load()
ds = sc.device_scene(0)
import os
H, W, S = cam.height, cam.width, int(os.environ.get("STRIP", "8"))
for parts in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    rows0 = dist.part_rows(H, S, 0, parts)
    band = torch.zeros((rows0, W, 3), dtype=torch.float32, device="cuda")
    times = []
    for r in range(parts):
        o = cam.opts(precision=abi.F32, seed=1, strip_rows=S, part_index=r, part_count=parts, flags=abi.FLAG_COMPACT_ROWS)
        st = abi.MirtStats()
        best = 1e9
        for _ in range(3):
            abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), band.data_ptr(), None, None, C.byref(st)))
            best = min(best, st.kernel_ms)
        times.append(best)
    print(f"strip {S} parts {parts}: kernel ms per rank " + " ".join(f"{t:.2f}" for t in times) + f"  (max {max(times):.3f})")
