"""Kernel time, wave iterations and lane utilisation of one rank's band (8-way split) per queue-order flag.  usage: band_stats.py [part]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, dist
import bench

part = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = abi.load()
scene, cam, name = bench.build_workload("balls1080p", 1920, 64, 50)
ds = scene.device_scene(0)
rows = dist.part_rows(cam.height, 8, part, 8)
band = torch.zeros((rows, cam.width, 3), dtype=torch.float32, device="cuda")
for flags, label in ((0, "row order"),):  # the ordered-queue experiment of round 3 added (0, "by expected path length") with the row order under flag 64
    for g in (0, 1, 64):
        cam.group_chunks = g
        o = cam.opts(precision=32, seed=1, strip_rows=8, part_index=part, part_count=8, flags=abi.FLAG_COMPACT_ROWS | flags)
        best = None
        for _ in range(4):
            st = abi.MirtStats()
            abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), band.data_ptr(), None, None, C.byref(st)))
            if best is None or st.kernel_ms < best.kernel_ms:
                best = st
        print(f"{label:26s} G={g or 'auto':4}: K2 {best.kernel_ms - best.resolve_ms:.3f} ms, wave iterations {best.wave_iterations}, lane utilisation {best.segments / (64.0 * best.wave_iterations):.4f}, "
              f"narrow tests per segment {best.narrow_tests / best.segments:.3f}", flush=True)
