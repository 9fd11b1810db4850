#!/usr/bin/env python3
"""Frames in flight: an 1/PARTS band of the bench frame (one rank's share), rendered FRAMES times
 (a) back to back on one stream, one scene handle;
 (b) alternating between two (three, four) scene handles on as many streams (each handle owns its work queue and partial sums),
     so the next frame's blocks take the CUs the last long paths of the previous frame leave idle.
usage: tools/inflight_probe.py [PARTS [FRAMES]]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, dist
from raytracer_amd.api import DeviceScene
import bench

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = abi.load()
scene, cam, name = bench.build_workload("balls1080p", 1920, 64, 50)
cam.chunk_samples = int(os.environ.get("PROBE_CHUNK", "0"))  # mirt_opts.chunk_samples (0 = the library's choice)
extra_flags = int(os.environ.get("PROBE_FLAGS", "0"))
cam.group_chunks = int(os.environ.get("PROBE_G", "0"))  # mirt_opts.group_chunks (0 = the library's choice)
S = dist.DEFAULT_STRIP_ROWS
handles = [DeviceScene(scene.flat, 0) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
for rank in sorted({0, parts // 2, parts - 1}):
    rows = dist.part_rows(cam.height, S, rank, parts)
    bands = [torch.zeros((rows, cam.width, 3), dtype=torch.float32, device="cuda") for _ in range(4)]
    bands8 = [torch.zeros((rows, cam.width, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
    opts = cam.opts(precision=32, seed=1, strip_rows=S, part_index=rank, part_count=parts, flags=abi.FLAG_COMPACT_ROWS | extra_flags)

    def run(nh):
        for w in range(2):
            for i in range(nh):
                abi.check(lib.mirt_render_device(handles[i].handle, C.byref(cam.pod), C.byref(opts), bands[i].data_ptr(), bands8[i].data_ptr(), streams[i].cuda_stream, None))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(frames):
            i = f % nh
            abi.check(lib.mirt_render_device(handles[i].handle, C.byref(cam.pod), C.byref(opts), bands[i].data_ptr(), bands8[i].data_ptr(), streams[i].cuda_stream, None))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3

    t = [run(k) for k in (1, 2, 3, 4)]
    same = all(bool(torch.equal(bands[0], bands[k])) for k in (1, 2, 3))
    print(f"part {rank}/{parts}: {rows} rows  ms/frame with 1 / 2 / 3 / 4 frames in flight: " + " / ".join(f"{x:.3f}" for x in t) + f"   frames identical: {same}", flush=True)
