"""PCIe-inclusive rate of the host-buffer entry point (mirt_render: device frame allocated, rendered, copied back) on the bench frame.
usage: PYTHONPATH=. python tools/host_buffer_rate.py"""
import ctypes as C
import time

import numpy as np

import bench
from raytracer_amd import abi

lib = abi.load()
scene, cam, name = bench.build_workload("balls1080p", 1920, 64, 50)
ds = scene.device_scene(0)
H, W = cam.height, cam.width
o = cam.opts(precision=abi.F32, seed=1)
lin = np.zeros((H, W, 3), np.float32)
rgb8 = np.zeros((H, W, 3), np.uint8)
for label, a, b in (("bytes only (the PPM payload, 6.2 MB)", None, rgb8.ctypes.data), ("linear f32 + bytes (31 MB)", lin.ctypes.data, rgb8.ctypes.data)):
    st = abi.MirtStats()
    for _ in range(2):
        abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), a, b, C.byref(st)))
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), a, b, C.byref(st)))
    dt = (time.perf_counter() - t0) / n
    print(f"mirt_render, {label}: {dt * 1e3:.2f} ms per frame, {st.segments / dt / 1e6:.0f} Mrays/s (kernel {st.kernel_ms:.2f} ms)", flush=True)
