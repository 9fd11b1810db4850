"""The tile cull of the matrix-pipe kernels on the bench frame: kernel time and tiles evaluated per wave-iteration with the cull
(default) and without (MIRT_FLAG_ALL_PAIRS), frames compared bit for bit.  usage: cull_probe.py [workload width spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "balls1080p"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lib = abi.load()
scene, cam, name = bench.build_workload(wl, width, spp, 50)
ds = scene.device_scene(0)
frames = {}
for prec in (32, 64):
    real = torch.float32 if prec == 32 else torch.float64
    for flags, label in ((abi.FLAG_ALL_PAIRS, "all pairs"), (0, "tile cull")):
        out = torch.zeros((cam.height, cam.width, 3), dtype=real, device="cuda")
        o = cam.opts(precision=prec, seed=1, flags=flags)
        best = None
        for _ in range(4 if prec == 32 else 2):
            st = abi.MirtStats()
            abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), out.data_ptr(), None, None, C.byref(st)))
            if best is None or st.kernel_ms < best.kernel_ms:
                best = st
        frames[(prec, flags)] = out
        print(f"f{prec} {label:10s}: K2 {best.kernel_ms - best.resolve_ms:8.3f} ms, {best.segments / (best.kernel_ms * 1e-3) / 1e6:9.1f} Mrays/s, tiles per wave-iteration "
              f"{best.mx_tiles / max(best.wave_iterations, 1):6.2f} of {(scene.flat.n_spheres + 31) // 32}(+1), segments {best.segments}", flush=True)
    print(f"f{prec} frames identical: {bool(torch.equal(frames[(prec, 0)], frames[(prec, abi.FLAG_ALL_PAIRS)]))}", flush=True)
