"""One rank's band of an 8-way split of the bench frame, FRAMES times, one frame at a time: run under
rocprofv3 --kernel-trace --stats to see what each kernel of a band launch costs.  usage: band_kernels.py FLAGS [FRAMES]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, dist
import bench

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = abi.load()
scene, cam, name = bench.build_workload("balls1080p", 1920, 64, 50)
ds = scene.device_scene(0)
rows = dist.part_rows(cam.height, 8, 4, 8)
band = torch.zeros((rows, cam.width, 3), dtype=torch.float32, device="cuda")
o = cam.opts(precision=32, seed=1, strip_rows=8, part_index=4, part_count=8, flags=abi.FLAG_COMPACT_ROWS | flags)
for _ in range(frames):
    abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), band.data_ptr(), None, None, None))
torch.cuda.synchronize()
