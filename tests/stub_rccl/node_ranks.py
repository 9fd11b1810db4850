"""Run by tests/test_gpu_node.py in a child process whose LD_LIBRARY_PATH puts tests/stub_rccl first, so that
mirt_node's dlopen("librccl.so.1") finds the stub (see stub_rccl.cpp).  torch must stay out of this process: it carries
the real librccl.so.1 and the dynamic loader would hand that one back.  Renders one frame on a plain mirt_scene and
then as N ranks sharing GPU 0, both exchanges, both precisions, odd sizes; every image must equal the one-GPU image."""
import sys

sys.modules["torch"] = None  # "import torch" now raises ImportError; abi.load() then uses the system HIP runtime

import numpy as np  # noqa: E402

from raytracer_amd import abi, presets  # noqa: E402
from raytracer_amd.api import NodeScene  # noqa: E402


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def main():
    sc = presets.bouncing_balls_scene(1)
    cases = 0
    for n in (2, 3, 5, 8):
        node = NodeScene(sc.flat, [0] * n, shared_devices=True)
        for (w, spp, depth, strip, prec, accel) in ((200, 4, 50, 8, abi.F32, None), (97, 2, 12, 3, abi.F64, None), (131, 3, 12, 1, abi.F32, "bvh"),
                                                  (64, 0, 8, 16, abi.F32, "valu")):
            cam = presets.bouncing_balls_camera(w, spp, depth)
            ref, ref8 = cam.render_arrays(sc, precision=prec, seed=9, accel=accel)
            seg = cam.last_stats["segments"]
            for gather in (abi.GATHER_TO_ROOT, abi.GATHER_ALLGATHER):
                lin, rgb8 = cam.render_arrays_node(node, precision=prec, seed=9, accel=accel, strip_rows=strip, gather=gather)
                st = cam.last_stats
                assert st["n_devices"] == n and st["segments"] == seg == sum(st["band_segments"][:n]), (n, st)
                assert np.array_equal(bits(lin), bits(ref)), ("linear", n, w, gather)
                assert np.array_equal(rgb8, ref8), ("bytes", n, w, gather)
                # rows of the frame really came from different ranks: every rank traced something when it owns a strip
                rows = [abi.load().mirt_part_rows(cam.height, strip, r, n) for r in range(n)]
                assert sum(rows) == cam.height
                assert all((st["band_segments"][r] > 0) == (rows[r] > 0) for r in range(n)), (rows, st["band_segments"])
                cases += 1
        node.close()
    print(f"STUB_NODE_OK {cases}")


if __name__ == "__main__":
    main()
