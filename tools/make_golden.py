"""Generate tests/golden/*.npz from the CPU oracle (oracle/oracle.cpp).

These are REGRESSION pins of the build's own oracle, not reference outputs: the reference (Rust) cannot be
built or run here and holds no golden data of its own -> "parity unpinned".
Run in the dev container:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from raytracer_amd import abi, presets  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)

CASES = {
    "mars": (presets.mars_texture_scene, lambda w, spp: presets.default_camera(w, spp, 50)),
    "balls": (lambda: presets.bouncing_balls_scene(1), lambda w, spp: presets.bouncing_balls_camera(w, spp, 50)),
}
data = {}
for name, (mk_scene, mk_cam) in CASES.items():
    sc = mk_scene()
    for w in (64, 400):
        cam = mk_cam(w, 0)
        for prec, tag in ((abi.F64, "f64"), (abi.F32, "f32")):
            ids, t = ob.primary_hits(sc.flat, cam.pod, prec, use_bvh=False)
            if prec == abi.F64:
                ids_b, t_b = ob.primary_hits(sc.flat, cam.pod, prec, use_bvh=True)
                assert np.array_equal(ids, ids_b) and np.array_equal(t.view(np.uint64), t_b.view(np.uint64)), "BVH != brute force"
            data[f"{name}_{w}_{tag}_id"] = ids
            data[f"{name}_{w}_{tag}_t"] = t
    cam = mk_cam(64, 16)
    for prec, tag in ((abi.F64, "f64"), (abi.F32, "f32")):
        lin, rgb8, seg, nan = ob.render(sc.flat, cam.pod, prec, seed=7, use_bvh=False, threads=8)
        assert nan == 0
        data[f"{name}_64x16spp_{tag}_lin"] = lin
        data[f"{name}_64x16spp_{tag}_rgb8"] = rgb8
        data[f"{name}_64x16spp_{tag}_segments"] = np.uint64(seg)
np.savez_compressed(os.path.join(out, "oracle_maps.npz"), **data)
print("wrote", len(data), "arrays,", os.path.getsize(os.path.join(out, "oracle_maps.npz")), "bytes")
