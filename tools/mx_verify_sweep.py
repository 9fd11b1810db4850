"""Run with MIRT_LIB pointing at a -DMIRT_VERIFY_BROAD build: renders random scenes (scales, dynamic ranges, camera distances up to
the limit of the matrix pipe's operand range) and lets the library print its count of exact candidates the broad phase missed
(stderr lines "[mirt verify broad] ... : N"; every N must be 0).  usage: PYTHONPATH=. python tools/mx_verify_sweep.py"""
import os
import sys

import numpy as np

from raytracer_amd import abi, api
from raytracer_amd.api import Camera, Dielectric, Lambertian, Metal, Scene, Sphere

api.DEFAULT_FLAGS = abi.FLAG_MX_SMALL  # matrix pipe also below its size threshold
PREC = abi.F64 if os.environ.get("MIRT_SWEEP_PRECISION") == "64" else abi.F32  # the f64 kernels use the same filter on balls around the f64 spheres
ACCEL = os.environ.get("MIRT_SWEEP_ACCEL", "brute")  # "valu": the vector-ALU broad phase (MIRT_SWEEP_NMIN >= 4096: its sparse group test)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
mats = [Lambertian.new((0.6, 0.6, 0.6), 1.0), Metal.new((0.9, 0.9, 0.9), None), Metal.new((0.8, 0.8, 0.8), 0.2), Dielectric.new(1.5)]
n_matrix = 0
for case in range(40):
    scale = float(2.0 ** rng.integers(-12, 13))
    n = int(rng.integers(int(os.environ.get("MIRT_SWEEP_NMIN", "24")), int(os.environ.get("MIRT_SWEEP_NMAX", "700"))))  # beyond 1024: the ring kernel and its top tiles
    ext = float(rng.uniform(2.0, 60.0))
    big = rng.random() < 0.6
    sc = Scene()
    if big:
        R = float(10.0 ** rng.uniform(1.5, 3.2))
        sc.add(Sphere.new_static(R * scale, (0.0, -R * scale, 0.0), mats[0]))
    off = rng.uniform(-1, 1, 3) * (0.0 if rng.random() < 0.5 else 3.0 * ext)
    for k in range(n):
        c = (rng.uniform(-ext, ext, 3) + off) * scale
        r = float(rng.uniform(0.02, 0.08) * ext) * scale
        m = mats[int(rng.integers(0, 4))]
        if rng.random() < 0.4:
            c1 = c + rng.uniform(-0.05, 0.05, 3) * ext * scale
            sc.add(Sphere.new(r, tuple(c), tuple(c1), m))
        else:
            sc.add(Sphere.new_static(r, tuple(c), m))
    sc.build_bvh()
    dist = float(rng.uniform(1.2, 5.0)) * ext
    d = rng.normal(size=3)
    d[1] = abs(d[1]) + 0.2
    d = d / np.linalg.norm(d)
    lookfrom = (off + d * dist) * scale
    cam = Camera.build(tuple(lookfrom), tuple(off * scale), None, 96, 1.0, 4, float(rng.uniform(0.3, 1.2)), float(dist * scale), float(rng.uniform(0.0, 0.05)), 50)
    kind = cam.closest_hit_kind(sc, precision=PREC)
    n_matrix += kind in ("matrix", "f64-matrix")
    print(f"case {case}: n={n} scale=2^{int(np.log2(scale))} ext={ext:.1f} big={big} kind={kind}", flush=True)
    cam.render_arrays(sc, precision=PREC, seed=case, accel=ACCEL)
    cam.primary_hits(sc, precision=PREC, accel=ACCEL)
print("cases on the matrix pipe:", n_matrix)
