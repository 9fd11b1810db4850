"""Fuzz: random scenes and cameras, hit-id / t maps of every closest-hit front end against the oracle (bit-exact), and
small rendered frames of the front ends against each other (bit-identical).
usage: PYTHONPATH=. python tools/fuzz_maps.py [first_seed] [n_cases]"""
import os
import sys

import numpy as np

from oracle import binding
from raytracer_amd import abi, api
from raytracer_amd.api import Camera, Dielectric, Lambertian, Metal, Scene, Sphere

binding.load()
api.DEFAULT_FLAGS = abi.FLAG_MX_SMALL  # matrix pipe also below its size threshold
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mats = [Lambertian.new((0.6, 0.6, 0.6), 1.0), Metal.new((0.9, 0.9, 0.9), None), Dielectric.new(1.5)]
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    scale = float(2.0 ** rng.integers(-10, 11))
    n = int(rng.integers(24, 1400))
    ext = float(rng.uniform(2.0, 60.0))
    sc = Scene()
    if rng.random() < 0.5:
        R = float(10.0 ** rng.uniform(1.5, 3.0))
        sc.add(Sphere.new_static(R * scale, (0.0, -R * scale, 0.0), mats[0]))
    off = rng.uniform(-1, 1, 3) * (0.0 if rng.random() < 0.6 else 2.0 * ext)
    for k in range(n):
        c = (rng.uniform(-ext, ext, 3) + off) * scale
        r = float(rng.uniform(0.02, 0.08) * ext) * scale
        if rng.random() < 0.4:
            sc.add(Sphere.new(r, tuple(c), tuple(c + rng.uniform(-0.05, 0.05, 3) * ext * scale), mats[k % 3]))
        else:
            sc.add(Sphere.new_static(r, tuple(c), mats[k % 3]))
    sc.build_bvh()
    dist = float(rng.uniform(1.2, 4.0)) * ext
    d = rng.normal(size=3)
    d = d / np.linalg.norm(d)
    cam = Camera.build(tuple((off + d * dist) * scale), tuple(off * scale), None, int(rng.integers(40, 200)), float(rng.uniform(0.7, 2.0)), 0,
                       float(rng.uniform(0.3, 1.2)), None, None, 5)
    for prec, ut in ((abi.F32, np.uint32), (abi.F64, np.uint64)):
        o_ids, o_t = binding.primary_hits(sc.flat, cam.pod, prec)
        for accel in (None, "valu") + (("bvh",) if prec == abi.F32 else ()):
            ids, t = cam.primary_hits(sc, precision=prec, accel=accel)
            if not (np.array_equal(ids, o_ids) and np.array_equal(t.view(ut), o_t.view(ut))):
                bad += 1
                print(f"MISMATCH seed {seed0 + case} prec {prec} accel {accel}: {int((ids != o_ids).sum())} ids differ", flush=True)
    # rendered frames (bounces, all materials, moving spheres at random times): bit-identical across the front ends
    rcam = Camera.build(tuple((off + d * dist) * scale), tuple(off * scale), None, 48, 1.0, 2, 0.8, float(dist * scale), 0.01, 20)
    for prec, ut in ((abi.F32, np.uint32), (abi.F64, np.uint64)):
        ref, ref8 = rcam.render_arrays(sc, precision=prec, seed=case, accel="valu")
        for accel in (None,) + (("bvh",) if prec == abi.F32 else ()):
            lin, rgb8 = rcam.render_arrays(sc, precision=prec, seed=case, accel=accel)
            if not (np.array_equal(lin.view(ut), ref.view(ut)) and np.array_equal(rgb8, ref8)):
                bad += 1
                print(f"RENDER MISMATCH seed {seed0 + case} prec {prec} accel {accel}", flush=True)
    print(f"case {seed0 + case}: n={n} scale=2^{int(np.log2(scale))} kinds {cam.closest_hit_kind(sc, precision=abi.F32)}/{cam.closest_hit_kind(sc, precision=abi.F64)} hits {int((o_ids != abi.MISS_ID).sum())}", flush=True)
print("mismatching maps:", bad)
sys.exit(1 if bad else 0)
