"""Fuzz of the shading path: random scenes with every material and texture kind (solid / checker / image Lambertian with
random reflectance, metals with and without fuzz, glass of random index, moving spheres), random cameras with defocus,
rendered with the same seeded counter RNG by the HIP library and by the oracle, in both precisions, through every
closest-hit front end.  Bars as in tests/test_gpu_parity.py::test_radiance_vs_oracle_same_rng: >= 99.5 % of the
channels within 1e-9 (f64) / 2e-5 (f32) relative-absolute, segment counts within 0.2 % (or two paths' worth), bytes within 1 LSB for 99.5 %.
(No bound on the largest difference: a Lambertian bounce that survives its reflectance test carries colour / reflectance,
material.rs:78, so one sample whose second segment hits on one side and misses on the other - the f32 sides differ in
the last bit of sin/cos - moves a channel by more than 1 / spp; seed 2386 of the first 600-case run was such a sample.)
usage: PYTHONPATH=. python tools/fuzz_radiance.py [first_seed] [n_cases]"""
import os
import sys

import numpy as np

from oracle import binding
from raytracer_amd import abi, api
from raytracer_amd.api import Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, Scene, Sphere

binding.load()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
worst = {abi.F64: 1.0, abi.F32: 1.0}
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    api.DEFAULT_FLAGS = abi.FLAG_MX_SMALL if rng.random() < 0.5 else 0  # matrix pipe also below its size threshold
    img = rng.integers(0, 256, (int(rng.integers(1, 24)), int(rng.integers(1, 48)), 3), dtype=np.uint8)

    def material():
        k = rng.random()
        if k < 0.25:
            return Lambertian.new(tuple(rng.uniform(0.05, 0.95, 3)), float(rng.uniform(0.3, 1.0)))
        if k < 0.40:
            return Lambertian.with_texture(CheckerTexture(float(rng.uniform(0.05, 2.0)), tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3))), float(rng.uniform(0.3, 1.0)))
        if k < 0.55:
            return Lambertian.with_texture(ImageTexture(img), float(rng.uniform(0.3, 1.0)))
        if k < 0.80:
            return Metal.new(tuple(rng.uniform(0.3, 1.0, 3)), None if rng.random() < 0.3 else float(rng.uniform(0.0, 1.3)))
        return Dielectric.new(float(rng.uniform(1.05, 2.4)) if rng.random() < 0.8 else float(rng.uniform(0.4, 0.95)))

    n = int(rng.integers(1, int(os.environ.get("FUZZ_MAX_N", "140"))))  # FUZZ_MAX_N=3000: beyond the LDS tables (640 materials, 1024 spheres)
    ext = float(rng.uniform(2.0, 12.0))
    sc = Scene()
    if rng.random() < 0.7:
        R = float(10.0 ** rng.uniform(1.5, 3.0))
        sc.add(Sphere.new_static(R, (0.0, -R, 0.0), material()))
    for k in range(n):
        c = rng.uniform(-ext, ext, 3) * np.array([1.0, 0.3, 1.0]) + np.array([0.0, 0.3 * ext, 0.0])
        r = float(rng.uniform(0.03, 0.25) * ext) * min(1.0, (140.0 / max(n, 1)) ** 0.5)
        m = material()
        if rng.random() < 0.3:
            sc.add(Sphere.new(r, tuple(c), tuple(c + rng.uniform(-0.1, 0.1, 3) * ext), m))
        else:
            sc.add(Sphere.new_static(r, tuple(c), m))
    sc.build_bvh()
    d = rng.normal(size=3)
    d[1] = abs(d[1]) * 0.5 + 0.1
    d = d / np.linalg.norm(d)
    dist = float(rng.uniform(1.5, 3.5)) * ext
    spp = int(rng.integers(1, 9))
    cam = Camera.build(tuple(d * dist), (0.0, 0.2 * ext, 0.0), None, int(rng.integers(24, 72)), float(rng.uniform(0.8, 1.8)), spp,
                       float(rng.uniform(0.3, 1.0)), dist, float(rng.choice([0.0, 0.01, 0.05])), int(rng.integers(1, 30)))
    line = []
    for prec, tol in ((abi.F64, 1e-9), (abi.F32, 2e-5)):
        o_lin, o_rgb8, o_seg, o_nan = binding.render(sc.flat, cam.pod, prec, seed=seed0 + case, threads=8)
        for accel in (None, "valu") + (("bvh",) if prec == abi.F32 else ()):
            lin, rgb8 = cam.render_arrays(sc, precision=prec, seed=seed0 + case, accel=accel)
            err = np.abs(lin.astype(np.float64) - o_lin.astype(np.float64))
            frac = float((err <= tol * (1.0 + np.abs(o_lin))).mean())
            seg = cam.last_stats["segments"]
            ok = o_nan == 0 and np.isfinite(lin).all() and frac >= 0.995 and abs(seg - o_seg) <= max(2 * cam.pod.max_bounce_depth, 0.002 * o_seg)  # one flipped path is up to `depth` segments
            ok = ok and (np.abs(rgb8.astype(int) - o_rgb8.astype(int)) <= 1).mean() >= 0.995
            worst[prec] = min(worst[prec], frac)
            if not ok:
                bad += 1
                print(f"RADIANCE MISMATCH seed {seed0 + case} prec {prec} accel {accel}: close {frac:.5f} max {err.max():.3g} segments {seg} vs {o_seg} nan {o_nan}", flush=True)
        line.append(f"f{prec} close {frac:.4f}")
    print(f"case {seed0 + case}: n={len(sc.flat.objects) if hasattr(sc.flat, 'objects') and sc.flat.objects is not None else n} {cam.width}x{cam.height}x{spp} depth {cam.pod.max_bounce_depth} " + " ".join(line), flush=True)
print(f"cases {n_cases}  failing renders: {bad}  lowest share of close channels: f64 {worst[abi.F64]:.5f} f32 {worst[abi.F32]:.5f}")
sys.exit(1 if bad else 0)
