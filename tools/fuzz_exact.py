#!/usr/bin/env python3
"""Fuzz of the bit-exact radiance bar (tests/test_gpu_exact.py): random scenes of every material and the two procedural textures
(image textures only with FUZZ_IMAGES=1, which renders in f32 alone: in f64 acos/atan2 are libm's), moving spheres, random cameras, depths and sample counts.
  f64: no lens (sin/cos are libm's); GPU (matrix-pipe filter and all-f64 loop) must equal the oracle's reference-order render bit for bit.
  f32: lens on or off (the f32 spec's own sin/cos); GPU (matrix pipe, vector ALU, tree) must equal the oracle's f32-spec render bit for bit,
       with a random lower bound on the samples per work item.
usage: fuzz_exact.py [first_seed] [cases]      (FUZZ_MAX_N: spheres per scene, default 140; FUZZ_IMAGES=1: f32 only, with random image textures)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402
from raytracer_amd import abi, api  # noqa: E402
from raytracer_amd.api import Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, Scene, Sphere  # noqa: E402

IMAGES = os.environ.get("FUZZ_IMAGES") == "1"

binding.load()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    api.DEFAULT_FLAGS = abi.FLAG_MX_SMALL if rng.random() < 0.5 else 0

    def material():
        k = rng.random()
        if IMAGES and k < 0.25:  # the f32 spec's own acos / atan2 pick the texel: odd sizes, 1 x 1 included
            img = rng.integers(0, 256, (int(rng.integers(1, 40)), int(rng.integers(1, 70)), 3), dtype=np.uint8)
            return Lambertian.with_texture(ImageTexture(img), float(rng.uniform(0.4, 1.0)))
        if k < 0.35:
            return Lambertian.new(tuple(rng.uniform(0.05, 0.95, 3)), float(rng.uniform(0.3, 1.0)))
        if k < 0.55:
            return Lambertian.with_texture(CheckerTexture(float(rng.uniform(0.1, 2.0)), tuple(rng.uniform(0, 1, 3)), tuple(rng.uniform(0, 1, 3))), float(rng.uniform(0.5, 1.0)))
        if k < 0.80:
            return Metal.new(tuple(rng.uniform(0.3, 1.0, 3)), None if rng.random() < 0.3 else float(rng.uniform(0.0, 1.3)))
        return Dielectric.new(float(rng.uniform(1.05, 2.4)) if rng.random() < 0.8 else float(rng.uniform(0.4, 0.95)))

    n = int(rng.integers(1, int(os.environ.get("FUZZ_MAX_N", "140"))))
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
    spp = int(rng.integers(0, 11))
    W, ratio, vfov, depth = int(rng.integers(24, 72)), float(rng.uniform(0.8, 1.8)), float(rng.uniform(0.3, 1.0)), int(rng.integers(1, 30))
    lens = float(rng.choice([0.0, 0.01, 0.05]))
    seed = seed0 + case
    msgs = []
    for prec in ((abi.F32,) if IMAGES else (abi.F64, abi.F32)):
        cam = Camera.build(tuple(d * dist), (0.0, 0.2 * ext, 0.0), None, W, ratio, spp, vfov, dist, 0.0 if prec == abi.F64 else lens, depth)
        cam.chunk_samples = int(rng.choice([0, 0, 2, 4])) if prec == abi.F32 else 0
        g = max(1, cam.pass_granularity(precision=prec))
        if prec == abi.F64:
            o_lin, o_rgb8, o_seg, _ = binding.render(sc.flat, cam.pod, prec, seed=seed, use_bvh=True, threads=8)
        else:
            o_lin, o_rgb8, o_seg, _ = binding.render(sc.flat, cam.pod, prec, seed=seed, threads=8, spec_chunk=g)
        for accel in (None, "valu") + (("bvh",) if prec == abi.F32 else ()):
            lin, rgb8 = cam.render_arrays(sc, precision=prec, seed=seed, accel=accel)
            same = bool(np.array_equal(bits(lin), bits(o_lin))) and bool(np.array_equal(rgb8, o_rgb8)) and cam.last_stats["segments"] == o_seg
            if not same:
                bad += 1
                diff = int((bits(lin) != bits(o_lin)).sum())
                msgs.append(f"MISMATCH f{prec} accel {accel}: {diff} channels differ, segments {cam.last_stats['segments']} vs {o_seg}")
    print(f"case {seed}: n={n} {W}x{cam.height}x{spp} depth {depth} lens(f32) {lens} Q {g} " + ("; ".join(msgs) if msgs else ("bit-equal (f32 x3, image textures)" if IMAGES else "bit-equal (f64 x2, f32 x3)")), flush=True)
print(f"cases {n_cases}  renders not bit-equal to the oracle: {bad}")
sys.exit(1 if bad else 0)
