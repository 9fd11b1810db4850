#!/usr/bin/env python3
"""Latency of a loop iteration for a wave that runs alone: a W x W image of the bench scene aimed at the glass ball (every path
bounces inside it, many to the depth limit), 2 spp, so the launch is W*W work items - one wave for W = 8 - and
kernel time / wave iterations is what one iteration costs with nothing else on the chip.  usage: tools/lone_wave_probe.py [accel]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, presets
from raytracer_amd.api import Camera

accel = sys.argv[1] if len(sys.argv) > 1 else None
scene = presets.bouncing_balls_scene(1)
for W, ratio, spp, vfov, at in ((1, 1.0, 1024, 0.002, (-4.0, 1.0, 0.0)), (2, 1.0, 1024, 0.004, (-4.0, 1.0, 0.0)), (4, 2.0, 1024, 0.008, (-4.0, 1.0, 0.0)),
                                (4, 1.0, 1024, 0.008, (-4.0, 1.0, 0.0)), (8, 1.0, 1024, 0.016, (-4.0, 1.0, 0.0)), (2, 1.0, 1024, 0.004, (0.0, 1.0, 0.0))):
    cam = Camera.build((13.0, 2.0, 3.0), at, (0.0, 1.0, 0.0), W, ratio, spp, vfov, 10.0, 0.0, 50)
    cam.precision = abi.F32
    cam.chunk_samples = spp
    best = None
    for rep in range(3):
        cam.render_arrays(scene, seed=1, want_rgb8=False, accel=accel)
        st = cam.last_stats
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = dict(st)
    it = best.get("wave_iterations", 0)
    print(f"at {at} {cam.width}x{cam.height} px = lanes, {spp} samples each: kernel {best['kernel_ms']:.4f} ms, segments {best['segments']}, wave iterations {it}, "
          f"{best['kernel_ms'] * 1e3 / max(it, 1):.2f} us per wave iteration", flush=True)
