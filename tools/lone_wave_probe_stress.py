#!/usr/bin/env python3
"""tools/lone_wave_probe.py for a scene beyond the LDS: the 10 000-sphere stress scene, 1, 2, 4, 8 pixels (= lanes of ONE wave) on a
glass sphere near the middle of the field, 256 samples per item.  usage: tools/lone_wave_probe_stress.py [n_spheres]"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, presets
from raytracer_amd.api import Camera

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
scene = presets.stress_scene(n, 1)
rng = np.random.Generator(np.random.Philox(1))  # as presets.stress_scene draws
c = rng.random((n, 3)); r = 0.1 + 0.4 * rng.random(n); kind = rng.random(n)
cen = np.stack([-50.0 + 100.0 * c[:, 0], 0.2 + 19.8 * c[:, 1], -50.0 + 100.0 * c[:, 2]], axis=1)
glass = np.where((kind >= 0.9) & (r > 0.4))[0]
g = glass[np.argmin(np.linalg.norm(cen[glass] - np.array([0.0, 8.0, 40.0]), axis=1))]  # a large glass sphere on the camera's side of the field
at = tuple(float(x) for x in cen[g])
dist = float(np.linalg.norm(cen[g] - np.array([0.0, 30.0, 90.0])))
print(f"glass sphere {g} at {at}, r {r[g]:.3f}, {dist:.1f} from the camera", flush=True)
for W, ratio in ((1, 1.0), (2, 1.0), (4, 2.0), (4, 1.0)):
    spp = 256
    cam = Camera.build((0.0, 30.0, 90.0), at, (0.0, 1.0, 0.0), W, ratio, spp, 0.8 * float(r[g]) / dist, dist, 0.0, 50)
    cam.precision = abi.F32
    cam.chunk_samples = spp
    best = None
    for rep in range(3):
        cam.render_arrays(scene, seed=1, want_rgb8=False)
        st = cam.last_stats
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = dict(st)
    it = best.get("wave_iterations", 0)
    print(f"{cam.width}x{cam.height} px = lanes, {spp} samples each: kernel {best['kernel_ms']:.4f} ms, segments {best['segments']}, wave iterations {it}, "
          f"{best['kernel_ms'] * 1e3 / max(it, 1):.2f} us per wave iteration, tiles per iteration {best.get('mx_tiles', 0) / max(it, 1):.1f}", flush=True)
