"""How far is the throughput mode (binary32, the f32 spec of DESIGN.md 3) from the reference's own arithmetic when both see
the SAME random numbers?  CPU only: the oracle's f32 instantiation of the f32 spec (which the HIP kernels reproduce bit for
bit - tests/test_gpu_exact.py) against oracle_render_ex(precision 64, draws 32) = Rust's f64 operations in Rust's order fed
with the f32 draw layout.  Prints, per case, the figures tests/test_gpu_f32_vs_reference.py turns into bars.

    python tools/f32_vs_reference.py [case ...]     cases: c1 bench textured stress all(default)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from raytracer_amd import abi, presets  # noqa: E402


def cases():
    return {
        # BASELINE configs[0] in full: main.rs:30-56,126-127
        "c1": (presets.mars_texture_scene, lambda: presets.default_camera(400, 100, 50), 1),
        # the bench scene and camera (lens 0.4 deg, shutter, image texture, checker, metal, glass), reduced
        "bench": (lambda: presets.bouncing_balls_scene(1), lambda: presets.bouncing_balls_camera(320, 16, 50), 2),
        "textured": (presets.textured_scene, lambda: presets.textured_camera(256, 16, 50), 3),
        "stress": (lambda: presets.stress_scene(2000, 1), lambda: presets.stress_camera(256, 8, 50), 5),
    }


# The bars an f32 frame is held to against the reference-arithmetic frame under the same uniforms (stated in DESIGN.md 2).
# Measured values (this file, CPU; the HIP frames are these frames bit for bit) are about 2-3x inside each bar.  What they
# fail: round 1's textbook f32 discriminant on the stress case (PSNR 32 dB, 28 % more segments, 5.5 % of the pixels branched,
# bias -4e-3) - the kind of f32-spec change that biases the image.
#   psnr      : PSNR (dB, peak 1.0) of the linear frame, all pixels
#   max_abs   : largest per-channel difference of any pixel (one sample of spp taking another path moves a pixel by ~1/spp)
#   branched  : share of pixels in which some sample took another discrete path (hit id / face / scatter outcome /
#               reflect-or-refract / checker parity / texel) - binary32 rounding amplified by the bounces off small convex
#               spheres; not a bias (see `bias`)
#   same_rel  : |difference| / (1 + |reference|) over the pixels whose samples all took the same path: pure rounding
#   within_1  : share of quantised channels within +-1 of the reference-arithmetic bytes
#   bias      : |mean signed difference| / mean reference radiance
#   segments  : |segments_f32 - segments_ref| / segments_ref
BARS = {
    "c1": dict(psnr=80.0, max_abs=0.03, branched=0.004, same_rel=5e-6, within_1=0.9999, bias=1e-5, segments=1e-5),
    "bench": dict(psnr=48.0, max_abs=0.25, branched=0.06, same_rel=1e-2, within_1=0.985, bias=5e-5, segments=5e-4),
    "textured": dict(psnr=46.0, max_abs=0.5, branched=0.06, same_rel=1e-4, within_1=0.985, bias=5e-5, segments=5e-4),
    "stress": dict(psnr=56.0, max_abs=0.3, branched=0.003, same_rel=1e-2, within_1=0.998, bias=2e-5, segments=1e-3),
}


def failures(st, bars):
    """The bars of `bars` that the statistics `st` (compare() + segment counts) miss, as strings; empty = pass."""
    got = {
        "psnr": (st["psnr_db"], ">="), "max_abs": (st["max_abs"], "<="), "branched": (st["branched_pixels"], "<="),
        "same_rel": (st["max_rel_same_path"], "<="), "within_1": (st["rgb8_within_1"], ">="),
        "bias": (abs(st["mean_signed"]) / st["mean_ref"], "<="),
        "segments": (abs(st["segments_f32"] - st["segments_ref"]) / st["segments_ref"], "<="),
    }
    bad = []
    for k, bar in bars.items():
        v, op = got[k]
        if not (v >= bar if op == ">=" else v <= bar):
            bad.append(f"{k}: {v:.3g} not {op} {bar:g}")
    return bad


def compare(lin32, rgb32, sig32, ref, rgb_ref, sig_ref):
    """Statistics of an f32 frame against the reference-arithmetic frame (f64, same uniforms)."""
    err = np.abs(lin32.astype(np.float64) - ref)
    px_err = err.max(axis=2)
    mse = float((err ** 2).mean())
    same_path = sig32 == sig_ref
    d8 = np.abs(rgb32.astype(int) - rgb_ref.astype(int))
    rel = err / (1.0 + np.abs(ref))
    out = {
        "max_abs": float(err.max()),
        "psnr_db": float(10.0 * np.log10(1.0 / mse)) if mse > 0 else float("inf"),
        "branched_pixels": float(1.0 - same_path.mean()),
        "max_abs_same_path": float(px_err[same_path].max()) if same_path.any() else 0.0,
        "max_rel_same_path": float(rel.max(axis=2)[same_path].max()) if same_path.any() else 0.0,
        "rgb8_equal": float((d8 == 0).mean()),
        "rgb8_within_1": float((d8 <= 1).mean()),
        "rgb8_max": int(d8.max()),
        "mean_signed": float((lin32.astype(np.float64) - ref).mean()),
        "mean_ref": float(ref.mean()),
    }
    return out


def run(name, threads=8):
    mk_scene, mk_cam, seed = cases()[name]
    sc, cam = mk_scene(), mk_cam()
    q = cam.pass_granularity(precision=abi.F32)
    t0 = time.time()
    lin32, rgb32, seg32, sig32 = ob.render_ex(sc.flat, cam.pod, abi.F32, 0, seed, threads=threads, spec_chunk=q)
    ref, rgb_ref, seg_ref, sig_ref = ob.render_ex(sc.flat, cam.pod, abi.F64, 32, seed, use_bvh=True, threads=threads)
    st = compare(lin32, rgb32, sig32, ref, rgb_ref, sig_ref)
    st.update(case=name, size=f"{cam.width}x{cam.height}x{cam.pod.samples_per_pixel}", segments_f32=seg32, segments_ref=seg_ref,
              seconds=round(time.time() - t0, 1))
    return st


if __name__ == "__main__":
    want = sys.argv[1:] or ["all"]
    names = list(cases()) if "all" in want else want
    for n in names:
        st = run(n)
        print(st, "FAILS " + "; ".join(failures(st, BARS[n])) if failures(st, BARS[n]) else "within the bars", flush=True)
