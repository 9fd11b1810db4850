"""The throughput mode (MIRT_F32, the bench's dtype) against THE REFERENCE'S OWN ARITHMETIC under the same random numbers.
Needs a GPU.

The reference computes in f64 (core/point3.rs:13-16, scene/sphere.rs:83-103, camera/camera.rs:268-286).  The f32 kernels
are bit-equal to an f32 spec this build wrote (perpendicular-form discriminant, reciprocals, its own
sin/cos/acos/atan2, running product) - a regression pin, not parity.  What ties the f32 mode to the reference is this file:
every frame is compared, pixel by pixel, with oracle_render_ex(precision 64, draws 32) - Real = double, Rust's operations
in Rust's order (recursion-order product, textbook discriminant, true divisions, libm, the BVH walk), each uniform the very
binary32 value the kernel draws for that (pixel, sample, bounce, dimension).  No Monte-Carlo noise is left between the two
frames; the bars (tools/f32_vs_reference.py::BARS, stated in DESIGN.md 2) bound binary32 rounding, its amplification by the
bounces, and any bias an f32-spec change would introduce (round 1's textbook f32 discriminant misses them by far).

Cases: BASELINE configs[0] in full (main.rs:30-56,126-127: mars, 400 x 225, 100 spp, depth 50); the bench scene and camera
(lens, shutter, image texture, checker, metal, glass) at 320 x 180 x 16 spp; the textured C4 scene; stress_scene(2000).
"""
import numpy as np
import pytest

from raytracer_amd import abi
from tools import f32_vs_reference as fr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c1", "bench", "textured", "stress"])
@pytest.mark.parametrize("accel", [None, "valu"])
def test_f32_frame_against_reference_arithmetic_same_uniforms(oracle, name, accel):
    mk_scene, mk_cam, seed = fr.cases()[name]
    sc, cam = mk_scene(), mk_cam()
    lin, rgb8 = cam.render_arrays(sc, precision=abi.F32, seed=seed, accel=accel)
    seg = cam.last_stats["segments"]
    # the kernel's frame is the f32 spec's frame (so the spec's path signatures are the kernel's paths) ...
    q = cam.pass_granularity(precision=abi.F32)
    s_lin, s_rgb8, s_seg, s_sig = oracle.render_ex(sc.flat, cam.pod, abi.F32, 0, seed, threads=8, spec_chunk=q)
    assert seg == s_seg and np.array_equal(lin.view(np.uint32), s_lin.view(np.uint32)) and np.array_equal(rgb8, s_rgb8)
    # ... and within the stated bars of the reference's f64 arithmetic on the same uniforms
    ref, ref8, ref_seg, ref_sig = oracle.render_ex(sc.flat, cam.pod, abi.F64, 32, seed, use_bvh=True, threads=8)
    st = fr.compare(lin, rgb8, s_sig, ref, ref8, ref_sig)
    st.update(segments_f32=seg, segments_ref=ref_seg)
    print(name, accel, {k: (round(v, 9) if isinstance(v, float) else v) for k, v in st.items()})
    assert not fr.failures(st, fr.BARS[name]), fr.failures(st, fr.BARS[name])
