"""Bit-exact RADIANCE parity (not only hit ids): HIP path through the C ABI against the CPU oracle.  Needs a GPU.

Where no transcendental function is on a path (no image texture: acos/atan2 of sphere.rs:72-76; no lens: sin/cos of
point3.rs:106-110) every operation of the hot path is an IEEE + - * / sqrt (or an explicit fma mirrored in the oracle),
so equal operation ORDER must give equal bits:
  * MIRT_F64 (parity mode) against the oracle's restatement of the reference's own order - the recursion's
    a0 (a1 (... sky)) of camera.rs:277 and the sequential sample sum of camera.rs:242-246;
  * MIRT_F32 (throughput mode, the bench kernels) against the oracle's f32 instantiation of the f32 spec - running
    product per path, per-work-item sums added in chunk order (oracle.cpp::color_running_product / render_rows).
In f64, scenes with image textures or a lens are held to "almost every channel bit-equal": OCML and glibc differ in the last
ulp of acos/atan2/sin/cos, which moves a texel choice or a ray origin now and then.  The f32 spec spells these functions
out (mirt_device.h::lens_sincos, m_acos, m_atan2; restated in the oracle), so every f32 frame is held to bit-equality.
"""
import ctypes as C
import math

import numpy as np
import pytest

from raytracer_amd import abi, presets
from raytracer_amd.api import Camera, ImageTexture, Lambertian, Scene, Sphere

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def balls_without_image(seed=1):
    """bouncing_balls_scene with the one image-textured sphere given a plain colour: moving spheres, checker ground, fuzzy
    and plain metal, glass - every material and texture kind except ImageTexture."""
    src = presets.bouncing_balls_scene(seed)
    plain = Lambertian.new((0.7, 0.4, 0.3), 0.9)
    sc = Scene()
    for o in src.objects:
        if isinstance(getattr(o.mat, "texture", None), ImageTexture):
            o.mat = plain
        sc.add(o)
    sc.build_bvh()
    return sc


def lensless_camera(w, spp, depth=50):
    return Camera.build((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), w, 16.0 / 9.0, spp, math.radians(25.0), 10.0, 0.0, depth)


@pytest.fixture(scope="module")
def plain_balls():
    return balls_without_image(1)


@pytest.mark.parametrize("accel", [None, "valu"])
def test_f64_radiance_bit_exact_in_the_reference_order(oracle, plain_balls, accel):
    """f64 linear radiance == the oracle's, every bit of every channel: attenuations multiplied innermost first, samples
    added one after the other (one sample per work item), both closest-hit front ends."""
    sc, cam = plain_balls, lensless_camera(160, 16)
    assert cam.pass_granularity(precision=abi.F64) == 1
    lin, rgb8 = cam.render_arrays(sc, precision=abi.F64, seed=11, accel=accel)
    o_lin, o_rgb8, o_seg, nan = oracle.render(sc.flat, cam.pod, abi.F64, seed=11, use_bvh=True, threads=8)
    assert nan == 0 and cam.last_stats["segments"] == o_seg
    assert np.array_equal(bits(lin), bits(o_lin))
    assert np.array_equal(rgb8, o_rgb8)


@pytest.mark.parametrize("depth", [1, 2, 7])
def test_f64_radiance_bit_exact_depth_limits_and_no_aa(oracle, plain_balls, depth):
    sc = plain_balls
    for spp in (0, 5):
        cam = lensless_camera(96, spp, depth)
        lin, _ = cam.render_arrays(sc, precision=abi.F64, seed=3)
        o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=3, use_bvh=True, threads=8)
        assert cam.last_stats["segments"] == o_seg
        assert np.array_equal(bits(lin), bits(o_lin))


@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
def test_waves_whose_lanes_all_end_in_the_scatter(oracle, prec):
    """Round 3's loop deals work after the segment and evaluates Philox once; a lane whose path ends INSIDE the scatter
    (absorbed) has to wait for the next iteration, and a wave in which every lane does so at once must still take what is left
    of its reserved batch and of the queue (a first version left the loop there: frames with holes, found by the segment
    count of a depth-1 render).  A ground that fills the view and absorbs 49 of 50 rays makes that the common case; depth
    limits 1 and 2 end every path at a hit without a scatter.  Frames and segment counts against the oracle, all front ends."""
    sc = Scene()
    sc.add(Sphere.new_static(1000.0, (0.0, -1000.0, 0.0), Lambertian.new((0.8, 0.7, 0.6), 0.02)))
    for k in range(40):
        sc.add(Sphere.new_static(0.3, (-6.0 + 0.3 * k, 0.3, -2.0 + 0.1 * (k % 7)), Lambertian.new((0.3, 0.6, 0.9), 0.05)))
    sc.build_bvh()
    # 640 x 360 x 16 one-sample items = 3.7 M: beyond the queue's tail (2.1 M items dealt one per lane), so that waves hold reserved batches
    for depth, spp, w in ((50, 16, 640), (1, 8, 160), (2, 4, 160)):
        cam = Camera.build((0.0, 6.0, 4.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), w, 16.0 / 9.0, spp, math.radians(50.0), 7.0, 0.0, depth)
        g = cam.pass_granularity(precision=prec)
        if prec == abi.F32:
            o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=21, threads=8, spec_chunk=g)
        else:
            o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=21, use_bvh=True, threads=8)
        if depth == 1:
            assert o_seg == cam.width * cam.height * spp
        for accel in ((None, "valu", "bvh") if prec == abi.F32 else (None, "valu")):
            lin, _ = cam.render_arrays(sc, precision=prec, seed=21, accel=accel)
            assert cam.last_stats["segments"] == o_seg, (depth, accel)
            assert np.array_equal(bits(lin), bits(o_lin)), (depth, accel)


@pytest.mark.parametrize("accel", [None, "valu", "bvh"])
def test_f32_radiance_bit_exact_against_the_f32_spec(oracle, plain_balls, accel):
    """The bench kernels (matrix-pipe broad phase, vector-ALU broad phase, tree) against the oracle's f32 instantiation in the
    f32 spec's order.  10 spp with 4 samples per work item: a pixel is the sum of the sums of samples 0-3, 4-7, 8-9."""
    sc = plain_balls
    for spp, q in ((16, 0), (10, 4)):
        cam = lensless_camera(160, spp)
        cam.chunk_samples = q
        g = cam.pass_granularity(precision=abi.F32)
        assert g == max(q, 1)
        lin, rgb8 = cam.render_arrays(sc, precision=abi.F32, seed=11, accel=accel)
        o_lin, o_rgb8, o_seg, nan = oracle.render(sc.flat, cam.pod, abi.F32, seed=11, threads=8, spec_chunk=g)
        assert nan == 0 and cam.last_stats["segments"] == o_seg
        assert np.array_equal(bits(lin), bits(o_lin))
        assert np.array_equal(rgb8, o_rgb8)


def test_f32_radiance_bit_exact_with_the_lens(oracle, plain_balls):
    """The thin-lens sample of the f32 spec (mirt_device.h::lens_sincos: quarter-turn reduction + two polynomial kernels, every
    step an IEEE operation) is restated in the oracle's f32 instantiation, so the bench camera - lens 0.4 deg, shutter times,
    moving spheres - is bit-exact in f32 as long as no image texture is hit."""
    sc, cam = plain_balls, presets.bouncing_balls_camera(160, 12, 50)
    assert cam.pod.defocus_angle > 0
    for accel in (None, "valu"):
        lin, rgb8 = cam.render_arrays(sc, precision=abi.F32, seed=6, accel=accel)
        o_lin, o_rgb8, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=6, threads=8, spec_chunk=cam.pass_granularity(precision=abi.F32))
        assert cam.last_stats["segments"] == o_seg
        assert np.array_equal(bits(lin), bits(o_lin)) and np.array_equal(rgb8, o_rgb8)


@pytest.mark.parametrize("prec", [abi.F64, abi.F32])
def test_stress_scene_radiance_bit_exact(oracle, prec):
    """2 000 static spheres of all three materials (the C5 generator; no texture, no lens), streaming-tile kernels."""
    sc, cam = presets.stress_scene(2000, 1), presets.stress_camera(128, 4, 50)
    lin, _ = cam.render_arrays(sc, precision=prec, seed=5)
    if prec == abi.F64:
        o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=5, use_bvh=True, threads=8)
    else:
        o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=5, threads=8, spec_chunk=cam.pass_granularity(precision=abi.F32))
    assert cam.last_stats["segments"] == o_seg
    assert np.array_equal(bits(lin), bits(o_lin))


@pytest.mark.parametrize("n", [505, 512, 513, 700, 1024])
def test_f32_radiance_bit_exact_across_the_lds_layouts(oracle, n):
    """The matrix-pipe kernel has two LDS layouts: up to 512 sorted slots with 512 material records, up to 1024 slots with
    184 (these scenes have ~0.9 n materials: LDS copy below 513, HBM/L2 above).  Either side of the boundary the frame is
    the oracle's f32-spec frame bit for bit, and the vector-ALU kernel's."""
    sc, cam = presets.stress_scene(n, 3), presets.stress_camera(96, 3, 12)
    assert cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    lin, rgb8 = cam.render_arrays(sc, precision=abi.F32, seed=8)
    seg = cam.last_stats["segments"]
    o_lin, o_rgb8, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=8, threads=8, spec_chunk=cam.pass_granularity(precision=abi.F32))
    assert seg == o_seg and np.array_equal(bits(lin), bits(o_lin)) and np.array_equal(rgb8, o_rgb8)
    v_lin, _ = cam.render_arrays(sc, precision=abi.F32, seed=8, accel="valu")
    assert cam.last_stats["segments"] == seg and np.array_equal(bits(v_lin), bits(lin))


def test_f32_paths_are_as_long_as_f64_paths_from_a_far_camera():
    """Regression for the f32 discriminant: with d = h^2 - a (|oc|^2 - r^2) in binary32 a camera 95 units from
    radius-0.1..0.5 spheres put hit points inside the spheres and the 10 000-sphere frame traced 60 % more segments than in
    binary64.  The two precisions draw different random numbers, so the counts agree statistically, not exactly."""
    sc, cam = presets.stress_scene(10000, 1), presets.stress_camera(480, 4, 50)
    cam.render_arrays(sc, precision=abi.F32, seed=1, want_linear=False)
    seg32 = cam.last_stats["segments"]
    cam.render_arrays(sc, precision=abi.F64, seed=1, want_linear=False)
    seg64 = cam.last_stats["segments"]
    assert abs(seg32 - seg64) < 0.01 * seg64, (seg32, seg64)


@pytest.mark.parametrize("prec", [abi.F64, abi.F32])
def test_textured_lens_scenes_are_bit_equal_almost_everywhere(oracle, prec):
    """With the image texture and the lens back in (the bench scene and camera), what is left between the two sides is the
    last ulp of acos/atan2/sin/cos: the lens makes every ray origin depend on sin/cos, so bit-equality holds only where
    OCML and glibc agree - the order of the arithmetic no longer contributes.  Bars: >= 90 % of the channels bit-equal,
    all within 1e-12 (f64) / 2e-5 (f32) relative except single samples that branch differently."""
    sc, cam = presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(160, 8, 50)
    lin, _ = cam.render_arrays(sc, precision=prec, seed=2)
    if prec == abi.F64:
        o_lin, _, _, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=2, use_bvh=True, threads=8)
        tol = 1e-12
    else:
        o_lin, _, _, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=2, threads=8, spec_chunk=cam.pass_granularity(precision=abi.F32))
        tol = 2e-5
    same = float((bits(lin) == bits(o_lin)).mean())
    close = float((np.abs(lin - o_lin) <= tol * (1 + np.abs(o_lin))).mean())
    print(f"precision {prec}: bit-equal channels {same:.4f}, within tolerance {close:.4f}")
    assert close >= 0.995
    if prec == abi.F32:  # the f32 spec spells out its own sin/cos (lens) and acos/atan2 (Sphere::uv): nothing is left to a math library
        assert same == 1.0
    # without the lens only the textured sphere's uv is left
    cam2 = lensless_camera(160, 8)
    lin2, _ = cam2.render_arrays(sc, precision=prec, seed=2)
    if prec == abi.F64:
        o2, _, _, _ = oracle.render(sc.flat, cam2.pod, abi.F64, seed=2, use_bvh=True, threads=8)
    else:
        o2, _, _, _ = oracle.render(sc.flat, cam2.pod, abi.F32, seed=2, threads=8, spec_chunk=cam2.pass_granularity(precision=abi.F32))
    same2 = float((bits(lin2) == bits(o2)).mean())
    print(f"precision {prec}, no lens: bit-equal channels {same2:.5f}")
    assert same2 >= 0.999
    if prec == abi.F32:
        assert same2 == 1.0


@pytest.mark.parametrize("accel", [None, "valu", "bvh"])
def test_f32_textured_presets_bit_exact(oracle, accel):
    """Round 2: every f32 frame is the oracle's bit for bit, image textures and lens included - the textured C4 scene (two
    image spheres, checker ground, lens) and the reference's default mars scene, through every closest-hit front end."""
    for sc, cam, seed in ((presets.textured_scene(), presets.textured_camera(192, 6, 50), 3), (presets.mars_texture_scene(), presets.default_camera(160, 9, 50), 4)):
        lin, rgb8 = cam.render_arrays(sc, precision=abi.F32, seed=seed, accel=accel)
        o_lin, o_rgb8, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=seed, threads=8, spec_chunk=cam.pass_granularity(precision=abi.F32))
        assert cam.last_stats["segments"] == o_seg
        assert np.array_equal(bits(lin), bits(o_lin)) and np.array_equal(rgb8, o_rgb8)


def test_reference_default_render_c1(oracle):
    """BASELINE.json configs[0] at its exact size: main.rs:30-56,126-127 - mars_texture_scene, 400 x 225, 100 spp, depth 50,
    lens 0.4 deg - f64 parity mode against the oracle's reference-order render (image texture + lens: tolerance bar)."""
    sc, cam = presets.mars_texture_scene(), presets.default_camera(400, 100, 50)
    assert (cam.width, cam.height) == (400, 225)
    lin, rgb8 = cam.render_arrays(sc, precision=abi.F64, seed=1)
    o_lin, o_rgb8, o_seg, nan = oracle.render(sc.flat, cam.pod, abi.F64, seed=1, use_bvh=True, threads=8)
    assert nan == 0
    assert abs(cam.last_stats["segments"] - o_seg) <= 2e-4 * o_seg
    err = np.abs(lin - o_lin)
    assert float((err <= 1e-9 * (1 + np.abs(o_lin))).mean()) >= 0.999 and err.max() <= 0.05
    assert float((rgb8 == o_rgb8).mean()) >= 0.9995 and int(np.abs(rgb8.astype(int) - o_rgb8.astype(int)).max()) <= 3


def test_two_calls_on_one_handle_on_two_streams_are_ordered():
    """A handle owns one work queue and one workspace: a second mirt_render_device on the same handle on another stream,
    with nothing waiting in between, must come out right (the library orders it behind the first with an event) - before
    the fix the second launch re-zeroed the queue head and overwrote the partial sums of the first."""
    import torch

    lib = abi.load()
    sc = presets.bouncing_balls_scene(1)
    ds = sc.device_scene(0)
    cam_a, cam_b = presets.bouncing_balls_camera(320, 8, 50), presets.bouncing_balls_camera(256, 4, 50)
    ref = {}
    for k, cam in (("a", cam_a), ("b", cam_b)):
        lin, _ = cam.render_arrays(sc, precision=abi.F32, seed=9)
        ref[k] = torch.from_numpy(lin).cuda()
    out_a = torch.zeros_like(ref["a"])
    out_b = torch.zeros_like(ref["b"])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    oa, ob = cam_a.opts(precision=abi.F32, seed=9), cam_b.opts(precision=abi.F32, seed=9)
    torch.cuda.synchronize()
    for _ in range(3):  # back to back, no host wait, alternating streams
        abi.check(lib.mirt_render_device(ds.handle, C.byref(cam_a.pod), C.byref(oa), out_a.data_ptr(), None, s1.cuda_stream, None))
        abi.check(lib.mirt_render_device(ds.handle, C.byref(cam_b.pod), C.byref(ob), out_b.data_ptr(), None, s2.cuda_stream, None))
    torch.cuda.synchronize()
    assert torch.equal(out_a, ref["a"]) and torch.equal(out_b, ref["b"])
    # a deferred call still has to be collected before the handle takes another launch
    od = cam_b.opts(precision=abi.F32, seed=9, flags=abi.FLAG_DEFER_STATS)
    abi.check(lib.mirt_render_device(ds.handle, C.byref(cam_b.pod), C.byref(od), out_b.data_ptr(), None, s2.cuda_stream, None))
    assert lib.mirt_render_device(ds.handle, C.byref(cam_a.pod), C.byref(oa), out_a.data_ptr(), None, s1.cuda_stream, None) == -1
    st = abi.MirtStats()
    abi.check(lib.mirt_render_wait(ds.handle, C.byref(st)))
    assert st.segments > 0


def test_two_host_threads_share_one_handle():
    """include/mirt.h: "host threads sharing a handle are serialised".  The host-buffer entry points grow, clear and copy
    back staging frames the handle owns, so two threads on one handle must queue up for the WHOLE call (the lock
    used to cover the enqueue only - a second thread could free or overwrite a frame the first was still copying from).
    ctypes releases the GIL for the duration of a call, so the two threads really are inside the library together."""
    import threading

    sc = presets.bouncing_balls_scene(1)
    sc.device_scene(0)
    cams = [presets.bouncing_balls_camera(256, 4, 20), presets.bouncing_balls_camera(96, 2, 8)]  # different frame sizes: the staging frames regrow
    refs = [c.render_arrays(sc, precision=abi.F32, seed=13) for c in cams]
    bad = []

    def worker(k):
        cam = presets.bouncing_balls_camera(256, 4, 20) if k == 0 else presets.bouncing_balls_camera(96, 2, 8)
        for _ in range(6):
            lin, rgb8 = cam.render_arrays(sc, precision=abi.F32, seed=13)
            if not (np.array_equal(bits(lin), bits(refs[k][0])) and np.array_equal(rgb8, refs[k][1])):
                bad.append(k)

    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1, 0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad


def test_f64_deep_bounce_limit_shrinks_the_grid_instead_of_failing(oracle):
    """the f64 attenuation stacks (24 B x max_bounce_depth per lane) used to refuse bounce limits whose stacks for
    the FULL grid passed 8 GiB; any grid size renders the same image, so the grid now shrinks to what fits."""
    sc, cam = balls_without_image(1), lensless_camera(64, 2, 400000)  # 2.4 GB of stacks per block: 3 blocks fit
    lin, _ = cam.render_arrays(sc, precision=abi.F64, seed=3)
    ref_cam = lensless_camera(64, 2, 400000)
    o_lin, _, o_seg, _ = oracle.render(sc.flat, ref_cam.pod, abi.F64, seed=3, use_bvh=True, threads=8)
    assert cam.last_stats["segments"] == o_seg and np.array_equal(bits(lin), bits(o_lin))


def test_image_depends_on_arguments_only(monkeypatch):
    """No entry point reads the environment: the variables round 1's library obeyed change nothing any more, and the
    work-item size travels in mirt_opts."""
    sc, cam = presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(128, 8, 20)
    a, _ = cam.render_arrays(sc, precision=abi.F32, seed=4)
    for k, v in (("MIRT_CHUNK", "4"), ("MIRT_GROUP_CHUNKS", "1"), ("MIRT_BROAD", "valu"), ("MIRT_BVH_KEEP", "3")):
        monkeypatch.setenv(k, v)
    assert cam.pass_granularity(precision=abi.F32) == 1 and cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    b, _ = cam.render_arrays(sc, precision=abi.F32, seed=4)
    assert np.array_equal(bits(a), bits(b))
    bad = cam.opts(precision=abi.F32)
    bad._reserved[0] = 7
    assert abi.load().mirt_closest_hit_kind(sc.device_scene(0).handle, C.byref(cam.pod), C.byref(bad)) == -1


def test_fuzz_bit_exact_radiance():
    """A slice of tools/fuzz_exact.py (run over 2 000 scenes, profiles/r02_microbench.txt): random scenes, cameras, depths, sample
    counts and work-item sizes; f64 (both front ends) and f32 (three front ends) bit-equal to the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_exact.py"), "5000", "12"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
