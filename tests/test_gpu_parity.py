"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden maps.  Needs a GPU.

Bars (BASELINE.json):
  * hit-id / t maps of deterministic primary rays: BIT-EXACT at equal precision (f64 vs f64 oracle, f32 vs f32 oracle)
  * quantiser: byte-exact given equal linear input
  * radiance with the shared counter-based RNG: per-pixel tolerance stated in each test (the GPU multiplies the
    attenuations front to back, the recursion back to front, and libm differs from OCML in the last ulp, so a
    handful of samples may take a different branch)
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

from raytracer_amd import abi, presets
from raytracer_amd.api import Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, Scene, Sphere

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * math.log10(1.0 / mse)


SCENES = {
    "mars": (presets.mars_texture_scene, lambda w, s, d=50: presets.default_camera(w, s, d)),
    "balls": (lambda: presets.bouncing_balls_scene(1), lambda w, s, d=50: presets.bouncing_balls_camera(w, s, d)),
}


@pytest.fixture(scope="module")
def scenes():
    return {k: mk() for k, (mk, _) in SCENES.items()}


# ---- K1: hit-id maps, bit exact ------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mars", "balls"])
@pytest.mark.parametrize("prec", [abi.F64, abi.F32])
def test_hit_id_map_bit_exact_vs_oracle(oracle, scenes, name, prec):
    sc = scenes[name]
    cam = SCENES[name][1](400, 0)
    ids, t = cam.primary_hits(sc, precision=prec)
    o_ids, o_t = oracle.primary_hits(sc.flat, cam.pod, prec, use_bvh=(prec == abi.F64))
    assert np.array_equal(ids, o_ids)
    assert np.array_equal(bits(t), bits(o_t))
    assert (ids != abi.MISS_ID).any() and (ids == abi.MISS_ID).any()


@pytest.mark.parametrize("name", ["mars", "balls"])
def test_hit_id_map_matches_committed_golden(golden_dir, scenes, name):
    g = np.load(os.path.join(golden_dir, "oracle_maps.npz"))
    for w in (64, 400):
        for prec, tag in ((abi.F64, "f64"), (abi.F32, "f32")):
            ids, t = SCENES[name][1](w, 0).primary_hits(scenes[name], precision=prec)
            assert np.array_equal(ids, g[f"{name}_{w}_{tag}_id"])
            assert np.array_equal(bits(t), bits(g[f"{name}_{w}_{tag}_t"]))


def test_hit_id_map_1080p_bit_exact(oracle, scenes):
    """BASELINE.json configs[1]/[2] size: full 1920x1080 map of the ~485-sphere scene, f64 and f32."""
    sc = scenes["balls"]
    cam = presets.bouncing_balls_camera(1920, 0, 50)
    for prec in (abi.F64, abi.F32):
        ids, t = cam.primary_hits(sc, precision=prec)
        o_ids, o_t = oracle.primary_hits(sc.flat, cam.pod, prec, use_bvh=(prec == abi.F64))
        assert int((ids != o_ids).sum()) == 0
        assert np.array_equal(bits(t), bits(o_t))


def test_f32_vs_f64_id_mismatch_is_a_handful_of_edge_pixels(scenes):
    cam = presets.bouncing_balls_camera(1920, 0, 50)
    a, _ = cam.primary_hits(scenes["balls"], precision=abi.F64)
    b, _ = cam.primary_hits(scenes["balls"], precision=abi.F32)
    assert (a != b).mean() < 1e-3


# ---- closest-hit edge cases --------------------------------------------------------------------------------------------
def _cam(w=32, ratio=1.0, spp=0, depth=10, lookfrom=None, lookat=None, vfov=None):
    return Camera.build(lookfrom, lookat, None, w, ratio, spp, vfov, None, None, depth)


def _check_map(oracle, sc, cam):
    for prec in (abi.F64, abi.F32):
        ids, t = cam.primary_hits(sc, precision=prec)
        o_ids, o_t = oracle.primary_hits(sc.flat, cam.pod, prec)
        assert np.array_equal(ids, o_ids) and np.array_equal(bits(t), bits(o_t))
    return ids


def test_empty_scene_is_all_miss(oracle):
    sc = Scene()
    sc.build_bvh()
    ids = _check_map(oracle, sc, _cam(19, 1.7))
    assert (ids == abi.MISS_ID).all()
    lin, rgb8 = _cam(19, 1.7, spp=2).render_arrays(sc, precision=abi.F32, seed=1)
    assert np.isfinite(lin).all() and (rgb8 > 100).all()  # sky everywhere


def test_ties_go_to_lowest_id_and_duplicates(oracle):
    sc = Scene()
    m = Lambertian.new((1, 1, 1), 1.0)
    # ids: 0 moving (sorted AFTER the statics on the device), 1 and 2 static twins of it at tm = 0
    sc.add(Sphere.new(1.0, (0, 0, -5), (0, 0, -5), m))
    sc.add(Sphere.new_static(1.0, (0, 0, -5), m))
    sc.add(Sphere.new_static(1.0, (0, 0, -5), m))
    sc.build_bvh()
    ids = _check_map(oracle, sc, _cam(24))
    assert set(np.unique(ids)) == {0, abi.MISS_ID}


def test_candidate_list_overflow_flush(oracle):
    """More than MIRT_CAND_CAP spheres along one ray: 40 concentric shells + 30 spheres in a row behind each other."""
    sc = Scene()
    m = Lambertian.new((1, 1, 1), 1.0)
    for k in range(40):
        sc.add(Sphere.new_static(3.0 - 0.05 * k, (0, 0, -6), m))
    for k in range(30):
        sc.add(Sphere.new(0.4, (0.5, 0.5, -12.0 - k), (0.5, 0.6, -12.0 - k), m))
    sc.build_bvh()
    ids = _check_map(oracle, sc, _cam(40))
    assert (ids == 0).any()
    # from inside all shells: far roots, the innermost shell (id 39) wins
    ids = _check_map(oracle, sc, _cam(16, lookfrom=(0, 0, -6), lookat=(0, 0, -7)))
    assert (ids == 39).all()


def test_entry_list_overflow_many_groups(oracle):
    """The f32 path keeps one list entry per group of 16 records with a candidate (16 entries per lane): 600 spheres
    strung along the view axis put candidates into 38 groups for the central rays, so the list flushes twice per ray.
    Looked at from the front, from inside the string and from the far end (far-to-near order of the sorted array)."""
    sc = Scene()
    m = Lambertian.new((1, 1, 1), 1.0)
    for k in range(600):
        if k % 7 == 3:
            sc.add(Sphere.new(0.4, (0.0, 0.0, -12.0 - 0.9 * k), (0.0, 0.05, -12.0 - 0.9 * k), m))
        else:
            sc.add(Sphere.new_static(0.4, (0.0, 0.0, -12.0 - 0.9 * k), m))
    sc.build_bvh()
    ids = _check_map(oracle, sc, _cam(24, vfov=0.2))
    assert (ids == 0).any()
    ids = _check_map(oracle, sc, _cam(24, lookfrom=(0, 0, -12.0 - 0.9 * 300.5), lookat=(0, 0, -1000), vfov=0.3))
    assert (ids == 301).any()
    ids = _check_map(oracle, sc, _cam(24, lookfrom=(0, 0.01, -700), lookat=(0, 0, 0), vfov=0.2))
    assert (ids == 599).any()


def test_many_tiles_moving_and_static_mix(oracle):
    rng = np.random.default_rng(5)
    sc = Scene()
    for k in range(333):  # > 5 tiles of 64, odd counts to exercise the group padding
        c = tuple(rng.uniform(-6, 6, 3) + np.array([0, 0, -14.0]))
        mat = [Lambertian.new((0.5, 0.5, 0.5), 0.9), Metal.new((0.8, 0.8, 0.8), 0.1), Dielectric.new(1.5)][k % 3]
        if k % 5 < 2:
            sc.add(Sphere.new(float(rng.uniform(0.1, 0.6)), c, (c[0], c[1] + 0.3, c[2]), mat))
        else:
            sc.add(Sphere.new_static(float(rng.uniform(0.1, 0.6)), c, mat))
    sc.build_bvh()
    ids = _check_map(oracle, sc, _cam(97, 1.3))  # width not a multiple of the 8x8 wave tile
    assert len(np.unique(ids)) > 50


def test_large_scene_streams_wave_private_tiles(oracle):
    """More spheres than the block-resident LDS table holds (1024 records): the f32 broad phase streams tiles."""
    sc = presets.stress_scene(1500, seed=3)
    cam = presets.stress_camera(160, 0, 50)
    ids = _check_map(oracle, sc, cam)
    assert len(np.unique(ids)) > 200
    cam = presets.stress_camera(64, 4, 8)
    lin, _ = cam.render_arrays(sc, precision=abi.F32, seed=2)
    o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=2, threads=8)
    err = np.abs(lin - o_lin)
    assert float((err <= 2e-5 * (1 + np.abs(o_lin))).mean()) >= 0.995 and abs(cam.last_stats["segments"] - o_seg) <= 0.002 * o_seg


def test_one_pixel_image_and_tall_image(oracle, scenes):
    _check_map(oracle, scenes["mars"], Camera.build((0, 0, 12), (0, 0, 0), None, 1, 1.0, 0, 0.4, 10.0, None, 5))
    _check_map(oracle, scenes["mars"], Camera.build((0, 0, 12), (0, 0, 0), None, 3, 0.02, 0, 0.4, 10.0, None, 5))


# ---- K2: radiance -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mars", "balls"])
@pytest.mark.parametrize("prec", [abi.F64, abi.F32])
def test_radiance_vs_oracle_same_rng(oracle, scenes, name, prec):
    """Same seeded counter RNG on both sides.  Tolerance: >= 99.5 % of the channels agree to 1e-9 (f64) / 2e-5 (f32)
    relative-absolute; whole-image PSNR >= 50 dB (f64) / 45 dB (f32); no pixel off by more than 0.25 (a single
    sample of 16 taking another branch moves a channel by at most 1.25/16)."""
    sc = scenes[name]
    cam = SCENES[name][1](96, 16)
    lin, rgb8 = cam.render_arrays(sc, precision=prec, seed=7)
    o_lin, o_rgb8, o_seg, o_nan = oracle.render(sc.flat, cam.pod, prec, seed=7, threads=8)
    assert o_nan == 0 and np.isfinite(lin).all()
    err = np.abs(lin.astype(np.float64) - o_lin.astype(np.float64))
    tol = 1e-9 if prec == abi.F64 else 2e-5
    frac_close = float((err <= tol * (1.0 + np.abs(o_lin))).mean())
    assert frac_close >= 0.995, frac_close
    assert err.max() <= 0.25, err.max()
    assert psnr(lin, o_lin) >= (50.0 if prec == abi.F64 else 45.0)
    # exact segment count unless a path branched differently
    assert abs(cam.last_stats["segments"] - o_seg) <= 0.002 * o_seg
    assert (np.abs(rgb8.astype(int) - o_rgb8.astype(int)) <= 1).mean() >= 0.995


@pytest.mark.parametrize("name", ["mars", "balls"])
def test_radiance_matches_committed_golden(golden_dir, scenes, name):
    g = np.load(os.path.join(golden_dir, "oracle_maps.npz"))
    cam = SCENES[name][1](64, 16)
    for prec, tag, tol in ((abi.F64, "f64", 1e-9), (abi.F32, "f32", 2e-5)):
        lin, _ = cam.render_arrays(scenes[name], precision=prec, seed=7)
        ref = g[f"{name}_64x16spp_{tag}_lin"]
        err = np.abs(lin.astype(np.float64) - ref.astype(np.float64))
        assert float((err <= tol * (1.0 + np.abs(ref))).mean()) >= 0.995 and err.max() <= 0.25


def test_no_aa_render_is_deterministic_branch(oracle, scenes):
    """samples_per_pixel == 0 (Some(0)): one path per pixel through the pixel CORNER, tm = 0 (camera.rs:247-254)."""
    sc = scenes["balls"]
    cam = presets.bouncing_balls_camera(160, 0, 50)
    lin, _ = cam.render_arrays(sc, precision=abi.F64, seed=3)
    o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=3, threads=8)
    err = np.abs(lin - o_lin)
    assert float((err <= 1e-12).mean()) >= 0.999


def test_depth_limits(oracle, scenes):
    sc = scenes["balls"]
    for depth in (0, 1, 2):
        cam = presets.bouncing_balls_camera(64, 4, depth)
        lin, _ = cam.render_arrays(sc, precision=abi.F64, seed=1)
        o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=1, threads=4)
        assert cam.last_stats["segments"] == o_seg
        assert np.abs(lin - o_lin).max() <= 1e-12
        if depth == 0:
            assert (lin == 0).all() and o_seg == 0


def test_image_independent_of_partition(scenes):
    """Interleaved strips over 1, 2, 3 parts give bit-identical pixels (counter RNG keyed on the global pixel)."""
    sc = scenes["balls"]
    cam = presets.bouncing_balls_camera(80, 4, 50)
    lib = abi.load()
    ds = sc.device_scene(0)
    H, W = cam.height, cam.width
    full, _ = cam.render_arrays(sc, precision=abi.F32, seed=9)
    for n in (2, 3):
        acc = np.zeros_like(full)
        for r in range(n):
            part = np.zeros_like(full)
            o = cam.opts(precision=abi.F32, seed=9, strip_rows=8, part_index=r, part_count=n)
            abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), part.ctypes.data, None, None))
            acc += part  # rows not owned stay zero
        assert np.array_equal(acc, full)


def test_run_to_run_bit_stable(scenes):
    cam = presets.bouncing_balls_camera(64, 8, 50)
    a, _ = cam.render_arrays(scenes["balls"], precision=abi.F32, seed=2)
    b, _ = cam.render_arrays(scenes["balls"], precision=abi.F32, seed=2)
    c, _ = cam.render_arrays(scenes["balls"], precision=abi.F32, seed=3)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_textured_scene_psnr_f32_vs_f64_oracle(oracle):
    """Throughput mode against the reference's arithmetic (f64 oracle).  The two precisions draw different
    uniforms (f32: 4 per Philox block, f64: 2), so this is Monte Carlo agreement: two independent 128-spp
    estimates of the same image -> PSNR >= 30 dB and no bias (mean difference < 2e-3)."""
    sc = presets.textured_scene()
    cam = presets.textured_camera(96, 128, 50)
    lin, _ = cam.render_arrays(sc, precision=abi.F32, seed=4)
    o_lin, _, _, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=4, threads=8)
    assert psnr(lin, o_lin) >= 30.0
    assert abs(float(lin.mean()) - float(o_lin.mean())) < 2e-3


# ---- K3 / K4 / device entry points -------------------------------------------------------------------------------------
def test_quantiser_byte_exact(oracle):
    import torch

    lib = abi.load()
    rng = np.random.default_rng(0)
    for dt, prec in ((np.float64, abi.F64), (np.float32, abi.F32)):
        x = np.concatenate([rng.random(100001) * 1.3 - 0.1, [0.0, -0.0, 1.0, 4.0, 0.25, np.nan, np.inf, 1e-30, (127.5 / 254.999) ** 2]]).astype(dt)
        want, _ = oracle.quantize(x)
        d_x = torch.from_numpy(x).cuda()
        d_o = torch.empty(x.size, dtype=torch.uint8, device="cuda")
        nan = C.c_uint64(123)
        abi.check(lib.mirt_quantize_device(d_x.data_ptr(), prec, x.size, d_o.data_ptr(), None, C.byref(nan)))
        assert np.array_equal(d_o.cpu().numpy(), want) and nan.value == 0


def test_device_entry_points_and_deinterleave(scenes):
    import torch

    from raytracer_amd import dist

    lib = abi.load()
    sc = scenes["balls"]
    cam = presets.bouncing_balls_camera(72, 2, 50)
    ds = sc.device_scene(0)
    H, W = cam.height, cam.width
    full, full8 = cam.render_arrays(sc, precision=abi.F32, seed=5)
    n, S = 3, 8
    rows0 = dist.part_rows(H, S, 0, n)
    bands = torch.zeros((n, rows0, W, 3), dtype=torch.float32, device="cuda")
    bands8 = torch.zeros((n, rows0, W, 3), dtype=torch.uint8, device="cuda")
    for r in range(n):
        o = cam.opts(precision=abi.F32, seed=5, strip_rows=S, part_index=r, part_count=n, flags=abi.FLAG_COMPACT_ROWS)
        st = abi.MirtStats()
        abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), bands[r].data_ptr(), bands8[r].data_ptr(), None, C.byref(st)))
        assert st.segments > 0 and st.kernel_ms > 0
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    frame8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    abi.check(lib.mirt_deinterleave_device(bands.data_ptr(), bands[0].numel() * 4, n, S, W, H, 12, frame.data_ptr(), None))
    abi.check(lib.mirt_deinterleave_device(bands8.data_ptr(), bands8[0].numel(), n, S, W, H, 3, frame8.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), full) and np.array_equal(frame8.cpu().numpy(), full8)
    assert np.array_equal(dist.deinterleave_host(bands.cpu().numpy(), H, S), full)
    # compact primary hits
    ids_full, t_full = cam.primary_hits(sc, precision=abi.F32)
    d_id = torch.full((rows0, W), 7, dtype=torch.int32, device="cuda")
    d_t = torch.zeros((rows0, W), dtype=torch.float32, device="cuda")
    o = cam.opts(precision=abi.F32, strip_rows=S, part_index=1, part_count=n, flags=abi.FLAG_COMPACT_ROWS)
    abi.check(lib.mirt_primary_hits_device(ds.handle, C.byref(cam.pod), C.byref(o), d_id.data_ptr(), d_t.data_ptr(), None))
    torch.cuda.synchronize()
    rows = dist.owned_rows(H, S, 1, n)
    assert np.array_equal(d_id.cpu().numpy().view(np.uint32)[: len(rows)], ids_full[rows])


def test_error_codes_on_gpu(scenes):
    lib = abi.load()
    cam = presets.default_camera(16, 1, 5)
    ds = scenes["mars"].device_scene(0)
    o = cam.opts(precision=16)
    assert lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), None, None, None) == -1
    o = cam.opts(precision=abi.F32, part_index=2, part_count=2)
    assert lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), None, None, None) == -1
    h = C.c_void_p()
    assert lib.mirt_scene_create(C.byref(scenes["mars"].flat.desc), 99, C.byref(h)) == -2
    assert lib.mirt_device_count() >= 1


def test_ppm_output_matches_oracle_bytes(oracle, scenes):
    """Camera::render drop-in: header + one 'r g b' line per pixel; bytes equal the oracle's for the no-AA branch."""
    import io

    from raytracer_amd.api import ppm_body

    sc = scenes["mars"]
    cam = presets.default_camera(64, 0, 50)
    buf = io.BytesIO()
    cam.precision = abi.F64
    cam.render(sc, out=buf)
    _, o_rgb8, _, _ = oracle.render(sc.flat, cam.pod, abi.F64, seed=0, threads=4)
    assert buf.getvalue() == b"P3\n64 36\n255\n" + ppm_body(o_rgb8)


# ---- progressive passes / checkpointing ------------------------------------------------------------------
@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
@pytest.mark.parametrize("accel", [None, "bvh", "valu"])
def test_progressive_passes_equal_the_one_call_render(scenes, prec, accel):
    """Any cut of the sample range into passes - also through a checkpoint (the sums saved and reloaded) - ends in the image
    of the one-call render, bit for bit; intermediate images are the running means."""
    if accel == "bvh" and prec == abi.F64:
        pytest.skip("the tree is an f32-mode accelerator")
    sc, cam = scenes["balls"], SCENES["balls"][1](96, 12)
    cam.precision = prec
    ref, ref8 = cam.render_arrays(sc, seed=5, accel=accel)
    seg_ref = cam.last_stats["segments"]
    g = cam.pass_granularity()
    assert g >= 1 and 12 % g == 0
    real = np.float64 if prec == abi.F64 else np.float32
    for cuts in ([0, 12], [0, 4, 12], [0, 1 * g, 2 * g, 7 * g if 7 * g < 12 else 12, 12]):
        cuts = sorted(set(cuts))
        acc = np.full((cam.height, cam.width, 3), np.nan, dtype=real)  # the first pass must not read it
        seg = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            lin, rgb8 = cam.render_pass(sc, a, b - a, acc, seed=5, accel=accel)
            seg += cam.last_stats["segments"]
            acc = np.frombuffer(acc.tobytes(), dtype=real).reshape(acc.shape).copy()  # "checkpoint": through bytes and back
            if b < 12:
                assert np.allclose(lin, acc / b, rtol=1e-6, atol=0)
        assert np.array_equal(lin.view(np.uint8), ref.view(np.uint8)) and np.array_equal(rgb8, ref8)
        assert seg == seg_ref


def test_progressive_pass_argument_errors(scenes):
    sc, cam = scenes["balls"], SCENES["balls"][1](32, 8)
    cam.precision = abi.F32
    acc = np.zeros((cam.height, cam.width, 3), np.float32)
    for first, count in ((0, 0), (0, 9), (8, 1), (4, 8)):
        with pytest.raises(abi.MirtError):
            cam.render_pass(sc, first, count, acc, seed=1)
    cam0 = SCENES["balls"][1](32, 0)
    cam0.precision = abi.F32
    assert cam0.pass_granularity() == 0
    with pytest.raises(abi.MirtError):
        cam0.render_pass(sc, 0, 1, acc, seed=1)


def test_progressive_passes_with_larger_work_items(scenes):
    """With several samples per work item (large images; here asked for through mirt_opts.chunk_samples) pass boundaries are
    multiples of that size, and the result is still the one-call image."""
    sc, cam = scenes["balls"], SCENES["balls"][1](64, 10)
    cam.precision = abi.F32
    cam.chunk_samples = 4
    assert cam.pass_granularity() == 4
    ref, ref8 = cam.render_arrays(sc, seed=2)
    acc = np.zeros((cam.height, cam.width, 3), np.float32)
    with pytest.raises(abi.MirtError):
        cam.render_pass(sc, 0, 6, acc, seed=2)
    for a, b in ((0, 4), (4, 8), (8, 10)):
        lin, rgb8 = cam.render_pass(sc, a, b - a, acc, seed=2)
    assert np.array_equal(lin.view(np.uint8), ref.view(np.uint8)) and np.array_equal(rgb8, ref8)


def test_progressive_passes_on_compact_bands(scenes):
    """The device entry point with the multi-GPU partition options: each part accumulates its own compact band over two
    passes; reassembled, the frame is the one-call frame."""
    import ctypes as C

    import torch

    from raytracer_amd import dist

    sc, cam = scenes["balls"], SCENES["balls"][1](80, 6, 12)
    cam.precision = abi.F32
    ref, _ = cam.render_arrays(sc, seed=8)
    lib = abi.load()
    ds = sc.device_scene(0)
    H, W, S, parts = cam.height, cam.width, 8, 3
    rows0 = dist.part_rows(H, S, 0, parts)
    bands = torch.zeros((parts, rows0, W, 3), dtype=torch.float32, device="cuda")
    acc = torch.full((parts, rows0, W, 3), float("nan"), dtype=torch.float32, device="cuda")
    for r in range(parts):
        o = cam.opts(precision=abi.F32, seed=8, strip_rows=S, part_index=r, part_count=parts, flags=abi.FLAG_COMPACT_ROWS)
        for first, count in ((0, 2), (2, 4)):
            abi.check(lib.mirt_render_pass_device(ds.handle, C.byref(cam.pod), C.byref(o), first, count, acc[r].data_ptr(), bands[r].data_ptr(), None, None, None))
    torch.cuda.synchronize()
    frame = dist.deinterleave_host(bands.cpu().numpy(), H, S)
    assert np.array_equal(frame.view(np.uint8), ref.view(np.uint8))


def test_deferred_statistics_and_frames_in_flight(scenes):
    """MIRT_FLAG_DEFER_STATS: the call returns behind its launch, mirt_render_wait collects the same statistics later.  Three
    frames in flight on three scene handles and streams (what bench.py does on several GPUs) are the frames of three
    calls made one after the other, and a handle refuses a second launch before the first has been collected."""
    import torch

    from raytracer_amd.api import DeviceScene

    sc, cam = scenes["balls"], SCENES["balls"][1](160, 8, 30)
    cam.precision = abi.F32
    ref, ref8 = cam.render_arrays(sc, seed=4)
    seg_ref = cam.last_stats["segments"]
    lib = abi.load()
    H, W = cam.height, cam.width
    handles = [DeviceScene(sc.flat, 0) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    lin = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(3)]
    rgb = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(3)]
    o = cam.opts(precision=abi.F32, seed=4, flags=abi.FLAG_DEFER_STATS)
    st = abi.MirtStats()
    with pytest.raises(abi.MirtError):  # nothing outstanding yet
        abi.check(lib.mirt_render_wait(handles[0].handle, C.byref(st)))
    for rounds in range(2):
        for i in range(3):
            abi.check(lib.mirt_render_device(handles[i].handle, C.byref(cam.pod), C.byref(o), lin[i].data_ptr(), rgb[i].data_ptr(), streams[i].cuda_stream, None))
        with pytest.raises(abi.MirtError):  # handle 1 still owes its statistics
            abi.check(lib.mirt_render_device(handles[1].handle, C.byref(cam.pod), C.byref(o), lin[1].data_ptr(), rgb[1].data_ptr(), streams[1].cuda_stream, None))
        for i in (2, 0, 1):  # any order
            abi.check(lib.mirt_render_wait(handles[i].handle, C.byref(st)))
            assert st.segments == seg_ref and st.kernel_ms > 0.0
        with pytest.raises(abi.MirtError):
            abi.check(lib.mirt_render_wait(handles[0].handle, None))
        torch.cuda.synchronize()
        for i in range(3):
            assert np.array_equal(lin[i].cpu().numpy().view(np.uint8), ref.view(np.uint8)) and np.array_equal(rgb[i].cpu().numpy(), ref8)
            lin[i].zero_()
    with pytest.raises(abi.MirtError):  # host-buffer calls return when the frame is done: the flag is refused
        abi.check(lib.mirt_render(handles[0].handle, C.byref(cam.pod), C.byref(o), None, None, None))
    for h in handles:
        h.close()


def test_fuzz_radiance_against_the_oracle():
    """A slice of tools/fuzz_radiance.py (run over 1200 scenes, profiles/r01_microbench.txt): random scenes with every material
    and texture kind, random cameras, both precisions, every front end, same seeded RNG on both sides."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_radiance.py"), "7000", "12"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and "failing renders: 0" in r.stdout, r.stdout[-800:] + r.stderr[-400:]


@pytest.mark.parametrize("width,elem", [(72, 12), (72, 3), (71, 3), (71, 12), (5, 24), (1920, 12), (64, 1)])
def test_deinterleave_unit_sizes(width, elem):
    """K4 moves 16-, 4- or 1-byte units depending on the row size and the alignment; every choice is the host scatter."""
    import torch

    from raytracer_amd import dist

    lib = abi.load()
    H, S = 43, 8
    rng = np.random.default_rng(width * 31 + elem)
    for n in (1, 3):
        rows0 = dist.part_rows(H, S, 0, n)
        bands = rng.integers(0, 256, (n, rows0, width, elem), dtype=np.uint8)
        want = dist.deinterleave_host(bands, H, S)
        d_bands = torch.from_numpy(bands).cuda()
        for off in (0, 4 if (width * elem) % 4 == 0 else 1):  # a frame pointer that is not 16-byte aligned falls back to narrower units
            buf = torch.zeros(H * width * elem + 16, dtype=torch.uint8, device="cuda")
            abi.check(lib.mirt_deinterleave_device(d_bands.data_ptr(), rows0 * width * elem, n, S, width, H, elem, buf.data_ptr() + off, None))
            torch.cuda.synchronize()
            got = buf.cpu().numpy()[off : off + H * width * elem].reshape(H, width, elem)
            assert np.array_equal(got, want)


@pytest.mark.parametrize("workload", ["balls-f32", "balls-f64", "textured-f32"])
def test_bench_frame_full_size_properties(scenes, workload):
    """BASELINE's configurations (1920x1080, 64 spp, depth 50, ~485 spheres; the textured scene at 3840x2160, 256 spp)
    are too large for the oracle; at that size the
    frame is checked through properties that do not depend on the size: the 8 interleaved bands of an 8-GPU run,
    reassembled by K4, are the one-call frame bit for bit (linear radiance and bytes, equal segment counts); so is the
    frame accumulated in two progressive passes; so is the frame of the vector-ALU broad phase; a second run repeats it."""
    import torch

    from raytracer_amd import dist

    lib = abi.load()
    prec = abi.F64 if workload.endswith("f64") else abi.F32
    real, rsz = (torch.float64, 8) if prec == abi.F64 else (torch.float32, 4)
    if workload.startswith("balls"):
        sc, cam, spp = scenes["balls"], presets.bouncing_balls_camera(1920, 64, 50), 64
        assert (cam.height, cam.width) == (1080, 1920)
    else:
        sc, cam, spp = presets.textured_scene(), presets.textured_camera(3840, 256, 50), 256
        assert (cam.height, cam.width) == (2160, 3840)
    ds = sc.device_scene(0)
    H, W = cam.height, cam.width

    def whole(flags=0):
        lin = torch.zeros((H, W, 3), dtype=real, device="cuda")
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        st = abi.MirtStats()
        o = cam.opts(precision=prec, seed=1, flags=flags)
        abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), lin.data_ptr(), rgb.data_ptr(), None, C.byref(st)))
        return lin, rgb, st.segments

    ref, ref8, seg_ref = whole()
    again, again8, seg2 = whole()
    assert torch.equal(ref, again) and torch.equal(ref8, again8) and seg2 == seg_ref and seg_ref > 1.5 * H * W * spp
    valu, valu8, seg3 = whole(abi.FLAG_VALU_BROAD)
    assert torch.equal(ref, valu) and torch.equal(ref8, valu8) and seg3 == seg_ref
    # bands -> K4: the 8-GPU split of bench.py, and (bench frame) other rank counts and strip heights
    for n, S in ((8, 8),) + (((2, 8), (4, 16), (3, 5)) if workload == "balls-f32" else ()):
        rows0 = dist.part_rows(H, S, 0, n)
        bands = torch.zeros((n, rows0, W, 3), dtype=real, device="cuda")
        bands8 = torch.zeros((n, rows0, W, 3), dtype=torch.uint8, device="cuda")
        seg = 0
        for r in range(n):
            o = cam.opts(precision=prec, seed=1, strip_rows=S, part_index=r, part_count=n, flags=abi.FLAG_COMPACT_ROWS)
            st = abi.MirtStats()
            abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), bands[r].data_ptr(), bands8[r].data_ptr(), None, C.byref(st)))
            seg += st.segments
        frame = torch.empty((H, W, 3), dtype=real, device="cuda")
        frame8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        abi.check(lib.mirt_deinterleave_device(bands.data_ptr(), bands[0].numel() * rsz, n, S, W, H, 3 * rsz, frame.data_ptr(), None))
        abi.check(lib.mirt_deinterleave_device(bands8.data_ptr(), bands8[0].numel(), n, S, W, H, 3, frame8.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(frame, ref) and torch.equal(frame8, ref8) and seg == seg_ref, (n, S)
        del bands, bands8, frame, frame8
    # two progressive passes
    acc = torch.full((H, W, 3), float("nan"), dtype=real, device="cuda")
    lin = torch.zeros((H, W, 3), dtype=real, device="cuda")
    rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    o = cam.opts(precision=prec, seed=1)
    g = lib.mirt_pass_granularity(C.byref(cam.pod), C.byref(o))
    assert g >= 1 and spp % g == 0
    cut = max(g, spp * 3 // 8 // g * g)
    for first, count in ((0, cut), (cut, spp - cut)):
        abi.check(lib.mirt_render_pass_device(ds.handle, C.byref(cam.pod), C.byref(o), first, count, acc.data_ptr(), lin.data_ptr(), rgb.data_ptr(), None, None))
    torch.cuda.synchronize()
    assert torch.equal(lin, ref) and torch.equal(rgb, ref8)


def test_stress_frame_full_size_front_ends_agree():
    """BASELINE's stress configuration (10 000 spheres, 3840x2160, 16 spp, depth 50) at full size: the streaming matrix-pipe
    loop, the streaming vector-ALU loop and the tree render the same frame bit for bit, with equal segment counts."""
    import torch

    lib = abi.load()
    sc, cam = presets.stress_scene(10000, 1), presets.stress_camera(3840, 16, 50)
    ds = sc.device_scene(0)
    H, W = cam.height, cam.width
    assert (H, W) == (2160, 3840)
    out = {}
    for name, flags in (("matrix", 0), ("valu", abi.FLAG_VALU_BROAD), ("tree", abi.FLAG_BVH)):
        lin = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        st = abi.MirtStats()
        o = cam.opts(precision=abi.F32, seed=1, flags=flags)
        abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), lin.data_ptr(), rgb.data_ptr(), None, C.byref(st)))
        out[name] = (lin, rgb, st.segments)
    assert lib.mirt_closest_hit_kind(ds.handle, C.byref(cam.pod), C.byref(cam.opts(precision=abi.F32, seed=1))) == 1
    for name in ("valu", "tree"):
        assert torch.equal(out[name][0], out["matrix"][0]) and torch.equal(out[name][1], out["matrix"][1]) and out[name][2] == out["matrix"][2], name
    assert out["matrix"][2] > H * W * 16


@pytest.mark.parametrize("n", [31623, 100000])
def test_large_stress_frames_full_size_tree_equals_matrix_pipe(n):
    """The upper end of BASELINE's stress configuration (31 623 and 100 000 spheres, 3840x2160, 16 spp, depth 50) at FULL size:
    the bounding-volume tree and the brute-force streaming matrix-pipe loop render the same frame bit for bit, with equal
    segment counts (1.0 s / 4.1 s of brute force per frame; the 10 s vector-ALU pass at 100 000 is left to the reduced-size test
    below)."""
    import torch

    lib = abi.load()
    sc, cam = presets.stress_scene(n, 1), presets.stress_camera(3840, 16, 50)
    ds = sc.device_scene(0)
    H, W = cam.height, cam.width
    assert (H, W) == (2160, 3840)
    out = {}
    for name, flags in (("tree", abi.FLAG_BVH), ("matrix", 0)):
        lin = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        st = abi.MirtStats()
        o = cam.opts(precision=abi.F32, seed=1, flags=flags)
        abi.check(lib.mirt_render_device(ds.handle, C.byref(cam.pod), C.byref(o), lin.data_ptr(), rgb.data_ptr(), None, C.byref(st)))
        out[name] = (lin, rgb, st.segments, st.tree_segments)
    assert lib.mirt_closest_hit_kind(ds.handle, C.byref(cam.pod), C.byref(cam.opts(precision=abi.F32, seed=1))) == 1
    assert torch.equal(out["tree"][0], out["matrix"][0]) and torch.equal(out["tree"][1], out["matrix"][1])
    assert out["tree"][2] == out["matrix"][2] > H * W * 16 and out["tree"][3] == out["tree"][2] and out["matrix"][3] == 0
    ds.close()
    sc._device_scene.clear()


def test_hundred_thousand_spheres_front_ends_agree():
    """The largest stress scene of BASELINE (100 000 spheres; brute force streams 3 MB of operands per wave and segment) at a
    reduced frame size: matrix-pipe loop, vector-ALU loop and tree give the same maps and the same rendered frame."""
    sc = presets.stress_scene(100000, 1)
    cam = presets.stress_camera(256, 0, 50)
    maps = {a: cam.primary_hits(sc, precision=abi.F32, accel=a) for a in (None, "valu", "bvh")}
    assert (maps[None][0] != abi.MISS_ID).mean() > 0.2
    for a in ("valu", "bvh"):
        assert np.array_equal(maps[a][0], maps[None][0]) and np.array_equal(maps[a][1].view(np.uint32), maps[None][1].view(np.uint32)), a
    rcam = presets.stress_camera(160, 2, 12)
    ref, ref8 = rcam.render_arrays(sc, precision=abi.F32, seed=3)
    for a in ("valu", "bvh"):
        lin, rgb8 = rcam.render_arrays(sc, precision=abi.F32, seed=3, accel=a)
        assert np.array_equal(lin.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgb8, ref8), a


@pytest.mark.parametrize("width,ratio", [(1, 1.0), (1, 1 / 7), (3, 3.0), (63, 21.0), (64, 64.0), (65, 65 / 3), (129, 64.5), (200, 0.9)])
def test_render_odd_image_sizes_vs_oracle(oracle, scenes, width, ratio):
    """Rows shorter than, equal to and just over a wave of the resolve kernel, single rows and single columns."""
    sc = scenes["balls"]
    cam = Camera.build((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), None, width, ratio, 5, math.radians(25.0), 10.0, math.radians(0.4), 8)
    for prec, tol in ((abi.F64, 1e-9), (abi.F32, 2e-5)):
        lin, rgb8 = cam.render_arrays(sc, precision=prec, seed=11)
        o_lin, o_rgb8, o_seg, _ = oracle.render(sc.flat, cam.pod, prec, seed=11, threads=4)
        assert lin.shape == o_lin.shape == (cam.height, cam.width, 3)
        err = np.abs(lin.astype(np.float64) - o_lin.astype(np.float64))
        assert float((err <= tol * (1.0 + np.abs(o_lin))).mean()) >= 0.99, (width, ratio, prec)
        assert abs(cam.last_stats["segments"] - o_seg) <= max(16, 0.002 * o_seg)


def test_many_samples_few_pixels_vs_oracle(oracle, scenes):
    """777 samples per pixel on a 12x7 image: many chunks per pixel, several passes of the queue order, a last pass that is
    not full; the sums are added in sample order on both sides."""
    sc = scenes["balls"]
    cam = Camera.build((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), None, 12, 12 / 7, 777, math.radians(25.0), 10.0, math.radians(0.4), 6)
    for prec, tol in ((abi.F64, 1e-9), (abi.F32, 2e-5)):
        lin, _ = cam.render_arrays(sc, precision=prec, seed=21)
        o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, prec, seed=21, threads=8)
        err = np.abs(lin.astype(np.float64) - o_lin.astype(np.float64))
        assert float((err <= tol * (1.0 + np.abs(o_lin))).mean()) >= 0.97 and err.max() < 0.02, (prec, err.max())
        assert abs(cam.last_stats["segments"] - o_seg) <= 0.002 * o_seg


@pytest.mark.parametrize("n", [641, 800, 1100])
def test_more_materials_than_the_lds_table_holds(oracle, n):
    """One material per sphere: beyond the LDS table (512 / 184 records by layout) the matrix-pipe kernel reads material records from HBM/L2 instead of
    its LDS copy (and beyond 1024 spheres everything streams); radiance against the oracle either way."""
    rng = np.random.default_rng(n)
    sc = Scene()
    sc.add(Sphere.new_static(300.0, (0.0, -300.0, 0.0), Lambertian.with_texture(CheckerTexture(0.7, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 0.8)))
    for k in range(n - 1):
        c = rng.uniform(-9, 9, 3) * np.array([1.0, 0.0, 1.0]) + np.array([0.0, 0.25, 0.0])
        kind = k % 3
        m = (Lambertian.new(tuple(rng.uniform(0.1, 0.9, 3)), float(rng.uniform(0.5, 1.0))) if kind == 0
             else Metal.new(tuple(rng.uniform(0.4, 1.0, 3)), float(rng.uniform(0.0, 0.5))) if kind == 1 else Dielectric.new(float(rng.uniform(1.2, 1.9))))
        sc.add(Sphere.new_static(0.25, tuple(c), m))
    sc.build_bvh()
    cam = Camera.build((12.0, 3.0, 4.0), (0.0, 0.0, 0.0), None, 96, 1.6, 6, math.radians(35.0), 12.0, math.radians(0.3), 10)
    assert cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    for prec, tol in ((abi.F64, 1e-9), (abi.F32, 2e-5)):
        o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, prec, seed=4, threads=8)
        for accel in (None, "valu"):
            lin, _ = cam.render_arrays(sc, precision=prec, seed=4, accel=accel)
            err = np.abs(lin.astype(np.float64) - o_lin.astype(np.float64))
            assert float((err <= tol * (1.0 + np.abs(o_lin))).mean()) >= 0.995, (prec, accel)
            assert abs(cam.last_stats["segments"] - o_seg) <= max(20, 0.002 * o_seg)


def test_render_empty_scene_is_the_sky(oracle):
    """No spheres: every path is one segment into the sky gradient (camera.rs:282-284), through every front end."""
    sc = Scene()
    sc.build_bvh()
    cam = _cam(w=40, ratio=1.5, spp=3, depth=5)
    for prec, tol in ((abi.F64, 1e-12), (abi.F32, 1e-6)):
        o_lin, o_rgb8, o_seg, _ = oracle.render(sc.flat, cam.pod, prec, seed=2, threads=2)
        assert o_seg == cam.height * cam.width * 3
        for accel in (None, "valu") + (("bvh",) if prec == abi.F32 else ()):
            lin, rgb8 = cam.render_arrays(sc, precision=prec, seed=2, accel=accel)
            assert np.abs(lin.astype(np.float64) - o_lin.astype(np.float64)).max() <= tol and np.array_equal(rgb8, o_rgb8), (prec, accel)
            assert cam.last_stats["segments"] == o_seg


def test_fuzz_partitions_and_passes():
    """A slice of tools/fuzz_partition.py (run over 5 400 cases): random sizes, sample counts, rank counts, strip heights, pass
    cuts and queue knobs - bands and passes assemble into the one-call frame bit for bit."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_partition.py"), "9000", "40"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and "mismatching frames: 0" in r.stdout, r.stdout[-800:] + r.stderr[-400:]
