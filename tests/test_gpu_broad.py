"""The two broad phases of the f32 brute-force kernel - vector ALU (MIRT_FLAG_VALU_BROAD) and matrix pipe (the default
for resident scenes inside the f16 range, split-f16 operands through v_mfma_f32_32x32x16_f16) - must select supersets of
the exact candidates, i.e. give the same (id, t) for every ray: maps bit-exact against the f32 oracle and against each
other, rendered images bit-identical, and the host must fall back to the vector ALU where the operands would not fit."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from raytracer_amd import abi, presets
from raytracer_amd.api import Camera, Dielectric, Lambertian, Metal, Scene, Sphere

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _matrix_pipe_for_small_scenes_too(monkeypatch):
    """Scenes of fewer than 24 spheres default to the vector ALU (cheaper than preparing a ray's operands); the tests here
    want the matrix pipe wherever the operands fit, so every mirt_opts they build carries MIRT_FLAG_MX_SMALL."""
    from raytracer_amd import api

    monkeypatch.setattr(api, "DEFAULT_FLAGS", abi.FLAG_MX_SMALL)


def bits(a):
    return a.view(np.uint32)


def _maps(oracle, sc, cam, expect="matrix", check_oracle=True):
    assert cam.closest_hit_kind(sc, precision=abi.F32) == expect
    assert cam.closest_hit_kind(sc, precision=abi.F32, accel="valu") in ("valu", "tiny")  # tiny: <= 8 slots, exact test on every sphere
    ids, t = cam.primary_hits(sc, precision=abi.F32)
    v_ids, v_t = cam.primary_hits(sc, precision=abi.F32, accel="valu")
    assert np.array_equal(ids, v_ids) and np.array_equal(bits(t), bits(v_t))
    if check_oracle:
        o_ids, o_t = oracle.primary_hits(sc.flat, cam.pod, abi.F32)
        assert np.array_equal(ids, o_ids) and np.array_equal(bits(t), bits(o_t))
    return ids


@pytest.mark.parametrize("width", [64, 400, 1920])
def test_matrix_and_valu_maps_bouncing_balls(oracle, width):
    """400 wide holds the two pixels that exposed a double f32 -> f16 conversion (rounding ties) in the first version."""
    ids = _maps(oracle, presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(width, 0, 50), check_oracle=width <= 400)
    assert (ids != abi.MISS_ID).any() and (ids == abi.MISS_ID).any()


def test_matrix_maps_presets(oracle):
    _maps(oracle, presets.mars_texture_scene(), presets.default_camera(400, 0, 50))
    _maps(oracle, presets.scene(), presets.plain_camera(320, 0, 10))
    _maps(oracle, presets.blur_scene(), presets.plain_camera(320, 0, 10))


@pytest.mark.parametrize("n,seed", [(31, 1), (32, 2), (33, 3), (500, 4), (1000, 5), (1024, 6)])
def test_matrix_maps_random_scenes(oracle, n, seed):
    """Tile boundaries (31 / 32 / 33 spheres) and the largest resident scene (1024)."""
    sc = presets.stress_scene(n, seed=seed)
    cam = presets.stress_camera(160, 0, 50)
    ids = _maps(oracle, sc, cam)
    assert len(np.unique(ids)) > min(n, 100) // 8


@pytest.mark.parametrize("n", [1025, 1100, 6000])
def test_larger_scenes_stream_their_operands(oracle, n):
    """More than 1024 spheres: the operands no longer fit the LDS copy and are read tile by tile through L1/L2."""
    sc = presets.stress_scene(n, seed=7)
    cam = presets.stress_camera(128, 0, 50)
    ids = _maps(oracle, sc, cam)
    assert len(np.unique(ids)) > 100


@pytest.mark.parametrize("n", [3000, 4000, 4200, 7000])
def test_streaming_render_is_bit_identical_on_both_broad_phases(n, oracle):
    """Either side of the size (4 096 sorted slots) from which the vector-ALU streaming loop takes a group's maximum before it
    collects signs and compares with the ray's k2 once per group: matrix pipe, vector ALU and the f32 oracle, frame and map."""
    sc, cam = presets.stress_scene(n, seed=5), presets.stress_camera(96, 4, 12)
    assert cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    a, a8 = cam.render_arrays(sc, precision=abi.F32, seed=4)
    seg = cam.last_stats["segments"]
    b, b8 = cam.render_arrays(sc, precision=abi.F32, seed=4, accel="valu")
    assert cam.last_stats["segments"] == seg
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(a8, b8)
    cam0 = presets.stress_camera(96, 0, 12)
    ids, t = cam0.primary_hits(sc, precision=abi.F32, accel="valu")
    o_ids, o_t = oracle.primary_hits(sc.flat, cam0.pod, abi.F32)
    assert np.array_equal(ids, o_ids) and np.array_equal(bits(t), bits(o_t))


def test_continuation_pass_frame_bands_and_passes_are_bit_identical():
    """Scenes beyond the LDS, launches of >= 64 items per lane of the chip: the waves that are left with paths when the queue is
    dry hand them to a second launch of the kernel (render_body, T::CONTINUES).  A frame large enough for that - whole, as two
    bands, and as two progressive passes - is the vector-ALU kernel's frame bit for bit, with the same number of segments."""
    import ctypes as C

    sc = presets.stress_scene(1500, seed=12)
    cam = presets.stress_camera(2560, 16, 50)  # 3.7 M pixels x 16 one-sample items = 225 per lane; a band or a pass: 112
    assert cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    a, a8 = cam.render_arrays(sc, precision=abi.F32, seed=9)
    seg = cam.last_stats["segments"]
    b, b8 = cam.render_arrays(sc, precision=abi.F32, seed=9, accel="valu")
    assert cam.last_stats["segments"] == seg
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(a8, b8)
    lib, ds = abi.load(), sc.device_scene(0)
    acc = np.zeros_like(a)
    segs = 0
    for r in range(2):
        part = np.zeros_like(a)
        o = cam.opts(precision=abi.F32, seed=9, strip_rows=8, part_index=r, part_count=2)
        st = abi.MirtStats()
        abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), part.ctypes.data, None, C.byref(st)))
        acc += part
        segs += st.segments
    assert segs == seg and np.array_equal(bits(acc), bits(a))
    g = cam.pass_granularity(precision=abi.F32)
    assert g <= 2
    accum = np.full(a.shape, np.nan, np.float32)
    for first, count in ((0, 8), (8, 8)):
        lin, _ = cam.render_pass(sc, first, count, accum, precision=abi.F32, seed=9)
    assert np.array_equal(bits(lin), bits(a))


def test_ring_kernel_bands_passes_and_tiny_frames():
    """Scenes beyond the LDS run render_kernel_f32_mx_ring, whose waves move through the tiles in step behind block barriers
    and leave the loop together.  Bands of a partition, progressive passes, a frame with fewer items than one block has
    lanes (most waves never get work) and a one-pixel frame all give the vector-ALU kernel's frame."""
    import ctypes as C

    sc = presets.stress_scene(1500, seed=11)
    lib, ds = abi.load(), sc.device_scene(0)
    cam = presets.stress_camera(72, 4, 10)
    full, full8 = cam.render_arrays(sc, precision=abi.F32, seed=6)
    seg = cam.last_stats["segments"]
    v, v8 = cam.render_arrays(sc, precision=abi.F32, seed=6, accel="valu")
    assert cam.last_stats["segments"] == seg and np.array_equal(bits(full), bits(v)) and np.array_equal(full8, v8)
    acc = np.zeros_like(full)
    for r in range(3):
        part = np.zeros_like(full)
        o = cam.opts(precision=abi.F32, seed=6, strip_rows=4, part_index=r, part_count=3)
        abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), part.ctypes.data, None, None))
        acc += part
    assert np.array_equal(bits(acc), bits(full))
    g = cam.pass_granularity(precision=abi.F32)
    accum = np.full(full.shape, np.nan, np.float32)
    for a, b in ((0, g), (g, 4)) if g < 4 else ((0, 4),):
        lin, _ = cam.render_pass(sc, a, b - a, accum, precision=abi.F32, seed=6)
    assert np.array_equal(bits(lin), bits(full))
    for w, spp in ((5, 1), (1, 0), (3, 7)):
        small = presets.stress_camera(w, spp, 6)
        a, _ = small.render_arrays(sc, precision=abi.F32, seed=2)
        b, _ = small.render_arrays(sc, precision=abi.F32, seed=2, accel="valu")
        assert np.array_equal(bits(a), bits(b))


def _moving_mix(scale, n=300, seed=9):
    rng = np.random.default_rng(seed)
    sc = Scene()
    mats = [Lambertian.new((0.5, 0.5, 0.5), 1.0), Metal.new((0.8, 0.8, 0.8), 0.1), Dielectric.new(1.5)]
    sc.add(Sphere.new_static(1000.0 * scale, (0.0, -1000.0 * scale, 0.0), mats[0]))
    for k in range(n):
        c = rng.uniform(-8, 8, 3) * scale
        c[1] = abs(c[1]) * 0.3 + 0.2 * scale
        r = float(rng.uniform(0.05, 0.4)) * scale
        if k % 3 == 0:
            c1 = c + rng.uniform(-0.5, 0.5, 3) * scale
            sc.add(Sphere.new(r, tuple(c), tuple(c1), mats[k % 3]))
        else:
            sc.add(Sphere.new_static(r, tuple(c), mats[k % 3]))
    sc.build_bvh()
    cam = Camera.build((13.0 * scale, 2.0 * scale, 3.0 * scale), (0.0, 0.0, 0.0), None, 240, 16.0 / 9.0, 0, 0.35, None, None, 10)
    return sc, cam


@pytest.mark.parametrize("scale", [1.0, 2.0 ** -10, 1e-3, 1e3, 3e4])
def test_matrix_maps_are_scale_free(oracle, scale):
    """The operands live in the scene's own power-of-two scale: tiny and huge scenes behave like the unit-sized one."""
    sc, cam = _moving_mix(scale)
    ids = _maps(oracle, sc, cam)
    assert len(np.unique(ids)) > 50


@pytest.mark.parametrize("ground,r,expect", [(1e3, 0.02, "matrix"), (1e5, 0.01, "valu")])
def test_small_spheres_next_to_a_huge_one(oracle, ground, r, expect):
    """Dynamic range: the absolute part of the split-f16 margin (f16 subnormal spacing in the scene scale) against r^2.
    Radius 0.02 on a radius-1e3 ground still runs on the matrix pipe; radius 0.01 on a radius-1e5 ground would make every
    sphere a candidate for every nearby ray, so the host keeps that scene on the vector ALU."""
    rng = np.random.default_rng(5)
    sc = Scene()
    m = Lambertian.new((0.5, 0.5, 0.5), 1.0)
    sc.add(Sphere.new_static(ground, (0.0, -ground, 0.0), m))
    for k in range(400):
        c = rng.uniform(-1, 1, 3)
        sc.add(Sphere.new_static(r, (float(c[0]), r, float(c[2])), m))
    sc.build_bvh()
    cam = Camera.build((2.0, 0.6, 1.0), (0.0, 0.0, 0.0), None, 400, 1.5, 0, 0.9, None, None, 10)
    ids = _maps(oracle, sc, cam, expect=expect)
    assert len(np.unique(ids)) > 200


def test_far_cameras_rescale_and_very_far_ones_fall_back(oracle):
    """Rays start on spheres or at the lens, so the operand scale follows the camera position: a camera several scene
    radii away still runs on the matrix pipe (coarser scale), one so far away that the absolute part of the margin
    swamps r^2 for most spheres stays on the vector ALU."""
    rng = np.random.default_rng(2)
    sc = Scene()
    m = Metal.new((0.8, 0.8, 0.8), None)
    for k in range(200):
        c = rng.uniform(-30, 30, 3)
        sc.add(Sphere.new_static(float(rng.uniform(0.5, 2.0)), tuple(c), m))
    sc.build_bvh()
    near = Camera.build((150.0, 40.0, 30.0), (0.0, 0.0, 0.0), None, 200, 1.0, 0, 0.5, None, None, 5)
    far = Camera.build((4000.0, 400.0, 300.0), (0.0, 0.0, 0.0), None, 200, 1.0, 0, 0.02, 4030.0, None, 5)
    very_far = Camera.build((4.0e5, 4.0e4, 3.0e4), (0.0, 0.0, 0.0), None, 200, 1.0, 0, 2.5e-4, 4.03e5, None, 5)
    assert len(np.unique(_maps(oracle, sc, near, expect="matrix"))) > 50
    assert len(np.unique(_maps(oracle, sc, far, expect="matrix"))) > 50
    assert len(np.unique(_maps(oracle, sc, very_far, expect="valu"))) > 5


def test_off_origin_cluster_of_small_spheres_falls_back(oracle):
    """Small spheres far from the coordinate origin: the absolute part of the margin grows with |M| (f16 subnormal lo parts)
    and would exceed r^2 here, so the host keeps the scene on the vector ALU; ten times larger spheres use the matrix pipe."""
    rng = np.random.default_rng(11)
    sc = Scene()
    for k in range(200):
        c = rng.uniform(-3, 3, 3) + np.array([500.0, 20.0, -300.0])
        sc.add(Sphere.new_static(float(rng.uniform(0.02, 0.1)), tuple(c), Metal.new((0.8, 0.8, 0.8), None)))
    sc.build_bvh()
    cam = Camera.build((480.0, 25.0, -290.0), (500.0, 20.0, -300.0), None, 200, 1.0, 0, 0.35, None, None, 5)
    ids = _maps(oracle, sc, cam, expect="valu")
    assert len(np.unique(ids)) > 50
    sc = Scene()
    for k in range(200):
        c = rng.uniform(-30, 30, 3) + np.array([500.0, 20.0, -300.0])
        sc.add(Sphere.new_static(float(rng.uniform(0.5, 1.0)), tuple(c), Metal.new((0.8, 0.8, 0.8), None)))
    sc.build_bvh()
    cam = Camera.build((400.0, 45.0, -250.0), (500.0, 20.0, -300.0), None, 200, 1.0, 0, 0.5, None, None, 5)
    ids = _maps(oracle, sc, cam, expect="matrix")
    assert len(np.unique(ids)) > 50


@pytest.mark.parametrize("name", ["balls", "mars", "mix"])
def test_render_is_bit_identical_on_both_broad_phases(name):
    if name == "balls":
        sc, cam = presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(160, 8, 50)
    elif name == "mars":
        sc, cam = presets.mars_texture_scene(), presets.default_camera(160, 8, 20)
    else:
        sc, cam = _moving_mix(1.0)
        cam = Camera.build((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), None, 128, 16.0 / 9.0, 6, 0.35, 10.0, 0.02, 30)
    assert cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    a, a8 = cam.render_arrays(sc, precision=abi.F32, seed=4)
    st_a = dict(cam.last_stats)
    b, b8 = cam.render_arrays(sc, precision=abi.F32, seed=4, accel="valu")
    st_b = dict(cam.last_stats)
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(a8, b8)
    assert st_a["segments"] == st_b["segments"] and st_a["sphere_tests"] == st_b["sphere_tests"]
    # the matrix pipe's margins are a little wider: a few more candidates, never fewer hits
    assert st_a["narrow_tests"] <= 1.1 * st_b["narrow_tests"] + 64


@pytest.mark.parametrize("n,n_mats", [(300, 6), (300, 300), (600, 6), (600, 600)])
def test_material_records_from_lds_and_from_hbm(n, n_mats):
    """The matrix-pipe kernels have one body for material records resident in LDS and one for records in HBM / L2, picked per
    launch by the number of materials (512 fit the layout for scenes of up to 512 sorted slots, 155 that for up to 1024):
    few shared materials and one material per sphere, either side of the layout boundary - every frame equal to the
    vector-ALU kernel's, which fetches its records another way."""
    rng = np.random.Generator(np.random.Philox(n + n_mats))
    mats = []
    for k in range(n_mats):
        u = rng.random(5)
        mats.append(Lambertian.new(tuple(0.2 + 0.8 * u[:3]), 0.9) if k % 3 == 0 else Metal.new(tuple(0.5 + 0.5 * u[:3]), float(0.4 * u[3])) if k % 3 == 1 else Dielectric.new(1.3 + 0.4 * float(u[4])))
    sc = Scene()
    c = rng.random((n, 3))
    for i in range(n):
        sc.add(Sphere.new_static(0.15 + 0.2 * float(rng.random()), (-8.0 + 16.0 * c[i, 0], 0.2 + 3.0 * c[i, 1], -8.0 + 16.0 * c[i, 2]), mats[i % n_mats]))
    sc.build_bvh()
    cam = Camera.build((13.0, 4.0, 3.0), (0.0, 1.0, 0.0), None, 160, 16.0 / 9.0, 6, 0.5, 10.0, 0.0, 20)
    assert cam.closest_hit_kind(sc, precision=abi.F32) == "matrix"
    a, a8 = cam.render_arrays(sc, precision=abi.F32, seed=11)
    b, b8 = cam.render_arrays(sc, precision=abi.F32, seed=11, accel="valu")
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(a8, b8)
    assert len(np.unique(a8.reshape(-1, 3), axis=0)) > 100  # a picture, not a constant


def test_tiny_scenes_default_to_the_vector_alu(monkeypatch):
    from raytracer_amd import api

    monkeypatch.setattr(api, "DEFAULT_FLAGS", 0)
    # 1-3 spheres: no broad phase at all (render_kernel_f32_tiny) - the kind is reported so that a change of the slot padding
    # cannot silently send these scenes through the 16-record groups of the general kernel again (it did, for one r02 build)
    assert presets.default_camera(64, 0, 5).closest_hit_kind(presets.mars_texture_scene(), precision=abi.F32) == "tiny"
    assert presets.textured_camera(64, 1, 5).closest_hit_kind(presets.textured_scene(), precision=abi.F32) == "tiny"
    sc9 = presets.stress_scene(9, 1)
    assert presets.stress_camera(64, 0, 5).closest_hit_kind(sc9, precision=abi.F32) == "valu"
    assert presets.bouncing_balls_camera(64, 0, 5).closest_hit_kind(presets.bouncing_balls_scene(1), precision=abi.F32) == "matrix"


def test_inside_a_sphere_and_degenerate_inputs(oracle):
    sc = Scene()
    m = Dielectric.new(1.5)
    sc.add(Sphere.new_static(5.0, (0.0, 0.0, 0.0), m))       # camera inside
    sc.add(Sphere.new_static(0.0, (0.0, 0.0, -2.0), m))      # zero radius
    sc.add(Sphere.new_static(-1.0, (1.0, 0.0, -2.0), m))     # negative radius clamps to zero (sphere.rs:31)
    sc.add(Sphere.new(0.3, (-1.0, 0.0, -2.0), (-1.0, 0.0, -2.0), m))  # "moving" without travel
    for k in range(6):  # (degenerate spheres count as "too small for the f16 margins": keep them a minority)
        sc.add(Sphere.new_static(0.2, (-1.5 + 0.6 * k, 1.0, -3.0), m))
    sc.build_bvh()
    ids = _maps(oracle, sc, Camera.build((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), None, 96, 1.0, 0, None, None, None, 10))
    assert 0 in ids and 3 in ids


# ---- the f64 parity mode with the (f32) broad phase on the matrix pipe ---------------------------------------------------
def _maps64(oracle, sc, cam, expect="f64-matrix", check_oracle=True):
    assert cam.closest_hit_kind(sc, precision=abi.F64) == expect
    assert cam.closest_hit_kind(sc, precision=abi.F64, accel="valu") == "f64"
    ids, t = cam.primary_hits(sc, precision=abi.F64)
    v_ids, v_t = cam.primary_hits(sc, precision=abi.F64, accel="valu")
    assert np.array_equal(ids, v_ids) and np.array_equal(t.view(np.uint64), v_t.view(np.uint64))
    if check_oracle:
        o_ids, o_t = oracle.primary_hits(sc.flat, cam.pod, abi.F64)
        assert np.array_equal(ids, o_ids) and np.array_equal(t.view(np.uint64), o_t.view(np.uint64))
    return ids


@pytest.mark.parametrize("width", [400, 1920])
def test_f64_matrix_maps_bouncing_balls(oracle, width):
    ids = _maps64(oracle, presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(width, 0, 50), check_oracle=width <= 400)
    assert (ids != abi.MISS_ID).any() and (ids == abi.MISS_ID).any()


@pytest.mark.parametrize("scale", [1.0, 1e-3, 3e4])
def test_f64_matrix_maps_are_scale_free(oracle, scale):
    sc, cam = _moving_mix(scale)
    assert len(np.unique(_maps64(oracle, sc, cam))) > 50


def test_f64_matrix_maps_with_centres_that_f32_cannot_hold(oracle):
    """Centres and radii with bits below f32's last place: the balls behind the operands are built around the f64 spheres
    (the f32-rounded centre plus the rounding distance), so silhouettes that differ between the two precisions stay exact."""
    rng = np.random.default_rng(21)
    sc = Scene()
    m = Lambertian.new((0.5, 0.5, 0.5), 1.0)
    for k in range(300):
        c = rng.uniform(-40, 40, 3) + rng.uniform(-1e-7, 1e-7, 3)
        r = float(rng.uniform(0.3, 1.5)) * (1.0 + 1e-9 * k)
        if k % 4 == 0:
            sc.add(Sphere.new(r, tuple(c), tuple(c + rng.uniform(-1, 1, 3)), m))
        else:
            sc.add(Sphere.new_static(r, tuple(c), m))
    sc.build_bvh()
    cam = Camera.build((90.0 + 1e-9, 30.0, 20.0), (0.0, 0.0, 0.0), None, 300, 1.0, 0, 0.8, None, None, 5)
    assert len(np.unique(_maps64(oracle, sc, cam))) > 100


def test_f64_matrix_streams_large_scenes(oracle):
    sc = presets.stress_scene(1500, seed=3)
    assert len(np.unique(_maps64(oracle, sc, presets.stress_camera(128, 0, 50)))) > 100


@pytest.mark.parametrize("name", ["balls", "mix"])
def test_f64_render_is_bit_identical_on_both_broad_phases(name):
    if name == "balls":
        sc, cam = presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(128, 4, 50)
    else:
        sc, _ = _moving_mix(1.0)
        cam = Camera.build((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), None, 96, 16.0 / 9.0, 4, 0.35, 10.0, 0.02, 30)
    assert cam.closest_hit_kind(sc, precision=abi.F64) == "f64-matrix"
    a, a8 = cam.render_arrays(sc, precision=abi.F64, seed=4)
    seg = cam.last_stats["segments"]
    b, b8 = cam.render_arrays(sc, precision=abi.F64, seed=4, accel="valu")
    assert cam.last_stats["segments"] == seg
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) and np.array_equal(a8, b8)


def test_fuzz_all_front_ends_against_the_oracle():
    """A slice of tools/fuzz_maps.py (which was run over 360 scenes): random scenes, scales and cameras; every front end in
    both precisions against the oracle."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_maps.py"), "4000", "8"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and "mismatching maps: 0" in r.stdout, r.stdout[-800:] + r.stderr[-400:]


# ---- round 3: the tile cull of the matrix-pipe kernels ------------------------------------------------------------------------
def _cull_scenes():
    rng = np.random.default_rng(5)
    mats = [Lambertian.new((0.6, 0.5, 0.4), 0.9), Metal.new((0.8, 0.8, 0.9), 0.1), Dielectric.new(1.5)]
    out = [("bench scene", presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(192, 4, 30))]
    # two far-apart clusters and a huge ground: most tiles are out of every ray's way
    sc = Scene()
    sc.add(Sphere.new_static(2000.0, (0.0, -2000.0, 0.0), mats[0]))
    for c0 in ((-40.0, 1.0, 0.0), (40.0, 1.0, 5.0)):
        for k in range(150):
            c = np.array(c0) + rng.uniform(-6, 6, 3) * np.array([1.0, 0.1, 1.0])
            m = mats[k % 3]
            if k % 4 == 0:
                sc.add(Sphere.new(0.3, tuple(c), tuple(c + rng.uniform(-0.3, 0.3, 3)), m))
            else:
                sc.add(Sphere.new_static(0.3, tuple(c), m))
    sc.build_bvh()
    out.append(("two clusters", sc, Camera.build((0.0, 6.0, 60.0), (0.0, 0.0, 0.0), None, 160, 2.0, 3, math.radians(70.0), 60.0, 0.0, 20)))
    # 700 spheres: the 1024-slot layout (one ball per tile); everything in one spot: nothing to cull
    sc = Scene()
    for k in range(700):
        c = rng.uniform(-1, 1, 3)
        sc.add(Sphere.new_static(0.05, tuple(c), mats[k % 3]))
    sc.build_bvh()
    out.append(("one spot", sc, Camera.build((0.0, 0.5, 5.0), (0.0, 0.0, 0.0), None, 96, 1.0, 3, math.radians(40.0), 5.0, 0.0, 12)))
    # a static sphere whose centre has a -0.0 component keeps the scene in two sections (statics, movings)
    sc = Scene()
    sc.add(Sphere.new_static(1.0, (-0.0, 0.0, -0.0), mats[1]))
    for k in range(80):
        c = rng.uniform(-8, 8, 3)
        sc.add(Sphere.new(0.4, tuple(c), tuple(c + rng.uniform(-0.5, 0.5, 3)), mats[k % 3]))
    sc.build_bvh()
    out.append(("negative zero", sc, Camera.build((0.0, 2.0, 20.0), (0.0, 0.0, 0.0), None, 96, 1.0, 3, math.radians(50.0), 20.0, 0.0, 12)))
    # beyond the LDS: every wave streams its own tiles, one top tile per 32 tiles (3 000 spheres: 94 tiles, 3 top tiles)
    out.append(("stress 3000", presets.stress_scene(3000, 2), presets.stress_camera(160, 3, 12)))
    # the same with a ground, statics and movings in their two sections (a tile at the seam holds both kinds)
    sc = Scene()
    sc.add(Sphere.new_static(500.0, (0.0, -500.0, 0.0), mats[0]))
    for k in range(1500):
        c = rng.uniform(-30, 30, 3) * np.array([1.0, 0.0, 1.0]) + np.array([0.0, 0.3, 0.0])
        if k % 3 == 0:
            sc.add(Sphere.new(0.3, tuple(c), tuple(c + np.array([0.0, rng.uniform(0, 0.6), 0.0])), mats[k % 2]))
        else:
            sc.add(Sphere.new_static(0.3, tuple(c), mats[k % 3]))
    sc.build_bvh()
    out.append(("field of 1500", sc, Camera.build((0.0, 8.0, 40.0), (0.0, 0.0, 0.0), None, 128, 1.6, 3, math.radians(40.0), 40.0, math.radians(0.3), 16)))
    return out


@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
def test_tile_cull_changes_nothing_but_the_work(oracle, prec):
    """Round 3: the matrix-pipe kernels test a ball around every (half) tile of spatially ordered spheres first and skip the
    tiles no ray of a wave can touch.  Against MIRT_FLAG_ALL_PAIRS (round 2's loop): the same (id, t) for every primary ray, the
    same frame bit for bit, fewer tiles evaluated; and the culled frame is the oracle's (f32: bit for bit)."""
    lib = abi.load()
    for name, sc, cam in _cull_scenes():
        assert cam.closest_hit_kind(sc, precision=prec) in ("matrix", "f64-matrix"), name
        ds = sc.device_scene(0)
        res = {}
        for flags in (0, abi.FLAG_ALL_PAIRS):
            H, W = cam.height, cam.width
            lin = np.empty((H, W, 3), np.float64 if prec == abi.F64 else np.float32)
            st = abi.MirtStats()
            o = cam.opts(precision=prec, seed=3, flags=flags)
            abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), lin.ctypes.data, None, C.byref(st)))
            ids = np.empty((H, W), np.uint32)
            t = np.empty((H, W), lin.dtype)
            cam0 = Camera(cam.pod)
            abi.check(lib.mirt_primary_hits(ds.handle, C.byref(cam0.pod), C.byref(o), ids.ctypes.data, t.ctypes.data))
            res[flags] = (lin, ids, t, st.mx_tiles, st.wave_iterations, st.segments)
        a, b = res[0], res[abi.FLAG_ALL_PAIRS]
        assert np.array_equal(a[0].view(np.uint8), b[0].view(np.uint8)), name
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint8), b[2].view(np.uint8)), name
        assert a[5] == b[5] and a[4] > 0
        ntiles = (sc.flat.n_spheres + 31) // 32
        # all pairs: every wave-iteration that goes through the tile loop evaluates every tile (waves down to a few live rays
        # take trace_few instead: no tiles)
        assert b[3] % ntiles == 0 and 0 < b[3] <= b[4] * ntiles, (name, b[3], b[4], ntiles)
        ntop = 1 if ntiles <= 32 else (ntiles + 31) // 32
        assert a[3] <= b[3] + (b[3] // ntiles) * ntop, name        # the cull adds its top tile(s) and drops what it can
        print(f"{name}: tiles per wave-iteration {a[3] / a[4]:.2f} of {ntiles} (+1)")
        if name in ("bench scene", "two clusters", "stress 3000", "field of 1500"):
            assert a[3] < 0.8 * b[3], (name, a[3], b[3])
        if prec == abi.F32:
            o_lin, _, o_seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=3, threads=8, spec_chunk=cam.pass_granularity(precision=abi.F32))
            assert o_seg == a[5] and np.array_equal(a[0].view(np.uint32), o_lin.view(np.uint32)), name


@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
def test_tile_cull_lanes_that_keep_every_record(oracle, prec):
    """A ray the matrix-pipe filter cannot judge (NaN; an origin outside the operand range) is marked "every record is a
    candidate" by its own operands.  Such a lane passes every row of a top tile - also the empty rows behind the last half tile,
    which stand for no spheres at all - so the loop over the needed half tiles must stop at the last real one.  A camera whose
    pixel grid has a NaN component makes every ray such a lane: the culled loop, MIRT_FLAG_ALL_PAIRS, the vector-ALU kernel
    and the oracle must agree on every hit and on the frame (bench scene and a scene of 10 tiles: the packed loop over half
    tiles; 2 500 spheres: beyond the LDS).  What the exact test makes of a NaN ray is the specification's business (its max / min drop NaNs); this test only
    holds the front ends to each other and to the oracle."""
    lib = abi.load()
    two_clusters = _cull_scenes()[1]  # 10 tiles: rows 20..31 of its top tile are empty
    for sc, cam in ((presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(96, 2, 8)), two_clusters[1:], (presets.stress_scene(2500, 3), presets.stress_camera(96, 2, 8))):
        ds = sc.device_scene(0)
        bad = Camera(cam.pod)
        bad.pod.px00_loc[1] = float("nan")
        assert bad.closest_hit_kind(sc, precision=prec) in ("matrix", "f64-matrix")
        H, W = bad.height, bad.width
        o_ids, o_t = oracle.primary_hits(sc.flat, bad.pod, prec)
        res = {}
        for flags in (0, abi.FLAG_ALL_PAIRS, abi.FLAG_VALU_BROAD):
            o = bad.opts(precision=prec, seed=4, flags=flags)
            ids = np.full((H, W), 7, np.uint32)
            t = np.empty((H, W), np.float64 if prec == abi.F64 else np.float32)
            abi.check(lib.mirt_primary_hits(ds.handle, C.byref(bad.pod), C.byref(o), ids.ctypes.data, t.ctypes.data))
            lin = np.empty((H, W, 3), t.dtype)
            st = abi.MirtStats()
            abi.check(lib.mirt_render(ds.handle, C.byref(bad.pod), C.byref(o), lin.ctypes.data, None, C.byref(st)))
            res[flags] = (ids, t, lin, st.segments)
            assert np.array_equal(ids, o_ids), flags
            assert np.array_equal(t[ids != 0xFFFFFFFF], o_t[ids != 0xFFFFFFFF]), flags
        for flags in (abi.FLAG_ALL_PAIRS, abi.FLAG_VALU_BROAD):
            assert res[0][3] == res[flags][3]
            assert np.array_equal(res[0][2].view(np.uint8), res[flags][2].view(np.uint8)), flags


@pytest.mark.parametrize("prec", [abi.F32, abi.F64])
def test_drift_and_directly_tested_spheres(oracle, prec):
    """Round 3: the matrix-pipe filter takes the common part of the spheres' shutter motion out of its balls (the ray's origin
    moves instead) and leaves spheres far larger than the rest of the scene to an exact test by every ray.  Scenes built to lean
    on both: (a) a floor AND a wall of radius 3 000 / 2 500 (two directly tested spheres) under 220 small spheres that all move
    along one diagonal; (b) no giant at all, half the spheres static, the others moving in opposite directions (the midrange of
    the displacements is then 0 in one axis and the drift only half helps); (c) a MOVING giant.  Hit maps against the oracle and
    the vector-ALU kernel, frames against the vector-ALU kernel and MIRT_FLAG_ALL_PAIRS, bit for bit."""
    rng = np.random.default_rng(9)
    mats = [Lambertian.new((0.7, 0.6, 0.5), 0.9), Metal.new((0.9, 0.8, 0.7), 0.05), Dielectric.new(1.5)]
    scenes = []
    sc = Scene()
    sc.add(Sphere.new_static(3000.0, (0.0, -3000.0, 0.0), mats[0]))
    sc.add(Sphere.new_static(2500.0, (0.0, 0.0, -2510.0), mats[1]))
    for k in range(220):
        c = rng.uniform(-9, 9, 3) * np.array([1.0, 0.0, 1.0]) + np.array([0.0, 0.25, 0.0])
        sc.add(Sphere.new(0.25, tuple(c), tuple(c + np.array([1.0, 0.5, 0.0]) * rng.uniform(0.0, 0.8)), mats[k % 3]))
    sc.build_bvh()
    scenes.append(("two giants, one diagonal", sc, Camera.build((10.0, 4.0, 14.0), (0.0, 0.5, 0.0), None, 160, 1.6, 3, math.radians(45.0), 16.0, math.radians(0.3), 12)))
    sc = Scene()
    for k in range(300):
        c = rng.uniform(-7, 7, 3)
        if k % 2:
            sc.add(Sphere.new_static(0.3, tuple(c), mats[k % 3]))
        else:
            sc.add(Sphere.new(0.3, tuple(c), tuple(c + np.array([0.0, 1.0 if k % 4 else -1.0, 0.3]) * rng.uniform(0.2, 0.9)), mats[k % 3]))
    sc.build_bvh()
    scenes.append(("opposite motions", sc, Camera.build((0.0, 3.0, 22.0), (0.0, 0.0, 0.0), None, 128, 1.0, 3, math.radians(40.0), 22.0, 0.0, 12)))
    sc = Scene()
    sc.add(Sphere.new(1500.0, (0.0, -1500.0, 0.0), (0.0, -1500.5, 0.0), mats[0]))
    for k in range(150):
        c = rng.uniform(-6, 6, 3) * np.array([1.0, 0.0, 1.0]) + np.array([0.0, 0.3, 0.0])
        sc.add(Sphere.new(0.3, tuple(c), tuple(c + np.array([0.0, 0.4, 0.0])), mats[k % 3]))
    sc.build_bvh()
    scenes.append(("a moving giant", sc, Camera.build((0.0, 5.0, 15.0), (0.0, 0.0, 0.0), None, 128, 1.0, 3, math.radians(45.0), 15.0, 0.0, 12)))
    lib = abi.load()
    for name, sc, cam in scenes:
        assert cam.closest_hit_kind(sc, precision=prec) in ("matrix", "f64-matrix"), name
        ds = sc.device_scene(0)
        H, W = cam.height, cam.width
        o_ids, o_t = oracle.primary_hits(sc.flat, cam.pod, prec)
        res = {}
        for flags in (0, abi.FLAG_ALL_PAIRS, abi.FLAG_VALU_BROAD):
            o = cam.opts(precision=prec, seed=6, flags=flags)
            ids = np.empty((H, W), np.uint32)
            t = np.empty((H, W), np.float64 if prec == abi.F64 else np.float32)
            abi.check(lib.mirt_primary_hits(ds.handle, C.byref(cam.pod), C.byref(o), ids.ctypes.data, t.ctypes.data))
            lin = np.empty((H, W, 3), t.dtype)
            st = abi.MirtStats()
            abi.check(lib.mirt_render(ds.handle, C.byref(cam.pod), C.byref(o), lin.ctypes.data, None, C.byref(st)))
            res[flags] = (lin, st.segments)
            assert np.array_equal(ids, o_ids), (name, flags)
            assert np.array_equal(t[ids != 0xFFFFFFFF], o_t[ids != 0xFFFFFFFF]), (name, flags)
        for flags in (abi.FLAG_ALL_PAIRS, abi.FLAG_VALU_BROAD):
            assert res[0][1] == res[flags][1], (name, flags)
            assert np.array_equal(res[0][0].view(np.uint8), res[flags][0].view(np.uint8)), (name, flags)
