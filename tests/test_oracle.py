"""CPU tests of the oracle (oracle/oracle.cpp): known answers, the reference's single unit test, the independent
Python restatement, BVH-walk == brute-force, and the committed golden maps.  No GPU."""
import math

import numpy as np
import pytest

from oracle import oracle_np
from raytracer_amd import abi, presets
from raytracer_amd.api import Camera, Dielectric, Lambertian, Metal, Scene, Sphere, flatten, CheckerTexture, ImageTexture


# ---- RNG: Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution) ----------
PHILOX_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_kat(oracle, ctr, key, want):
    assert oracle.philox(ctr, key) == want


def test_draw_contract(oracle):
    # f64: dim d = 64-bit word pair d%2 of block d/2, (u64 >> 11) * 2^-53
    # f32: block 0 serves dims 0..4: dims 0..3 = word d >> 8, dim 4 = the low bytes of words 0..2
    for d in range(6):
        o = oracle.philox([5, 6, 7, d // 2], [9, 0])
        u = (o[0] | (o[1] << 32)) if d % 2 == 0 else (o[2] | (o[3] << 32))
        assert oracle.draw_dim(9, 5, 6, 7, d, abi.F64) == (u >> 11) * 2.0**-53
    o = oracle.philox([5, 6, 7, 0], [9, 0])
    for d in range(4):
        assert oracle.draw_dim(9, 5, 6, 7, d, abi.F32) == (o[d] >> 8) * 2.0**-24
    assert oracle.draw_dim(9, 5, 6, 7, 4, abi.F32) == ((o[0] & 0xFF) | ((o[1] & 0xFF) << 8) | ((o[2] & 0xFF) << 16)) * 2.0**-24
    o1 = oracle.philox([5, 6, 7, 1], [9, 0])
    assert oracle.draw_dim(9, 5, 6, 7, 5, abi.F32) == (o1[0] >> 8) * 2.0**-24
    vals = [oracle.draw_dim(3, 1, s, 0, d, p) for s in range(50) for d in range(5) for p in (abi.F64, abi.F32)]
    assert all(0.0 <= v < 1.0 for v in vals) and 0.4 < sum(vals) / len(vals) < 0.6


# ---- the reference's only unit test: src/scene/aabb.rs:123-137 ---------------------------------------------------
def test_reference_aabb_unit_test(oracle):
    assert oracle.aabb_slab((0.0, 0.0, 0.0), (2.0, 2.0, 2.0), (1.0, 1.0, -1.0), (0.0, 0.0, 1.0), 2, 0.0, float("inf"))
    # and the complementary case: the slab behind the ray is rejected
    assert not oracle.aabb_slab((0.0, 0.0, 0.0), (2.0, 2.0, 2.0), (1.0, 1.0, 3.0), (0.0, 0.0, 1.0), 2, 0.0, float("inf"))


# ---- camera (a1) ----------------------------------------------------------------------------------------------------
def test_camera_kat(oracle):
    # default camera of main.rs:30-56
    c = oracle.camera_build((0, 0, 12), (0, 0, 0), (0, 1, 0), 400, 16.0 / 9.0, 100, math.radians(25.0), 10.0, math.radians(0.4), 50)
    assert (c.img_width, c.img_height, c.samples_per_pixel, c.max_bounce_depth) == (400, 225, 100, 50)
    h = math.tan(math.radians(25.0) / 2.0)
    assert h == 0.22169466264293988
    vp_h = 2.0 * h * 10.0
    assert vp_h == 4.433893252858797
    assert vp_h * (400.0 / 225.0) == 7.882476893971194
    # px_du = u*vp_w/W with u = (1,0,0) here
    assert c.px_du[0] == 7.882476893971194 / 400.0 and c.px_du[1] == 0.0
    assert c.px_dv[1] == -4.433893252858797 / 225.0
    assert 10.0 * math.tan(math.radians(0.4) * 0.5) == 0.034906726815962505
    assert c.disk_u_r[0] == 0.034906726815962505


def _cam_fields(c):
    return [tuple(getattr(c, f)) if hasattr(getattr(c, f), "__len__") else getattr(c, f) for f, _ in abi.MirtCamera._fields_]


@pytest.mark.parametrize(
    "args",
    [
        ((0, 0, 12), (0, 0, 0), (0, 1, 0), 400, 16.0 / 9.0, 100, math.radians(25.0), 10.0, math.radians(0.4), 50),
        ((13, 2, 3), (0, 0, 0), (0, 1, 0), 1920, 16.0 / 9.0, 64, math.radians(25.0), 10.0, math.radians(0.4), 50),
        (None, None, None, 7, 3.0, None, None, None, None, None),  # all defaults: 100 spp, depth 10, vfov pi/2
        ((1, 2, 3), (-1, 0.5, 2), (0.1, 1, 0.2), 33, 0.01, 0, 1.0, 2.5, 0.0, 3),  # Some(0): AA off; tall image
        ((0, 0, 0), (0, 0, -1), (0, 1, 0), 2, 5.0, 1, 0.5, 1.0, 0.1, 1),  # W/ratio < 1 -> height 1
    ],
)
def test_host_camera_matches_oracle(oracle, args):
    """api.Camera.build (host product code) and oracle_camera_build are two restatements of camera.rs:123-206."""
    a = Camera.build(*args).pod
    b = oracle.camera_build(*args)
    assert _cam_fields(a) == _cam_fields(b)


def test_camera_unnormalised_u(oracle):
    # quirk: u = vup x w is not normalised (camera.rs:164): |u| = sin(angle(vup, w)) = 0.98895.. for (13,2,3)
    c = oracle.camera_build((13, 2, 3), (0, 0, 0), (0, 1, 0), 400, 16.0 / 9.0, 0, math.radians(25.0), 10.0, 0.0, 50)
    vp_w = 2.0 * math.tan(math.radians(25.0) / 2.0) * 10.0 * (400.0 / 225.0)
    ulen = math.sqrt(sum(x * x for x in c.px_du)) * 400.0 / vp_w
    assert abs(ulen - 0.98895) < 1e-5


# ---- sphere hit (a9, a10) -------------------------------------------------------------------------------------------
def _sphere(r=1.0, c0=(0, 0, 0), c1=None):
    s = abi.MirtSphere()
    s.center0[:] = c0
    s.center1[:] = c1 if c1 is not None else c0
    s.radius = r
    s.moving = 0 if c1 is None else 1
    return s


def test_sphere_hit_roots(oracle):
    s = _sphere(1.0, (0, 0, -5))
    h = oracle.sphere_hit(s, (0, 0, 0), (0, 0, -1))
    assert h["t"] == 4.0 and h["front"] and h["n"] == (0.0, 0.0, 1.0) and h["p"] == (0.0, 0.0, -4.0)
    # from inside: near root negative -> far root, Inside face, normal flipped
    h = oracle.sphere_hit(s, (0, 0, -5), (0, 0, -1))
    assert h["t"] == 1.0 and not h["front"] and h["n"] == (0.0, 0.0, 1.0)
    # strict interval: a root exactly at t_min is rejected (interval.rs:37-39) -> far root taken instead
    h = oracle.sphere_hit(s, (0, 0, 0), (0, 0, -1), tmin=4.0)
    assert h["t"] == 6.0
    # both roots outside -> None; miss -> None; unnormalised direction scales t
    assert oracle.sphere_hit(s, (0, 0, 0), (0, 0, -1), tmin=0.001, tmax=4.0) is None
    assert oracle.sphere_hit(s, (0, 0, 0), (0, 1, 0)) is None
    assert oracle.sphere_hit(s, (0, 0, 0), (0, 0, -2))["t"] == 2.0
    # grazing, d == 0 exactly: tangent ray at x = 1
    h = oracle.sphere_hit(s, (1, 0, 0), (0, 0, -1))
    assert h["t"] == 5.0
    # negative radius is stored as 0 (sphere.rs:31): a point sphere can still be "hit" exactly through its centre
    assert oracle.sphere_hit(_sphere(-3.0, (0, 0, -5)), (0, 0.5, 0), (0, 0, -1)) is None


def test_sphere_moving_center(oracle):
    s = _sphere(0.5, (0, 0, -5), (0, 2, -5))
    assert oracle.sphere_hit(s, (0, 0, 0), (0, 0, -1), tm=0.0)["t"] == 4.5
    assert oracle.sphere_hit(s, (0, 0, 0), (0, 0, -1), tm=1.0) is None
    assert oracle.sphere_hit(s, (0, 1, 0), (0, 0, -1), tm=0.5)["t"] == 4.5


def test_sphere_uv(oracle):
    # sphere.rs:72-76: theta = -acos(y), phi = -atan2(z, x) + pi, u = phi/(2pi), v = theta/pi  (v in [-1, 0])
    s = _sphere(1.0)
    h = oracle.sphere_hit(s, (5, 0, 0), (-1, 0, 0))  # hits (1,0,0)
    assert h["u"] == 0.5 * math.pi * 0.3183098861837907 and h["v"] == -(math.pi / 2) * 0.3183098861837907
    h = oracle.sphere_hit(s, (0, 5, 0), (0, -1, 0))  # north pole: acos(1) = 0
    assert h["v"] == 0.0 or h["v"] == -0.0


def test_np_restatement_agrees_on_random_rays(oracle):
    rng = np.random.default_rng(3)
    n_hit = 0
    for _ in range(3000):
        c0, c1 = rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3)
        mv = bool(rng.integers(2))
        r = float(rng.uniform(0.05, 2.5))
        o, d = rng.uniform(-4, 4, 3), rng.uniform(-1, 1, 3)
        tm = float(rng.random())
        a = oracle.sphere_hit(_sphere(r, tuple(c0), tuple(c1) if mv else None), tuple(o), tuple(d), tm=tm)
        b = oracle_np.sphere_hit(tuple(c0), tuple(c1), mv, r, tuple(o), tuple(d), tm, 0.001, float("inf"))
        assert (a is None) == (b is None)
        if a is not None:
            n_hit += 1
            n_flip = tuple(x if b[3] else -x for x in b[2])
            assert (a["t"], a["p"], a["n"], a["front"], a["u"], a["v"]) == (b[0], b[1], n_flip, b[3], b[4], b[5])
    assert n_hit > 100


# ---- materials (a11) -------------------------------------------------------------------------------------------------
def _one_mat_scene(mat):
    sc = Scene()
    sc.add(Sphere.new_static(1.0, (0, 0, 0), mat))
    sc.build_bvh()
    return sc.flat


def test_lambertian_scatter(oracle):
    fl = _one_mat_scene(Lambertian.new((0.5, 0.25, 1.0), 0.8))
    p, n = (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)
    seen_absorb = seen_scatter = False
    for sample in range(64):
        ua, ub = oracle.draws(11, 3, sample, 1, 0)
        uc, ud = oracle.draws(11, 3, sample, 1, 1)
        ok, att, so, sd, tm = oracle.scatter(fl, 0, (0, 5, 0), (0, -1, 0), 0.25, p, n, True, 0.0, 0.0, 11, 3, sample, 1)
        assert ok == (ua < 0.8)
        if ok:
            seen_scatter = True
            ln = math.sqrt(ub * ub + uc * uc + ud * ud)
            want = (n[0] + ub / ln, n[1] + uc / ln, n[2] + ud / ln)  # positive-octant quirk of point3.rs:88-93
            assert sd == want and so == p and tm == 0.25
            assert att == (0.5 / 0.8, 0.25 / 0.8, 1.0 / 0.8)  # attenuation = texture / reflectance (material.rs:77-78)
        else:
            seen_absorb = True
    assert seen_absorb and seen_scatter


def test_metal_scatter(oracle):
    fl = _one_mat_scene(Metal.new((0.7, 0.6, 0.5), None))
    d = (1.0, -2.0, 0.5)  # unnormalised on purpose: reflect uses D as is (material.rs:105)
    ok, att, so, sd, tm = oracle.scatter(fl, 0, (0, 5, 0), d, 0.0, (0, 1, 0), (0, 1, 0), True, 0, 0, 1, 2, 3, 1)
    dn = d[1]
    assert ok and att == (0.7, 0.6, 0.5) and sd == (d[0] - 0.0 * dn, d[1] - 2.0 * dn, d[2] - 0.0 * dn)
    # fuzzy: unit(refl) + unit(U,U,U)*fuzz, kept only if it leaves the surface
    fl = _one_mat_scene(Metal.new((1, 1, 1), 0.3))
    ua, ub = oracle.draws(1, 2, 3, 1, 0)
    uc, _ = oracle.draws(1, 2, 3, 1, 1)
    ok, att, so, sd, tm = oracle.scatter(fl, 0, (0, 5, 0), (0.0, -1.0, 0.0), 0.0, (0, 1, 0), (0, 1, 0), True, 0, 0, 1, 2, 3, 1)
    ln = math.sqrt(ua * ua + ub * ub + uc * uc)
    assert sd == (0.0 + ua / ln * 0.3, 1.0 + ub / ln * 0.3, 0.0 + uc / ln * 0.3) and ok


def test_dielectric_scatter(oracle):
    fl = _one_mat_scene(Dielectric.new(1.5))
    # head-on from outside: ri = 1/1.5, cos = 1, schlick = r0 = 0.04 -> refracts unless ua < 0.04; direction stays (0,-1,0)
    for sample in range(40):
        ua, _ = oracle.draws(5, 0, sample, 2, 0)
        ok, att, so, sd, tm = oracle.scatter(fl, 0, (0, 5, 0), (0, -2, 0), 0.0, (0, 1, 0), (0, 1, 0), True, 0, 0, 5, 0, sample, 2)
        assert ok and att == (1.0, 1.0, 1.0)
        r0 = ((1.0 - 1.0 / 1.5) / (1.0 + 1.0 / 1.5)) ** 2
        if r0 > ua:
            assert sd == (0.0, 1.0, 0.0)
        else:
            assert abs(sd[1] + 1.0) < 1e-15 and sd[0] == 0.0
    # total internal reflection from inside at a grazing angle: no draw is consumed, always reflects
    d = (1.0, 0.2, 0.0)
    ok, att, so, sd, tm = oracle.scatter(fl, 0, (0, 0, 0), d, 0.0, (0, 1, 0), (0, -1, 0), False, 0, 0, 5, 0, 0, 2)
    ln = math.sqrt(1.04)
    ud = (1.0 / ln, 0.2 / ln, 0.0 / ln)
    dot = ud[0] * 0.0 + ud[1] * -1.0 + ud[2] * 0.0
    assert sd == (ud[0] - (0.0 * 2.0) * dot, ud[1] - (-1.0 * 2.0) * dot, ud[2] - (0.0 * 2.0) * dot)


# ---- textures (a12) --------------------------------------------------------------------------------------------------
def test_checker_texture(oracle):
    fl = _one_mat_scene(Lambertian.with_texture(CheckerTexture(0.32, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0))
    even, odd = (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)
    assert oracle.texture_color(fl, 0, 0, 0, (0.1, 0.1, 0.1)) == even  # 0+0+0
    assert oracle.texture_color(fl, 0, 0, 0, (0.4, 0.1, 0.1)) == odd  # 1+0+0
    assert oracle.texture_color(fl, 0, 0, 0, (-0.1, 0.1, 0.1)) == odd  # floor(-0.3125) = -1: -1 % 2 = -1 != 0 in Rust
    assert oracle.texture_color(fl, 0, 0, 0, (-0.1, -0.1, 0.1)) == even  # -2
    assert oracle.texture_color(fl, 0, 0, 0, (-0.1, -0.1, -0.1)) == odd  # -3 % 2 = -1


def test_image_texture_index_clamps(oracle):
    img = np.zeros((4, 8, 3), np.uint8)
    img[..., 0] = np.arange(8)[None, :] * 10
    img[..., 1] = np.arange(4)[:, None] * 20
    fl = _one_mat_scene(Lambertian.with_texture(ImageTexture(img), 1.0))
    col = lambda u, v: oracle.texture_color(fl, 0, u, v, (0, 0, 0))
    assert col(0.0, 0.0) == (0.0, 0.0, 0.0)
    assert col(1.0, -1.0) == (70 / 255.0, 60 / 255.0, 0.0)  # u*W = 8 -> min(., W-1); |v| = 1 -> H-1
    assert col(0.999, -0.26) == (70 / 255.0, 20 / 255.0, 0.0)  # truncating casts
    assert col(-0.5, -7.0) == (0.0, 60 / 255.0, 0.0)  # clamp(u,0,1), clamp(|v|,0,1)
    assert col(0.5, 0.5) == (40 / 255.0, 40 / 255.0, 0.0)  # sign of v removed by abs
    # height 0 -> cyan (texture.rs:78-80)
    fl0 = _one_mat_scene(Lambertian.with_texture(ImageTexture(np.zeros((0, 0, 3), np.uint8)), 1.0))
    assert oracle.texture_color(fl0, 0, 0.3, 0.3, (0, 0, 0)) == (0.0, 1.0, 1.0)


# ---- quantiser (a13) -------------------------------------------------------------------------------------------------
def test_quantiser_edges(oracle):
    x = np.array([0.0, -1.0, 1e-300, 1.0, 4.0, 0.25, (127.5 / 254.999) ** 2, 1e308, float("inf"), -0.0], np.float64)
    q, nan = oracle.quantize(x)
    want = [oracle_np.quantize(float(v)) for v in x]
    assert nan == 0 and q.tolist() == want
    assert q[0] == 0 and q[1] == 0 and q[3] == 255 and q[4] == 255 and q[5] == 127  # sqrt(.25)*254.999 = 127.4995
    # a NaN never reaches the `expect("f64 is nan!")` of rgb.rs:33: linear_to_gamma's `linear > 0.0` is false for NaN -> 0.0
    q, nan = oracle.quantize(np.array([float("nan"), 0.5], np.float64))
    assert nan == 0 and q[0] == 0 and oracle_np.quantize(float("nan")) == 0
    rng = np.random.default_rng(0)
    x = rng.random(20000) * 1.2
    q, nan = oracle.quantize(x)
    assert q.tolist() == [oracle_np.quantize(float(v)) for v in x]


# ---- closest hit: BVH walk (reference) == brute-force loop (what the GPU runs) ------------------------------------------
@pytest.mark.parametrize("name", ["mars", "balls"])
def test_bvh_equals_brute_force_and_np(oracle, name):
    sc = presets.mars_texture_scene() if name == "mars" else presets.bouncing_balls_scene(1)
    cam = (presets.default_camera if name == "mars" else presets.bouncing_balls_camera)(96, 0, 50)
    ids_b, t_b = oracle.primary_hits(sc.flat, cam.pod, abi.F64, use_bvh=True)
    ids_f, t_f = oracle.primary_hits(sc.flat, cam.pod, abi.F64, use_bvh=False)
    assert np.array_equal(ids_b, ids_f) and np.array_equal(t_b.view(np.uint64), t_f.view(np.uint64))
    assert (ids_f != abi.MISS_ID).any() and (ids_f == abi.MISS_ID).any()
    # independent restatement on a coarse image
    cam = (presets.default_camera if name == "mars" else presets.bouncing_balls_camera)(32, 0, 50)
    ids_f, t_f = oracle.primary_hits(sc.flat, cam.pod, abi.F64, use_bvh=False)
    sph = [(tuple(s.center0), tuple(s.center1), bool(s.moving), s.radius) for s in sc.flat.sph[: sc.flat.n_spheres]]
    c = cam.pod
    ids_n, t_n = oracle_np.primary_hits(
        sph, dict(lookfrom=tuple(c.lookfrom), px_du=tuple(c.px_du), px_dv=tuple(c.px_dv), vp_upper_left=tuple(c.vp_upper_left),
                  W=c.img_width, H=c.img_height))
    assert ids_f.tolist() == ids_n and t_f.tolist() == t_n


def test_bvh_equals_brute_force_radiance(oracle):
    sc = presets.bouncing_balls_scene(2)
    cam = presets.bouncing_balls_camera(64, 4, 50)
    a = oracle.render(sc.flat, cam.pod, abi.F64, seed=3, use_bvh=True, threads=4)
    b = oracle.render(sc.flat, cam.pod, abi.F64, seed=3, use_bvh=False, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_ties_go_to_lowest_id(oracle):
    sc = Scene()
    m = Lambertian.new((1, 1, 1), 1.0)
    sc.add(Sphere.new_static(1.0, (0, 0, -5), m))
    sc.add(Sphere.new_static(1.0, (0, 0, -5), Lambertian.new((1, 0, 0), 1.0)))  # distinct object, same geometry
    sc.build_bvh()
    cam = Camera.build(None, None, None, 16, 1.0, 0, None, None, None, None)
    ids, _ = oracle.primary_hits(sc.flat, cam.pod, abi.F64, use_bvh=False)
    assert set(np.unique(ids)) <= {0, abi.MISS_ID} and (ids == 0).any()


def test_threads_do_not_change_the_image(oracle):
    sc = presets.bouncing_balls_scene(1)
    cam = presets.bouncing_balls_camera(48, 3, 50)
    a = oracle.render(sc.flat, cam.pod, abi.F32, seed=5, threads=1)
    b = oracle.render(sc.flat, cam.pod, abi.F32, seed=5, threads=8)
    assert np.array_equal(a[0], b[0]) and a[2] == b[2]


# ---- committed golden maps --------------------------------------------------------------------------------------------
def test_oracle_reproduces_golden(oracle, golden_dir):
    import os

    g = np.load(os.path.join(golden_dir, "oracle_maps.npz"))
    for name, mk_scene, mk_cam in (
        ("mars", presets.mars_texture_scene, lambda w, s: presets.default_camera(w, s, 50)),
        ("balls", lambda: presets.bouncing_balls_scene(1), lambda w, s: presets.bouncing_balls_camera(w, s, 50)),
    ):
        sc = mk_scene()
        for prec, tag in ((abi.F64, "f64"), (abi.F32, "f32")):
            ids, t = oracle.primary_hits(sc.flat, mk_cam(64, 0).pod, prec)
            assert np.array_equal(ids, g[f"{name}_64_{tag}_id"])
            assert np.array_equal(t.view(np.uint8), g[f"{name}_64_{tag}_t"].view(np.uint8))
            lin, rgb8, seg, nan = oracle.render(sc.flat, mk_cam(64, 16).pod, prec, seed=7, threads=8)
            assert np.array_equal(lin.view(np.uint8), g[f"{name}_64x16spp_{tag}_lin"].view(np.uint8))
            assert np.array_equal(rgb8, g[f"{name}_64x16spp_{tag}_rgb8"]) and seg == int(g[f"{name}_64x16spp_{tag}_segments"])


def test_f32_map_differs_from_f64_only_at_a_few_edge_pixels(oracle, golden_dir):
    import os

    g = np.load(os.path.join(golden_dir, "oracle_maps.npz"))
    for name in ("mars", "balls"):
        a, b = g[f"{name}_400_f64_id"], g[f"{name}_400_f32_id"]
        assert (a != b).mean() < 2e-3


def _np_scene(sc):
    """The flattened scene as oracle_np wants it (plain tuples), from the same mirt_scene_desc the C++ oracle reads."""
    f = sc.flat
    imgs = f._img_arrays
    texs = []
    for t in f.tex[: f.desc.n_textures]:
        if t.kind == abi.TEX_SOLID:
            texs.append(("solid", tuple(t.color)))
        elif t.kind == abi.TEX_CHECKER:
            texs.append(("checker", t.inv_scale, tuple(t.color), tuple(t.odd)))
        else:
            texs.append(("image", imgs[t.image] if imgs[t.image].ndim == 3 else None))
    mats = []
    for m in f.mat[: f.desc.n_materials]:
        if m.kind == abi.MAT_LAMBERTIAN:
            mats.append(("lambertian", texs[m.texture], m.reflectance))
        elif m.kind == abi.MAT_METAL:
            mats.append(("metal", tuple(m.albedo), m.fuzz if m.has_fuzz else None))
        else:
            mats.append(("dielectric", m.ior))
    return [(tuple(s.center0), tuple(s.center1), bool(s.moving), s.radius, mats[s.material]) for s in f.sph[: f.n_spheres]]


def _np_cam(c):
    return dict(lookfrom=tuple(c.lookfrom), px_du=tuple(c.px_du), px_dv=tuple(c.px_dv), vp_upper_left=tuple(c.vp_upper_left), px00_loc=tuple(c.px00_loc),
                disk_u_r=tuple(c.disk_u_r), disk_v_r=tuple(c.disk_v_r), defocus_angle=c.defocus_angle, W=c.img_width, H=c.img_height,
                spp=c.samples_per_pixel, depth=c.max_bounce_depth)


@pytest.mark.parametrize("name,width,spp,depth", [("balls", 24, 3, 12), ("balls", 20, 0, 6), ("mars", 24, 4, 8), ("textured", 20, 3, 10), ("scene", 20, 3, 10)])
def test_independent_python_render_equals_the_cpp_oracle(oracle, name, width, spp, depth):
    """The whole stochastic path - Philox and its dimension layout, camera ray with lens and shutter time, moving spheres, the three
    materials with their draw orders and quirks, the three textures, the recursion's product order, the sample sum, the quantiser -
    restated a second time in plain Python floats (oracle/oracle_np.py, written from the Rust source and the RNG contract):
    linear radiance, bytes and segment count bit-equal to oracle.cpp's f64 render (BVH walk and brute force)."""
    sc, cam = {
        "balls": lambda: (presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(width, spp, depth)),
        "mars": lambda: (presets.mars_texture_scene(), presets.default_camera(width, spp, depth)),
        "textured": lambda: (presets.textured_scene(), presets.textured_camera(width, spp, depth)),
        "scene": lambda: (presets.scene(), presets.plain_camera(width, spp, depth)),
    }[name]()
    lin, q8, seg = oracle_np.render(_np_scene(sc), _np_cam(cam.pod), seed=5)
    for use_bvh in (True, False):
        o_lin, o_rgb8, o_seg, nan = oracle.render(sc.flat, cam.pod, abi.F64, seed=5, use_bvh=use_bvh, threads=2)
        assert nan == 0 and o_seg == seg
        assert np.array_equal(np.asarray(lin, np.float64).view(np.uint64), o_lin.view(np.uint64))
        assert np.array_equal(np.asarray(q8, np.uint8), o_rgb8)
    for k, (ctr, key) in enumerate([((0, 0, 0, 0), (0, 0)), ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2), ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0))]):
        assert list(oracle_np.philox4x32_10(ctr, key)) == oracle.philox(ctr, key)


def test_f32_spec_paths_are_as_long_as_f64_paths_from_a_far_camera():
    """The f32 spec's discriminant goes through the perpendicular to the line (oracle.cpp Disc<float>): with the textbook
    form in binary32 this frame (camera 95 units from radius-0.1..0.5 spheres) traced 49 023 segments against 30 596 in
    binary64 - hit points inside the spheres, rays re-hitting them from within.  Different random numbers in the two
    precisions: the counts agree statistically."""
    from oracle import binding as ob
    from raytracer_amd import abi, presets

    sc, cam = presets.stress_scene(10000, 1), presets.stress_camera(96, 4, 50)
    seg = {}
    for prec, chunk in ((abi.F64, 0), (abi.F32, 4)):
        _, _, seg[prec], nan = ob.render(sc.flat, cam.pod, precision=prec, seed=1, use_bvh=True, threads=8, want_rgb8=False, spec_chunk=chunk)
        assert nan == 0
    assert abs(seg[abi.F32] - seg[abi.F64]) < 0.02 * seg[abi.F64], seg


# ---- the f32 spec against the reference's arithmetic under the same uniforms ---------------------------
def test_f32_layout_draws_in_the_f64_instantiation(oracle):
    """oracle_render_ex(precision 64, draws 32) feeds Real = double with the binary32 uniforms of the f32 layout: on a scene
    with nothing to hit the frame is a pure function of the two anti-aliasing draws, so it can be recomputed here."""
    cam = presets.plain_camera(8, 3, 5)
    sc = Scene()
    sc.add(Sphere.new_static(0.5, (0.0, 0.0, 50.0), Lambertian.new((0.5, 0.5, 0.5), 1.0)))  # behind the camera
    sc.build_bvh()
    lin, _, seg, sig = oracle.render_ex(sc.flat, cam.pod, abi.F64, 32, 7, use_bvh=True)
    assert seg == cam.width * cam.height * 3 and len(np.unique(sig)) == 1
    p = cam.pod
    for (hn, wn) in ((0, 0), (3, 5)):
        acc = np.zeros(3)
        for s in range(3):
            ox = oracle.draw_dim(7, hn * cam.width + wn, s, 0, 0, abi.F32) - 0.5
            oy = oracle.draw_dim(7, hn * cam.width + wn, s, 0, 1, abi.F32) - 0.5
            d = [p.px00_loc[i] + p.px_du[i] * (wn + ox) + p.px_dv[i] * (hn + oy) - p.lookfrom[i] for i in range(3)]
            a = 0.5 * (d[1] / math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1.0)
            acc += np.array([1.0 * (1.0 - a) + 0.5 * a, 1.0 * (1.0 - a) + 0.7 * a, 1.0 * (1.0 - a) + 1.0 * a])
        assert np.array_equal(lin[hn, wn], acc * (1.0 / 3.0))
    # the f64 layout gives another frame; binary32 arithmetic on 53-bit uniforms is refused
    lin64, _, _, _ = oracle.render_ex(sc.flat, cam.pod, abi.F64, 0, 7, use_bvh=True)
    assert not np.array_equal(lin, lin64)
    with pytest.raises(ValueError):
        oracle.render_ex(sc.flat, cam.pod, abi.F32, 64, 7)


def test_path_signatures_tell_branches_from_rounding(oracle):
    """Equal frames have equal signatures; BVH walk and brute force take the same discrete paths; another seed does not."""
    sc, cam = presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(48, 3, 20)
    a = oracle.render_ex(sc.flat, cam.pod, abi.F64, 32, 5, use_bvh=True, threads=4)
    b = oracle.render_ex(sc.flat, cam.pod, abi.F64, 32, 5, use_bvh=False, threads=2)
    c = oracle.render_ex(sc.flat, cam.pod, abi.F64, 32, 6, use_bvh=True, threads=4)
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0])
    assert (a[3] != c[3]).mean() > 0.5  # the sky rows take the same one-segment path under any seed
    # signatures ride along: the frame is render()'s
    lin, _, seg, _ = oracle.render(sc.flat, cam.pod, abi.F32, seed=5, threads=4, spec_chunk=2)
    e = oracle.render_ex(sc.flat, cam.pod, abi.F32, 0, 5, threads=4, spec_chunk=2)
    assert np.array_equal(lin.view(np.uint32), e[0].view(np.uint32)) and seg == e[2]


@pytest.mark.parametrize("name", ["c1", "bench", "textured", "stress"])
def test_f32_spec_against_reference_arithmetic_same_uniforms(name):
    """The f32 SPEC (which the HIP kernels reproduce bit for bit, tests/test_gpu_exact.py) within the stated bars of the
    reference's f64 arithmetic fed with the same uniforms - the CPU half of tests/test_gpu_f32_vs_reference.py, so that a change
    of the f32 spec that biases the image fails here before it reaches a GPU."""
    from tools import f32_vs_reference as fr

    st = fr.run(name)
    assert not fr.failures(st, fr.BARS[name]), (fr.failures(st, fr.BARS[name]), st)
