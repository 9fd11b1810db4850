"""Host logic in front of the C ABI: flattening, presets, PPM text, band bookkeeping.  No GPU."""
import io

import numpy as np
import pytest

from raytracer_amd import abi, dist, presets
from raytracer_amd.api import Camera, Dielectric, Lambertian, Metal, Scene, Sphere, ppm_body


def test_flatten_dedupes_shared_objects():
    sc = Scene()
    glass = Dielectric.new(1.5)
    ground = Sphere.new_static(1000.0, (0, -1000, 0), Lambertian.new((0.5, 0.5, 0.5), 1.0))
    a = Sphere.new_static(1.0, (0, 1, 0), glass)
    b = Sphere.new(0.2, (1, 0.2, 0), (1, 0.5, 0), glass)
    for o in (a, ground, b, ground, ground):
        sc.add(o)
    with pytest.raises(RuntimeError, match="expected bvh to exist"):
        sc.flat
    sc.build_bvh()
    f = sc.flat
    assert f.n_spheres == 3 and f.desc.n_materials == 2 and f.objects.tolist() == [0, 1, 2, 1, 1]
    assert f.sph[2].moving == 1 and tuple(f.sph[2].center1) == (1.0, 0.5, 0.0) and f.sph[0].material == f.sph[2].material
    sc.build_bvh()  # idempotent (hittable.rs:86-90)
    assert sc.flat is f


def test_bouncing_balls_preset_shape():
    sc = presets.bouncing_balls_scene(1)
    f = sc.flat
    assert len(f.objects) == f.n_spheres + 21  # the ground sphere is added 22 times (main.rs:225)
    kinds = [f.mat[f.sph[i].material].kind for i in range(f.n_spheres)]
    moving = sum(f.sph[i].moving for i in range(f.n_spheres))
    assert 470 <= f.n_spheres <= 488 and 0.7 < moving / f.n_spheres < 0.9
    assert kinds.count(abi.MAT_DIELECTRIC) > 5 and kinds.count(abi.MAT_METAL) > 30
    assert presets.bouncing_balls_scene(1).flat.n_spheres == f.n_spheres  # seeded: reproducible
    # every moving sphere is a Lambertian with reflectance in [0.8, 1]
    for i in range(f.n_spheres):
        if f.sph[i].moving:
            m = f.mat[f.sph[i].material]
            assert m.kind == abi.MAT_LAMBERTIAN and 0.8 <= m.reflectance <= 1.0


def test_default_camera_is_main_rs():
    c = presets.default_camera()
    assert (c.width, c.height, c.pod.samples_per_pixel, c.pod.max_bounce_depth) == (400, 225, 100, 50)
    assert Camera.build(None, None, None, 100, 2.0, 0, None, None, None, None).pod.samples_per_pixel == 0
    assert Camera.build(None, None, None, 100, 2.0, None, None, None, None, None).pod.samples_per_pixel == 100


def test_ppm_body_format():
    img = np.array([[[0, 1, 2], [255, 254, 3]]], np.uint8)
    assert ppm_body(img) == b"0 1 2\n255 254 3\n"


@pytest.mark.parametrize("H,S,n", [(1080, 8, 8), (225, 8, 2), (17, 4, 3), (5, 8, 4), (64, 16, 1)])
def test_band_bookkeeping_roundtrip(H, S, n):
    W = 7
    frame = np.arange(H * W * 3, dtype=np.float32).reshape(H, W, 3)
    rows_max = dist.part_rows(H, S, 0, n)
    bands = np.zeros((n, rows_max, W, 3), np.float32)
    seen = np.zeros(H, int)
    for r in range(n):
        rows = dist.owned_rows(H, S, r, n)
        assert len(rows) == dist.part_rows(H, S, r, n) == abi.load().mirt_part_rows(H, S, r, n)
        seen[rows] += 1
        bands[r, : len(rows)] = frame[rows]
    assert (seen == 1).all()
    assert np.array_equal(dist.deinterleave_host(bands, H, S), frame)


def test_bench_helpers_import_and_parse_the_committed_profiles():
    """bench.py is only ever run on the GPU box; its host-side helpers are exercised here so that an edit cannot lose one of
    them unnoticed (it did once: a NameError surfaced only on the box)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for name in ("build_workload", "cpu_baseline", "cpu_baseline_c1", "profile_traffic", "profile_valu_rate", "run_node", "main"):
        assert callable(getattr(b, name)), name
    sc, cam, label = b.build_workload("stress16", 64, 1, 2)
    assert sc.flat.n_spheres == 16 and cam.width == 64 and label == "stress(16)"
    t = b.profile_traffic("r02_mx", "render_kernel_f32_mx")
    assert t and t["file"] == "profiles/r02_mx_hbm_traffic.txt" and 7.0e8 < t["bytes"] < 9.5e8 and t["write"] > 20 * t["fetch"]
    r = b.profile_valu_rate("r02_textured")
    assert r and 0.2 < r < 0.5
    assert b.MX_CYCLES_PER_WAVE_TILE == 192
