"""The C++ host mirror (raytracer_amd/host/raytracer.hpp) against the Python mirror (raytracer_amd/api.py).

CPU: Camera::build must produce the same POD bit for bit (both restate camera.rs:135-182 in binary64).
GPU: the example program's PPM (through mirt_render) equals the PPM of the same scene built in Python.
"""
import io
import os
import subprocess

import pytest

from raytracer_amd import abi
from raytracer_amd.api import Camera, CheckerTexture, Dielectric, Lambertian, Metal, Scene, Sphere

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "raytracer_amd", "host")
EXE = os.path.join(HOST, "example_main")


def _exe():
    if not os.path.exists(EXE):
        r = subprocess.run(["make", "-C", HOST], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("C++ host example not built: " + r.stderr[-300:])
    return EXE


def _example_camera(width, spp=8, depth=10):
    return Camera.build((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), width, 16.0 / 9.0, spp,
                        0.3490658503988659, 10.0, 0.010471975511965976, depth)


def _pod_lines(cam):
    p = cam.pod
    groups = [p.lookfrom, p.px_du, p.px_dv, p.vp_upper_left, p.px00_loc, p.disk_u_r, p.disk_v_r]
    vals = [[float(x) for x in g] for g in groups]
    return vals, float(p.defocus_angle), (p.img_width, p.img_height, p.samples_per_pixel, p.max_bounce_depth)


def _parse(lines):
    vals = [[float.fromhex(x) for x in ln.split()] for ln in lines[:7]]
    return vals, float.fromhex(lines[7]), tuple(int(x) for x in lines[8].split())


@pytest.mark.parametrize("width", [160, 401, 1920])
def test_cpp_camera_build_bit_equal_to_python(width):
    out = subprocess.run([_exe(), "camera", str(width)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert _parse(out[0:9]) == _pod_lines(_example_camera(width))
    # all-defaults camera (camera.rs:135-149): vfov pi/2, 100 spp, depth 10
    assert _parse(out[9:18]) == _pod_lines(Camera.build(None, None, None, width, 1.5, None, None, None, None, None))


def _example_scene():
    sc = Scene()
    checker = CheckerTexture(0.32, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    ground = Sphere.new_static(1000.0, (0.0, -1000.0, 0.0), Lambertian.with_texture(checker, 0.9))
    sc.add(ground)
    sc.add(ground)
    sc.add(Sphere.new_static(1.0, (0.0, 1.0, 0.0), Dielectric.new(1.5)))
    sc.add(Sphere.new_static(1.0, (-4.0, 1.0, 0.0), Lambertian.new((0.4, 0.2, 0.1), 0.8)))
    sc.add(Sphere.new_static(1.0, (4.0, 1.0, 0.0), Metal.new((0.7, 0.6, 0.5), 0.1)))
    sc.add(Sphere.new_static(0.5, (2.0, 0.5, 2.0), Metal.new((0.8, 0.8, 0.9), None)))
    sc.add(Sphere.new(0.3, (-1.5, 0.3, 2.5), (-1.5, 0.6, 2.5), Lambertian.new((0.1, 0.2, 0.7), 1.0)))
    sc.build_bvh()
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [64, 32])
def test_cpp_example_ppm_equals_python_mirror(prec):
    r = subprocess.run([_exe(), "96", "4", "6", "11", str(prec)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    cam = _example_camera(96, spp=4, depth=6)
    cam.seed, cam.precision = 11, (abi.F64 if prec == 64 else abi.F32)
    buf = io.BytesIO()
    cam.render(_example_scene(), out=buf)
    assert r.stdout == buf.getvalue()
    assert r.stdout.startswith(b"P3\n96 54\n255\n")


# ---- image loader and preset front end (f4) ------------------------------------------------------------
CLI = os.path.join(HOST, "mirt_cli")
ASSETS = os.path.join(ROOT, "raytracer_amd", "assets")


def _cli():
    if not os.path.exists(CLI):
        r = subprocess.run(["make", "-C", HOST], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("mirt_cli not built: " + r.stderr[-300:])
    return CLI


def _read_p6(path):
    import numpy as np

    d = open(path, "rb").read()
    magic, dims, maxv, body = d.split(b"\n", 3)
    w, h = map(int, dims.split())
    assert magic == b"P6" and maxv == b"255"
    return np.frombuffer(body, dtype=np.uint8).reshape(h, w, 3)


def _synthetic_image(w, h):
    import numpy as np

    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(3)
    img = np.stack([127 + 120 * np.sin(x / 9.0) * np.cos(y / 13.0), (x * 255 / w), 255 - (y * 255 / h)], axis=-1)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling,w,h,tol_max,tol_mean", [(0, 200, 120, 4, 0.2), (0, 67, 45, 4, 0.2), (2, 160, 96, 64, 3.0), (-1, 64, 64, 1, 0.05)])
def test_jpeg_decoder_against_libjpeg(tmp_path, subsampling, w, h, tol_max, tol_mean):
    """Baseline decode of PIL-written files vs PIL's own (libjpeg) decode.  4:4:4 and grayscale: within the +-1 of the
    IDCT times the colour matrix; 4:2:0: libjpeg interpolates chroma where this decoder replicates it (loose bound).
    subsampling -1 = grayscale; odd sizes exercise the partial MCUs; one case uses restart markers."""
    import numpy as np
    from PIL import Image

    img = _synthetic_image(w, h)
    src = tmp_path / "t.jpg"
    if subsampling < 0:
        Image.fromarray(img[..., 0]).save(src, quality=92)
    else:
        Image.fromarray(img).save(src, quality=92, subsampling=subsampling, **({"restart_marker_rows": 1} if w == 67 else {}))
    want = np.asarray(Image.open(src).convert("RGB")).astype(int)
    out = tmp_path / "t.ppm"
    subprocess.run([_cli(), "decode", str(src), str(out)], check=True, capture_output=True)
    got = _read_p6(out).astype(int)
    assert got.shape == want.shape
    d = np.abs(got - want)
    assert d.max() <= tol_max and d.mean() <= tol_mean, (d.max(), d.mean())


def test_jpeg_decoder_rejects_progressive_and_garbage(tmp_path):
    from PIL import Image

    src = tmp_path / "p.jpg"
    Image.fromarray(_synthetic_image(32, 32)).save(src, progressive=True)
    r = subprocess.run([_cli(), "decode", str(src), str(tmp_path / "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 1 and "baseline" in r.stderr
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"\xff\xd8\xff\xdb\x00\x02garbage")
    r = subprocess.run([_cli(), "decode", str(bad), str(tmp_path / "o.ppm")], capture_output=True, text=True)
    assert r.returncode == 1 and "error" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/presets/textures"), reason="the reference's textures are only in the build container")
@pytest.mark.parametrize("stem", ["earthmap", "mars_1k_color"])
def test_jpeg_decoder_on_the_reference_textures(tmp_path, stem):
    """The files ImageTexture is fed with in main.rs:138-141,154-157, against the libjpeg decode shipped in assets/."""
    import numpy as np

    out = tmp_path / "o.ppm"
    subprocess.run([_cli(), "decode", f"/root/reference/presets/textures/{stem}.jpg", str(out)], check=True, capture_output=True)
    got = _read_p6(out).astype(int)
    with np.load(os.path.join(ASSETS, "textures.npz")) as z:
        want = z[stem].astype(int)
    d = np.abs(got - want)
    # the IJG-style fixed-point IDCT and colour conversion reproduce libjpeg on these 4:4:4 files byte for byte (round 1's
    # floating-point IDCT was within 4); zune-jpeg itself, the reference's decoder, stays unpinned
    assert got.shape == (512, 1024, 3) and d.max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", [("mars", ["--width", "80", "--spp", "2", "--depth", "8", "--precision", "64"]),
                                       ("bouncing_balls", ["--width", "96", "--spp", "2", "--depth", "12", "--precision", "32"]),
                                       ("bouncing_balls", ["--width", "96", "--spp", "0", "--depth", "12", "--precision", "32", "--accel", "bvh"]),
                                       ("scene", ["--width", "64", "--spp", "3", "--depth", "10", "--precision", "64"]),
                                       ("scene2", ["--width", "64", "--spp", "0", "--depth", "10", "--precision", "64"]),
                                       ("blur_scene", ["--width", "64", "--spp", "3", "--depth", "10", "--precision", "32"])])
def test_cli_presets_equal_python_presets(name, args):
    """Both front ends must build the same scene (same seeded layout, same flattening) and camera: equal PPM bytes."""
    from raytracer_amd import presets

    if not os.path.exists(os.path.join(ASSETS, "earthmap.ppm")):
        import __graft_entry__

        __graft_entry__.export_textures()
    r = subprocess.run([_cli(), "render", "--scene", name, "--seed", "5", "--layout-seed", "1", "--textures", ASSETS] + args, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    kv = dict(zip(args[::2], args[1::2]))
    w, spp, depth = int(kv["--width"]), int(kv["--spp"]), int(kv["--depth"])
    sc, cam = {
        "mars": lambda: (presets.mars_texture_scene(), presets.default_camera(w, spp, depth)),
        "bouncing_balls": lambda: (presets.bouncing_balls_scene(1), presets.bouncing_balls_camera(w, spp, depth)),
        "scene": lambda: (presets.scene(), presets.plain_camera(w, spp, depth)),
        "scene2": lambda: (presets.scene2(), presets.plain_camera(w, spp, depth)),
        "blur_scene": lambda: (presets.blur_scene(), presets.plain_camera(w, spp, depth)),
    }[name]()
    cam.seed, cam.precision = 5, (abi.F64 if kv["--precision"] == "64" else abi.F32)
    _, rgb8 = cam.render_arrays(sc, want_linear=False, accel="bvh" if kv.get("--accel") == "bvh" else None)
    from raytracer_amd.api import ppm_body

    want = f"P3\n{cam.width} {cam.height}\n255\n".encode() + ppm_body(rgb8)
    assert r.stdout == want


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["32", "64"])
def test_cli_progressive_passes_and_checkpoint_resume(tmp_path, prec):
    """the frame rendered in passes - and across two runs through a checkpoint file - is the frame of one run."""
    if not os.path.exists(os.path.join(ASSETS, "earthmap.ppm")):
        import __graft_entry__

        __graft_entry__.export_textures()
    base = [_cli(), "render", "--scene", "bouncing_balls", "--width", "80", "--spp", "10", "--depth", "12", "--precision", prec, "--seed", "3",
            "--textures", ASSETS]
    one = subprocess.run(base, capture_output=True, timeout=120)
    assert one.returncode == 0, one.stderr.decode()[-400:]
    out = tmp_path / "p.ppm"
    r = subprocess.run(base + ["--passes", "4", "--out", str(out)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    assert out.read_bytes() == one.stdout and r.stderr.count(b"pass ") == 4
    ck, out2 = tmp_path / "job.ck", tmp_path / "q.ppm"
    r1 = subprocess.run(base + ["--passes", "3", "--checkpoint", str(ck), "--out", str(out2), "--stop-after", "1"], capture_output=True, timeout=120)
    assert r1.returncode == 0 and ck.exists() and out2.read_bytes() != one.stdout  # a quarter of the samples so far
    r2 = subprocess.run(base + ["--passes", "3", "--checkpoint", str(ck), "--out", str(out2)], capture_output=True, timeout=120)
    assert r2.returncode == 0 and b"resuming" in r2.stderr, r2.stderr.decode()[-400:]
    assert out2.read_bytes() == one.stdout
    other = subprocess.run(base[:-4] + ["--seed", "4", "--textures", ASSETS, "--checkpoint", str(ck)], capture_output=True, timeout=120)
    assert other.returncode == 1 and b"another job" in other.stderr
    # sums written with one work-item size cannot be continued with another: the size is part of the job
    ck2 = tmp_path / "job2.ck"
    r3 = subprocess.run(base + ["--passes", "2", "--checkpoint", str(ck2), "--stop-after", "1", "--out", str(out2)], capture_output=True, timeout=120)
    assert r3.returncode == 0 and ck2.exists()
    r4 = subprocess.run(base + ["--passes", "2", "--checkpoint", str(ck2), "--chunk", "2"], capture_output=True, timeout=120)
    assert r4.returncode == 1 and b"another job" in r4.stderr


@pytest.mark.gpu
def test_cli_loads_jpeg_textures_directly(tmp_path):
    """SURVEY.md 8 f3 / image_loader.rs:5-8: `--textures DIR` with the .jpg files themselves (here a 4:4:4 baseline JPEG written
    by libjpeg under the reference's file name).  The built-in decoder reproduces libjpeg's bytes, so the rendered PPM equals the
    Python front end's render of the libjpeg-decoded image."""
    import numpy as np
    from PIL import Image

    from raytracer_amd import presets
    from raytracer_amd.api import ImageTexture, Lambertian, Scene, Sphere, ppm_body

    img = _synthetic_image(256, 128)
    Image.fromarray(img).save(tmp_path / "earthmap.jpg", quality=92, subsampling=0)
    decoded = np.asarray(Image.open(tmp_path / "earthmap.jpg").convert("RGB"))
    r = subprocess.run([_cli(), "render", "--scene", "mars", "--width", "96", "--spp", "4", "--depth", "8", "--precision", "64", "--seed", "3",
                        "--textures", str(tmp_path)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    sc = Scene()
    sc.add(Sphere.new_static(1.8, (0.0, 0.0, 0.0), Lambertian.with_texture(ImageTexture(decoded), 1.0)))
    sc.build_bvh()
    cam = presets.default_camera(96, 4, 8)
    cam.seed, cam.precision = 3, abi.F64
    _, rgb8 = cam.render_arrays(sc, want_linear=False)
    assert r.stdout == f"P3\n{cam.width} {cam.height}\n255\n".encode() + ppm_body(rgb8)
