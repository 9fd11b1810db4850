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
