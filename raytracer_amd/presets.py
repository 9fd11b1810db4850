"""Scene presets: seeded restatements of the reference's hard-coded scenes (src/main.rs) plus the stress scene
of SURVEY.md §8(d).  The reference seeds its layout RNG from the OS (main.rs:199), so only the distribution
is contractual; here every layout is a pure function of `seed`.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np

from .api import Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, Scene, Sphere

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "textures.npz")
_cache = {}

MARS_TEXTURE = "mars_1k_color.jpg"  # main.rs:25
EARTH_TEXTURE = "earthmap.jpg"  # main.rs:26


def load_image_to_rgb(filename: str) -> np.ndarray:
    """scene/image_loader.rs:5-8.  The two reference textures are shipped pre-decoded (tools/make_textures.py);
    returns an (H, W, 3) uint8 array = image::RgbImage."""
    key = os.path.splitext(os.path.basename(filename))[0]
    if key not in _cache:
        with np.load(_ASSETS) as z:
            if key not in z.files:
                raise FileNotFoundError(f"failed to load image {filename!r}")
            _cache[key] = np.ascontiguousarray(z[key])
    return _cache[key]


class SplitMix64:
    """Tiny stable generator for layouts (independent of numpy's stream definitions)."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def u01(self) -> float:
        return (self.next_u64() >> 11) * 2.0**-53

    def range(self, lo: float, hi: float) -> float:
        return lo + (hi - lo) * self.u01()


# ---- cameras ------------------------------------------------------------------------------------------
def default_camera(img_width=400, aa_samples_per_px: Optional[int] = 100, max_depth=50, lookfrom=(0.0, 0.0, 12.0)) -> Camera:
    """The camera main() builds (main.rs:30-56): look_from_mars -> origin, vfov 25 deg, focus 10, defocus 0.4 deg."""
    return Camera.build(
        lookfrom, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), img_width, 16.0 / 9.0, aa_samples_per_px,
        math.radians(25.0), 10.0, math.radians(0.4), max_depth,
    )


def bouncing_balls_camera(img_width=1920, aa_samples_per_px: Optional[int] = 64, max_depth=50) -> Camera:
    """The camera intended for bouncing_balls_scene: the dead `lookfrom` of main.rs:32, (13,2,3)."""
    return default_camera(img_width, aa_samples_per_px, max_depth, lookfrom=(13.0, 2.0, 3.0))


# ---- scenes -------------------------------------------------------------------------------------------
def mars_texture_scene() -> Scene:
    """main.rs:136-149 (C1/C2): one sphere r=1.8 at the origin, Lambertian(ImageTexture(earthmap), 1.0)."""
    scene = Scene()
    tex = ImageTexture(load_image_to_rgb(EARTH_TEXTURE))  # NB: the "mars" scene loads the EARTH texture
    mat = Lambertian.with_texture(tex, 1.0)
    scene.add(Sphere.new_static(1.8, (0.0, 0.0, 0.0), mat))
    scene.build_bvh()
    return scene


def bouncing_balls_scene(seed: int = 1) -> Scene:
    """main.rs:151-229 (C3) with the draw order of :199-221; the ground sphere is added 22 times (:225)."""
    rng = SplitMix64(seed)
    scene = Scene()
    mars_material = Lambertian.with_texture(ImageTexture(load_image_to_rgb(MARS_TEXTURE)), 1.0)
    near3, near1, near2 = (-4.0, 1.0, 0.0), (4.0, 1.0, 0.0), (0.0, 1.0, 0.0)
    mars_surface = Sphere.new_static(1.2, near1, mars_material)
    ground_material = Lambertian.with_texture(CheckerTexture(0.32, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)
    ground_sphere = Sphere.new_static(1000.0, (0.0, -1000.0, 0.0), ground_material)
    glass_mat = Dielectric.new(1.5)
    glass_sphere = Sphere.new_static(1.0, near2, glass_mat)
    metallic_sphere = Sphere.new_static(1.0, near3, Metal.new((0.7, 0.6, 0.5), None))
    for s in (glass_sphere, mars_surface, metallic_sphere):
        scene.add(s)
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.u01()
            center = (float(a) + 0.9 * rng.u01(), 0.2, float(b) + 0.9 * rng.u01())
            d = (center[0] - 4.0, center[1] - 0.2, center[2] - 0.0)
            if math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > 0.9:
                center2 = None
                if choose_mat < 0.8:
                    a1 = (rng.u01(), rng.u01(), rng.u01())
                    a2 = (rng.u01(), rng.u01(), rng.u01())
                    albedo = (a1[0] * a2[0], a1[1] * a2[1], a1[2] * a2[2])
                    reflectance = rng.range(0.8, 1.0)
                    center2 = (center[0] + 0.0, center[1] + rng.range(0.0, 0.5), center[2] + 0.0)
                    mat = Lambertian.new(albedo, reflectance)
                elif choose_mat < 0.95:
                    albedo = (rng.range(0.5, 1.0), rng.range(0.5, 1.0), rng.range(0.5, 1.0))
                    mat = Metal.new(albedo, rng.range(0.0, 0.5))
                else:
                    mat = glass_mat
                scene.add(Sphere.new(0.2, center, center2, mat))
        scene.add(ground_sphere)
    scene.build_bvh()
    return scene


def plain_camera(img_width=400, aa_samples_per_px: Optional[int] = 100, max_depth=50) -> Camera:
    """The all-defaults camera (camera.rs:135-149) the three small scenes of main.rs:74-121 were written for."""
    return Camera.build(None, None, None, img_width, 16.0 / 9.0, aa_samples_per_px, None, None, None, max_depth)


def _small_scene_materials():
    return (Lambertian.new((0.8, 0.8, 0.1), 1.0), Lambertian.new((0.7, 0.2, 0.8), 1.0), Dielectric.new(1.5),
            Dielectric.new(1.0 / 1.5), Metal.new((0.8, 0.6, 0.2), 0.5))


def scene2() -> Scene:
    """main.rs:74-87 (built by main(), never rendered)."""
    r = math.cos(math.pi / 4.0)
    sc = Scene()
    sc.add(Sphere.new_static(r, (-r, 0.0, -1.0), Lambertian.new((1.0, 1.0, 0.0), 1.0)))
    sc.add(Sphere.new_static(r, (r, 0.0, -1.0), Lambertian.new((0.0, 1.0, 1.0), 1.0)))
    sc.build_bvh()
    return sc


def scene() -> Scene:
    """main.rs:95-107."""
    g, c, l, b, r = _small_scene_materials()
    sc = Scene()
    for rad, ctr, m in ((100.0, (0.0, -100.5, -1.0), g), (0.5, (0.0, 0.0, -3.2), c), (0.5, (-1.0, 0.0, -2.5), l), (0.3, (-1.0, 0.0, -2.5), b),
                        (0.4, (1.0, 0.0, -1.0), r), (0.4, (6.0, 0.0, -10.0), r), (0.5, (-3.0, 0.0, -1.6), c)):
        sc.add(Sphere.new_static(rad, ctr, m))
    sc.build_bvh()
    return sc


def blur_scene() -> Scene:
    """main.rs:109-121."""
    g, c, l, b, r = _small_scene_materials()
    sc = Scene()
    for rad, ctr, m in ((100.0, (0.0, -100.5, -1.0), g), (0.5, (0.0, 0.0, -1.2), c), (0.5, (-1.0, 0.0, -1.0), l), (0.3, (-1.0, 0.0, -1.0), b),
                        (0.4, (1.0, 0.0, -1.0), r), (0.4, (6.0, 0.0, -10.0), r), (0.5, (-3.0, 0.0, -1.6), c)):
        sc.add(Sphere.new_static(rad, ctr, m))
    sc.build_bvh()
    return sc


def textured_scene() -> Scene:
    """C4: image + procedural textures from presets/textures: a mars sphere, an earth sphere, checker ground."""
    scene = Scene()
    mars = Lambertian.with_texture(ImageTexture(load_image_to_rgb(MARS_TEXTURE)), 1.0)
    earth = Lambertian.with_texture(ImageTexture(load_image_to_rgb(EARTH_TEXTURE)), 1.0)
    ground = Lambertian.with_texture(CheckerTexture(0.32, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)
    scene.add(Sphere.new_static(1.8, (-2.0, 1.8, 0.0), mars))
    scene.add(Sphere.new_static(1.8, (2.0, 1.8, 0.0), earth))
    scene.add(Sphere.new_static(1000.0, (0.0, -1000.0, 0.0), ground))
    scene.build_bvh()
    return scene


def stress_scene(n: int, seed: int = 1) -> Scene:
    """C5: n static spheres, centres U([-50,50] x [0.2,20] x [-50,50]), r U[0.1,0.5],
    70/20/10 % Lambertian / metal / glass."""
    rng = np.random.Generator(np.random.Philox(seed))
    c = rng.random((n, 3))
    r = 0.1 + 0.4 * rng.random(n)
    kind = rng.random(n)
    col = rng.random((n, 3))
    fuzz = 0.5 * rng.random(n)
    scene = Scene()
    glass = Dielectric.new(1.5)
    for i in range(n):
        center = (-50.0 + 100.0 * c[i, 0], 0.2 + 19.8 * c[i, 1], -50.0 + 100.0 * c[i, 2])
        if kind[i] < 0.7:
            mat = Lambertian.new(tuple(col[i]), 1.0)
        elif kind[i] < 0.9:
            mat = Metal.new(tuple(0.5 + 0.5 * col[i]), float(fuzz[i]))
        else:
            mat = glass
        scene.add(Sphere.new_static(float(r[i]), center, mat))
    scene.build_bvh()
    return scene


def stress_camera(img_width=3840, aa_samples_per_px: Optional[int] = 16, max_depth=50) -> Camera:
    return Camera.build(
        (0.0, 30.0, 90.0), (0.0, 8.0, 0.0), (0.0, 1.0, 0.0), img_width, 16.0 / 9.0, aa_samples_per_px,
        math.radians(40.0), 90.0, 0.0, max_depth,
    )


def textured_camera(img_width=3840, aa_samples_per_px: Optional[int] = 256, max_depth=50) -> Camera:
    return Camera.build(
        (0.0, 3.0, 14.0), (0.0, 1.5, 0.0), (0.0, 1.0, 0.0), img_width, 16.0 / 9.0, aa_samples_per_px,
        math.radians(25.0), 14.0, math.radians(0.4), max_depth,
    )
