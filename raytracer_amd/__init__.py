"""raytracer_amd - MI355X (gfx950) render backend for the sphere path tracer's per-pixel hot path.

csrc/   hand-written HIP kernels + the C ABI (include/mirt.h) -> libmirt.so
api.py  host-side mirror of the reference's Scene / Camera / Sphere / Material / Texture interface
presets.py  seeded restatements of the reference's hard-coded scenes
"""
from . import abi  # noqa: F401
from .api import (  # noqa: F401
    Camera, CheckerTexture, Dielectric, ImageTexture, InitError, Lambertian, Material, Metal, RenderError,
    Scene, SolidColor, Sphere, Texture,
)

__all__ = [
    "abi", "Camera", "CheckerTexture", "Dielectric", "ImageTexture", "InitError", "Lambertian", "Material", "Metal",
    "RenderError", "Scene", "SolidColor", "Sphere", "Texture",
]
