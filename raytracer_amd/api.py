"""Host-side mirror of the reference's scene/camera interface, in front of the C ABI (include/mirt.h).

Names, argument meaning and error behaviour follow the reference (all file:line are /root/reference/src):
  Camera.build / Camera.render           camera/camera.rs:123-206, 232-259
  Scene.add / Scene.build_bvh            scene/hittable.rs:79-90
  Sphere.new_static / Sphere.new         scene/sphere.rs:28-59
  Lambertian / Metal / Dielectric        scene/material.rs:35-57, 91-95, 120-124
  SolidColor / CheckerTexture / ImageTexture   scene/texture.rs:18-21, 36-44, 65-68

All host arithmetic is Python float = IEEE binary64 without contraction, i.e. the reference's f64.
The per-pixel work itself runs only in the HIP library; nothing here falls back to a CPU renderer.
"""
from __future__ import annotations

import ctypes as C
import math
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import abi

Vec = Tuple[float, float, float]


# ---- core/point3.rs helpers (f64, same operation order) -------------------------------------------
def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def _mul(a, l):
    return (a[0] * l, a[1] * l, a[2] * l)


def _div(a, l):
    return (a[0] / l, a[1] / l, a[2] / l)


def _neg(a):
    return (-a[0], -a[1], -a[2])


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _unit(a):
    return _div(a, math.sqrt(_dot(a, a)))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


# ---- textures (scene/texture.rs) -------------------------------------------------------------------
class Texture:
    pass


@dataclass(eq=False)
class SolidColor(Texture):
    albedo: Vec


@dataclass(eq=False)
class CheckerTexture(Texture):
    """CheckerTexture::new(scale, even_color, odd_color): stores inv_scale = scale.recip() (texture.rs:39)."""

    scale: float
    even_color: Vec
    odd_color: Vec

    @property
    def inv_scale(self) -> float:
        return 1.0 / self.scale


@dataclass(eq=False)
class ImageTexture(Texture):
    """ImageTexture::new(Arc<RgbImage>): px_colors is an (H, W, 3) uint8 array (image::RgbImage)."""

    px_colors: np.ndarray


# ---- materials (scene/material.rs) ------------------------------------------------------------------
class Material:
    pass


@dataclass(eq=False)
class Lambertian(Material):
    texture: Texture
    reflectance: float

    @staticmethod
    def new(albedo: Vec, reflectance: float) -> "Lambertian":
        return Lambertian(SolidColor(tuple(albedo)), reflectance)

    @staticmethod
    def with_texture(texture: Texture, reflectance: float) -> "Lambertian":
        return Lambertian(texture, reflectance)


@dataclass(eq=False)
class Metal(Material):
    albedo: Vec
    fuzz: Optional[float] = None

    @staticmethod
    def new(albedo: Vec, fuzz: Optional[float]) -> "Metal":
        return Metal(tuple(albedo), fuzz)


@dataclass(eq=False)
class Dielectric(Material):
    refraction_index: float

    @staticmethod
    def new(refraction_index: float) -> "Dielectric":
        return Dielectric(refraction_index)


# ---- sphere (scene/sphere.rs) ------------------------------------------------------------------------
@dataclass(eq=False)
class Sphere:
    r: float
    center: Vec
    moving_center: Optional[Vec]
    mat: Material

    @staticmethod
    def new_static(r: float, center: Vec, mat: Material) -> "Sphere":
        return Sphere(r, tuple(center), None, mat)

    @staticmethod
    def new(r: float, center: Vec, moving_center: Optional[Vec], mat: Material) -> "Sphere":
        return Sphere(r, tuple(center), None if moving_center is None else tuple(moving_center), mat)


# ---- scene (scene/hittable.rs:72-99) -----------------------------------------------------------------
class FlatScene:
    """A mirt_scene_desc plus the numpy/ctypes storage that keeps its pointers alive."""

    def __init__(self, spheres, materials, textures, images, objects):
        self.sph = (abi.MirtSphere * max(1, len(spheres)))(*spheres)
        self.mat = (abi.MirtMaterial * max(1, len(materials)))(*materials)
        self.tex = (abi.MirtTexture * max(1, len(textures)))(*textures)
        self._img_arrays = [np.ascontiguousarray(a, dtype=np.uint8) for a in images]
        imgs = []
        for a in self._img_arrays:
            h, w = (a.shape[0], a.shape[1]) if a.ndim == 3 else (0, 0)
            imgs.append(abi.MirtImage(w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.img = (abi.MirtImage * max(1, len(imgs)))(*imgs)
        self.desc = abi.MirtSceneDesc(
            len(spheres), len(materials), len(textures), len(imgs),
            C.cast(self.sph, C.POINTER(abi.MirtSphere)), C.cast(self.mat, C.POINTER(abi.MirtMaterial)),
            C.cast(self.tex, C.POINTER(abi.MirtTexture)), C.cast(self.img, C.POINTER(abi.MirtImage)),
        )
        # insertion-order object list with duplicates (what the reference's BVH is built over)
        self.objects = np.asarray(objects, dtype=np.uint32)

    @property
    def n_spheres(self):
        return self.desc.n_spheres


class Scene:
    """`Scene::default()`, `add`, `build_bvh` (idempotent).  Rendering without build_bvh raises, where the
    reference panics ("expected bvh to exist when hit called", hittable.rs:93-96)."""

    def __init__(self):
        self.objects: List[Sphere] = []
        self._flat: Optional[FlatScene] = None
        self._device_scene = {}

    def add(self, obj: Sphere) -> None:
        self.objects.append(obj)

    def build_bvh(self) -> None:
        if self._flat is None:
            self._flat = flatten(self.objects)

    @property
    def flat(self) -> FlatScene:
        if self._flat is None:
            raise RuntimeError("expected bvh to exist when hit called")
        return self._flat

    def device_scene(self, device: int = 0) -> "DeviceScene":
        if device not in self._device_scene:
            self._device_scene[device] = DeviceScene(self.flat, device)
        return self._device_scene[device]


def flatten(objects: Sequence[Sphere]) -> FlatScene:
    """Flatten the Arc graph: one id per distinct object (an object added twice keeps one id - the preset
    main.rs:225 adds the same ground sphere 22 times), one slot per distinct material/texture/image."""
    sph_ids, mat_ids, tex_ids, img_ids = {}, {}, {}, {}
    spheres, materials, textures, images, order = [], [], [], [], []

    def tex_id(t: Texture) -> int:
        if id(t) in tex_ids:
            return tex_ids[id(t)]
        m = abi.MirtTexture()
        if isinstance(t, SolidColor):
            m.kind = abi.TEX_SOLID
            m.color[:] = t.albedo
        elif isinstance(t, CheckerTexture):
            m.kind = abi.TEX_CHECKER
            m.color[:] = t.even_color
            m.odd[:] = t.odd_color
            m.inv_scale = t.inv_scale
        elif isinstance(t, ImageTexture):
            m.kind = abi.TEX_IMAGE
            key = id(t.px_colors)
            if key not in img_ids:
                img_ids[key] = len(images)
                images.append(t.px_colors)
            m.image = img_ids[key]
        else:
            raise TypeError(f"unknown texture {t!r}")
        tex_ids[id(t)] = len(textures)
        textures.append(m)
        return tex_ids[id(t)]

    def mat_id(mt: Material) -> int:
        if id(mt) in mat_ids:
            return mat_ids[id(mt)]
        m = abi.MirtMaterial()
        if isinstance(mt, Lambertian):
            m.kind = abi.MAT_LAMBERTIAN
            m.texture = tex_id(mt.texture)
            m.reflectance = mt.reflectance
        elif isinstance(mt, Metal):
            m.kind = abi.MAT_METAL
            m.albedo[:] = mt.albedo
            m.has_fuzz = 0 if mt.fuzz is None else 1
            m.fuzz = 0.0 if mt.fuzz is None else mt.fuzz
        elif isinstance(mt, Dielectric):
            m.kind = abi.MAT_DIELECTRIC
            m.ior = mt.refraction_index
        else:
            raise TypeError(f"unknown material {mt!r}")
        mat_ids[id(mt)] = len(materials)
        materials.append(m)
        return mat_ids[id(mt)]

    for o in objects:
        if id(o) not in sph_ids:
            s = abi.MirtSphere()
            s.center0[:] = o.center
            s.center1[:] = o.moving_center if o.moving_center is not None else o.center
            s.radius = o.r  # clamped to max(r,0) inside the library, like sphere.rs:31
            s.material = mat_id(o.mat)
            s.moving = 0 if o.moving_center is None else 1
            sph_ids[id(o)] = len(spheres)
            spheres.append(s)
        order.append(sph_ids[id(o)])
    fs = FlatScene(spheres, materials, textures, images, order)
    fs._keepalive = list(objects)
    return fs


class DeviceScene:
    """Owns a mirt_scene handle (device-resident SoA buffers)."""

    def __init__(self, flat: FlatScene, device: int = 0):
        self.flat = flat
        self.device = device
        self.handle = C.c_void_p()
        abi.check(abi.load().mirt_scene_create(C.byref(flat.desc), device, C.byref(self.handle)))

    def close(self):
        if self.handle:
            abi.load().mirt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NodeScene:
    """Owns a mirt_node: one scene replica, stream and band buffer per GPU of `devices`; the root is devices[0]."""

    def __init__(self, flat: FlatScene, devices: Sequence[int], shared_devices: bool = False):
        self.flat = flat
        self.devices = list(devices)
        self.handle = C.c_void_p()
        ids = (C.c_int * len(self.devices))(*self.devices)
        flags = abi.NODE_SHARED_DEVICES if shared_devices else 0  # only with tests/stub_rccl: RCCL wants one rank per GPU
        abi.check(abi.load().mirt_node_create(C.byref(flat.desc), ids, len(self.devices), flags, C.byref(self.handle)))

    def close(self):
        if self.handle:
            abi.load().mirt_node_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- camera (camera/camera.rs) ------------------------------------------------------------------------
class InitError(Exception):
    """InitError::Antialiaser (camera.rs:26-29).  Uniform::new(-0.5, 0.5) cannot fail, so never raised today."""


class RenderError(Exception):
    """RenderError::{WriteHeader, WritePx} (camera.rs:21-24)."""


def f64_to_u32(value: float) -> Optional[int]:
    """utils/math.rs:5-11."""
    if math.isnan(value) or value < 0.0 or value > float(0xFFFFFFFF):
        return None
    return int(value)  # truncation toward zero, like `as u32`


DEFAULT_FLAGS = 0  # OR-ed into every mirt_opts.flags built here (tests set abi.FLAG_MX_SMALL)


def _accel_flags(accel):
    if accel not in (None, "brute", "bvh", "valu", "matrix"):
        raise ValueError(f"unknown accel {accel!r}")
    return abi.FLAG_BVH if accel == "bvh" else abi.FLAG_VALU_BROAD if accel == "valu" else abi.FLAG_MX_SMALL if accel == "matrix" else 0


@dataclass
class Camera:
    pod: abi.MirtCamera
    seed: int = 0
    precision: int = abi.F64
    device: int = 0
    last_stats: dict = field(default_factory=dict)
    chunk_samples: int = 0  # mirt_opts.chunk_samples (0 = the library's choice)
    group_chunks: int = 0  # mirt_opts.group_chunks (0 = automatic)

    @staticmethod
    def build(
        lookfrom: Optional[Vec],
        lookat: Optional[Vec],
        vup: Optional[Vec],
        img_width: int,
        ratio: float,
        aa_samples_per_px: Optional[int],
        vfov: Optional[float],
        focus_dist: Optional[float],
        defocus_angle: Optional[float],
        max_bounce_depth: Optional[int],
    ) -> "Camera":
        """Same ten positional Options and defaults as camera.rs:123-149; same arithmetic as :151-182."""
        lookfrom = (0.0, 0.0, 0.0) if lookfrom is None else tuple(map(float, lookfrom))
        lookat = (0.0, 0.0, -1.0) if lookat is None else tuple(map(float, lookat))
        vup = (0.0, 1.0, 0.0) if vup is None else tuple(map(float, vup))
        vfov = math.pi / 2.0 if vfov is None else float(vfov)
        focus_dist = 1.0 if focus_dist is None else float(focus_dist)
        defocus_angle = 0.0 if defocus_angle is None else float(defocus_angle)
        spp = 0 if aa_samples_per_px == 0 else (100 if aa_samples_per_px is None else int(aa_samples_per_px))
        max_bounce_depth = 10 if max_bounce_depth is None else int(max_bounce_depth)

        if float(img_width) / ratio < 1.0:
            img_height = 1
        else:
            img_height = f64_to_u32(float(img_width) / ratio)
            if img_height is None:
                raise OverflowError("img height in pixels should fit u32")

        h = math.tan(vfov / 2.0)
        vp_height = 2.0 * h * focus_dist
        vp_width = vp_height * (float(img_width) / float(img_height))
        w = _unit(_sub(lookfrom, lookat))
        u = _cross(vup, w)  # not normalised, as in the reference (camera.rs:164)
        v = _cross(w, u)
        vp_v = _mul(_neg(v), vp_height)
        vp_u = _mul(u, vp_width)
        px_du = _div(vp_u, float(img_width))
        px_dv = _div(vp_v, float(img_height))
        vp_upper_left = _sub(_sub(lookfrom, _mul(w, focus_dist)), _mul(_add(vp_u, vp_v), 0.5))
        px00_loc = _add(vp_upper_left, _mul(_add(px_dv, px_du), 0.5))
        defocus_radius = focus_dist * math.tan(defocus_angle * 0.5)

        pod = abi.MirtCamera()
        pod.lookfrom[:] = lookfrom
        pod.px_du[:] = px_du
        pod.px_dv[:] = px_dv
        pod.vp_upper_left[:] = vp_upper_left
        pod.px00_loc[:] = px00_loc
        pod.disk_u_r[:] = _mul(u, defocus_radius)
        pod.disk_v_r[:] = _mul(v, defocus_radius)
        pod.defocus_angle = defocus_angle
        pod.img_width = img_width
        pod.img_height = img_height
        pod.samples_per_pixel = spp
        pod.max_bounce_depth = max_bounce_depth
        return Camera(pod)

    # -- geometry helpers -------------------------------------------------------------------------
    @property
    def width(self):
        return self.pod.img_width

    @property
    def height(self):
        return self.pod.img_height

    def opts(self, precision=None, seed=None, strip_rows=0, part_index=0, part_count=1, flags=0, chunk_samples=None, group_chunks=None,
             bvh_keep=0) -> abi.MirtOpts:
        """mirt_opts for a call.  Tuning travels in the struct (the library reads no environment variable): this camera's
        `chunk_samples` / `group_chunks` attributes and the module-wide DEFAULT_FLAGS fill what the caller leaves open."""
        return abi.MirtOpts(
            self.precision if precision is None else precision, flags | DEFAULT_FLAGS, self.seed if seed is None else seed,
            strip_rows, part_index, part_count, self.chunk_samples if chunk_samples is None else chunk_samples,
            self.group_chunks if group_chunks is None else group_chunks, bvh_keep,
        )

    # -- the hot path: everything below runs in the HIP library -------------------------------------
    def render_arrays(self, scene: Scene, precision=None, seed=None, want_linear=True, want_rgb8=True, accel=None):
        """Camera::render without the text emission: returns (linear (H,W,3) real | None, rgb8 (H,W,3) u8 | None).
        accel="bvh" traverses the bounding-ball tree (abi.FLAG_BVH, f32 only) instead of testing every sphere;
        accel="valu" keeps the brute-force broad phase on the vector ALU (abi.FLAG_VALU_BROAD)."""
        ds = scene.device_scene(self.device)
        prec = self.precision if precision is None else precision
        H, W = self.height, self.width
        lin = np.empty((H, W, 3), dtype=np.float64 if prec == abi.F64 else np.float32) if want_linear else None
        rgb8 = np.empty((H, W, 3), dtype=np.uint8) if want_rgb8 else None
        st = abi.MirtStats()
        o = self.opts(precision=prec, seed=seed, flags=_accel_flags(accel))
        abi.check(
            abi.load().mirt_render(
                ds.handle, C.byref(self.pod), C.byref(o),
                None if lin is None else lin.ctypes.data, None if rgb8 is None else rgb8.ctypes.data, C.byref(st),
            )
        )
        self.last_stats = st.asdict()
        return lin, rgb8

    def render_arrays_node(self, node: "NodeScene", precision=None, seed=None, want_linear=True, want_rgb8=True, accel=None,
                           gather=abi.GATHER_TO_ROOT, strip_rows=0):
        """render_arrays on all GPUs of `node` (mirt_node_render: interleaved strips, one RCCL exchange, K4 on the root);
        the same image bit for bit.  Returns (linear, rgb8); self.last_stats holds the per-device figures."""
        prec = self.precision if precision is None else precision
        H, W = self.height, self.width
        lin = np.empty((H, W, 3), dtype=np.float64 if prec == abi.F64 else np.float32) if want_linear else None
        rgb8 = np.empty((H, W, 3), dtype=np.uint8) if want_rgb8 else None
        st = abi.MirtNodeStats()
        o = self.opts(precision=prec, seed=seed, flags=_accel_flags(accel), strip_rows=strip_rows, part_count=0)
        abi.check(abi.load().mirt_node_render(node.handle, C.byref(self.pod), C.byref(o), gather,
                                              None if lin is None else lin.ctypes.data, None if rgb8 is None else rgb8.ctypes.data, C.byref(st)))
        n = st.n_devices
        self.last_stats = {"segments": st.segments, "n_devices": n, "gather": st.gather, "wall_ms": st.wall_ms, "gather_ms": st.gather_ms,
                           "scatter_ms": st.scatter_ms, "band_kernel_ms": list(st.band_kernel_ms[:n]), "band_segments": list(st.band_segments[:n])}
        return lin, rgb8

    def pass_granularity(self, precision=None) -> int:
        """Samples per work item for this image (mirt_pass_granularity): progressive passes begin and end on multiples of it."""
        o = self.opts(precision=self.precision if precision is None else precision)
        return int(abi.load().mirt_pass_granularity(C.byref(self.pod), C.byref(o)))

    def render_pass(self, scene: Scene, first: int, count: int, accum: np.ndarray, precision=None, seed=None, want_linear=True, want_rgb8=True, accel=None):
        """Progressive rendering / checkpointing (mirt_render_pass): add samples [first, first + count) of every pixel to
        `accum` ((H,W,3) radiance sums, in place; save it and `first + count` to checkpoint) and return the image so far as
        render_arrays would.  After the last pass the image equals render_arrays' bit for bit."""
        ds = scene.device_scene(self.device)
        prec = self.precision if precision is None else precision
        H, W = self.height, self.width
        real = np.float64 if prec == abi.F64 else np.float32
        if accum.shape != (H, W, 3) or accum.dtype != real or not accum.flags["C_CONTIGUOUS"]:
            raise ValueError(f"accum must be a contiguous ({H},{W},3) {np.dtype(real).name} array")
        lin = np.empty((H, W, 3), dtype=real) if want_linear else None
        rgb8 = np.empty((H, W, 3), dtype=np.uint8) if want_rgb8 else None
        st = abi.MirtStats()
        o = self.opts(precision=prec, seed=seed, flags=_accel_flags(accel))
        abi.check(
            abi.load().mirt_render_pass(
                ds.handle, C.byref(self.pod), C.byref(o), first, count, accum.ctypes.data,
                None if lin is None else lin.ctypes.data, None if rgb8 is None else rgb8.ctypes.data, C.byref(st),
            )
        )
        self.last_stats = st.asdict()
        return lin, rgb8

    def primary_hits(self, scene: Scene, precision=None, accel=None):
        """Hit-id / t map of the deterministic branch (camera.rs:247-254)."""
        ds = scene.device_scene(self.device)
        prec = self.precision if precision is None else precision
        H, W = self.height, self.width
        ids = np.empty((H, W), dtype=np.uint32)
        t = np.empty((H, W), dtype=np.float64 if prec == abi.F64 else np.float32)
        o = self.opts(precision=prec, flags=_accel_flags(accel))
        abi.check(abi.load().mirt_primary_hits(ds.handle, C.byref(self.pod), C.byref(o), ids.ctypes.data, t.ctypes.data))
        return ids, t

    def closest_hit_kind(self, scene: Scene, precision=None, accel=None) -> str:
        """Which closest-hit front end a call with these arguments runs (mirt_closest_hit_kind)."""
        ds = scene.device_scene(self.device)
        o = self.opts(precision=self.precision if precision is None else precision, flags=_accel_flags(accel))
        k = abi.load().mirt_closest_hit_kind(ds.handle, C.byref(self.pod), C.byref(o))
        if k < 0:
            abi.check(k)
        return ("valu", "matrix", "bvh", "f64", "f64-matrix", "tiny")[k]

    def render(self, scene: Scene, out=None) -> None:
        """Camera::render: PPM P3 on stdout - "P3\\nW H\\n255\\n" then one "r g b\\n" line per pixel
        (camera.rs:233-235,255; rgb.rs:28-36).  A NaN pixel makes the library return MIRT_ERR_NAN
        where the reference panics."""
        out = sys.stdout.buffer if out is None else out
        try:
            out.write(f"P3\n{self.width} {self.height}\n255\n".encode())
        except OSError as e:
            raise RenderError(f"WriteHeader: {e}") from e
        _, rgb8 = self.render_arrays(scene, want_linear=False)
        try:
            out.write(ppm_body(rgb8))
        except OSError as e:
            raise RenderError(f"WritePx: {e}") from e


_DEC = np.zeros((256, 4), np.uint8)  # decimal digits of 0..255 followed by a separator slot, and their lengths
_DEC_LEN = np.zeros(256, np.int64)
for _v in range(256):
    _d = str(_v).encode()
    _DEC[_v, : len(_d)] = np.frombuffer(_d, np.uint8)
    _DEC_LEN[_v] = len(_d)


def ppm_body(rgb8: np.ndarray) -> bytes:
    """The "r g b\\n" lines of a frame (rgb.rs:28-36 through Display), vectorised: digits from a 256-entry table, a
    separator after each value, the unused digit slots dropped."""
    v = np.ascontiguousarray(rgb8, np.uint8).reshape(-1)
    cells = _DEC[v].copy()  # (n, 4): up to three digits, then room for the separator
    n = _DEC_LEN[v]
    sep = np.where(np.arange(v.size) % 3 == 2, ord("\n"), ord(" ")).astype(np.uint8)
    cells[np.arange(v.size), n] = sep
    keep = np.arange(4)[None, :] <= n[:, None]
    return cells[keep].tobytes()
