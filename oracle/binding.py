"""ctypes binding of oracle/_build/liboracle.so for tests/, smoke() and bench.py's cpu_baseline leg ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _DIR, "-s"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double
    P = C.POINTER
    lib.oracle_philox4x32_10.argtypes = [P(u32), P(u32), P(u32)]
    lib.oracle_philox4x32_10.restype = None
    lib.oracle_camera_build.argtypes = [P(dbl), P(dbl), P(dbl), u32, dbl, C.c_int64, C.c_int, dbl, C.c_int, dbl, C.c_int, dbl,
                                        C.c_int, u32, P(abi.MirtCamera)]
    lib.oracle_primary_hits.argtypes = [P(abi.MirtSceneDesc), P(abi.MirtCamera), u32, vp, u32, vp, vp]
    lib.oracle_render.argtypes = [P(abi.MirtSceneDesc), P(abi.MirtCamera), u32, u64, vp, u32, C.c_int, u32, u32, vp, vp, P(u64)]
    lib.oracle_render_spec.argtypes = [P(abi.MirtSceneDesc), P(abi.MirtCamera), u32, u64, vp, u32, C.c_int, u32, u32, u32, vp, vp, P(u64)]
    lib.oracle_render_ex.argtypes = [P(abi.MirtSceneDesc), P(abi.MirtCamera), u32, u32, u64, vp, u32, C.c_int, u32, u32, u32, vp, vp, P(u64), vp]
    lib.oracle_quantize.argtypes = [vp, u32, C.c_size_t, vp]
    lib.oracle_sphere_hit.argtypes = [P(abi.MirtSphere), P(dbl), P(dbl), dbl, dbl, dbl, P(dbl)]
    lib.oracle_aabb_slab.argtypes = [P(dbl), P(dbl), P(dbl), P(dbl), C.c_int, dbl, dbl]
    lib.oracle_scatter.argtypes = [P(abi.MirtSceneDesc), u32, P(dbl), P(dbl), dbl, P(dbl), u64, u32, u32, u32, P(dbl)]
    lib.oracle_texture_color.argtypes = [P(abi.MirtSceneDesc), u32, dbl, dbl, P(dbl), P(dbl)]
    lib.oracle_draw_dim.argtypes = [u64, u32, u32, u32, u32, u32]
    lib.oracle_draw_dim.restype = dbl
    _lib = lib
    return lib


def _d3(v):
    return (C.c_double * 3)(*v) if v is not None else None


def _real(precision):
    return np.float64 if precision == abi.F64 else np.float32


def camera_build(lookfrom, lookat, vup, img_width, ratio, aa, vfov, focus_dist, defocus_angle, max_depth) -> abi.MirtCamera:
    out = abi.MirtCamera()
    rc = load().oracle_camera_build(
        _d3(lookfrom), _d3(lookat), _d3(vup), img_width, ratio, -1 if aa is None else aa,
        vfov is not None, vfov or 0.0, focus_dist is not None, focus_dist or 0.0,
        defocus_angle is not None, defocus_angle or 0.0, max_depth is not None, max_depth or 0, C.byref(out))
    if rc != 0:
        raise OverflowError("img height in pixels should fit u32")
    return out


def primary_hits(flat, cam_pod, precision=abi.F64, use_bvh=False):
    H, W = cam_pod.img_height, cam_pod.img_width
    ids = np.empty((H, W), np.uint32)
    t = np.empty((H, W), _real(precision))
    objs = flat.objects if use_bvh else None
    load().oracle_primary_hits(C.byref(flat.desc), C.byref(cam_pod), precision,
                               None if objs is None else objs.ctypes.data, 0 if objs is None else len(objs),
                               ids.ctypes.data, t.ctypes.data)
    return ids, t


def render(flat, cam_pod, precision=abi.F64, seed=0, use_bvh=False, threads=1, rows=None, want_rgb8=True, spec_chunk=0):
    """spec_chunk = 0: the reference's pixel arithmetic (recursion-order product, sequential sample sum).
    spec_chunk = Q >= 1: the throughput kernels' (running product; sums of Q samples added in chunk order) - pass
    mirt_pass_granularity's value (1 when anti-aliasing is off)."""
    H, W = cam_pod.img_height, cam_pod.img_width
    lin = np.zeros((H, W, 3), _real(precision))
    rgb8 = np.zeros((H, W, 3), np.uint8) if want_rgb8 else None
    seg = C.c_uint64(0)
    objs = flat.objects if use_bvh else None
    r0, r1 = (0, 0) if rows is None else rows
    common = (C.byref(flat.desc), C.byref(cam_pod), precision, seed, None if objs is None else objs.ctypes.data, 0 if objs is None else len(objs), threads, r0, r1)
    outs = (lin.ctypes.data, None if rgb8 is None else rgb8.ctypes.data, C.byref(seg))
    if spec_chunk:
        nan = load().oracle_render_spec(*common, spec_chunk, *outs)
    else:
        nan = load().oracle_render(*common, *outs)
    return lin, rgb8, int(seg.value), nan


def render_ex(flat, cam_pod, precision=abi.F64, draws=0, seed=0, use_bvh=False, threads=1, spec_chunk=0, want_rgb8=True):
    """oracle_render_ex: as render(), plus `draws` = 32 with precision F64 - the reference's f64 arithmetic fed with the
    uniforms of the f32 mode (the frame an MIRT_F32 render is held to, per pixel, in tests/test_gpu_f32_vs_reference.py) -
    and the per-pixel path signatures.  Returns (linear, rgb8, segments, signatures)."""
    H, W = cam_pod.img_height, cam_pod.img_width
    lin = np.zeros((H, W, 3), _real(precision))
    rgb8 = np.zeros((H, W, 3), np.uint8) if want_rgb8 else None
    sig = np.zeros((H, W), np.uint64)
    seg = C.c_uint64(0)
    objs = flat.objects if use_bvh else None
    rc = load().oracle_render_ex(C.byref(flat.desc), C.byref(cam_pod), precision, draws, seed, None if objs is None else objs.ctypes.data,
                                 0 if objs is None else len(objs), threads, 0, 0, spec_chunk, lin.ctypes.data,
                                 None if rgb8 is None else rgb8.ctypes.data, C.byref(seg), sig.ctypes.data)
    if rc < 0:
        raise ValueError("oracle_render_ex: unsupported precision / draws combination")
    return lin, rgb8, int(seg.value), sig


def quantize(lin: np.ndarray):
    prec = abi.F64 if lin.dtype == np.float64 else abi.F32
    lin = np.ascontiguousarray(lin)
    out = np.empty(lin.size, np.uint8)
    nan = load().oracle_quantize(lin.ctypes.data, prec, lin.size, out.ctypes.data)
    return out.reshape(lin.shape), nan


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    load().oracle_philox4x32_10(c, k, o)
    return list(o)


def draw_dim(seed, pixel, sample, bounce, dim, precision=abi.F64):
    """Dimension `dim` of the RNG stream (seed; pixel, sample, bounce) as the contract defines it for `precision`."""
    return load().oracle_draw_dim(seed, pixel, sample, bounce, dim, precision)


def draws(seed, pixel, sample, bounce, blk, precision=abi.F64):
    """f64 helper kept for the tests: the two uniforms of Philox block `blk` = dims 2*blk and 2*blk+1."""
    return draw_dim(seed, pixel, sample, bounce, 2 * blk, precision), draw_dim(seed, pixel, sample, bounce, 2 * blk + 1, precision)


def sphere_hit(sphere: abi.MirtSphere, o, d, tm=0.0, tmin=0.001, tmax=float("inf")):
    out = (C.c_double * 11)()
    hit = load().oracle_sphere_hit(C.byref(sphere), _d3(o), _d3(d), tm, tmin, tmax, out)
    if not hit:
        return None
    return dict(t=out[1], p=tuple(out[2:5]), n=tuple(out[5:8]), front=bool(out[8]), u=out[9], v=out[10])


def aabb_slab(bmin, bmax, o, d, axis, tmin, tmax):
    return bool(load().oracle_aabb_slab(_d3(bmin), _d3(bmax), _d3(o), _d3(d), axis, tmin, tmax))


def scatter(flat, material, rin_o, rin_d, tm, p, n, front, u, v, seed, pixel, sample, bounce):
    hit = (C.c_double * 9)(*p, *n, 1.0 if front else 0.0, u, v)
    out = (C.c_double * 11)()
    ok = load().oracle_scatter(C.byref(flat.desc), material, _d3(rin_o), _d3(rin_d), tm, hit, seed, pixel, sample, bounce, out)
    return bool(ok), tuple(out[1:4]), tuple(out[4:7]), tuple(out[7:10]), out[10]


def texture_color(flat, texture, u, v, p):
    rgb = (C.c_double * 3)()
    load().oracle_texture_color(C.byref(flat.desc), texture, u, v, _d3(p), rgb)
    return tuple(rgb)
