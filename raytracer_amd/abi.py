"""ctypes mirror of include/mirt.h and loader for the HIP library (libmirt.so).

There is no CPU fallback: if the in-tree library is missing, loading fails loudly with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

MISS_ID = 0xFFFFFFFF
F64, F32 = 64, 32
FLAG_COMPACT_ROWS = 1
FLAG_BVH = 2  # closest hit through the bounding-ball tree (f32 only); same image bit for bit
FLAG_DEFER_STATS = 8  # _device render calls: return at once, mirt_render_wait() collects the statistics
FLAG_VALU_BROAD = 4  # f32 brute force: broad phase on the vector ALU instead of the matrix pipe; same image bit for bit
FLAG_ALL_PAIRS = 32  # matrix-pipe kernels: no tile cull, every (ray, sphere) pair is tested; same image bit for bit
FLAG_MX_SMALL = 16  # matrix-pipe broad phase also for scenes below its size threshold (tests, tuning); same image bit for bit
STATIC_SCRATCH_BYTES = 128 << 20  # MIRT_STATIC_SCRATCH_BYTES: the device array behind mirt_render_device_static
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2
TEX_SOLID, TEX_CHECKER, TEX_IMAGE = 0, 1, 2

_D3 = C.c_double * 3


class MirtImage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class MirtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("image", C.c_uint32), ("color", _D3), ("odd", _D3), ("inv_scale", C.c_double)]


class MirtMaterial(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("texture", C.c_uint32),
        ("reflectance", C.c_double),
        ("albedo", _D3),
        ("fuzz", C.c_double),
        ("has_fuzz", C.c_uint32),
        ("_pad", C.c_uint32),
        ("ior", C.c_double),
    ]


class MirtSphere(C.Structure):
    _fields_ = [("center0", _D3), ("center1", _D3), ("radius", C.c_double), ("material", C.c_uint32), ("moving", C.c_uint32)]


class MirtSceneDesc(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_uint32),
        ("n_materials", C.c_uint32),
        ("n_textures", C.c_uint32),
        ("n_images", C.c_uint32),
        ("spheres", C.POINTER(MirtSphere)),
        ("materials", C.POINTER(MirtMaterial)),
        ("textures", C.POINTER(MirtTexture)),
        ("images", C.POINTER(MirtImage)),
    ]


class MirtCamera(C.Structure):
    _fields_ = [
        ("lookfrom", _D3),
        ("px_du", _D3),
        ("px_dv", _D3),
        ("vp_upper_left", _D3),
        ("px00_loc", _D3),
        ("disk_u_r", _D3),
        ("disk_v_r", _D3),
        ("defocus_angle", C.c_double),
        ("img_width", C.c_uint32),
        ("img_height", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("max_bounce_depth", C.c_uint32),
    ]


class MirtOpts(C.Structure):
    _fields_ = [
        ("precision", C.c_uint32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint64),
        ("strip_rows", C.c_uint32),
        ("part_index", C.c_uint32),
        ("part_count", C.c_uint32),
        ("chunk_samples", C.c_uint32),  # lower bound of the samples per work item (0 = 1): the one tuning field that reaches the image's last bits
        ("group_chunks", C.c_uint32),  # image-neutral tuning: chunks of a pixel kept together in the queue order (0 = automatic)
        ("bvh_keep", C.c_uint32),  # image-neutral tuning, FLAG_BVH: suspension threshold in 64ths (0 = automatic)
        ("_reserved", C.c_uint32 * 2),
    ]


class MirtStats(C.Structure):
    _fields_ = [
        ("segments", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("static_tests", C.c_uint64),
        ("moving_tests", C.c_uint64),
        ("narrow_tests", C.c_uint64),
        ("wave_iterations", C.c_uint64),
        ("nan_pixels", C.c_uint64),
        ("kernel_ms", C.c_double),  # K2 + K2b
        ("tree_segments", C.c_uint64),
        ("resolve_ms", C.c_double),  # of kernel_ms: K2b alone
        ("mx_tiles", C.c_uint64),  # (wave, tile) evaluations of the matrix-pipe kernels: what the tile cull left
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


NODE_MAX_DEVICES = 16
NODE_SHARED_DEVICES = 1  # mirt_node_create flag (test rigs only)
GATHER_TO_ROOT, GATHER_ALLGATHER = 0, 1  # mirt_gather_kind: grouped ncclSend/ncclRecv to the root | ncclAllGather


class MirtNodeStats(C.Structure):
    _fields_ = [
        ("segments", C.c_uint64),
        ("nan_pixels", C.c_uint64),
        ("n_devices", C.c_uint32),
        ("gather", C.c_uint32),
        ("wall_ms", C.c_double),
        ("gather_ms", C.c_double),
        ("scatter_ms", C.c_double),
        ("band_kernel_ms", C.c_double * NODE_MAX_DEVICES),
        ("band_segments", C.c_uint64 * NODE_MAX_DEVICES),
    ]


# every symbol include/mirt.h declares (tests/test_abi.py checks the built library exports them all)
EXPORTS = [
    "mirt_abi_version",
    "mirt_device_count",
    "mirt_last_error",
    "mirt_scene_create",
    "mirt_scene_destroy",
    "mirt_primary_hits",
    "mirt_render",
    "mirt_render_device",
    "mirt_primary_hits_device",
    "mirt_quantize_device",
    "mirt_deinterleave_device",
    "mirt_part_rows",
    "mirt_closest_hit_kind",
    "mirt_render_pass_device",
    "mirt_render_pass",
    "mirt_pass_granularity",
    "mirt_render_wait",
    "mirt_render_device_static",
    "mirt_philox4x32_10",
    "mirt_measure_fma_peak",
    "mirt_node_create",
    "mirt_node_destroy",
    "mirt_node_device_count",
    "mirt_node_render",
    "mirt_node_render_device",
]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRT_LIB") or os.path.join(_PKG_DIR, "csrc", "libmirt.so")  # MIRT_LIB: A/B builds of the kernels
_lib = None


class MirtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"mirt error {code}: {text}")
        self.code = code


def load():
    """Load libmirt.so (built in-tree by __graft_entry__.build() / raytracer_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found - the HIP extension is not built. Run `make -C raytracer_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 and cannot initialise the GPU once
    # /opt/rocm's copy has been loaded (and vice versa).  Importing torch first makes the dynamic linker resolve
    # libmirt.so's libamdhip64.so.7 to the copy torch already loaded, so torch tensors (device memory, streams,
    # torch.distributed/RCCL) and the mirt kernels share a runtime.  Without torch the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER
    lib.mirt_abi_version.restype = C.c_int
    lib.mirt_device_count.restype = C.c_int
    lib.mirt_last_error.restype = C.c_char_p
    lib.mirt_scene_create.argtypes = [P(MirtSceneDesc), C.c_int, P(vp)]
    lib.mirt_scene_destroy.argtypes = [vp]
    lib.mirt_scene_destroy.restype = None
    lib.mirt_primary_hits.argtypes = [vp, P(MirtCamera), P(MirtOpts), vp, vp]
    lib.mirt_render.argtypes = [vp, P(MirtCamera), P(MirtOpts), vp, vp, P(MirtStats)]
    lib.mirt_render_device.argtypes = [vp, P(MirtCamera), P(MirtOpts), vp, vp, vp, P(MirtStats)]
    lib.mirt_primary_hits_device.argtypes = [vp, P(MirtCamera), P(MirtOpts), vp, vp, vp]
    lib.mirt_quantize_device.argtypes = [vp, u32, sz, vp, vp, P(u64)]
    lib.mirt_deinterleave_device.argtypes = [vp, sz, u32, u32, u32, u32, u32, vp, vp]
    lib.mirt_part_rows.argtypes = [u32, u32, u32, u32]
    lib.mirt_part_rows.restype = u32
    lib.mirt_render_pass_device.argtypes = [vp, P(MirtCamera), P(MirtOpts), u32, u32, vp, vp, vp, vp, P(MirtStats)]
    lib.mirt_render_pass.argtypes = [vp, P(MirtCamera), P(MirtOpts), u32, u32, vp, vp, vp, P(MirtStats)]
    lib.mirt_pass_granularity.argtypes = [P(MirtCamera), P(MirtOpts)]
    lib.mirt_pass_granularity.restype = u32
    lib.mirt_render_device_static.argtypes = [vp, P(MirtCamera), P(MirtOpts), vp, vp, vp, P(MirtStats)]
    lib.mirt_render_wait.argtypes = [vp, P(MirtStats)]
    lib.mirt_render_wait.restype = C.c_int
    lib.mirt_closest_hit_kind.argtypes = [vp, P(MirtCamera), P(MirtOpts)]
    lib.mirt_closest_hit_kind.restype = C.c_int
    lib.mirt_node_create.argtypes = [P(MirtSceneDesc), P(C.c_int), u32, u32, P(vp)]
    lib.mirt_node_destroy.argtypes = [vp]
    lib.mirt_node_destroy.restype = None
    lib.mirt_node_device_count.argtypes = [vp]
    lib.mirt_node_device_count.restype = u32
    lib.mirt_node_render.argtypes = [vp, P(MirtCamera), P(MirtOpts), u32, vp, vp, P(MirtNodeStats)]
    lib.mirt_node_render_device.argtypes = [vp, P(MirtCamera), P(MirtOpts), u32, vp, vp, P(MirtNodeStats)]
    lib.mirt_measure_fma_peak.argtypes = [C.c_int, P(C.c_double)]
    lib.mirt_philox4x32_10.argtypes = [P(u32), P(u32), P(u32)]
    lib.mirt_philox4x32_10.restype = None
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise MirtError(code, load().mirt_last_error().decode("utf-8", "replace"))
