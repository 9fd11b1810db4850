//! Raw FFI declarations for include/mirt.h (ABI version 3).  Layouts are `#[repr(C)]` copies of the C structs;
//! `tests/test_abi.py` pins the C side's sizes and offsets, and the constants below restate them so that a
//! mismatch fails at compile time on the Rust side.  This file has never seen a compiler (no cargo / rustc in
//! the build image): `tests/test_ffi_drift.py` parses it next to the header and fails when the two disagree on
//! an entry point, its arity, a constant, a struct's fields or a struct's size.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const MIRT_ABI_VERSION: u32 = 3;
pub const MIRT_MISS_ID: u32 = 0xFFFF_FFFF;

pub const MIRT_OK: c_int = 0;
pub const MIRT_ERR_INVALID: c_int = -1;
pub const MIRT_ERR_NO_DEVICE: c_int = -2;
pub const MIRT_ERR_HIP: c_int = -3;
pub const MIRT_ERR_NAN: c_int = -4;
pub const MIRT_ERR_NOMEM: c_int = -5;
pub const MIRT_ERR_TOO_LARGE: c_int = -6;

pub const MIRT_MAT_LAMBERTIAN: u32 = 0;
pub const MIRT_MAT_METAL: u32 = 1;
pub const MIRT_MAT_DIELECTRIC: u32 = 2;
pub const MIRT_TEX_SOLID: u32 = 0;
pub const MIRT_TEX_CHECKER: u32 = 1;
pub const MIRT_TEX_IMAGE: u32 = 2;
pub const MIRT_F64: u32 = 64;
pub const MIRT_F32: u32 = 32;
pub const MIRT_FLAG_COMPACT_ROWS: u32 = 1;
pub const MIRT_FLAG_BVH: u32 = 2;
pub const MIRT_FLAG_VALU_BROAD: u32 = 4;
pub const MIRT_FLAG_DEFER_STATS: u32 = 8;
pub const MIRT_FLAG_MX_SMALL: u32 = 16; // tuning / test switch, image-neutral
pub const MIRT_FLAG_ALL_PAIRS: u32 = 32; // matrix-pipe kernels: no tile cull (measurements), image-neutral
pub const MIRT_STATIC_SCRATCH_BYTES: u64 = 128 << 20; // the device array behind mirt_render_device_static

#[repr(C)]
#[derive(Clone, Copy)]
pub struct mirt_image {
    pub width: u32,
    pub height: u32,
    pub rgb: *const u8,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mirt_texture {
    pub kind: u32,
    pub image: u32,
    pub color: [f64; 3],
    pub odd: [f64; 3],
    pub inv_scale: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mirt_material {
    pub kind: u32,
    pub texture: u32,
    pub reflectance: f64,
    pub albedo: [f64; 3],
    pub fuzz: f64,
    pub has_fuzz: u32,
    pub _pad: u32,
    pub ior: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mirt_sphere {
    pub center0: [f64; 3],
    pub center1: [f64; 3],
    pub radius: f64,
    pub material: u32,
    pub moving: u32,
}

#[repr(C)]
pub struct mirt_scene_desc {
    pub n_spheres: u32,
    pub n_materials: u32,
    pub n_textures: u32,
    pub n_images: u32,
    pub spheres: *const mirt_sphere,
    pub materials: *const mirt_material,
    pub textures: *const mirt_texture,
    pub images: *const mirt_image,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mirt_camera {
    pub lookfrom: [f64; 3],
    pub px_du: [f64; 3],
    pub px_dv: [f64; 3],
    pub vp_upper_left: [f64; 3],
    pub px00_loc: [f64; 3],
    pub disk_u_r: [f64; 3],
    pub disk_v_r: [f64; 3],
    pub defocus_angle: f64,
    pub img_width: u32,
    pub img_height: u32,
    pub samples_per_pixel: u32,
    pub max_bounce_depth: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct mirt_opts {
    pub precision: u32,
    pub flags: u32,
    pub seed: u64,
    pub strip_rows: u32,
    pub part_index: u32,
    pub part_count: u32,
    pub chunk_samples: u32, // lower bound of the samples per work item (0 = 1); the one tuning field that reaches the image
    pub group_chunks: u32,  // image-neutral tuning (0 = automatic)
    pub bvh_keep: u32,      // image-neutral tuning, MIRT_FLAG_BVH (0 = automatic)
    pub _reserved: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct mirt_stats {
    pub segments: u64,
    pub sphere_tests: u64,
    pub static_tests: u64,
    pub moving_tests: u64,
    pub narrow_tests: u64,
    pub wave_iterations: u64,
    pub nan_pixels: u64,
    pub kernel_ms: f64,
    pub tree_segments: u64,
    pub resolve_ms: f64, // of kernel_ms (K2 + K2b): the resolve kernel alone
    pub mx_tiles: u64,   // (wave, tile of 32 spheres) evaluations of the matrix-pipe kernels: what the tile cull left
}

pub const MIRT_NODE_MAX_DEVICES: usize = 16;
pub const MIRT_NODE_SHARED_DEVICES: u32 = 1; // mirt_node_create flag, test rigs only
pub const MIRT_GATHER_TO_ROOT: u32 = 0;
pub const MIRT_GATHER_ALLGATHER: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mirt_node_stats {
    pub segments: u64,
    pub nan_pixels: u64,
    pub n_devices: u32,
    pub gather: u32,
    pub wall_ms: f64,
    pub gather_ms: f64,
    pub scatter_ms: f64,
    pub band_kernel_ms: [f64; MIRT_NODE_MAX_DEVICES],
    pub band_segments: [u64; MIRT_NODE_MAX_DEVICES],
}

#[repr(C)]
pub struct mirt_node {
    _opaque: [u8; 0],
}

#[repr(C)]
pub struct mirt_scene {
    _opaque: [u8; 0],
}

// Sizes as pinned on the C side (tests/test_abi.py).
const _: () = assert!(std::mem::size_of::<mirt_image>() == 16);
const _: () = assert!(std::mem::size_of::<mirt_scene_desc>() == 48);
const _: () = assert!(std::mem::size_of::<mirt_texture>() == 64);
const _: () = assert!(std::mem::size_of::<mirt_material>() == 64);
const _: () = assert!(std::mem::size_of::<mirt_sphere>() == 64);
const _: () = assert!(std::mem::size_of::<mirt_camera>() == 192);
const _: () = assert!(std::mem::size_of::<mirt_opts>() == 48);
const _: () = assert!(std::mem::size_of::<mirt_stats>() == 88);
const _: () = assert!(std::mem::size_of::<mirt_node_stats>() == 304);

extern "C" {
    pub fn mirt_abi_version() -> c_int;
    pub fn mirt_measure_fma_peak(device: c_int, tflops: *mut f64) -> c_int;
    pub fn mirt_node_create(desc: *const mirt_scene_desc, device_ids: *const c_int, n_devices: u32, flags: u32, out: *mut *mut mirt_node) -> c_int;
    pub fn mirt_node_destroy(node: *mut mirt_node);
    pub fn mirt_node_device_count(node: *const mirt_node) -> u32;
    pub fn mirt_node_render(node: *mut mirt_node, cam: *const mirt_camera, opts: *const mirt_opts, gather: u32, rgb_linear: *mut c_void, rgb8: *mut u8, stats: *mut mirt_node_stats) -> c_int;
    pub fn mirt_node_render_device(node: *mut mirt_node, cam: *const mirt_camera, opts: *const mirt_opts, gather: u32, d_rgb_linear: *mut c_void, d_rgb8: *mut u8, stats: *mut mirt_node_stats) -> c_int;
    pub fn mirt_device_count() -> c_int;
    pub fn mirt_last_error() -> *const c_char;
    pub fn mirt_scene_create(desc: *const mirt_scene_desc, device: c_int, out: *mut *mut mirt_scene) -> c_int;
    pub fn mirt_scene_destroy(scene: *mut mirt_scene);
    pub fn mirt_primary_hits(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, id: *mut u32, t: *mut c_void) -> c_int;
    pub fn mirt_render(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, rgb_linear: *mut c_void, rgb8: *mut u8, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_render_device_static(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, d_rgb_linear: *mut c_void, d_rgb8: *mut u8, stream: *mut c_void, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_render_device(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, d_rgb_linear: *mut c_void, d_rgb8: *mut u8, stream: *mut c_void, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_primary_hits_device(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, d_id: *mut u32, d_t: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn mirt_quantize_device(d_linear: *const c_void, precision: u32, n: usize, d_out: *mut u8, stream: *mut c_void, nan_count: *mut u64) -> c_int;
    pub fn mirt_deinterleave_device(d_bands: *const c_void, band_stride_bytes: usize, part_count: u32, strip_rows: u32, width: u32, height: u32, elem_bytes: u32, d_frame: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn mirt_part_rows(height: u32, strip_rows: u32, part_index: u32, part_count: u32) -> u32;
    pub fn mirt_render_pass_device(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, first_sample: u32, sample_count: u32, d_accum: *mut c_void, d_rgb_linear: *mut c_void, d_rgb8: *mut u8, stream: *mut c_void, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_render_pass(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts, first_sample: u32, sample_count: u32, accum: *mut c_void, rgb_linear: *mut c_void, rgb8: *mut u8, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_render_wait(scene: *const mirt_scene, stats: *mut mirt_stats) -> c_int;
    pub fn mirt_pass_granularity(cam: *const mirt_camera, opts: *const mirt_opts) -> u32;
    pub fn mirt_closest_hit_kind(scene: *const mirt_scene, cam: *const mirt_camera, opts: *const mirt_opts) -> c_int;
    pub fn mirt_philox4x32_10(ctr: *const u32, key: *const u32, out: *mut u32);
}
