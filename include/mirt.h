/*
 * mirt.h — C ABI of the MI355X (gfx950) render backend for the sphere path tracer.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own; the seam this
 * library replaces is the Rust call pair
 *     Camera::build(..)            reference src/camera/camera.rs:123-206   (stays on the host side)
 *     Camera::render(&self,&Scene) reference src/camera/camera.rs:232-259   (replaced by mirt_render)
 * plus the scene container `Scene::{add,build_bvh,hit}` (src/scene/hittable.rs:72-99), which is
 * flattened into a mirt_scene_desc when `build_bvh`/`render` is called.
 *
 * Conventions: every entry point returns 0 (MIRT_OK) or a negative mirt_status; nothing aborts or
 * unwinds across the ABI; output buffers are caller-owned; mirt_last_error() is thread-local.
 * Plain pointers and sizes only - no C++/torch types.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_ABI_VERSION 3u /* 3: mirt_stats.resolve_ms */
#define MIRT_MISS_ID 0xFFFFFFFFu /* hit-id of a ray that hits nothing (SURVEY.md §8c "Hit id definition") */

typedef enum mirt_status {
    MIRT_OK = 0,
    MIRT_ERR_INVALID = -1,   /* bad argument / malformed scene description                         */
    MIRT_ERR_NO_DEVICE = -2, /* no HIP device, or device index out of range                        */
    MIRT_ERR_HIP = -3,       /* a HIP runtime call failed (text in mirt_last_error)                */
    MIRT_ERR_NAN = -4,       /* a NaN reached the quantiser: the reference panics here             */
                             /* (src/core/rgb.rs:33 `expect("f64 is nan!")`); we return a code     */
    MIRT_ERR_NOMEM = -5,
    MIRT_ERR_TOO_LARGE = -6  /* a size exceeds a documented limit                                  */
} mirt_status;

/* ---- scene description (host memory, copied by mirt_scene_create) --------------------------- */

typedef enum mirt_material_kind {
    MIRT_MAT_LAMBERTIAN = 0, /* src/scene/material.rs:29-84   */
    MIRT_MAT_METAL = 1,      /* src/scene/material.rs:86-114  */
    MIRT_MAT_DIELECTRIC = 2  /* src/scene/material.rs:116-154 */
} mirt_material_kind;

typedef enum mirt_texture_kind {
    MIRT_TEX_SOLID = 0,   /* src/scene/texture.rs:14-28 */
    MIRT_TEX_CHECKER = 1, /* src/scene/texture.rs:30-59 (children are always SolidColor, :37-43) */
    MIRT_TEX_IMAGE = 2    /* src/scene/texture.rs:61-94 */
} mirt_texture_kind;

/* RGB8 image, row-major, 3 bytes per texel, as `image::RgbImage` (src/scene/image_loader.rs:5-8). */
typedef struct mirt_image {
    uint32_t width;
    uint32_t height; /* 0 is legal: ImageTexture then returns cyan (texture.rs:78-80) */
    const uint8_t* rgb;
} mirt_image;

typedef struct mirt_texture {
    uint32_t kind;    /* mirt_texture_kind                                              */
    uint32_t image;   /* IMAGE: index into images[]                                     */
    double color[3];  /* SOLID: albedo; CHECKER: even colour                            */
    double odd[3];    /* CHECKER: odd colour                                            */
    double inv_scale; /* CHECKER: 1/scale, computed by the caller like texture.rs:39    */
} mirt_texture;

typedef struct mirt_material {
    uint32_t kind;      /* mirt_material_kind                                                 */
    uint32_t texture;   /* LAMBERTIAN: index into textures[]                                  */
    double reflectance; /* LAMBERTIAN: scatter probability and attenuation divisor (:68,:78)  */
    double albedo[3];   /* METAL                                                              */
    double fuzz;        /* METAL: used only if has_fuzz                                       */
    uint32_t has_fuzz;  /* METAL: `fuzz: Option<f64>` is Some (material.rs:88)                */
    uint32_t _pad;
    double ior;         /* DIELECTRIC: refraction_index                                       */
} mirt_material;

/* One sphere = one hit id (its index in spheres[]).  src/scene/sphere.rs:14-24. */
typedef struct mirt_sphere {
    double center0[3]; /* Center::Static(p) or Center::Moving(p, _)      */
    double center1[3]; /* Center::Moving(_, p); ignored unless moving    */
    double radius;     /* stored as max(r, 0) like sphere.rs:31,50       */
    uint32_t material; /* index into materials[]                         */
    uint32_t moving;   /* 0 = static                                     */
} mirt_sphere;

typedef struct mirt_scene_desc {
    uint32_t n_spheres;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t n_images;
    const mirt_sphere* spheres;
    const mirt_material* materials;
    const mirt_texture* textures;
    const mirt_image* images;
} mirt_scene_desc;

/* The POD left behind by Camera::build (struct Camera, src/camera/camera.rs:107-120).  The host
 * side keeps computing it exactly as camera.rs:135-182 does; the device only consumes it. */
typedef struct mirt_camera {
    double lookfrom[3];
    double px_du[3];
    double px_dv[3];
    double vp_upper_left[3];
    double px00_loc[3];
    double disk_u_r[3];   /* Defocuser::disk_u_r = u * defocus_radius (camera.rs:84) */
    double disk_v_r[3];
    double defocus_angle; /* <= 0 disables the lens sample (camera.rs:220)           */
    uint32_t img_width;
    uint32_t img_height;
    uint32_t samples_per_pixel; /* 0 = anti-aliasing disabled: the deterministic branch camera.rs:247-254 */
    uint32_t max_bounce_depth;
} mirt_camera;

typedef enum mirt_precision {
    MIRT_F64 = 64, /* parity mode: the reference's arithmetic, operation for operation       */
    MIRT_F32 = 32  /* throughput mode: binary32 in the "f32 spec" of DESIGN.md 3 (running product, perpendicular-form discriminant, reciprocals) */
} mirt_precision;

#define MIRT_FLAG_COMPACT_ROWS 1u /* outputs hold only the rows this call owns, packed in row order */
/* Closest hit through a bounding-ball tree built by mirt_scene_create instead of the brute-force tile loop (f32
 * precision only).  Same (id, t) for every ray, hence the same image bit for bit; the counterpart of the reference's
 * Bvh::hit (src/scene/bvh.rs:78-92) for scenes where testing every sphere is not affordable. */
#define MIRT_FLAG_BVH 2u
/* Brute-force loop (either precision): keep the broad phase on the vector ALU.  By default scenes whose extents fit the
 * split-f16 operands run the broad phase's conservative test on the matrix pipe (v_mfma_f32_32x32x16_f16); the candidates go
 * through the same exact test either way, so the (id, t) of every ray - and the image - do not depend on this flag. */
#define MIRT_FLAG_VALU_BROAD 4u
/* Device-buffer render calls: do not wait for the launch.  The call records its timing events and sends the counters to
 * pinned host memory behind the kernels, returns at once (its `stats` argument is ignored) and mirt_render_wait() on
 * the same scene handle collects them later.  For callers that keep several frames in flight - one scene handle and
 * one stream per frame in flight, since a handle owns one work queue and one set of partial sums. */
#define MIRT_FLAG_DEFER_STATS 8u
/* Brute-force loop: take the matrix-pipe broad phase also for scenes below its size threshold (24 spheres), where the
 * vector-ALU loop is faster.  A tuning / test switch; the image does not depend on it. */
#define MIRT_FLAG_MX_SMALL 16u
/* Matrix-pipe brute-force loop, scenes of up to 131 072 spheres: test every (ray, sphere) pair.  By default the kernel first
 * tests, with the same contraction and margins, a ball around each tile of 32 spheres (the spheres are kept in a spatial order)
 * and skips the tiles that no ray of the wave can touch - the (id, t) of every ray, hence the image, is the same bit for bit;
 * mirt_stats.mx_tiles tells how many tiles were evaluated.  The flag switches the cull off (measurements: the all-pairs rate). */
#define MIRT_FLAG_ALL_PAIRS 32u

/* Which rows a call renders: rows are grouped in strips of `strip_rows`; strip s is owned when
 * s % part_count == part_index (interleaved strips).  part_count <= 1 = whole image.
 * The image is a function of (scene, camera, precision, seed, chunk_samples) and of nothing else: no environment
 * variable is read by any entry point.  Zero-initialise the struct; 0 always means "the library's default". */
typedef struct mirt_opts {
    uint32_t precision;  /* mirt_precision */
    uint32_t flags;
    uint64_t seed;       /* key of the counter-based RNG; the image does not depend on the partition */
    uint32_t strip_rows; /* 0 -> 8 */
    uint32_t part_index;
    uint32_t part_count;
    /* Lower bound for the samples per work item (0 -> 1; the library doubles it until the per-item sums of the whole
     * frame fit their budget; mirt_pass_granularity returns the value in force).  A pixel's radiance is the sum, in
     * chunk order, of per-item sums, so this is the one tuning field that reaches the last bits of the image: callers
     * that resume a checkpoint (mirt_render_pass) must pass the value the checkpoint was written with. */
    uint32_t chunk_samples;
    uint32_t group_chunks; /* tuning, image-neutral: chunks of a pixel kept together in the queue order (0 -> automatic) */
    uint32_t bvh_keep;     /* tuning, image-neutral, MIRT_FLAG_BVH: share (1..64, in 64ths) of the traversing lanes below
                              which a wave suspends its tree queries (0 -> automatic) */
    uint32_t _reserved[2]; /* must be 0 */
} mirt_opts;

/* Counters written by a render call (device-side exact counts). */
typedef struct mirt_stats {
    uint64_t segments;        /* ray segments traced = closest-hit queries (camera.rs:273)        */
    uint64_t sphere_tests;    /* static_tests + moving_tests: (ray, sphere) pairs of the brute-force loop  */
    uint64_t static_tests;
    uint64_t moving_tests;
    uint64_t narrow_tests;    /* candidates (d >= 0) that reached the root computation            */
    uint64_t wave_iterations; /* wave-level passes of the segment loop (64 lanes each)            */
    uint64_t nan_pixels;      /* pixels whose radiance was NaN (reference would panic)            */
    double kernel_ms;         /* hipEvent time of the launch on the stream it ran on: the render kernel K2 */
                              /* AND the resolve kernel K2b behind it                                      */
    uint64_t tree_segments;   /* of `segments`: traced through the bounding-volume tree (MIRT_FLAG_BVH);  */
                              /* those are not counted in *_tests                                          */
    double resolve_ms;        /* of kernel_ms: the resolve kernel K2b alone (K2 = kernel_ms - resolve_ms) */
    uint64_t mx_tiles;        /* matrix-pipe kernels: (wave, tile of 32 spheres) evaluations actually made - the tile cull */
                              /* skips tiles no ray of a wave can touch; wave_iterations x tiles without it (0: other kernels) */
} mirt_stats;

typedef struct mirt_scene mirt_scene; /* opaque: device-resident flattened scene */

/* ---- entry points ---------------------------------------------------------------------------- */

int mirt_abi_version(void);
int mirt_device_count(void);
const char* mirt_last_error(void);

/* Replaces Scene::build_bvh (src/scene/hittable.rs:86-90): freezes the scene into SoA device
 * buffers on HIP device `device`.  The caller keeps its own buffers. */
int mirt_scene_create(const mirt_scene_desc* desc, int device, mirt_scene** out);
void mirt_scene_destroy(mirt_scene* scene);

/* Deterministic primary rays: the no-AA branch of Camera::render (camera.rs:247-254) + Scene::hit
 * (hittable.rs:92-98) in the interval (0.001, inf).  id[W*H] = sphere index or MIRT_MISS_ID,
 * t[W*H] = hit parameter as `precision` reals (float or double; +inf on a miss).  Host buffers. */
int mirt_primary_hits(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                      uint32_t* id, void* t);

/* Replaces the body of Camera::render (camera.rs:237-257) + ARgb's Display quantiser
 * (src/core/rgb.rs:20-36).  rgb_linear: W*H*3 reals (pixel-interleaved r,g,b), rgb8: W*H*3 bytes
 * ready for the "r g b\n" PPM lines; either may be NULL.  Host buffers.  stats may be NULL. */
int mirt_render(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                void* rgb_linear, uint8_t* rgb8, mirt_stats* stats);

/* Same as mirt_render / mirt_primary_hits but the outputs are DEVICE pointers on the scene's
 * device and the work is enqueued on `stream` (a hipStream_t, NULL = default stream).  The calls
 * synchronise the stream before returning only if `stats` is non-NULL.  Used by the multi-GPU
 * gather (the band is handed to RCCL without leaving HBM) and by bench.py.
 * A scene handle owns ONE work queue, ONE set of counters and ONE workspace of partial sums: render calls
 * on the same handle are ordered one behind the other whatever their streams (each launch records an event,
 * the next one's stream waits for it), so a second call on another stream is safe but does not overlap the
 * first - keep one handle per frame in flight for that.  Host threads sharing a handle are serialised: for
 * the duration of the enqueue in the _device entry points, for the whole call in mirt_render / mirt_render_pass
 * (whose staging frames belong to the handle) and in mirt_render_wait; distinct handles are independent. */
int mirt_render_device(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                       void* d_rgb_linear, uint8_t* d_rgb8, void* stream, mirt_stats* stats);
int mirt_primary_hits_device(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                             uint32_t* d_id, void* d_t, void* stream);
/* mirt_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has NO counterpart in the reference (whose render allocates freely on the CPU); it exists only
 * for those callers.  Same arguments, same kernels, same image bit for bit as mirt_render_device - but the launch's
 * counters / work-queue head, its per-item partial sums and (MIRT_F64) its per-lane attenuation stacks live in a
 * file-scope device array of MIRT_STATIC_SCRATCH_BYTES that each GPU the library runs on holds until the process ends,
 * and the call makes no hipMalloc / hipFree / event or host allocation of any kind (everything else it needs was made by
 * mirt_scene_create: operands, events, the occupancy figures of every render kernel; tests/test_zz_no_alloc.py runs the
 * call under stream capture and replays the graph).  Limits: a call whose partial sums (3 reals per pixel of the part and chunk of samples, see
 * mirt_pass_granularity; + 24 bytes x max_bounce_depth per lane of the grid in MIRT_F64) exceed the array is refused with
 * MIRT_ERR_TOO_LARGE - e.g. 400x225x100 spp in f32 fits, 1920x1080x64 spp does not; a camera so far outside the scene
 * that the matrix-pipe operands have to be rebuilt at another scale is refused with MIRT_ERR_INVALID until one
 * mirt_render_device call has been made with it (or MIRT_FLAG_VALU_BROAD is passed).  With stats == NULL and without
 * MIRT_FLAG_DEFER_STATS the call only enqueues one memset and two kernels on `stream`.
 * ALL calls on one GPU share that GPU's array, whatever their scene handles: calls that overlap on one GPU race. */
#define MIRT_STATIC_SCRATCH_BYTES (128ull << 20)
int mirt_render_device_static(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                              void* d_rgb_linear, uint8_t* d_rgb8, void* stream, mirt_stats* stats);
/* Waits for the render call made with MIRT_FLAG_DEFER_STATS on this scene handle (no reference counterpart: the
 * reference's render returns when the frame is done) and fills `stats` (may be NULL).  Until it has been called, further
 * render calls on the handle are refused; without the flag there is nothing to wait for and the call is an error. */
int mirt_render_wait(const mirt_scene* scene, mirt_stats* stats);

/* K3 alone: gamma-2 + quantise n reals (device pointers) exactly like rgb.rs:20-36 and
 * utils/math.rs:13-21.  *nan_count receives how many inputs were NaN (host pointer, may be NULL). */
int mirt_quantize_device(const void* d_linear, uint32_t precision, size_t n, uint8_t* d_out,
                         void* stream, uint64_t* nan_count);

/* Progressive rendering / checkpointing (the reference renders a frame in one go): add samples
 * [first_sample, first_sample + sample_count) of every pixel of the image `cam` describes (samples_per_pixel = the FINAL
 * count: it fixes the sample set and the work decomposition) to `accum` - the running radiance SUMS, precision-sized RGB
 * in the layout of the linear output, caller-owned, read when first_sample > 0 and always written - and, where the
 * pointers are not NULL, write the image so far: the sums divided by the samples accumulated, and its quantised bytes.
 * Pass boundaries must be multiples of mirt_pass_granularity() (the last pass ends at samples_per_pixel).  After the
 * last pass the image equals mirt_render's bit for bit, however the range was cut and whether or not the sums went
 * through a checkpoint file in between. */
int mirt_render_pass_device(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                            uint32_t first_sample, uint32_t sample_count, void* d_accum,
                            void* d_rgb_linear, uint8_t* d_rgb8, void* stream, mirt_stats* stats);
/* The same with host buffers (full frames only); `accum` is required. */
int mirt_render_pass(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts,
                     uint32_t first_sample, uint32_t sample_count, void* accum,
                     void* rgb_linear, uint8_t* rgb8, mirt_stats* stats);
/* Samples per work item for this image (1, 2, 4, ...): the unit of pass boundaries.  It follows from the size of the
 * whole frame and the sample count, not from the part a call renders: a pixel's radiance is a sum of per-item sums, so
 * its last bits depend on this size, and the image does not depend on the partition.  0 = no anti-aliasing. */
uint32_t mirt_pass_granularity(const mirt_camera* cam, const mirt_opts* opts);

/* K4: scatter `part_count` compact bands (as gathered with RCCL, band r at
 * d_bands + r * band_stride_bytes, each holding that part's rows packed in row order) into the
 * full frame.  elem_bytes = bytes per pixel (e.g. 12 for f32 RGB, 3 for rgb8). */
int mirt_deinterleave_device(const void* d_bands, size_t band_stride_bytes, uint32_t part_count,
                             uint32_t strip_rows, uint32_t width, uint32_t height,
                             uint32_t elem_bytes, void* d_frame, void* stream);

/* Rows owned by a part (helper for sizing compact buffers). */
uint32_t mirt_part_rows(uint32_t height, uint32_t strip_rows, uint32_t part_index, uint32_t part_count);

/* Which closest-hit front end a render / primary-hits call with these arguments runs (no reference counterpart; for
 * benchmarks and tests): 0 = brute force with the broad phase on the vector ALU, 1 = brute force with the broad phase
 * on the matrix pipe, 2 = bounding-volume tree, 3 = f64 brute force on the vector ALU, 4 = f64 brute force with the (f32)
 * broad phase on the matrix pipe, 5 = f32, a handful of spheres (<= 8 slots): the exact test on every sphere, no broad phase
 * (render_kernel_f32_tiny); negative = MIRT_ERR_* for invalid arguments. */
int mirt_closest_hit_kind(const mirt_scene* scene, const mirt_camera* cam, const mirt_opts* opts);

/* ---- one host thread, N GPUs of one node (SURVEY.md 8b "one host thread drives N devices", 8e) ---------------------
 * What a drop-in Camera::render (src/camera/camera.rs:232) calls to use the whole node: the scene is replicated on every
 * device of `device_ids`, the frame is cut into interleaved strips of mirt_opts.strip_rows rows (strip s -> device
 * s mod N), every device renders its band with the kernels of mirt_render_device, the bands meet on the root
 * (device_ids[0]) in ONE exchange over xGMI with RCCL (single process, ncclCommInitAll) and K4 scatters them into frame
 * order.  The image is mirt_render's bit for bit, whatever N and whichever exchange.  RCCL is loaded at run time on the
 * first node render (librccl.so.1); single-GPU entry points never touch it. */
#define MIRT_NODE_MAX_DEVICES 16u
typedef enum mirt_gather_kind {
    MIRT_GATHER_TO_ROOT = 0,  /* grouped ncclSend / ncclRecv: every peer sends its band to the root over its own xGMI link */
    MIRT_GATHER_ALLGATHER = 1 /* ncclAllGather (ring): every device ends up with all bands; only the root's copy is used */
} mirt_gather_kind;
typedef struct mirt_node_stats {
    uint64_t segments;   /* of the whole frame */
    uint64_t nan_pixels;
    uint32_t n_devices;
    uint32_t gather;     /* mirt_gather_kind of this call */
    double wall_ms;      /* host clock around the call */
    double gather_ms;    /* HIP events on the root's stream around the exchange (includes waiting for the slowest band) */
    double scatter_ms;   /* K4 on the root */
    double band_kernel_ms[MIRT_NODE_MAX_DEVICES]; /* per device: HIP-event time of its render launch */
    uint64_t band_segments[MIRT_NODE_MAX_DEVICES];
} mirt_node_stats;
typedef struct mirt_node mirt_node;
/* flags: 0, or MIRT_NODE_SHARED_DEVICES to let several ranks name the same GPU.  RCCL itself refuses that (one rank per
 * GPU), so the flag is only good with an exchange library that copes: tests/stub_rccl, which lets the N-rank partition,
 * exchange and scatter logic of this file be rehearsed on a one-GPU box. */
#define MIRT_NODE_SHARED_DEVICES 1u
int mirt_node_create(const mirt_scene_desc* desc, const int* device_ids, uint32_t n_devices, uint32_t flags, mirt_node** out);
void mirt_node_destroy(mirt_node* node);
uint32_t mirt_node_device_count(const mirt_node* node);
/* Host buffers, full frames (rgb_linear: W*H*3 reals, rgb8: W*H*3 bytes; either may be NULL, not both).  When only rgb8 is
 * asked for, the bands travel as bytes (a quarter of the f32 traffic).  opts: precision, seed, flags (closest-hit front end),
 * strip_rows and the tuning fields are honoured; part_index / part_count must be left 0 (the node partitions). */
int mirt_node_render(mirt_node* node, const mirt_camera* cam, const mirt_opts* opts, uint32_t gather,
                     void* rgb_linear, uint8_t* rgb8, mirt_node_stats* stats);
/* The same with DEVICE pointers on the root device. */
int mirt_node_render_device(mirt_node* node, const mirt_camera* cam, const mirt_opts* opts, uint32_t gather,
                            void* d_rgb_linear, uint8_t* d_rgb8, mirt_node_stats* stats);

/* Measurement helper (no reference counterpart; SURVEY.md 8d asks for the FP32 vector peak "measured on the box by an FMA
 * microbenchmark" next to the nominal figure): runs independent v_fma_f32 chains on every SIMD of `device` at 8 waves per
 * SIMD for a few milliseconds and returns the rate in TFLOP/s (2 flop per lane and instruction). */
int mirt_measure_fma_peak(int device, double* tflops);

/* Counter-based RNG, exposed for known-answer tests: Philox4x32-10 block.  out[4]. */
void mirt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
