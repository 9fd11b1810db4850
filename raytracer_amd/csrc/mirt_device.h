/*
 * mirt_device.h — device-side building blocks of the gfx950 render backend (product code; does not
 * include or call anything under oracle/).
 *
 * Everything is templated on Real:
 *   double : parity mode.  The reference's f64 operation sequence, one IEEE operation per source
 *            operation (the whole library is compiled with -ffp-contract=off).
 *   float  : throughput mode.  Same sequence in binary32, except the sphere discriminant, whose
 *            products are fused with explicit fmaf (the "f32 spec" of DESIGN.md).
 *
 * Reference citations are /root/reference/src/<file>:<line>.
 */
#ifndef MIRT_DEVICE_H
#define MIRT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mirt.h"

#define MIRT_WAVE 64
#define MIRT_WAVES_PER_BLOCK 4
#define MIRT_BLOCK (MIRT_WAVE * MIRT_WAVES_PER_BLOCK)
#define MIRT_TILE 64   /* spheres staged per LDS tile, one per lane */
#define MIRT_GROUP 4    /* spheres per broad-phase group */
#define MIRT_CGROUP 16  /* bounding records per iteration of the f32 conservative broad phase */
#define MIRT_CAND_CAP 32 /* deferred narrow-phase candidates kept per lane (16-bit; the f32 path keeps 16 32-bit entries) */

namespace mirt {

/* ---- small vector type (core/point3.rs) -------------------------------------------------------- */
template <class R> struct V3 {
    R x, y, z;
};
template <class R> __host__ __device__ inline V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> __host__ __device__ inline V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> __host__ __device__ inline V3<R> operator*(V3<R> a, R l) { return {a.x * l, a.y * l, a.z * l}; }
template <class R> __host__ __device__ inline V3<R> operator/(V3<R> a, R l) { return {a.x / l, a.y / l, a.z / l}; }
template <class R> __host__ __device__ inline V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> __host__ __device__ inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* point3.rs:52-58 */

__device__ inline float m_sqrt(float x) { return sqrtf(x); }
__device__ inline double m_sqrt(double x) { return sqrt(x); }
__device__ inline float m_fabs(float x) { return fabsf(x); }
__device__ inline double m_fabs(double x) { return fabs(x); }
__device__ inline float m_fmin(float a, float b) { return fminf(a, b); }
__device__ inline double m_fmin(double a, double b) { return fmin(a, b); }
__device__ inline float m_floor(float x) { return floorf(x); }
__device__ inline double m_floor(double x) { return floor(x); }
__device__ inline float m_round(float x) { return roundf(x); }
__device__ inline double m_round(double x) { return round(x); }
/* Sphere::uv (sphere.rs:72-76).  double: the math library's acos / atan2, as Rust's f64 methods.  float (f32 spec): the two
 * polynomial forms of Abramowitz & Stegun 4.4.46 / 4.4.49 (|error| <= 2.2e-8 / 1.4e-8 before rounding), every step an IEEE
 * operation or an explicit fma, restated operation for operation in the oracle - the angles only choose a texel, and with
 * them spelled out an image-textured f32 frame is the oracle's bit for bit (OCML's acosf / atan2f differ from glibc's in
 * the last ulp now and then, which moved a texel choice in ~0.1 % of the channels).  |x| > 1 gives NaN like acos itself. */
__device__ inline float m_acos(float x) {
    const float ax = __builtin_fabsf(x);
    float p = -0.0012624911f;
    p = __builtin_fmaf(p, ax, 0.0066700901f);
    p = __builtin_fmaf(p, ax, -0.0170881256f);
    p = __builtin_fmaf(p, ax, 0.0308918810f);
    p = __builtin_fmaf(p, ax, -0.0501743046f);
    p = __builtin_fmaf(p, ax, 0.0889789874f);
    p = __builtin_fmaf(p, ax, -0.2145988016f);
    p = __builtin_fmaf(p, ax, 1.5707963050f);
    const float r = sqrtf(1.0f - ax) * p;
    return x < 0.0f ? 3.14159265358979323846f - r : r;
}
__device__ inline double m_acos(double x) { return acos(x); }
__device__ inline float m_atan2(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float t = mx == 0.0f ? 0.0f : mn / mx;
    const float z = t * t;
    float q = 0.0028662257f;
    q = __builtin_fmaf(q, z, -0.0161657367f);
    q = __builtin_fmaf(q, z, 0.0429096138f);
    q = __builtin_fmaf(q, z, -0.0752896400f);
    q = __builtin_fmaf(q, z, 0.1065626393f);
    q = __builtin_fmaf(q, z, -0.1420889944f);
    q = __builtin_fmaf(q, z, 0.1999355085f);
    q = __builtin_fmaf(q, z, -0.3333314528f);
    q = __builtin_fmaf(q, z, 1.0f);
    float r = t * q;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return y < 0.0f ? -r : r;
}
__device__ inline double m_atan2(double y, double x) { return atan2(y, x); }
__device__ inline float m_sin(float x) { return sinf(x); }
__device__ inline double m_sin(double x) { return sin(x); }
__device__ inline void m_sincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ inline void m_sincos(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ inline float m_cos(float x) { return cosf(x); }
__device__ inline double m_cos(double x) { return cos(x); }

/* Lens sample angle theta = 2 pi u, u in [0,1) (point3.rs:107-109: `Uniform(0, 2 pi)`, then cos / sin).
 *   double: the reference's own expression - th = u * TAU rounded once, then libm's sin/cos of it (OCML here, glibc in the
 *           oracle: equal up to their last ulp).
 *   float (f32 spec): sin and cos of the quarter-turn count x = 4u directly.  k = rint(x) and r = x - k in [-1/2, 1/2] are
 *           exact in binary32, y = r * (pi/2) in [-pi/4, pi/4] goes through the two Cephes single-precision kernels
 *           (|error| < 1e-7), and k mod 4 picks the quadrant.  Every step is an IEEE operation or an explicit fma, restated
 *           one for one in the oracle's f32 instantiation: lens origins are bit-equal on both sides, and the routine costs
 *           a fifth of OCML's sincosf with its argument reduction for arbitrary floats (119 VALU instructions per wave and
 *           iteration of the render loop, 8 % of everything outside the broad phase). */
__host__ __device__ inline void lens_sincos(float u, float& s, float& c) {
    const float x = u * 4.0f;
    const float k = __builtin_rintf(x);
    const float y = (x - k) * 1.57079632679489661923f;
    const float z = y * y;
    const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * y, y);
    const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                                    __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3; /* k in {0, ..., 4} */
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}
__device__ inline void lens_sincos(double u, double& s, double& c) {
    const double TAU = 6.28318530717958647692528676655900577;
    sincos(u * TAU, &s, &c); /* one range reduction for both */
}

/* point3.rs:39-46; the f32 spec takes one division and three products instead of three divisions */
__device__ inline V3<double> unit(V3<double> a) { return a / m_sqrt(dot(a, a)); }
__device__ inline V3<float> unit(V3<float> a) { return a * (1.0f / m_sqrt(dot(a, a))); }
template <class R> __device__ inline V3<R> reflect(V3<R> v, V3<R> n) { return v - (n * (R)2.0) * dot(v, n); } /* point3.rs:116-118 */

/* ---- 4-wide sphere records ------------------------------------------------------------------------ */
template <class R> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<double> { typedef double4 type; };

/* ---- flattened scene in HBM (see DESIGN.md "Data layout") ---------------------------------------- */
struct DevImage {
    uint32_t w, h;
    uint32_t offset; /* first texel in the RGBA8 pool (one dword per texel, A unused) */
    uint32_t _pad;
};
/* One self-contained record per material: the texture of a Lambertian (solid colour, checker colours and 1/scale, or the
 * image's size and place in the texel pool) is folded in on the host, so a bounce costs ONE dependent fetch before the
 * texel itself instead of material -> texture -> image -> texel. */
template <class R> struct DevMaterial {
    uint32_t kind, tex_kind, has_fuzz, img_w; /* tex_kind: LAMBERTIAN only */
    uint32_t img_h, img_offset, _pad0, _pad1; /* IMAGE texture: size, first texel in the pool (img_h == 0: cyan, texture.rs:78-80) */
    R reflectance, fuzz, ior, inv_scale;
    V3<R> albedo; /* METAL albedo | SOLID colour / reflectance | CHECKER even colour / reflectance (the attenuation of material.rs:78) */
    V3<R> odd;    /* CHECKER odd colour / reflectance */
};
/* Node of the bounding-volume tree (MIRT_FLAG_BVH): the boxes of BOTH children, one 64-byte fetch per step.
 *   q0 = (lo0.x, lo0.y, lo0.z, hi0.x)  q1 = (hi0.y, hi0.z, lo1.x, lo1.y)  q2 = (lo1.z, hi1.x, hi1.y, hi1.z)
 *   q3 = (ref0, ref1, -, -) as bit patterns
 * A child's box is the box of its leaves' bounding balls, widened on the host by s * max_i(|M_i| + rho_i) (s = MIRT_BVH_S);
 * the ray adds s |O| at run time (trace_bvh has the derivation).
 * ref: bit 31 set = leaf, bits 3..30 first record, bits 0..2 count - 1; clear = node index; 0xFFFFFFFF = none */
struct BvhNode {
    float4 q0, q1, q2, q3;
};
#define MIRT_BVH_S 4.0e-3f
#define MIRT_BVH_NONE 0xFFFFFFFFu
#define MIRT_BVH_STACK 32

template <class R> struct DevScene {
    /* spheres, statics first: [0, n_static) static, [n_static, n) moving; both counts are padded to a
     * multiple of MIRT_GROUP with dummy records (r*r = -inf -> never a candidate, orig_id = ~0u) */
    const typename Vec4<R>::type* A; /* (c0.x, c0.y, c0.z, r*r)                 */
    const typename Vec4<R>::type* B; /* (c1-c0).x, .y, .z, r                    */
    const uint32_t* orig_id;         /* sorted index -> hit id (index in mirt_scene_desc.spheres) */
    const uint32_t* material;        /* sorted index -> material                */
    const DevMaterial<R>* mats;
    const uint32_t* texels;          /* RGBA8 pool                              */
    /* f32 only: conservative bounding record per sorted sphere for the broad phase, (M.x, M.y, M.z, w):
     * M = centre of the ball that contains the sphere over the whole shutter interval, w = |M|^2 - rho^2 - margin
     * (see trace_cons).  NULL in the f64 instantiation, whose broad phase evaluates the exact discriminant. */
    const float4* S;
    unsigned long long* dbg;         /* diagnostic builds: counter of broad-phase misses (must stay 0) */
    uint32_t n, n_static;
    uint32_t n_live;                 /* slots before the padding to whole 32-sphere tiles (statics and movings each padded to MIRT_GROUP): what
                                        the exact-test-on-every-sphere paths of tiny scenes walk and the host's kernel choice looks at */
    uint32_t n_materials;
    /* bounding-volume tree (f32 only): nodes, the bounding records in leaf order, leaf slot -> sorted sphere index */
    const BvhNode* bvh;
    const float4* bvh_S;
    const uint32_t* bvh_idx;
    uint32_t bvh_root; /* a ref like BvhNode::q3.x */
    uint32_t bvh_n_nodes, bvh_n_recs, bvh_depth;
    /* matrix-pipe broad phase (f32 only, trace_mx): split-f16 sphere operands, 64 x 16 B per tile of 32 sorted spheres,
     * in the scene's power-of-two scale mx_scale (NULL when the scene does not qualify) */
    const uint4* SM;
    float mx_scale;
    uint32_t n_big, big_idx[4]; /* sorted slots of the spheres the matrix-pipe filter leaves out (MIRT_MX_BIG: far larger than the rest of the scene - the ground): every ray tests them exactly, directly */
    float mx_drift[3]; /* the common part of the spheres' shutter motion (world units): the filter tests the ray with its origin moved by -tm * mx_drift against balls around what is left of each sphere's motion (mirt_scene_create) */
    uint32_t mx_cull; /* resident scenes: tile `ntiles` of SM is the top tile (a ball around each tile's balls) and trace_mx may skip tiles by it */
};
template <class R> struct DevCamera { /* struct Camera, camera/camera.rs:107-120, rounded once to Real */
    V3<R> lookfrom, du, dv, ul, px00, disk_u, disk_v;
    R defocus_angle;
    uint32_t W, H, spp, depth;
};
struct DevPart { /* interleaved row strips (mirt_opts) */
    uint32_t strip_rows, part_index, part_count, local_rows, compact;
};

/* unsigned division by a launch-invariant divisor (Granlund & Montgomery 1994, fig. 4.1): ~5 VALU instead of the ~40 of
 * a 32-bit hardware-less divide; exact for every 32-bit numerator */
struct FastDiv {
    uint32_t m, s1, s2;
};
__host__ inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < d) ++l;
    FastDiv f;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ inline uint32_t fastdiv(uint32_t x, const FastDiv& f) {
    const uint32_t t = __umulhi(x, f.m);
    return (t + ((x - t) >> f.s1)) >> f.s2;
}

__host__ __device__ inline uint32_t part_rows(uint32_t height, uint32_t S, uint32_t idx, uint32_t cnt) {
    if (cnt <= 1) return height;
    uint32_t rows = 0;
    uint32_t nstrips = (height + S - 1) / S;
    for (uint32_t s = idx; s < nstrips; s += cnt) {
        uint32_t r0 = s * S, r1 = r0 + S;
        rows += (r1 > height ? height : r1) - r0;
    }
    return rows;
}
/* local (compact) row -> global row */
__host__ __device__ inline uint32_t part_global_row(const DevPart& p, uint32_t lr) {
    if (p.part_count <= 1) return lr;
    uint32_t ls = lr / p.strip_rows, within = lr - ls * p.strip_rows;
    return (ls * p.part_count + p.part_index) * p.strip_rows + within;
}

/* ---- counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) --------------------------------------
 * ctr = (pixel, sample, bounce, block), key = (seed_lo, seed_hi).  Dimension d of a stream (pixel, sample, bounce):
 *   f64: block d/2, 64-bit word pair d%2 (x0|x1<<32, x2|x3<<32), u = (word >> 11) * 2^-53
 *   f32: block 0: dims 0..3 = (word d >> 8) * 2^-24, dim 4 = the low bytes of words 0..2 (24 bits) * 2^-24
 * (Philox's integer multiplies are quarter-rate on the VALU: the f32 mode gets all five draws of a camera ray, or the
 * four of a bounce, out of one block.) */
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
#if defined(__HIP_DEVICE_COMPILE__)
        /* a ^ b ^ c as ONE three-input boolean instruction (v_bitop3_b32, truth table 0x96) instead of two v_xor_b32: 20 vector-ALU
         * instructions less per evaluation, and the kernels evaluate this once per loop iteration */
        uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
#else
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#endif
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* The uniforms of one camera ray (bounce 0) or one scatter (bounce b >= 1) of a sample: Philox4x32-10 blocks with the counter
 * (pixel, sample, bounce, block).  fill() evaluates the blocks once - render_body makes ONE call per loop iteration in which a lane
 * asks for the camera ray's blocks or for a scatter's, whichever it needs - and the dims*() accessors turn the words into uniforms. */
template <class R> struct Draws;
template <> struct Draws<double> {
    uint32_t o[12];
    /* dims 2*blk and 2*blk+1 of block blk; a camera ray takes three blocks (five dims), a scatter two */
    __device__ inline void fill(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t k0, uint32_t k1, bool camera) {
        philox4x32_10(pixel, sample, bounce, 0u, k0, k1, o);
        philox4x32_10(pixel, sample, bounce, 1u, k0, k1, o + 4);
        if (camera) philox4x32_10(pixel, sample, bounce, 2u, k0, k1, o + 8);
    }
    __device__ inline void pair(uint32_t blk, double& a, double& b) const {
        a = (double)(((uint64_t)o[4 * blk] | ((uint64_t)o[4 * blk + 1] << 32)) >> 11) * 0x1.0p-53;
        b = (double)(((uint64_t)o[4 * blk + 2] | ((uint64_t)o[4 * blk + 3] << 32)) >> 11) * 0x1.0p-53;
    }
    __device__ inline void dims0123(double& a, double& b, double& c, double& d) const { pair(0, a, b); pair(1, c, d); }
    __device__ inline void dims01234(double& a, double& b, double& c, double& d, double& e) const {
        double f;
        pair(0, a, b);
        pair(1, c, d);
        pair(2, e, f);
    }
};
template <> struct Draws<float> {
    uint32_t o[4];
    __device__ inline void fill(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t k0, uint32_t k1, bool) {
        philox4x32_10(pixel, sample, bounce, 0u, k0, k1, o);
    }
    __device__ static inline float cvt(uint32_t w) { return (float)(w >> 8) * 0x1.0p-24f; }
    __device__ inline void dims0123(float& a, float& b, float& c, float& d) const { a = cvt(o[0]); b = cvt(o[1]); c = cvt(o[2]); d = cvt(o[3]); }
    /* dims 0..4 out of ONE block: dim 4 = the low bytes of words 0..2 (24 bits) */
    __device__ inline void dims01234(float& a, float& b, float& c, float& d, float& e) const {
        a = cvt(o[0]); b = cvt(o[1]); c = cvt(o[2]); d = cvt(o[3]);
        e = (float)((o[0] & 0xFFu) | ((o[1] & 0xFFu) << 8) | ((o[2] & 0xFFu) << 16)) * 0x1.0p-24f;
    }
};

} /* namespace mirt */
#endif
