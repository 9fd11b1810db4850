/*
 * mirt_kernels.h — the HIP kernels of the per-pixel hot path (gfx950, wave64).
 *
 *   K1 primary_hits_kernel  deterministic primary ray -> (hit id, t)         camera.rs:247-254 + hittable.rs:92-98
 *   K2 render_kernel        ray generation + bounce loop + accumulation      camera.rs:209-259, 268-286
 *                           (+ K3 gamma/quantise fused in the epilogue)      core/rgb.rs:20-36
 *   K3 quantize_kernel      stand-alone quantiser
 *   K4 deinterleave_kernel  gathered row bands -> frame
 *
 * Closest-hit query (`trace`): brute force over all spheres, as an n-body style tile loop.
 *   broad phase : every lane evaluates the discriminant d of sphere.rs:81-90 for every sphere; the sphere
 *                 records are staged per wave in LDS (one coalesced 16-B load per lane per 64 spheres) and
 *                 read back as wave-uniform broadcasts.  Lanes only remember WHICH spheres have d >= 0
 *                 (a short per-lane candidate list in LDS).
 *   narrow phase: each lane then walks its own list and runs the exact root selection of sphere.rs:91-101
 *                 (sqrt, true divisions, strict interval) - a few iterations per lane instead of one
 *                 divergent slow path per (wave, sphere).
 * Results are identical to testing the spheres one after the other in insertion order with a shrinking
 * interval (ties in t go to the lowest hit id), which is what the reference's BVH walk computes.
 */
#ifndef MIRT_KERNELS_H
#define MIRT_KERNELS_H

#include "mirt_device.h"

namespace mirt {

/* ---- discriminant of sphere.rs:83-90 ------------------------------------------------------------- */
__device__ inline double ray_a(V3<double> d) { return dot(d, d); }
__device__ inline float ray_a(V3<float> d) { return __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x)); }

__device__ inline void disc(V3<double> oc, V3<double> D, double a, double r2, double& h, double& d) {
    h = dot(D, oc);
    double c = dot(oc, oc) - r2;
    d = h * h - a * c;
}
__device__ inline void disc(V3<float> oc, V3<float> D, float a, float r2, float& h, float& d) {
    h = __builtin_fmaf(D.z, oc.z, __builtin_fmaf(D.y, oc.y, D.x * oc.x));
    float c = __builtin_fmaf(oc.x, oc.x, __builtin_fmaf(oc.y, oc.y, __builtin_fmaf(oc.z, oc.z, -r2)));
    d = __builtin_fmaf(h, h, -(a * c));
}
/* sphere.rs:61-70 */
__device__ inline V3<double> center_at(V3<double> c0, V3<double> dc, double tm) { return {c0.x + tm * dc.x, c0.y + tm * dc.y, c0.z + tm * dc.z}; }
__device__ inline V3<float> center_at(V3<float> c0, V3<float> dc, float tm) {
    return {__builtin_fmaf(tm, dc.x, c0.x), __builtin_fmaf(tm, dc.y, c0.y), __builtin_fmaf(tm, dc.z, c0.z)};
}

template <class R> struct RayT { /* core/ray.rs */
    V3<R> o, d;
    R tm;
};

template <class R> struct WaveLds {
    typename Vec4<R>::type tileA[MIRT_TILE];
    typename Vec4<R>::type tileB[MIRT_TILE];
    uint32_t cand[MIRT_CAND_CAP][MIRT_WAVE];
};

template <class R> struct Closest {
    R t;
    uint32_t idx;  /* sorted index, 0xFFFFFFFF = none */
    uint32_t orig; /* hit id */
};

/* exact per-candidate test: sphere.rs:91-101 against the interval (tmin, best.t), ties -> lowest id */
template <class R>
__device__ inline void narrow_test(const DevScene<R>& sc, uint32_t i, const RayT<R>& ray, R a, R tmin, Closest<R>& best) {
    typename Vec4<R>::type sa = sc.A[i];
    V3<R> C = {sa.x, sa.y, sa.z};
    if (i >= sc.n_static) {
        typename Vec4<R>::type sb = sc.B[i];
        C = center_at(C, V3<R>{sb.x, sb.y, sb.z}, ray.tm);
    }
    V3<R> oc = C - ray.o;
    R h, d;
    disc(oc, ray.d, a, sa.w, h, d);
    if (!(d >= (R)0.0)) return;
    uint32_t orig = sc.orig_id[i];
    R sq = m_sqrt(d);
    R t = (h - sq) / a;
    bool ok = (tmin < t) && (t < best.t || (t == best.t && orig < best.orig));
    if (!ok) {
        t = (h + sq) / a;
        ok = (tmin < t) && (t < best.t || (t == best.t && orig < best.orig));
    }
    if (ok) {
        best.t = t;
        best.idx = i;
        best.orig = orig;
    }
}

__device__ inline float m_max4(float a, float b, float c, float d) { return fmaxf(fmaxf(a, b), fmaxf(c, d)); }
__device__ inline double m_max4(double a, double b, double c, double d) { return fmax(fmax(a, b), fmax(c, d)); }

/* Closest hit in (tmin, +inf).  Wave-synchronous: every lane of the wave must call it; lanes with
 * active == false do no narrow-phase work.  Returns the winner in `best` (idx == ~0u on a miss).
 *
 * Spheres are visited in groups of MIRT_GROUP (the sorted arrays are padded to whole groups with dummy
 * records, r*r = -inf, whose discriminant can never be >= 0): one batch of broadcast LDS reads, four
 * independent discriminant chains, ONE wave-level test "does any lane have a candidate in this group". */
template <class R>
__device__ inline void trace(const DevScene<R>& sc, WaveLds<R>& lds, uint32_t lane, bool active, const RayT<R>& ray, R tmin,
                             Closest<R>& best, uint32_t& narrow_count) {
    typedef typename Vec4<R>::type R4;
    best.t = (R)INFINITY;
    best.idx = 0xFFFFFFFFu;
    best.orig = 0xFFFFFFFFu;
    /* an inactive lane gets a = NaN: every d is NaN and `d >= 0` is false for free */
    const R a_true = ray_a(ray.d);
    const R a = active ? a_true : (R)NAN;
    const V3<R> O = ray.o, D = ray.d;
    const R tm = ray.tm;
    const uint32_t ngroups = sc.n / MIRT_GROUP;         /* n and n_static are multiples of MIRT_GROUP */
    const uint32_t gstatic = sc.n_static / MIRT_GROUP;  /* groups [0, gstatic) static, the rest moving */
    const uint32_t groups_per_tile = MIRT_TILE / MIRT_GROUP;

    uint32_t g = 0;
    while (g < ngroups) { /* one pass unless some lane's candidate list fills up */
        uint32_t cnt = 0;
        bool flush = false;
        /* ---- broad phase ---------------------------------------------------------------------- */
        while (g < ngroups && !flush) {
            const uint32_t tile = g / groups_per_tile;
            const uint32_t base = tile * MIRT_TILE;
            {   /* stage the tile: one coalesced 16-B (f32) record per lane */
                uint32_t li = base + lane;
                uint32_t ci = li < sc.n ? li : sc.n - 1;
                lds.tileA[lane] = sc.A[ci];
                if (base + MIRT_TILE > sc.n_static) lds.tileB[lane] = sc.B[ci];
            }
            uint32_t gend = (tile + 1) * groups_per_tile;
            gend = gend < ngroups ? gend : ngroups;
            for (; g < gend; ++g) {
                const uint32_t j = g * MIRT_GROUP - base;
                R4 s0 = lds.tileA[j], s1 = lds.tileA[j + 1], s2 = lds.tileA[j + 2], s3 = lds.tileA[j + 3];
                V3<R> c0 = {s0.x, s0.y, s0.z}, c1 = {s1.x, s1.y, s1.z}, c2 = {s2.x, s2.y, s2.z}, c3 = {s3.x, s3.y, s3.z};
                if (g >= gstatic) { /* moving: centre at the ray's time first (sphere.rs:61-70) */
                    R4 q0 = lds.tileB[j], q1 = lds.tileB[j + 1], q2 = lds.tileB[j + 2], q3 = lds.tileB[j + 3];
                    c0 = center_at(c0, V3<R>{q0.x, q0.y, q0.z}, tm);
                    c1 = center_at(c1, V3<R>{q1.x, q1.y, q1.z}, tm);
                    c2 = center_at(c2, V3<R>{q2.x, q2.y, q2.z}, tm);
                    c3 = center_at(c3, V3<R>{q3.x, q3.y, q3.z}, tm);
                }
                R h, d0, d1, d2, d3;
                disc(c0 - O, D, a, s0.w, h, d0);
                disc(c1 - O, D, a, s1.w, h, d1);
                disc(c2 - O, D, a, s2.w, h, d2);
                disc(c3 - O, D, a, s3.w, h, d3);
                R dm = m_max4(d0, d1, d2, d3); /* fmax ignores NaN operands: NaN = miss */
                if (__ballot(dm >= (R)0.0) != 0ull) { /* wave-uniform branch, rarely taken */
                    const uint32_t i0 = g * MIRT_GROUP;
                    if (d0 >= (R)0.0) { lds.cand[cnt][lane] = i0; ++cnt; }
                    if (d1 >= (R)0.0) { lds.cand[cnt][lane] = i0 + 1; ++cnt; }
                    if (d2 >= (R)0.0) { lds.cand[cnt][lane] = i0 + 2; ++cnt; }
                    if (d3 >= (R)0.0) { lds.cand[cnt][lane] = i0 + 3; ++cnt; }
                    if (__ballot(cnt > MIRT_CAND_CAP - MIRT_GROUP) != 0ull) { /* no room for another group */
                        flush = true;
                        ++g;
                        break;
                    }
                }
            }
        }
        /* ---- narrow phase: each lane walks its own list ------------------------------------------ */
        for (uint32_t k = 0; k < MIRT_CAND_CAP; ++k) {
            if (__ballot(k < cnt) == 0ull) break;
            if (k < cnt) {
                uint32_t i = lds.cand[k][lane];
                narrow_test(sc, i, ray, a_true, tmin, best);
                ++narrow_count;
            }
        }
    }
}

/* ---- hit record of the winner: sphere.rs:102-107, hittable.rs:48-60 -------------------------------- */
template <class R> struct Hit {
    V3<R> p, n, on; /* on = outward normal (input of Sphere::uv) */
    bool front;
};
template <class R> __device__ inline Hit<R> make_hit(const DevScene<R>& sc, const Closest<R>& c, const RayT<R>& ray) {
    typename Vec4<R>::type sa = sc.A[c.idx];
    typename Vec4<R>::type sb = sc.B[c.idx];
    V3<R> C = {sa.x, sa.y, sa.z};
    if (c.idx >= sc.n_static) C = center_at(C, V3<R>{sb.x, sb.y, sb.z}, ray.tm);
    Hit<R> h;
    h.p = ray.o + ray.d * c.t;   /* ray.rs:27-29 */
    h.on = (h.p - C) / sb.w;     /* sphere.rs:103 */
    h.front = dot(ray.d, h.on) < (R)0.0;
    h.n = h.front ? h.on : -h.on;
    return h;
}

/* ---- textures: scene/texture.rs ----------------------------------------------------------------- */
template <class R> __device__ inline int32_t floor_to_i32(R f) { /* Rust `as i32`: saturating, NaN -> 0 */
    if (f != f) return 0;
    if (f >= (R)2147483647.0) return 2147483647;
    if (f <= (R)-2147483648.0) return (int32_t)0x80000000;
    return (int32_t)f;
}
template <class R> __device__ inline uint32_t to_u32_sat(R x) { /* Rust `as u32` */
    if (!(x > (R)0.0)) return 0u;
    if (x >= (R)4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
template <class R> __device__ inline V3<R> texture_color(const DevScene<R>& sc, uint32_t tex, const Hit<R>& h) {
    const DevTexture<R>& t = sc.texs[tex];
    if (t.kind == MIRT_TEX_SOLID) return t.even; /* texture.rs:24-28 */
    if (t.kind == MIRT_TEX_CHECKER) {            /* texture.rs:46-59 */
        uint32_t acc = (uint32_t)floor_to_i32(m_floor(h.p.x * t.inv_scale));
        acc += (uint32_t)floor_to_i32(m_floor(h.p.y * t.inv_scale));
        acc += (uint32_t)floor_to_i32(m_floor(h.p.z * t.inv_scale));
        return (((int32_t)acc) % 2 == 0) ? t.even : t.odd;
    }
    /* IMAGE: texture.rs:71-94 with (u,v) of sphere.rs:72-76 */
    DevImage im = sc.imgs[t.image];
    if (im.h == 0) return {(R)0.0, (R)1.0, (R)1.0};
    const R PI = (R)3.14159265358979323846264338327950288;
    const R FRAC_1_PI = (R)0.318309886183790671537767526745028724;
    R theta = -m_acos(h.on.y);
    R phi = -m_atan2(h.on.z, h.on.x) + PI;
    R u = (R)0.5 * phi * FRAC_1_PI;
    R v = theta * FRAC_1_PI;
    u = u < (R)0.0 ? (R)0.0 : (u > (R)1.0 ? (R)1.0 : u);
    v = m_fabs(v);
    v = v < (R)0.0 ? (R)0.0 : (v > (R)1.0 ? (R)1.0 : v);
    uint32_t i = to_u32_sat(u * (R)im.w);
    uint32_t j = to_u32_sat(v * (R)im.h);
    i = i < im.w - 1 ? i : im.w - 1;
    j = j < im.h - 1 ? j : im.h - 1;
    uint32_t px = sc.texels[im.offset + j * im.w + i];
    return {(R)(px & 0xFFu) / (R)255.0, (R)((px >> 8) & 0xFFu) / (R)255.0, (R)((px >> 16) & 0xFFu) / (R)255.0};
}

/* ---- materials: scene/material.rs ----------------------------------------------------------------- */
template <class R> __device__ inline R schlick(R cosv, R ri) { /* material.rs:173-177; powi(x,5) = x*((x*x)*(x*x)) */
    R r0 = ((R)1.0 - ri) / ((R)1.0 + ri);
    r0 = r0 * r0;
    R x = (R)1.0 - cosv;
    R x2 = x * x;
    R x4 = x2 * x2;
    return r0 + ((R)1.0 - r0) * (x * x4);
}
template <class R> __device__ inline V3<R> refract(V3<R> uv, V3<R> n, R e) { /* material.rs:157-166 */
    R cos_theta = m_fmin(dot(-uv, n), (R)1.0);
    V3<R> perp = (uv + n * cos_theta) * e;
    V3<R> par = n * (-m_sqrt(m_fabs((R)1.0 - dot(perp, perp))));
    return perp + par;
}

/* One scatter event.  Returns false when the path is absorbed (black for the whole path, camera.rs:279). */
template <class R>
__device__ inline bool scatter(const DevScene<R>& sc, uint32_t mat, const RayT<R>& rin, const Hit<R>& hit, const Draws<R>& dr, V3<R>& att,
                               RayT<R>& out) {
    const DevMaterial<R>& m = sc.mats[mat];
    if (m.kind == MIRT_MAT_LAMBERTIAN) { /* material.rs:59-84 */
        R ua, ub, uc, ud;
        dr.block(0, ua, ub);
        if (!(ua < m.reflectance)) return false;
        dr.block(1, uc, ud);
        V3<R> dir = hit.n + unit(V3<R>{ub, uc, ud}); /* positive-octant quirk of point3.rs:88-93 */
        const R e = (R)1e-8;
        if (m_fabs(dir.x) < e && m_fabs(dir.y) < e && m_fabs(dir.z) < e) dir = hit.n;
        out = {hit.p, dir, rin.tm};
        att = texture_color(sc, m.tex, hit) / m.reflectance;
        return true;
    }
    if (m.kind == MIRT_MAT_METAL) { /* material.rs:97-114 */
        V3<R> refl = reflect(rin.d, hit.n);
        if (m.has_fuzz) {
            R ua, ub, uc, ud;
            dr.block(0, ua, ub);
            dr.block(1, uc, ud);
            refl = unit(refl) + unit(V3<R>{ua, ub, uc}) * m.fuzz;
        }
        out = {hit.p, refl, rin.tm};
        att = m.albedo;
        return !m.has_fuzz || dot(refl, hit.n) > (R)0.0;
    }
    /* DIELECTRIC material.rs:126-154 */
    att = {(R)1.0, (R)1.0, (R)1.0};
    R ri = hit.front ? (R)1.0 / m.ior : m.ior;
    V3<R> ud = unit(rin.d);
    R cos_theta = m_fmin(dot(-ud, hit.n), (R)1.0);
    R sin_theta = m_sqrt((R)1.0 - cos_theta * cos_theta);
    bool refl = ri * sin_theta > (R)1.0;
    if (!refl) {
        R ua, ub;
        dr.block(0, ua, ub);
        refl = schlick(cos_theta, ri) > ua;
    }
    V3<R> dir = refl ? reflect(ud, hit.n) : refract(ud, hit.n, ri);
    out = {hit.p, dir, rin.tm};
    return true;
}

/* ---- camera rays: camera/camera.rs:209-230, 248-252, 261-264 --------------------------------------- */
template <class R> __device__ inline RayT<R> ray_for(const DevCamera<R>& c, uint32_t wn, uint32_t hn, const Draws<R>& dr) {
    R ua, ub;
    dr.block(0, ua, ub);
    R ox = ua - (R)0.5, oy = ub - (R)0.5;
    V3<R> px = c.px00 + (c.du * ((R)wn + ox)) + (c.dv * ((R)hn + oy));
    V3<R> orig = c.lookfrom;
    if (!(c.defocus_angle <= (R)0.0)) {
        R uc, ud;
        dr.block(1, uc, ud);
        const R TAU = (R)6.28318530717958647692528676655900577;
        R r = m_sqrt(uc);
        R th = ud * TAU;
        R dx = r * m_cos(th), dy = r * m_sin(th);
        orig = c.lookfrom + (c.disk_u * dx) + (c.disk_v * dy);
    }
    R ue, uf;
    dr.block(2, ue, uf);
    return {orig, px - orig, ue};
}
template <class R> __device__ inline RayT<R> primary_ray(const DevCamera<R>& c, uint32_t wn, uint32_t hn) {
    V3<R> pc = c.ul + (c.du * (R)wn) + (c.dv * (R)hn);
    return {c.lookfrom, pc - c.lookfrom, (R)0.0};
}

/* ---- quantiser: core/rgb.rs:20-36 + utils/math.rs:13-21 ---------------------------------------------- */
template <class R> __device__ inline uint32_t quantize1(R x, uint32_t& nan) {
    R g = x > (R)0.0 ? m_sqrt(x) : (R)0.0;
    R v = g * (R)254.999;
    if (v != v) { nan = 1; return 0; }
    R c = v < (R)0.0 ? (R)0.0 : (v > (R)255.0 ? (R)255.0 : v);
    return (uint32_t)m_round(c);
}

/* ---- wave tile -> pixel: a wave owns an 8x8 pixel tile of the (compact) band ----------------------- */
struct PixelMap {
    uint32_t lx, ly; /* position in the compact band */
    uint32_t gy;     /* global row */
    bool valid;
};
__device__ inline PixelMap map_pixel(uint32_t wave_global, uint32_t lane, uint32_t W, const DevPart& part) {
    uint32_t tiles_x = (W + 7) / 8;
    uint32_t ty = wave_global / tiles_x, tx = wave_global - ty * tiles_x;
    PixelMap p;
    p.lx = tx * 8 + (lane & 7);
    p.ly = ty * 8 + (lane >> 3);
    p.valid = p.lx < W && p.ly < part.local_rows;
    p.gy = part_global_row(part, p.valid ? p.ly : 0);
    return p;
}

struct DevStats { /* device-side counters, one atomic per wave at kernel end */
    unsigned long long segments, narrow, wave_iters, nan_pixels;
};

template <class R> __device__ inline R wave_sum_u(R v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

/* ---- K1 ------------------------------------------------------------------------------------------ */
template <class R>
__global__ __launch_bounds__(MIRT_BLOCK) void primary_hits_kernel(DevScene<R> sc, DevCamera<R> cam, DevPart part, uint32_t* __restrict__ out_id,
                                                                  R* __restrict__ out_t) {
    __shared__ WaveLds<R> lds_all[MIRT_WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    WaveLds<R>& lds = lds_all[w];
    PixelMap pm = map_pixel(blockIdx.x * MIRT_WAVES_PER_BLOCK + w, lane, cam.W, part);
    RayT<R> ray = primary_ray(cam, pm.lx, pm.gy);
    Closest<R> best;
    uint32_t narrow = 0;
    trace(sc, lds, lane, pm.valid, ray, (R)0.001, best, narrow);
    if (pm.valid) {
        uint32_t row = part.compact ? pm.ly : pm.gy;
        size_t o = (size_t)row * cam.W + pm.lx;
        out_id[o] = best.orig; /* 0xFFFFFFFF on a miss */
        if (out_t) out_t[o] = best.t;
    }
}

/* ---- K2 (+K3) ------------------------------------------------------------------------------------ */
template <class R>
__global__ __launch_bounds__(MIRT_BLOCK) void render_kernel(DevScene<R> sc, DevCamera<R> cam, DevPart part, uint32_t k0, uint32_t k1,
                                                            R* __restrict__ out_lin, uint8_t* __restrict__ out_rgb8, DevStats* __restrict__ stats) {
    __shared__ WaveLds<R> lds_all[MIRT_WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    WaveLds<R>& lds = lds_all[w];
    PixelMap pm = map_pixel(blockIdx.x * MIRT_WAVES_PER_BLOCK + w, lane, cam.W, part);
    const uint32_t pixel = pm.gy * cam.W + pm.lx; /* RNG key: independent of the partition */
    const bool aa = cam.spp > 0;
    const uint32_t nsamples = aa ? cam.spp : 1u;

    V3<R> acc = {(R)0.0, (R)0.0, (R)0.0};
    V3<R> thr = {(R)1.0, (R)1.0, (R)1.0};
    RayT<R> ray = {};
    uint32_t sample = 0, depth_left = cam.depth, bounce = 0;
    uint32_t n_seg = 0, n_narrow = 0, n_iter = 0;
    bool running = pm.valid && cam.depth > 0; /* depth 0: color() returns black before any hit test */
    if (!running && pm.valid) sample = nsamples;
    if (running) {
        Draws<R> dr{pixel, 0u, 0u, k0, k1};
        ray = aa ? ray_for(cam, pm.lx, pm.gy, dr) : primary_ray(cam, pm.lx, pm.gy);
    }

    while (__ballot(running) != 0ull) {
        ++n_iter;
        Closest<R> best;
        trace(sc, lds, lane, running, ray, (R)0.001, best, n_narrow);
        if (running) {
            ++n_seg;
            bool done;
            V3<R> radiance = {(R)0.0, (R)0.0, (R)0.0};
            if (best.idx != 0xFFFFFFFFu) {
                Hit<R> hit = make_hit(sc, best, ray);
                ++bounce;
                Draws<R> dr{pixel, sample, bounce, k0, k1};
                V3<R> att;
                RayT<R> nxt;
                if (scatter(sc, sc.material[best.idx], ray, hit, dr, att, nxt)) {
                    thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
                    ray = nxt;
                    --depth_left;
                    done = depth_left == 0; /* color(.., 0) = black */
                } else {
                    done = true;
                }
            } else { /* sky: camera.rs:282-284 */
                V3<R> ud = unit(ray.d);
                R a = (R)0.5 * (ud.y + (R)1.0);
                V3<R> sky = {(R)1.0 * ((R)1.0 - a) + (R)0.5 * a, (R)1.0 * ((R)1.0 - a) + (R)0.7 * a, (R)1.0 * ((R)1.0 - a) + (R)1.0 * a};
                radiance = {thr.x * sky.x, thr.y * sky.y, thr.z * sky.z};
                done = true;
            }
            if (done) {
                acc = acc + radiance;
                ++sample;
                if (sample < nsamples) { /* regenerate: this lane starts its next sample at once */
                    thr = {(R)1.0, (R)1.0, (R)1.0};
                    depth_left = cam.depth;
                    bounce = 0;
                    Draws<R> dr{pixel, sample, 0u, k0, k1};
                    ray = ray_for(cam, pm.lx, pm.gy, dr);
                } else {
                    running = false;
                }
            }
        }
    }

    uint32_t nan = 0;
    if (pm.valid) {
        if (aa) acc = acc * ((R)1.0 / (R)cam.spp); /* samples_scale, camera.rs:50,246 */
        uint32_t row = part.compact ? pm.ly : pm.gy;
        size_t o = ((size_t)row * cam.W + pm.lx) * 3;
        if (out_lin) {
            out_lin[o] = acc.x;
            out_lin[o + 1] = acc.y;
            out_lin[o + 2] = acc.z;
        }
        uint32_t r = quantize1(acc.x, nan), g = quantize1(acc.y, nan), b = quantize1(acc.z, nan);
        if (out_rgb8) {
            out_rgb8[o] = (uint8_t)r;
            out_rgb8[o + 1] = (uint8_t)g;
            out_rgb8[o + 2] = (uint8_t)b;
        }
    }
    if (stats) {
        uint32_t s1 = wave_sum_u(n_seg), s2 = wave_sum_u(n_narrow), s3 = wave_sum_u(nan);
        if (lane == 0) {
            atomicAdd(&stats->segments, (unsigned long long)s1);
            atomicAdd(&stats->narrow, (unsigned long long)s2);
            atomicAdd(&stats->wave_iters, (unsigned long long)n_iter);
            if (s3) atomicAdd(&stats->nan_pixels, (unsigned long long)s3);
        }
    }
}

/* ---- K3 ------------------------------------------------------------------------------------------ */
template <class R> __global__ void quantize_kernel(const R* __restrict__ lin, size_t n, uint8_t* __restrict__ out, unsigned long long* nan_count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t nan = 0;
    out[i] = (uint8_t)quantize1(lin[i], nan);
    if (nan && nan_count) atomicAdd(nan_count, 1ull);
}

/* ---- K4: gathered bands -> frame.  One thread per 4 bytes where possible. --------------------------- */
__global__ void deinterleave_kernel(const uint8_t* __restrict__ bands, size_t band_stride, uint32_t part_count, uint32_t strip_rows,
                                    uint32_t width, uint32_t height, uint32_t elem_bytes, uint8_t* __restrict__ frame) {
    size_t row_bytes = (size_t)width * elem_bytes;
    size_t total = row_bytes * height;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= total) return;
    uint32_t gy = (uint32_t)(i / row_bytes);
    size_t in_row = i - (size_t)gy * row_bytes;
    uint32_t strip = gy / strip_rows, within = gy - strip * strip_rows;
    uint32_t part = strip % part_count, ls = strip / part_count;
    /* rows of earlier local strips are all full strips (only the last global strip can be short) */
    size_t lr = (size_t)ls * strip_rows + within;
    frame[i] = bands[(size_t)part * band_stride + lr * row_bytes + in_row];
}

} /* namespace mirt */
#endif
