/*
 * mirt_kernels.h — the HIP kernels of the per-pixel hot path (gfx950, wave64).
 *
 *   K1 primary_hits_kernel  deterministic primary ray -> (hit id, t)         camera.rs:247-254 + hittable.rs:92-98
 *   K2 render_kernel        ray generation + bounce loop + accumulation      camera.rs:209-259, 268-286
 *                           (+ K3 gamma/quantise fused in the epilogue)      core/rgb.rs:20-36
 *   K3 quantize_kernel      stand-alone quantiser
 *   K4 deinterleave_kernel  gathered row bands -> frame
 *
 * Closest-hit query: brute force over all spheres, as an n-body style tile loop, or - on request, f32 only - a
 * bounding-volume tree (`trace_bvh`).
 *   broad phase : every lane tests every sphere.  `trace_mx` (the default, either precision): a conservative bounding-ball
 *                 test evaluated whole on the matrix pipe (split-f16 operands, two chained v_mfma_f32_32x32x16_f16), which
 *                 leaves 1 VALU per pair; `trace_cons` (f32): the same test in 8 VALU per pair, records from LDS as
 *                 wave-uniform broadcasts; `trace` (f64): the discriminant d of sphere.rs:81-90 itself.  Lanes only
 *                 remember WHICH spheres may be hit (a short per-lane list in LDS).
 *   narrow phase: each lane then walks its own list and runs the exact root selection of sphere.rs:91-101
 *                 (sqrt, true divisions in f64 / the ray's reciprocal in the f32 spec, strict interval) - a few iterations per lane instead of one
 *                 divergent slow path per (wave, sphere).
 * Results are identical to testing the spheres one after the other in insertion order with a shrinking
 * interval (ties in t go to the lowest hit id), which is what the reference's BVH walk computes.
 */
#ifndef MIRT_KERNELS_H
#define MIRT_KERNELS_H

#include "mirt_device.h"

namespace mirt {

/* Wave priority: the broad phases (long runs of independent MFMA / VALU work) issue at priority 0, everything else - the exact
 * tests, shading, ray generation: short dependent chains - at priority 1, so that on a SIMD whose waves are in different
 * phases the latency-bound ones are served first and the tile loops fill what is left (bench frame 12.97 -> 12.61 ms; a
 * raised priority around the MFMA cluster itself, or for the whole broad phase: null / +0.5 %).  s_setprio is a scalar
 * instruction: all call sites are wave-uniform. */
#ifndef MIRT_PHASE_PRIO
#define MIRT_PHASE_PRIO 1
#endif
__device__ inline void prio_broad() {
#if MIRT_PHASE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ inline void prio_other() {
#if MIRT_PHASE_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
}

/* ---- discriminant of sphere.rs:83-90 -------------------------------------------------------------
 * double: the reference's expressions, one IEEE operation at a time (d = h*h - a*c with c = |oc|^2 - r^2).
 * float (the f32 spec of the throughput kernels): the same quantity through the perpendicular from the centre to the
 * line, d = a (r^2 - |oc - (h/a) D|^2).  The textbook form subtracts two numbers of size |oc|^2 to get one of size r^2:
 * in binary32, for a sphere of radius 0.1 seen from 100 units away, that leaves d with an error of ~5 % of r^2, roots off
 * by up to ~0.03, hit points well inside the sphere - and scattered rays that hit the same sphere again from within (the
 * 10 000-sphere stress frame traced 60 % more segments than in binary64, and darker spheres).  The perpendicular form has
 * no such cancellation (Haines, Guenther, Akenine-Moeller, "Precision improvements for ray/sphere intersection", Ray
 * Tracing Gems ch. 7); h / a is taken as h * (1 / a) with one correctly rounded division per ray. */
template <class R> struct RayA;
template <> struct RayA<double> { double a; };
template <> struct RayA<float> { float a, inv; };
__device__ inline RayA<double> ray_a(V3<double> d) { return {dot(d, d)}; }
__device__ inline RayA<float> ray_a(V3<float> d) {
    const float a = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
    return {a, 1.0f / a};
}

/* sphere.rs:92 / :96, (h -+ sqrt(d)) / a: the f32 spec multiplies by the ray's 1 / a */
__device__ inline double root_over_a(double num, RayA<double> ra) { return num / ra.a; }
__device__ inline float root_over_a(float num, RayA<float> ra) { return num * ra.inv; }

__device__ inline void disc(V3<double> oc, V3<double> D, RayA<double> ra, double r2, double& h, double& d, double& c) {
    h = dot(D, oc);
    c = dot(oc, oc) - r2;
    d = h * h - ra.a * c;
}
__device__ inline void disc(V3<double> oc, V3<double> D, RayA<double> ra, double r2, double& h, double& d) {
    double c;
    disc(oc, D, ra, r2, h, d, c);
}
__device__ inline void disc(V3<float> oc, V3<float> D, RayA<float> ra, float r2, float& h, float& d) {
    h = __builtin_fmaf(D.z, oc.z, __builtin_fmaf(D.y, oc.y, D.x * oc.x));
    const float q = h * ra.inv;
    const float lx = __builtin_fmaf(-q, D.x, oc.x), ly = __builtin_fmaf(-q, D.y, oc.y), lz = __builtin_fmaf(-q, D.z, oc.z);
    d = ra.a * (r2 - __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx)));
}
/* |h| below which h*h may be subnormal (the behind-the-ray shortcut of narrow_test needs h*h normal) */
__device__ inline float h_normal_min(float) { return 1e-18f; }
__device__ inline double h_normal_min(double) { return 1e-150; }
/* sphere.rs:61-70 */
__device__ inline V3<double> center_at(V3<double> c0, V3<double> dc, double tm) { return {c0.x + tm * dc.x, c0.y + tm * dc.y, c0.z + tm * dc.z}; }
__device__ inline V3<float> center_at(V3<float> c0, V3<float> dc, float tm) {
    return {__builtin_fmaf(tm, dc.x, c0.x), __builtin_fmaf(tm, dc.y, c0.y), __builtin_fmaf(tm, dc.z, c0.z)};
}

template <class R> struct RayT { /* core/ray.rs */
    V3<R> o, d;
    R tm;
};

/* Per-wave LDS: one tile of (centre, r^2) records and the per-lane candidate lists (indices relative to the
 * first sphere of the current pass, 16 bit).  f32: 1 KiB + 2 KiB per wave. */
template <class R> struct WaveLds {
    typename Vec4<R>::type tileA[MIRT_TILE];
    uint16_t cand[MIRT_CAND_CAP][MIRT_WAVE];
};

template <class R> struct Closest {
    R t;
    uint32_t idx;  /* sorted index, 0xFFFFFFFF = none */
};

/* Where the exact sphere records of the narrow phase / hit record come from: HBM (through L2), or the block's LDS
 * copy (f32 kernels, scenes of up to MIRT_LDS_RECORDS spheres: ~100 cycles instead of ~600+ per dependent load). */
template <class R> struct GeomGlobal {
    const DevScene<R>& sc;
    __device__ inline typename Vec4<R>::type a(uint32_t i) const { return sc.A[i]; }
    __device__ inline typename Vec4<R>::type b(uint32_t i) const { return sc.B[i]; }
    __device__ inline uint32_t material(uint32_t i) const { return sc.material[i]; }
};

struct DevStats { /* device-side counters, one atomic per wave at kernel end */
    unsigned long long segments, narrow, wave_iters, nan_pixels;
    unsigned int queue_head; /* next unclaimed work item of the running render launch */
    unsigned int cont_count; /* paths handed to the continuation pass (render_body, T::CONTINUES) */
    unsigned long long mx_tiles; /* matrix-pipe kernels: (wave, tile of 32 spheres) evaluations, i.e. what the tile cull left of wave_iters x tiles */
    unsigned long long phase_cycles[6]; /* diagnostic builds only (MIRT_PHASE_TIMING) */
    unsigned long long broad_misses;    /* diagnostic builds only (MIRT_VERIFY_BROAD): must be 0 */
    /* diagnostic builds only (MIRT_PHASE_TIMING), in s_memrealtime ticks (100 MHz): */
    unsigned long long inv_start_max;   /* ~(earliest wave start) */
    unsigned long long end_max;         /* latest wave end */
    unsigned long long life_sum;        /* sum over waves of (end - start) */
    unsigned long long n_waves;
    unsigned long long start_sum;       /* sum over waves of their start tick minus the launch's base tick (stats->queue time): see host */
    unsigned long long idle_sum;        /* sum over waves of (end of the wave - its last iteration with a full complement of work): the drain */
    unsigned long long bvh_lane_steps, bvh_wave_steps, bvh_leaf_rounds, bvh_lane_leaves; /* -DMIRT_PHASE_TIMING -DMIRT_BVH_COUNTERS */
    unsigned long long diag[48]; /* -DMIRT_PHASE_TIMING: MIRT_DIAG, [k] wave-level entries of block k, [24 + k] lanes that entered it */
    unsigned int waves_done;     /* T::CONTINUES, first launch: waves that have ended (render_body) */
    unsigned int _pad2;
};

/* Diagnostic build only (-DMIRT_PHASE_TIMING -DMIRT_BLOCK_COUNTS): wave-level execution counts of the divergent blocks of the segment loop
 * (DevStats::diag[k] += 1 whenever at least one lane of a wave enters block k); printed by the host next to the phase cycles. */
/* sc.dbg points at DevStats::broad_misses; diag[] sits this many 64-bit words behind it (round 2 added two fields in between and left
 * the constant at 9: every label of profiles/r02_*_phase_split.txt printed after that is two rows off) */
#define MIRT_DIAG_BASE ((offsetof(DevStats, diag) - offsetof(DevStats, broad_misses)) / 8)
#ifdef MIRT_BLOCK_COUNTS /* a build of its own (with -DMIRT_PHASE_TIMING for the printing): the atomics distort the cycle split */
#define MIRT_DIAG(sc, k, cond)                                                                                   \
    do {                                                                                                         \
        const unsigned long long b_ = __ballot(cond);                                                            \
        if (b_ != 0ull && (cond) && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)b_) - 1)) {             \
            atomicAdd((unsigned long long*)(sc).dbg + MIRT_DIAG_BASE + (k), 1ull);                               \
            atomicAdd((unsigned long long*)(sc).dbg + MIRT_DIAG_BASE + 24 + (k), (unsigned long long)__popcll(b_)); \
        }                                                                                                        \
    } while (0)
#else
#define MIRT_DIAG(sc, k, cond) do { } while (0)
#endif
enum { DG_NARROW_ROUND = 0, DG_NARROW_MOVING, DG_NARROW_ROOTS, DG_NARROW_FAR, DG_RAYGEN, DG_HIT, DG_LAMBERT, DG_METAL, DG_FUZZ, DG_GLASS, DG_IMAGE,
       DG_CHECKER, DG_SKY, DG_STORE, DG_DEAL, DG_ATOMIC, DG_SCATTERED, DG_PASS, DG_N };

/* exact per-candidate test: sphere.rs:91-101 against the interval (tmin, best.t), ties -> lowest hit id.  The hit
 * id of a sphere is only looked up on an exact tie in t (and once for the winner, by the caller). */
template <class R, class G>
__device__ inline void narrow_test(const DevScene<R>& sc, const G& geo, uint32_t i, const RayT<R>& ray, RayA<R> ra, R tmin, Closest<R>& best) {
    typename Vec4<R>::type sa = geo.a(i);
    V3<R> C = {sa.x, sa.y, sa.z};
    if (i >= sc.n_static) {
        MIRT_DIAG(sc, DG_NARROW_MOVING, true);
        typename Vec4<R>::type sb = geo.b(i);
        C = center_at(C, V3<R>{sb.x, sb.y, sb.z}, ray.tm);
    }
    V3<R> oc = C - ray.o;
    R h, d;
    /* Sphere entirely behind the ray (h < 0, origin outside): sqrt(d) <= -h, so both roots (h -+ sqrt(d))/a are <= 0 < tmin.
     * Exact shortcuts: they reject only what the root test below would reject, before paying for the sqrt and the two
     * divisions.  double: c > 0 gives d = rn(h*h - a*c) <= rn(h*h), and sqrt(rn(h*h)) == |h| exactly when h*h is normal.
     * float: d <= rn(h*h) is tested as such (the f32 discriminant does not go through c). */
    bool behind;
    if constexpr (sizeof(R) == 8) {
        R c;
        disc(oc, ray.d, ra, sa.w, h, d, c);
        behind = c > (R)0.0;
    } else {
        disc(oc, ray.d, ra, sa.w, h, d);
        behind = d <= h * h;
    }
    if (!(d >= (R)0.0)) return;
    if (h < -h_normal_min(h) && behind && tmin >= (R)0.0) return;
    MIRT_DIAG(sc, DG_NARROW_ROOTS, true);
    R sq = m_sqrt(d);
    /* strict interval (utils/interval.rs:37-39); an exact tie with the current best goes to the lower hit id
     * (never taken while there is no best: t < +inf is required, as in the reference) */
    auto inside = [&](R t) {
        return (tmin < t) && (t < best.t || (t == best.t && best.idx != 0xFFFFFFFFu && sc.orig_id[i] < sc.orig_id[best.idx]));
    };
    R t = root_over_a(h - sq, ra);
    bool ok = inside(t);
    if (!ok) {
        MIRT_DIAG(sc, DG_NARROW_FAR, true);
        t = root_over_a(h + sq, ra);
        ok = inside(t);
    }
    if (ok) {
        best.t = t;
        best.idx = i;
    }
}

/* Diagnostic build only (-DMIRT_PHASE_TIMING): per-wave cycle stamps (s_memtime) around the phases of the segment
 * loop, summed into DevStats::phase_cycles.  The product build compiles this to nothing. */
struct PhaseTimer {
#ifdef MIRT_PHASE_TIMING
    unsigned long long last, acc[6];
    __device__ inline void start() { last = __builtin_amdgcn_s_memtime(); for (int i = 0; i < 6; ++i) acc[i] = 0; }
    __device__ inline void stamp(int i) { unsigned long long now = __builtin_amdgcn_s_memtime(); acc[i] += now - last; last = now; }
#else
    __device__ inline void start() {}
    __device__ inline void stamp(int) {}
#endif
};
enum { PH_QUEUE = 0, PH_RAYGEN = 1, PH_BROAD = 2, PH_NARROW = 3, PH_SHADE = 4, PH_OTHER = 5 };

__device__ inline float m_max4(float a, float b, float c, float d) { return fmaxf(fmaxf(a, b), fmaxf(c, d)); }
__device__ inline double m_max4(double a, double b, double c, double d) { return fmax(fmax(a, b), fmax(c, d)); }

/* Closest hit in (tmin, +inf).  Wave-synchronous: every lane of the wave must call it; lanes with
 * active == false do no narrow-phase work.  Returns the winner in `best` (idx == ~0u on a miss).
 *
 * Spheres are visited in groups of MIRT_GROUP (the sorted arrays are padded to whole groups with dummy
 * records, r*r = -inf, whose discriminant can never be >= 0): one batch of wave-uniform operand fetches, four
 * independent discriminant chains, ONE wave-level test "does any lane have a candidate in this group".
 *
 * Where the wave-uniform sphere data comes from (gfx9 VALU instructions take at most ONE scalar operand):
 *   (centre0, r^2)  LDS tile, broadcast ds_read_b128 -> VGPRs      (moving spheres; statics too unless STATIC_SMEM)
 *   (centre1-centre0) scalar loads (s_load through the scalar cache) -> SGPRs, used directly as the one scalar
 *                   operand of v_fma_f32(tm, dc, c0): no LDS bandwidth, no VALU moves
 *   STATIC_SMEM:    static spheres need one scalar operand per instruction only, so their records can come
 *                   through the scalar cache as well and skip LDS altogether. */
template <class R, bool STATIC_SMEM>
__device__ inline void trace(const DevScene<R>& sc, WaveLds<R>& lds, uint32_t lane, bool active, const RayT<R>& ray, R tmin,
                             Closest<R>& best, uint32_t& narrow_count, PhaseTimer& pt) {
    typedef typename Vec4<R>::type R4;
    typedef R R4v __attribute__((ext_vector_type(4)));
    typedef const R4v __attribute__((address_space(4))) CR4; /* constant address space: uniform index -> s_load */
    CR4* Ac = (CR4*)sc.A;
    CR4* Bc = (CR4*)sc.B;
    best.t = (R)INFINITY;
    best.idx = 0xFFFFFFFFu;
    /* an inactive lane gets a = NaN: every d is NaN and `d >= 0` is false for free */
    const RayA<R> a_true = ray_a(ray.d);
    RayA<R> a = a_true;
    if (!active) a = ray_a(V3<R>{(R)NAN, (R)NAN, (R)NAN});
    const V3<R> O = ray.o, D = ray.d;
    const R tm = ray.tm;
    const uint32_t ngroups = sc.n / MIRT_GROUP;         /* n and n_static are multiples of MIRT_GROUP */
    const uint32_t gstatic = sc.n_static / MIRT_GROUP;  /* groups [0, gstatic) static, the rest moving */
    const uint32_t groups_per_tile = MIRT_TILE / MIRT_GROUP;

    uint32_t g = 0;
    while (g < ngroups) { /* one pass unless some lane's candidate list fills up */
        uint32_t cnt = 0;
        bool flush = false;
        const uint32_t pass_base = g * MIRT_GROUP;
        uint32_t glimit = g + 0x10000u / MIRT_GROUP; /* candidate indices are 16 bit, relative to pass_base */
        glimit = glimit < ngroups ? glimit : ngroups;
        /* ---- broad phase ---------------------------------------------------------------------- */
        while (g < glimit && !flush) {
            const uint32_t tile = g / groups_per_tile;
            const uint32_t base = tile * MIRT_TILE;
            if (!STATIC_SMEM || base + MIRT_TILE > sc.n_static) { /* stage the tile: one coalesced record per lane */
                uint32_t li = base + lane;
                lds.tileA[lane] = sc.A[li < sc.n ? li : sc.n - 1];
            }
            uint32_t gend = (tile + 1) * groups_per_tile;
            gend = gend < glimit ? gend : glimit;
            for (; g < gend; ++g) {
                const uint32_t i0 = g * MIRT_GROUP;
                const uint32_t j = i0 - base;
                R h, d0, d1, d2, d3;
                if (g >= gstatic) { /* moving: centre at the ray's time first (sphere.rs:61-70) */
                    R4 s0 = lds.tileA[j], s1 = lds.tileA[j + 1], s2 = lds.tileA[j + 2], s3 = lds.tileA[j + 3];
                    R4v q0 = Bc[i0], q1 = Bc[i0 + 1], q2 = Bc[i0 + 2], q3 = Bc[i0 + 3];
                    disc(center_at(V3<R>{s0.x, s0.y, s0.z}, V3<R>{q0.x, q0.y, q0.z}, tm) - O, D, a, s0.w, h, d0);
                    disc(center_at(V3<R>{s1.x, s1.y, s1.z}, V3<R>{q1.x, q1.y, q1.z}, tm) - O, D, a, s1.w, h, d1);
                    disc(center_at(V3<R>{s2.x, s2.y, s2.z}, V3<R>{q2.x, q2.y, q2.z}, tm) - O, D, a, s2.w, h, d2);
                    disc(center_at(V3<R>{s3.x, s3.y, s3.z}, V3<R>{q3.x, q3.y, q3.z}, tm) - O, D, a, s3.w, h, d3);
                } else if (STATIC_SMEM) {
                    R4v s0 = Ac[i0], s1 = Ac[i0 + 1], s2 = Ac[i0 + 2], s3 = Ac[i0 + 3];
                    disc(V3<R>{s0.x, s0.y, s0.z} - O, D, a, s0.w, h, d0);
                    disc(V3<R>{s1.x, s1.y, s1.z} - O, D, a, s1.w, h, d1);
                    disc(V3<R>{s2.x, s2.y, s2.z} - O, D, a, s2.w, h, d2);
                    disc(V3<R>{s3.x, s3.y, s3.z} - O, D, a, s3.w, h, d3);
                } else {
                    R4 s0 = lds.tileA[j], s1 = lds.tileA[j + 1], s2 = lds.tileA[j + 2], s3 = lds.tileA[j + 3];
                    disc(V3<R>{s0.x, s0.y, s0.z} - O, D, a, s0.w, h, d0);
                    disc(V3<R>{s1.x, s1.y, s1.z} - O, D, a, s1.w, h, d1);
                    disc(V3<R>{s2.x, s2.y, s2.z} - O, D, a, s2.w, h, d2);
                    disc(V3<R>{s3.x, s3.y, s3.z} - O, D, a, s3.w, h, d3);
                }
                R dm = m_max4(d0, d1, d2, d3); /* fmax ignores NaN operands: NaN = miss */
                if (__ballot(dm >= (R)0.0) != 0ull) { /* wave-uniform branch, rarely taken */
                    const uint16_t r0 = (uint16_t)(i0 - pass_base);
                    if (d0 >= (R)0.0) { lds.cand[cnt][lane] = r0; ++cnt; }
                    if (d1 >= (R)0.0) { lds.cand[cnt][lane] = r0 + 1; ++cnt; }
                    if (d2 >= (R)0.0) { lds.cand[cnt][lane] = r0 + 2; ++cnt; }
                    if (d3 >= (R)0.0) { lds.cand[cnt][lane] = r0 + 3; ++cnt; }
                    if (__ballot(cnt > MIRT_CAND_CAP - MIRT_GROUP) != 0ull) { /* no room for another group */
                        flush = true;
                        ++g;
                        break;
                    }
                }
            }
        }
        pt.stamp(PH_BROAD);
        /* ---- narrow phase: each lane walks its own list ------------------------------------------ */
        for (uint32_t k = 0; k < MIRT_CAND_CAP; ++k) {
            if (__ballot(k < cnt) == 0ull) break;
            if (k < cnt) {
                uint32_t i = pass_base + lds.cand[k][lane];
                narrow_test(sc, GeomGlobal<R>{sc}, i, ray, a_true, tmin, best);
                ++narrow_count;
            }
        }
        pt.stamp(PH_NARROW);
    }
}

/* ---- f32 throughput mode: conservative broad phase ---------------------------------------------------
 * The broad phase only has to find a SUPERSET of the spheres whose exact discriminant (disc(), the f32 spec) is
 * >= 0; the narrow phase re-evaluates the exact discriminant and runs the exact root selection, so the final
 * (id, t) is unchanged.  That freedom buys a cheaper test:
 *   - every sphere, static or moving, is replaced by a ball (M, rho) that contains it for all shutter times
 *     tm in [0,1]:  M = c0 + dc/2, rho = r + |dc|/2 (+ rounding slack)                     [host, in double]
 *   - with a NORMALISED direction Dn = D/|D| the line-ball test is  (Dn.(M-O))^2 - |M-O|^2 + rho^2 >= 0,
 *     expanded so that everything that depends on the ray alone or the sphere alone is hoisted:
 *         h = Dn.M - k1                k1 = Dn.O                      3 fma
 *         q = h*h + 2 O.M - k2 - w     k2 = |O|^2 (1 - 160 eps)       1 sub + 4 fma
 *         candidate  <=>  q >= 0       w  = |M|^2 - rho^2 - 160 eps (|M|^2 + rho^2)
 *     8 VALU instructions per (ray, sphere) instead of 12 (static) / 15 (moving), one 16-byte record per sphere.
 *   - the records come from LDS as wave-uniform broadcast ds_read_b128 into VGPRs, sixteen per loop iteration (a VALU
 *     instruction with an SGPR source issues at HALF rate on gfx950 - bench/sgpr_rate.hip - and the CU has ONE scalar
 *     ALU for its four SIMDs, so neither the scalar cache nor per-sphere scalar work is affordable).
 *   - what a lane remembers: the sign bit of every q is shifted into a per-lane word (v_alignbit_b32, ONE VALU op
 *     per pair), and once per group of sixteen the word (group number << 16 | sign bits) is written to the lane's
 *     entry list in LDS UNCONDITIONALLY; the list cursor advances only if some sign bit says "candidate".  No
 *     branch, no exec-mask juggling per sphere: with incoherent bounce rays some lane has a candidate in almost
 *     every group, so a "rarely taken" slow path does not exist (the first version had one: 8 branches per group).
 *   - the expansion cancels (|M|^2 + |O|^2 - 2 O.M), so both sides carry a rigorous rounding margin: the exact
 *     discriminant can be >= 0 only if the real line passes within r^2 + 32 eps (|C-O|^2 + r^2) of the centre,
 *     and the evaluation above is off by at most 40 eps ((|M|+|O|)^2 + rho^2); 160 eps (|M|^2 + rho^2) on the
 *     sphere side plus 160 eps |O|^2 on the ray side covers both (eps = 2^-24).  A diagnostic build
 *     (-DMIRT_VERIFY_BROAD) re-tests every pair exactly and counts misses: 0 on every workload in bench.py.
 *   - scenes of MIRT_CONS_SPARSE_FROM sorted slots and more (the streaming loop): a group of sixteen holds a candidate for some
 *     lane of the wave only now and then, so the loop keeps the sixteen values, takes their maximum three at a time and collects
 *     signs only in a group whose maximum says so; and with one comparison per group, k2 stays out of the per-record sum:
 *         q + k2 = h*h + 2 O.M - w  >=  k2        7 instructions per pair, the comparison itself is exact
 *     - the same real quantity with one rounding fewer, and the partial sums no longer carry |O|^2, so the evaluation error is
 *     at most the one derived above (10 000 spheres at 4K: 338 -> 298 ms, 0.67 -> 0.76 of the FP32 peak; verify build 0 misses). */
#define MIRT_RAY_SHRINK 0.99999f /* <= 1 - 160 * 2^-24 */
/* The broad phases only FILTER (their survivors go through the exact test), so the ray-side constants need not be correctly
 * rounded: v_rsq_f32 / v_sqrt_f32 (1 ulp) instead of sqrtf + division (~25 VALU instructions per ray and segment).  The
 * normalised direction is then off by <= 2.5 eps instead of <= 2 eps, i.e. +1 eps ((|M|+|O|)^2) on h^2 - far inside the slack
 * between the derived evaluation errors (40 eps / 50.4 eps) and the margins carried (160 eps / 256 eps); the -DMIRT_VERIFY_BROAD
 * build still counts 0 misses (profiles/r02_microbench.txt). */
__device__ inline float filter_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double filter_rsq(double x) { return 1.0 / sqrt(x); }
__device__ inline float filter_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

/* LDS of an f32 block: ONE block of 16 waves per CU (4 waves per SIMD), so that one copy of the scene serves them all.
 *   resident scenes (<= MIRT_LDS_RECORDS spheres): bounding records S for the broad phase, exact records A, B and
 *     material ids for the narrow phase and the hit record; nothing about a sphere is read from HBM in the loop
 *   larger scenes: wave w stages 64-record tiles of S in S[64 w .. 64 w + 63]; A, B, material come from HBM/L2
 *   ent: per-wave, per-lane entry lists (group number << 16 | inverted sign bits) */
#ifndef MIRT_F32_WAVES
#define MIRT_F32_WAVES 16
#endif
#ifndef MIRT_F32_EU
#define MIRT_F32_EU 4 /* waves per SIMD the render kernel is compiled for (VGPR budget = 512 / this) */
#endif
#define MIRT_F32_BLOCK (MIRT_F32_WAVES * MIRT_WAVE)
#define MIRT_LDS_RECORDS 1024
#ifndef MIRT_ENT_CAP
#define MIRT_ENT_CAP 16
#endif
#define MIRT_F32_LDS_MATS 512u
struct F32Lds {
    float4 S[MIRT_LDS_RECORDS];
    float4 A[MIRT_LDS_RECORDS];
    float4 B[MIRT_LDS_RECORDS];
    uint16_t mat[MIRT_LDS_RECORDS];
    uint32_t ent[MIRT_F32_WAVES][MIRT_ENT_CAP][MIRT_WAVE];
    DevMaterial<float> mats[MIRT_F32_LDS_MATS]; /* the scene's materials when they fit (36 KB): a bounce starts with this record */
    /* all threads of the block; true when the scene is resident */
    __device__ inline bool fill(const DevScene<float>& sc) {
        const uint32_t nrec = (sc.n + MIRT_CGROUP - 1) / MIRT_CGROUP * MIRT_CGROUP;
        const bool res = nrec <= MIRT_LDS_RECORDS && sc.n_materials <= 0x10000u;
        if (sc.n_materials <= MIRT_F32_LDS_MATS) {
            const uint32_t nd = sc.n_materials * (uint32_t)(sizeof(DevMaterial<float>) / 4);
            const uint32_t* src = (const uint32_t*)sc.mats;
            uint32_t* dst = (uint32_t*)mats;
            for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        }
        if (res) {
            for (uint32_t i = threadIdx.x; i < nrec; i += blockDim.x) {
                S[i] = sc.S[i]; /* padded to whole tiles on the host */
                if (i < sc.n) {
                    A[i] = sc.A[i];
                    B[i] = sc.B[i];
                    mat[i] = (uint16_t)sc.material[i];
                }
            }
        }
        __syncthreads();
        return res;
    }
};
template <class LDS> struct GeomLdsT {
    const LDS& L;
    __device__ inline float4 a(uint32_t i) const { return L.A[i]; }
    __device__ inline float4 b(uint32_t i) const { return L.B[i]; }
    __device__ inline uint32_t material(uint32_t i) const { return L.mat[i]; }
};
typedef GeomLdsT<F32Lds> GeomLds;

#ifndef MIRT_CONS_SPARSE_FROM
#define MIRT_CONS_SPARSE_FROM 4096u /* sorted slots from which the streaming vector-ALU loop takes the maximum of a group's q before it collects signs */
#endif
template <bool RES>
__device__ inline void trace_cons(const DevScene<float>& sc, F32Lds& L, uint32_t w, uint32_t lane, bool active, const RayT<float>& ray, float tmin,
                                  Closest<float>& best, uint32_t& narrow_count, PhaseTimer& pt) {
    best.t = INFINITY;
    best.idx = 0xFFFFFFFFu;
    const V3<float> O = ray.o;
    const RayA<float> a_true = ray_a(ray.d);
    const float inv = filter_rsq(a_true.a);
    const V3<float> Dn = {ray.d.x * inv, ray.d.y * inv, ray.d.z * inv};
    const float k1 = __builtin_fmaf(Dn.z, O.z, __builtin_fmaf(Dn.y, O.y, Dn.x * O.x));
    /* an inactive lane gets k2 = +inf: q = -inf, never a candidate */
    const float k2 = active ? __builtin_fmaf(O.z, O.z, __builtin_fmaf(O.y, O.y, O.x * O.x)) * MIRT_RAY_SHRINK : INFINITY;
    const V3<float> O2 = {O.x * 2.0f, O.y * 2.0f, O.z * 2.0f};
    const float nk2 = -k2;
    const uint32_t ngroups = (sc.n + MIRT_CGROUP - 1) / MIRT_CGROUP; /* the record array is padded to whole tiles */
    uint32_t(*ent)[MIRT_WAVE] = L.ent[w];
    const bool sparse = !RES && sc.n >= MIRT_CONS_SPARSE_FROM; /* see the group loop */

/* q for one record: 1 sub + 3 fma (e chain) + 3 fma (h chain) + 1 fma = 8 VALU.  q is never -0 (h*h >= +0), so its
 * sign bit is clear exactly for q >= +0 and for positive NaNs (kept: the exact test sorts them out). */
#define MIRT_CONS_H(s) __builtin_fmaf(Dn.z, s.z, __builtin_fmaf(Dn.y, s.y, __builtin_fmaf(Dn.x, s.x, -k1)))
#define MIRT_CONS_QW(s)                                                                                                  \
    __builtin_fmaf(MIRT_CONS_H(s), MIRT_CONS_H(s),                                                                       \
                   __builtin_fmaf(O2.z, s.z, __builtin_fmaf(O2.y, s.y, __builtin_fmaf(O2.x, s.x, nk2 - s.w))))
/* acc = (acc << 1) | signbit(q) */
#define MIRT_CONS_PUSH(s) acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(MIRT_CONS_QW(s)), 31)

    float4 pre = {0.f, 0.f, 0.f, 0.f}; /* streaming mode: prefetched record of tile `pre_tile` */
    uint32_t pre_tile = 0xFFFFFFFFu;
    uint32_t g = 0;
    while (g < ngroups) { /* one pass unless some lane's entry list fills up */
        prio_broad();
        uint32_t cnt = 0;
        const uint32_t gbase = g;
        uint32_t glimit = g + 0x10000u; /* group numbers in the entries are 16 bit, relative to gbase */
        glimit = glimit < ngroups ? glimit : ngroups;
        bool flush = false;
        while (g < glimit && !flush) {
            /* records of this tile: the block-resident LDS table, or (big scenes) a wave-private tile staged here */
            const uint32_t tile = g / (MIRT_TILE / MIRT_CGROUP);
            const float4* tp;
            if (RES) {
                tp = L.S + tile * MIRT_TILE;
            } else {
                /* one coalesced record per lane; the NEXT tile's record is already on its way from L2 while this one is
                 * tested (the array is padded to whole tiles plus one spare, so tile + 1 always exists) */
                if (pre_tile != tile) pre = sc.S[tile * MIRT_TILE + lane]; /* first tile of a pass, or resuming after a flush */
                L.S[w * MIRT_TILE + lane] = pre;
                pre = sc.S[(tile + 1) * MIRT_TILE + lane];
                pre_tile = tile + 1;
                tp = L.S + w * MIRT_TILE;
            }
            uint32_t gend = (tile + 1) * (MIRT_TILE / MIRT_CGROUP);
            gend = gend < glimit ? gend : glimit;
            for (; g < gend && !flush; ++g) {
                const float4* p = tp + (g * MIRT_CGROUP - tile * MIRT_TILE);
                uint32_t acc = g - gbase;
                if (sparse) {
                    /* Large scenes: a group of 16 records holds a candidate for SOME lane of the wave only now and then (10 000
                     * spheres: one group in six), and collecting a sign costs 3.4 issue cycles against the 2.2 of the
                     * instructions that compute q.  So the sixteen values are kept, their maximum is taken three at a time (8
                     * instructions instead of 16), and only a group in which some lane's maximum reaches the threshold collects its
                     * signs; a group without one writes nothing, which is what an all-ones word amounts to.  With one comparison per
                     * group instead of a sign per record the ray's k2 can stay out of the sum: q + k2 = h^2 + 2 O.M - w >= k2, one
                     * rounding less than q >= 0 (the margins cover either order of the same few roundings; the verify build re-tests
                     * every pair).  A NaN is skipped by the maximum and fails the comparison: such a sphere would have been a
                     * candidate only for the exact test to reject it. */
                    float q[MIRT_CGROUP]; /* q + k2: the ray's k2 is left out of the sum (7 instructions per record) and compared with once per group */
#define MIRT_CONS_QK(s) __builtin_fmaf(MIRT_CONS_H(s), MIRT_CONS_H(s), __builtin_fmaf(O2.z, s.z, __builtin_fmaf(O2.y, s.y, __builtin_fmaf(O2.x, s.x, -s.w))))
                    {
                        const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3], s4 = p[4], s5 = p[5], s6 = p[6], s7 = p[7];
                        q[0] = MIRT_CONS_QK(s0); q[1] = MIRT_CONS_QK(s1); q[2] = MIRT_CONS_QK(s2); q[3] = MIRT_CONS_QK(s3);
                        q[4] = MIRT_CONS_QK(s4); q[5] = MIRT_CONS_QK(s5); q[6] = MIRT_CONS_QK(s6); q[7] = MIRT_CONS_QK(s7);
                    }
                    {
                        const float4 s0 = p[8], s1 = p[9], s2 = p[10], s3 = p[11], s4 = p[12], s5 = p[13], s6 = p[14], s7 = p[15];
                        q[8] = MIRT_CONS_QK(s0); q[9] = MIRT_CONS_QK(s1); q[10] = MIRT_CONS_QK(s2); q[11] = MIRT_CONS_QK(s3);
                        q[12] = MIRT_CONS_QK(s4); q[13] = MIRT_CONS_QK(s5); q[14] = MIRT_CONS_QK(s6); q[15] = MIRT_CONS_QK(s7);
                    }
#undef MIRT_CONS_QK
                    float m = __builtin_fmaxf(q[0], q[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, q[r]), q[r + 1]); /* v_max3_f32 */
                    if (__ballot(m >= k2) == 0ull) { /* k2 = +inf for a lane without a ray */
#ifdef MIRT_VERIFY_BROAD
                        acc = (acc << 16) | 0xFFFFu;
#else
                        continue;
#endif
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(q[r] - k2), 31); /* the difference's sign is exact */
                    }
                } else {
                {
                    const float4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3], s4 = p[4], s5 = p[5], s6 = p[6], s7 = p[7];
                    MIRT_CONS_PUSH(s0); MIRT_CONS_PUSH(s1); MIRT_CONS_PUSH(s2); MIRT_CONS_PUSH(s3);
                    MIRT_CONS_PUSH(s4); MIRT_CONS_PUSH(s5); MIRT_CONS_PUSH(s6); MIRT_CONS_PUSH(s7);
                }
                {
                    const float4 s0 = p[8], s1 = p[9], s2 = p[10], s3 = p[11], s4 = p[12], s5 = p[13], s6 = p[14], s7 = p[15];
                    MIRT_CONS_PUSH(s0); MIRT_CONS_PUSH(s1); MIRT_CONS_PUSH(s2); MIRT_CONS_PUSH(s3);
                    MIRT_CONS_PUSH(s4); MIRT_CONS_PUSH(s5); MIRT_CONS_PUSH(s6); MIRT_CONS_PUSH(s7);
                }
                }
#ifdef MIRT_VERIFY_BROAD
                for (uint32_t k_ = 0; k_ < MIRT_CGROUP; ++k_) { /* exact discriminant >= 0 must imply the candidate bit */
                    const uint32_t i_ = g * MIRT_CGROUP + k_;
                    if (i_ >= sc.n) break;
                    float4 sa_ = sc.A[i_];
                    V3<float> C_ = {sa_.x, sa_.y, sa_.z};
                    if (i_ >= sc.n_static) {
                        float4 sb_ = sc.B[i_];
                        C_ = center_at(C_, V3<float>{sb_.x, sb_.y, sb_.z}, ray.tm);
                    }
                    float hh_, dd_;
                    disc(C_ - O, ray.d, a_true, sa_.w, hh_, dd_);
                    if (active && dd_ >= 0.0f && ((acc >> (MIRT_CGROUP - 1 - k_)) & 1u)) atomicAdd(sc.dbg, 1ull);
                }
#endif
                /* unconditional write, conditional advance (cnt < MIRT_ENT_CAP here: a full list flushes below) */
                ent[cnt][lane] = acc;
                cnt += ((acc & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
                flush = __ballot(cnt >= MIRT_ENT_CAP) != 0ull; /* no room for another entry */
            }
        }
        prio_other();
        pt.stamp(PH_BROAD);
        /* ---- narrow phase: each lane walks its own entries, exact test (sphere.rs:81-101) ---------------- */
        {
            uint32_t k = 0, m = 0, gi = 0;
            for (;;) {
                if (m == 0 && k < cnt) {
                    const uint32_t e = ent[k][lane];
                    ++k;
                    m = ~e & 0xFFFFu; /* never 0: the cursor only advanced past entries with a candidate */
                    gi = (gbase + (e >> 16)) * MIRT_CGROUP + (MIRT_CGROUP - 1);
                }
                if (__ballot(m != 0) == 0ull) break;
                if (m != 0) {
                    const uint32_t bit = (uint32_t)__ffs((int)m) - 1u; /* sign of record j sits at bit 15 - j */
                    m &= m - 1u;
                    if (RES) narrow_test(sc, GeomLds{L}, gi - bit, ray, a_true, tmin, best);
                    else narrow_test(sc, GeomGlobal<float>{sc}, gi - bit, ray, a_true, tmin, best);
                    ++narrow_count;
                }
            }
        }
        pt.stamp(PH_NARROW);
    }
#undef MIRT_CONS_QW
#undef MIRT_CONS_H
#undef MIRT_CONS_PUSH
}

/* ---- the broad phase on the matrix pipe ---------------------------------------------------------------
 * The conservative test of trace_cons, q = h^2 + e >= 0 with h = Dn.M - k1 and e = 2 O.M - k2 - w, is a polynomial in the
 * sphere's M whose coefficients depend on the ray alone:
 *     q = sum_{i<=j} c_ij M_i M_j + sum_i l_i M_i + c0 - w,
 *     c_ii = Dn_i^2, c_ij = 2 Dn_i Dn_j, l = 2 (O - k1 Dn), c0 = k1^2 - k2          (k1 = Dn.O, k2 ~ |O|^2 less its margins)
 * i.e. ONE (rays x K) x (K x spheres) contraction: sphere side M_i M_j (6), M_i (3), 1, w; ray side the coefficients.  Two
 * chained v_mfma_f32_32x32x16_f16 (K = 32) evaluate it for 32 rays x 32 spheres in 64 cycles and leave ONE VALU instruction
 * per pair, the v_alignbit that collects the sign of q (round 1 had the two dot products on the matrix pipe and h*h + e on
 * the vector ALU: 2 per pair; the vector-ALU loop needs 9).  About 4.5 VALU instructions per MFMA issue for free in its
 * shadow (bench/mfma_coexec.hip), so a tile of 4 MFMA + 39 VALU costs ~170 SIMD-cycles against ~230 before.
 *
 * Precision.  f16 operands carry 11 bits, so every value x is split x ~ xh + xl (xh = f16(x), xl = f16(x - xh),
 * |x - xh - xl| <= max(2^-22 |x|, 2^-25) =: delta(x), |xl| <= 2^-11 |x| + 2^-25) and every product x y takes the three
 * partial products xh yh + xh yl + xl yh: off by xl yl + delta(x) |y| + |x| delta(y) <= 3 * 2^-22 |x||y| + 2^-25 (|x| + |y|)
 * (+ terms of order 2^-36).  f16 x f16 products are exact in the f32 accumulator, f16 subnormals are honoured, and the
 * accumulation of one MFMA is off by <= 2^-22.8 of the sum of |terms| (bench/mfma_probe.hip); two chained ones by
 * <= 2^-21.8, budgeted as a = 2^-21.  Everything is evaluated in the scene's power-of-two scale (mx_scale: exact in f32),
 * chosen so that max(|M|+rho) <= 64 (products M_i M_j <= 4096, l_i <= 380, c0 <= 36000: all inside the f16 range) as long
 * as |O|^2 <= MIRT_MX_K2MAX, which holds for every ray that starts on a sphere; a ray that violates it (far camera: the
 * host then picks the VALU kernel anyway; NaN) simply keeps every sphere as a candidate.
 *   K slots (ray value x sphere value); p runs over the nine products xx yy zz xy xz yz | x y z:
 *     3p, 3p+1, 3p+2   (ch, Ph) (ch, Pl) (cl, Ph)         27-29  (c0h, 1) (c0m, 1) (c0l, 1)  (c0 split three ways)
 *     30, 31           (-1, wh) (-1, wl)                   (w rounded one-sidedly on the host: wh + wl <= w)
 *   Error of the evaluated q against the real-number q of the exact Dn, O, M, with eps = 2^-24, in scaled units, |Dn| <= 1.0001:
 *     sum of |terms| <= |M|^2 (quadratic) + 2 |O||M| (linear) + |O|^2 + |M|^2 + rho^2  <=  2 (|M|+|O|)^2 + rho^2
 *     partial products       12 eps (|M|^2 + 2|O||M|)  <= 12 eps (|M|+|O|)^2      + 2^-25 (4.1 + 1.8 |M| + 3.6 |O|)
 *     accumulation            8 eps (2 (|M|+|O|)^2 + rho^2)
 *     f32 coefficients        c_ij: 9 eps |M|^2 (Dn = d * rsq(a): 4 eps per component), l: 26 eps |O||M|, c0: 19 eps |O|^2
 *                             <= 19 eps (|M|+|O|)^2
 *   Together with the 32 eps of "exact discriminant >= 0 => the line passes the ball" the relative part is
 *   <= 79 eps (|M|+|O|)^2 + 8 eps rho^2, inside the 128 eps ((|M|+|O|)^2 + rho^2) <= 256 eps (|M|^2 + |O|^2 + rho^2) that
 *   the margins provide; the absolute part 2^-25 (4.1 + 1.8 |M| + 3.6 |O|) is linear.  So the records carry
 *   w = |M|^2 - rho^2 - 256 eps (|M|^2 + rho^2) - 2^-21 |M| (rounded down in the split) and the ray
 *   k2 = |O|^2 (1 - 335 eps) - 2^-20.9 |O| - 5.4 * 2^-25.  The absolute part is what limits the dynamic range: where it is
 *   not small against rho^2 for most spheres (radius-0.01 spheres on a radius-1e5 ground) the host keeps the scene on the
 *   VALU kernel.  -DMIRT_VERIFY_BROAD re-tests every pair exactly here as well.
 *   Operand lanes: sphere rows come from the A operand (lane l: row l % 32, K half l / 32), rays are the columns (B
 *   operand, lane l: ray l % 32, K half l / 32), so the wave's 64 rays take two chains per tile and each lane supplies one
 *   K half of its own ray and receives the other lane's half by v_permlane32_swap; a result lane holds 16 of the 32 rows of
 *   one column - rows are ordered so that these are 16 consecutive sorted spheres - and the same swap brings the two
 *   16-bit words of a ray together.  The entry lists and the narrow phase are trace_cons's. */
#define MIRT_MX_K2MAX 36000.0f
#ifndef MIRT_MX_MARGIN_TEST /* diagnostic builds: scale every margin of trace_mx (1 = the derived ones) to see how much slack they carry */
#define MIRT_MX_MARGIN_TEST 1.0
#endif
#define MIRT_MX_SHRINK (1.0f - 2.0e-5f * (float)MIRT_MX_MARGIN_TEST) /* <= 1 - 335 * 2^-24 */
#define MIRT_MX_FLOOR_LIN (5.2e-7f * (float)MIRT_MX_MARGIN_TEST) /* >= 2^-20.9 */
#define MIRT_MX_FLOOR_ABS (1.6e-7f * (float)MIRT_MX_MARGIN_TEST) /* >= 5.1 x 2^-25 */
typedef _Float16 mirt_h8 __attribute__((ext_vector_type(8)));
typedef float mirt_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t mirt_u4 __attribute__((ext_vector_type(4)));
/* Two layouts of the 160 KB: scenes of up to 512 sorted slots (the bench scene) keep 512 material records (one per sphere at most, unless the description carries unused ones) beside their
 * operands; up to 1024 slots the operands take twice the room and the materials get what is left. */
template <uint32_t RECORDS, uint32_t MATS, uint32_t ENT_WORDS = 12, uint32_t PAIRS = 0> struct MxLdsT {
    static constexpr bool GEOM_LDS = true; /* exact records and material ids of a resident scene are here as well */
    static constexpr uint32_t N_RECORDS = RECORDS, N_MATS = MATS;
    /* PAIRS > 0: the narrow phase of a resident scene is balanced across the lanes of the wave (narrow_balanced): a ring of
     * PAIRS (ray lane, sphere) pairs per wave, one (t, hit id, sphere) result slot per lane, the hit ids of the sorted slots */
    static constexpr uint32_t PAIR_CAP = PAIRS;
    static_assert(PAIRS == 0 || (PAIRS >= 128 && (PAIRS & (PAIRS - 1)) == 0 && RECORDS <= 1024), "pair ring: a power of two >= 128; 10-bit sphere index");
    uint4 SM[RECORDS * 4 + 128]; /* 2 x 64 x 16 B per tile of 32 spheres, + the top tile (the balls around the tiles) behind the last one */
    float4 A[RECORDS];
    float4 B[RECORDS];
    unsigned long long slot[PAIRS ? MIRT_F32_WAVES : 1][PAIRS ? MIRT_WAVE : 1];
    uint16_t pairs[PAIRS ? MIRT_F32_WAVES : 1][PAIRS ? PAIRS : 1];
    uint16_t oid[PAIRS ? RECORDS : 1];
    uint16_t mat[RECORDS];
    static constexpr uint32_t LDS_ENT_CAP = ENT_WORDS; /* in 4-byte words: trace_mx stores ENT_WORDS / 2 tile entries of 8 bytes */
    /* 16-byte alignment spelled out: trace_mx writes 8-byte entries and stage_params 16-byte words; with the lists at a 4-byte
     * offset the compiler splits every entry access in two and the bench frame takes 13.5 ms instead of 12.5 */
    alignas(16) uint32_t ent[MIRT_F32_WAVES][LDS_ENT_CAP][MIRT_WAVE];
    alignas(16) DevMaterial<float> mats[MATS]; /* the scene's materials when they fit: the shading of a segment starts with this record */
    alignas(16) uint32_t params[120]; /* the block's copy of camera, partition and launch parameters (render_kernel_f32_mx) */
    /* all threads of the block; true when the scene is resident (larger scenes read their operands through L1/L2) */
    __device__ inline bool fill(const DevScene<float>& sc) {
        const bool res = sc.n <= RECORDS && sc.n_materials <= 0x10000u;
        if (sc.n_materials <= MATS) {
            const uint32_t nd = sc.n_materials * (uint32_t)(sizeof(DevMaterial<float>) / 4);
            const uint32_t* src = (const uint32_t*)sc.mats;
            uint32_t* dst = (uint32_t*)mats;
            for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
        }
        if (res) {
            const uint32_t ntiles = (sc.n + 31) / 32;
            for (uint32_t i = threadIdx.x; i < (ntiles + 1u) * 128; i += blockDim.x) SM[i] = sc.SM[i]; /* + the top tile */
            for (uint32_t i = threadIdx.x; i < sc.n; i += blockDim.x) {
                A[i] = sc.A[i];
                B[i] = sc.B[i];
                mat[i] = (uint16_t)sc.material[i];
                if (PAIRS) oid[i] = (uint16_t)sc.orig_id[i]; /* < RECORDS real spheres: the ids fit (a dummy's 0xFFFF never wins) */
            }
        } else if (top_tiles_resident(sc)) {
            /* a scene beyond the LDS whose waves stream their own tiles (mx_cull 3): the operand area is free, and the TOP tiles -
             * one per 32 tiles, all of them tested by every wave in every segment - fit it */
            const uint32_t ntiles = (sc.n + 31) / 32, ntop = (ntiles + 31u) / 32u;
            for (uint32_t i = threadIdx.x; i < ntop * 128u; i += blockDim.x) SM[i] = sc.SM[ntiles * 128u + i];
        }
        __syncthreads();
        return res;
    }
    /* wave-uniform: the top tiles of this (streamed) scene sit in SM[0 .. ntop * 128) */
    __device__ static inline bool top_tiles_resident(const DevScene<float>& sc) {
        const uint32_t ntiles = (sc.n + 31) / 32, ntop = (ntiles + 31u) / 32u;
        return sc.n > RECORDS && sc.mx_cull == 3u && ntop * 128u <= RECORDS * 4u + 128u;
    }
};
#define MIRT_MX_SMALL_RECORDS 512u
#ifndef MIRT_MX_SMALL_ENT
#define MIRT_MX_SMALL_ENT 14
#endif
#ifndef MIRT_MX_PAIRS
#define MIRT_MX_PAIRS 128 /* 0: every lane walks its own candidates (the narrow phase of rounds 1 and 2) */
#endif
typedef MxLdsT<MIRT_MX_SMALL_RECORDS, 512u, MIRT_MX_SMALL_ENT, MIRT_MX_PAIRS> MxLdsSmall; /* 32 + 8 + 8 + (8 + 4 + 1) + 1 + 56 + 36 KB; 7 entries per lane: 13.58 -> 13.53 ms against 6 */
typedef MxLdsT<MIRT_LDS_RECORDS, 155u> MxLds;            /* 66 + 16 + 16 + 2 + 48 + 10.9 KB */
static_assert(sizeof(MxLds) <= 160u * 1024u && sizeof(MxLdsSmall) <= 160u * 1024u, "MxLds layouts must fit the CU's LDS");
/* f64 kernels (256-thread blocks): only the operands and the entry lists; the exact f64 records come through L2 */
#define MIRT_MX64_RECORDS 512u /* three blocks per CU: 32 KB of operands + 12 KB of entry lists each */
struct MxLds64 {
    static constexpr bool GEOM_LDS = false;
    static constexpr uint32_t LDS_ENT_CAP = MIRT_ENT_CAP;
    uint4 SM[MIRT_MX64_RECORDS * 4 + 128];
    uint32_t ent[MIRT_WAVES_PER_BLOCK][MIRT_ENT_CAP][MIRT_WAVE];
    __device__ inline bool fill(const DevScene<double>& sc) {
        const bool res = sc.n <= MIRT_MX64_RECORDS;
        if (res) {
            const uint32_t ntiles = (sc.n + 31) / 32;
            for (uint32_t i = threadIdx.x; i < (ntiles + 1u) * 128; i += blockDim.x) SM[i] = sc.SM[i]; /* + the top tile */
        }
        __syncthreads();
        return res;
    }
};
/* The empty asm statements pin ONE conversion per value: without them instruction selection converts the same f32 twice
 * for the two users of hi - v_fma_mixlo_f16 (the exact product d * inv rounded once) for the residual and
 * v_cvt_pk_f16_f32 (the f32 product rounded again) for the operand - and the two disagree by one f16 ulp whenever the
 * f32 value sits on a rounding tie: hi + lo is then off by 2^-11 |x| (found as 2 wrong pixels in 90 000). */
__device__ inline void mx_split(float x, _Float16& hi, _Float16& lo) {
    asm volatile("" : "+v"(x));
    hi = (_Float16)x;
    asm volatile("" : "+v"(hi));
    /* lo = f16(x - hi): the difference is exact in binary32, so there is one rounding, to f16, however it is computed - and
     * v_fma_mixlo_f16 (f32 x * 1.0 - f16 hi, result rounded to f16) is ONE instruction where cvt, sub, cvt are three */
    asm volatile("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x), "v"(hi)); /* s_nop: the wait state the compiler puts between v_cvt_f16_f32 and a reader of its result (it does not look inside asm) */
}
/* The ray side of the contraction: the ten coefficients of q, split, packed and exchanged between the half-waves (see the
 * K-slot table above).  bx0 / bx1: B operands of the two chained MFMAs for rays 0-31, by0 / by1 for rays 32-63. */
template <class R>
__device__ inline void mx_ray_operands(const DevScene<R>& sc, const RayT<R>& ray, RayA<R> a_true, bool active, mirt_h8& bx0, mirt_h8& bx1, mirt_h8& by0,
                                       mirt_h8& by1) {
        const R inv = filter_rsq(a_true.a);
        const V3<float> Dn = {(float)(ray.d.x * inv), (float)(ray.d.y * inv), (float)(ray.d.z * inv)};
        /* the drift: every sphere is tested where it is at the ray's time MINUS tm * mx_drift, so the origin moves by the same
         * amount (one rounding per component, <= eps |O|: a line displaced by that much costs 2 eps (|M| + |O|)^2 of the spare
         * margin, see above); the balls on the host are built around the spheres' motion relative to the drift */
        const R tmn = -ray.tm;
        const V3<float> O = {(float)((ray.o.x + tmn * (R)sc.mx_drift[0]) * (R)sc.mx_scale), (float)((ray.o.y + tmn * (R)sc.mx_drift[1]) * (R)sc.mx_scale),
                             (float)((ray.o.z + tmn * (R)sc.mx_drift[2]) * (R)sc.mx_scale)};
        const float k1 = __builtin_fmaf(Dn.z, O.z, __builtin_fmaf(Dn.y, O.y, Dn.x * O.x));
        const float o2 = __builtin_fmaf(O.z, O.z, __builtin_fmaf(O.y, O.y, O.x * O.x));
        const float k2 = o2 * MIRT_MX_SHRINK - __builtin_fmaf(MIRT_MX_FLOOR_LIN, filter_sqrt(o2), MIRT_MX_FLOOR_ABS);
        const bool ok = o2 <= MIRT_MX_K2MAX; /* false for NaN */
        /* the nine coefficients and the constant of q as a polynomial in the sphere's M */
        const float dx2 = Dn.x + Dn.x, dy2 = Dn.y + Dn.y, m2k1 = -(k1 + k1);
        float cf[10] = {Dn.x * Dn.x, Dn.y * Dn.y, Dn.z * Dn.z, dx2 * Dn.y, dx2 * Dn.z, dy2 * Dn.z,
                        __builtin_fmaf(m2k1, Dn.x, O.x + O.x), __builtin_fmaf(m2k1, Dn.y, O.y + O.y), __builtin_fmaf(m2k1, Dn.z, O.z + O.z),
                        __builtin_fmaf(k1, k1, -k2)};
        if (!active || !ok) {
            /* the two special lanes are decided by their operands, not by masks in the tile loop: with every coefficient 0 the
             * test reads q = c0 - w, and |w| <= 64^2 for a real record (60000 for a dummy).  No ray: c0 = -60000, q < 0 for
             * every record, never a candidate.  Origin out of range (or NaN): c0 = +60000, q >= 0 for every record, all
             * of them candidates (a dummy too: the exact test rejects it). */
#pragma unroll
            for (int k = 0; k < 9; ++k) cf[k] = 0.0f;
            cf[9] = active ? 60000.0f : -60000.0f;
        }
        _Float16 rv[32];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            _Float16 h, l;
            mx_split(cf[k], h, l);
            rv[3 * k] = h; rv[3 * k + 1] = h; rv[3 * k + 2] = l; /* against the sphere's (hi, lo, hi) */
        }
        {
            _Float16 h, m, l;
            mx_split(cf[9], h, m);
            asm volatile("s_nop 0\n\tv_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(cf[9] - (float)h), "v"(m)); /* both differences are exact in f32 */
            rv[27] = h; rv[28] = m; rv[29] = l;
            rv[30] = rv[31] = (_Float16)-1.0f;
        }
        /* K halves of this lane's own ray for the two chained MFMAs.
         * X operand: lanes 0-31 their own low half, lanes 32-63 the high half of ray (lane - 32);
         * Y operand: lanes 0-31 the low half of ray (lane + 32), lanes 32-63 their own high half */
        const mirt_h8 lo0 = {rv[0], rv[1], rv[2], rv[3], rv[4], rv[5], rv[6], rv[7]}, hi0 = {rv[8], rv[9], rv[10], rv[11], rv[12], rv[13], rv[14], rv[15]};
        const mirt_h8 lo1 = {rv[16], rv[17], rv[18], rv[19], rv[20], rv[21], rv[22], rv[23]}, hi1 = {rv[24], rv[25], rv[26], rv[27], rv[28], rv[29], rv[30], rv[31]};
        mirt_u4 x0 = __builtin_bit_cast(mirt_u4, lo0), y0 = __builtin_bit_cast(mirt_u4, hi0);
        mirt_u4 x1 = __builtin_bit_cast(mirt_u4, lo1), y1 = __builtin_bit_cast(mirt_u4, hi1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            auto s0 = __builtin_amdgcn_permlane32_swap(x0[k], y0[k], false, false);
            x0[k] = s0[0]; y0[k] = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(x1[k], y1[k], false, false);
            x1[k] = s1[0]; y1[k] = s1[1];
        }
        bx0 = __builtin_bit_cast(mirt_h8, x0); by0 = __builtin_bit_cast(mirt_h8, y0);
        bx1 = __builtin_bit_cast(mirt_h8, x1); by1 = __builtin_bit_cast(mirt_h8, y1);
}

/* ---- the narrow phase balanced across the lanes of a wave (f32, resident scenes, LDS layouts with a pair ring) --------
 * After a broad phase every lane holds its own short list of candidates - 2.4 per lane and segment on the bench frame, but
 * unevenly: a wave ran 7.1 rounds of the exact test with a third of its lanes busy.  Here the candidates of all 64 rays go
 * through ONE ring of (ray lane, sphere) pairs per wave in LDS, every lane takes a pair - whoever's ray it belongs to,
 * fetching that ray from its owner's registers with ds_bpermute - and a round of exact tests is full as long as 64 pairs
 * are waiting.  A pair's result goes back to the ray's owner through a 64-bit ds_min on the owner's slot.
 *
 * Why the result is the same bit for bit.  narrow_test walks a ray's candidates in list order against a shrinking
 * interval: sphere i contributes t_i = its near root if tmin < near root, else its far root if tmin < far root (a root
 * beyond the best so far is rejected either way: far >= near), and the best is the minimum of (t_i, hit id) in
 * lexicographic order (strict <, exact ties to the lower hit id).  That minimum does not depend on the order of the
 * candidates, and (t bits, hit id, sorted index) packed into 64 bits orders the same way (t > tmin > 0: positive floats
 * order like their bit patterns), so an unordered ds_min_u64 per pair computes it.  Every t_i is evaluated by the same
 * instructions on the same inputs as before (the fetched ray is the owner's, bit for bit). */
template <class LDS, class = void> struct mx_pair_cap { static constexpr uint32_t value = 0; };
template <class LDS> struct mx_pair_cap<LDS, decltype((void)LDS::PAIR_CAP)> { static constexpr uint32_t value = LDS::PAIR_CAP; };
#define MIRT_KEY_NONE 0x7F800000FFFFFFFFull /* t = +inf, no sphere */
__device__ inline uint32_t wave_scan_incl(uint32_t x) { /* inclusive prefix sum over the 64 lanes (checked on the hardware by bench/imul_rate.hip) */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false); /* row_bcast:15 -> rows 1, 3 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false); /* row_bcast:31 -> rows 2, 3 */
    return x;
}
/* one (ray, sphere) pair: sphere.rs:81-101 against (tmin, +inf); MIRT_KEY_NONE when the sphere contributes no root */
template <class G>
__device__ inline unsigned long long pair_key(const DevScene<float>& sc, const G& geo, uint32_t i, uint32_t oid, const RayT<float>& ray, RayA<float> ra, float tmin) {
    const float4 sa = geo.a(i);
    V3<float> C = {sa.x, sa.y, sa.z};
    if (i >= sc.n_static) {
        const float4 sb = geo.b(i);
        C = center_at(C, V3<float>{sb.x, sb.y, sb.z}, ray.tm);
    }
    const V3<float> oc = C - ray.o;
    float h, d;
    disc(oc, ray.d, ra, sa.w, h, d);
    unsigned long long key = MIRT_KEY_NONE;
    /* the same shortcuts as narrow_test: they reject only what the root test would reject */
    if (d >= 0.0f && !(h < -h_normal_min(h) && d <= h * h && tmin >= 0.0f)) {
        const float sq = m_sqrt(d);
        float t = root_over_a(h - sq, ra);
        bool ok = tmin < t && t < INFINITY;
        if (!ok) {
            t = root_over_a(h + sq, ra);
            ok = tmin < t && t < INFINITY;
        }
        if (ok) key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((oid << 16) | i);
    }
    return key;
}
/* one round: lanes [0, count) take the pairs at ring positions head, head + 1, ... */
template <class LDS>
__device__ inline void narrow_round(const DevScene<float>& sc, LDS& L, uint32_t w, uint32_t lane, uint32_t head, uint32_t count, const RayT<float>& ray,
                                    RayA<float> ra, float tmin) {
    const bool valid = lane < count;
    uint32_t pr = lane << 10;
    if (valid) pr = L.pairs[w][(head + lane) & (LDS::PAIR_CAP - 1u)];
    const uint32_t r = pr >> 10, i = pr & 1023u;
    /* the ray of lane r, from its owner's registers (all lanes take part: a lane outside exec would hand out zeros) */
    const int ba = (int)(r << 2);
    auto get = [&](float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(ba, __float_as_int(v))); };
    const RayT<float> q = {{get(ray.o.x), get(ray.o.y), get(ray.o.z)}, {get(ray.d.x), get(ray.d.y), get(ray.d.z)}, get(ray.tm)};
    const RayA<float> qa = {get(ra.a), get(ra.inv)};
    MIRT_DIAG(sc, DG_NARROW_ROUND, valid);
    if (valid) {
        const unsigned long long key = pair_key(sc, GeomLdsT<LDS>{L}, i, (uint32_t)L.oid[i], q, qa, tmin);
        if (key != MIRT_KEY_NONE) (void)__hip_atomic_fetch_min(&L.slot[w][r], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}
/* the candidates of one pass of the broad phase (entry lists ent2[0 .. cnt) of every lane) -> pairs -> rounds.  head / tail
 * are wave-uniform ring counters; whatever is left in the ring when the last pass is through is run by narrow_drain. */
template <class LDS>
__device__ inline void narrow_balanced(const DevScene<float>& sc, LDS& L, uint32_t w, uint32_t lane, uint2 (*ent2)[MIRT_WAVE], uint32_t gbase, uint32_t cnt,
                                       const RayT<float>& ray, RayA<float> ra, float tmin, uint32_t& head, uint32_t& tail, uint32_t& narrow_count) {
    uint32_t k = 0, m = 0, ga = 0, gb = 0;
    for (;;) {
        if (m == 0 && k < cnt) {
            const uint2 e = ent2[k][lane];
            ++k;
            m = (~e.x & 0xFFFFu) | (~e.y << 16);
            ga = (gbase + (e.x >> 16)) * MIRT_CGROUP + (MIRT_CGROUP - 1);
            gb = (gbase + (e.y >> 16)) * MIRT_CGROUP + (2 * MIRT_CGROUP - 1);
        }
        const unsigned long long have = __ballot(m != 0);
        if (have == 0ull) break;
        const uint32_t n = (uint32_t)__popcll(have);
        if (tail - head + n > LDS::PAIR_CAP) { /* no room for this batch: a full round first (>= 64 are waiting: PAIR_CAP >= 128) */
            narrow_round(sc, L, w, lane, head, 64u, ray, ra, tmin);
            head += 64u;
        }
        if (m != 0) {
            const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            const uint32_t idx = (bit >= 16u ? gb : ga) - bit;
            const uint32_t pos = tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0u));
            L.pairs[w][pos & (LDS::PAIR_CAP - 1u)] = (uint16_t)((lane << 10) | idx);
            ++narrow_count;
        }
        tail += n;
    }
}
template <class LDS>
__device__ inline void narrow_drain(const DevScene<float>& sc, LDS& L, uint32_t w, uint32_t lane, const RayT<float>& ray, RayA<float> ra, float tmin,
                                    uint32_t& head, uint32_t tail) {
    while (head != tail) {
        const uint32_t c = tail - head < 64u ? tail - head : 64u;
        narrow_round(sc, L, w, lane, head, c, ray, ra, tmin);
        head += c;
    }
}

/* one tile: q for 32 rays x 32 spheres per chain of two MFMAs - all that is left to the vector ALU is collecting the signs.
 * mx / my come in holding the 16-bit group numbers and leave as (group << 16) | signs: after the swap mx belongs to this lane's
 * ray and the 16 spheres of the rows of sub 0, my to the rows of sub 1 (bit 15 - k set = sphere k of the group ruled out). */
__device__ inline void mx_tile(const mirt_h8& a0, const mirt_h8& a1, const mirt_h8& bx0, const mirt_h8& bx1, const mirt_h8& by0, const mirt_h8& by1,
                               uint32_t& mx, uint32_t& my) {
    const mirt_f16v zero = {};
    mirt_f16v qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bx0, zero, 0, 0, 0);
    mirt_f16v qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, by0, zero, 0, 0, 0);
    qx = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bx1, qx, 0, 0, 0);
    qy = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, by1, qy, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = __builtin_amdgcn_alignbit(mx, __float_as_uint(qx[r]), 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) my = __builtin_amdgcn_alignbit(my, __float_as_uint(qy[r]), 31);
    /* lanes 0-31 hold (ray l, rows of sub 0) and (ray l + 32, sub 0); lanes 32-63 (ray l - 32, sub 1) and (ray l, sub 1) */
    auto sw = __builtin_amdgcn_permlane32_swap(mx, my, false, false);
    mx = sw[0];
    my = sw[1];
}
/* one top tile -> the wave's need word: bit k set = some ray of the wave may touch the ball of row k (wave-uniform) */
__device__ inline uint32_t mx_need_word(const mirt_h8& a0, const mirt_h8& a1, const mirt_h8& bx0, const mirt_h8& bx1, const mirt_h8& by0, const mirt_h8& by1) {
    uint32_t mx = 0, my = 0;
    mx_tile(a0, a1, bx0, bx1, by0, by1, mx, my);
    /* mx: this lane's ray against rows 0..15 of the top tile (bit 15 - k set = row k ruled out), my: rows 16..31;
     * reversed and complemented: bit k set = the ray may touch the ball of row k */
    uint32_t mine = __builtin_bitreverse32((~mx << 16) | (~my & 0xFFFFu));
    mine |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x111, 0xF, 0xF, false); /* OR over the wave, as wave_scan_incl adds */
    mine |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x112, 0xF, 0xF, false);
    mine |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x114, 0xF, 0xF, false);
    mine |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x118, 0xF, 0xF, false);
    mine |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x142, 0xA, 0xF, false);
    mine |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x143, 0xC, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)mine, 63);
}
#ifdef MIRT_VERIFY_BROAD
/* diagnostic build: the 16 spheres of group g against this lane's ray, exactly - a sphere whose discriminant is >= 0 must have
 * its bit clear in `word` (word = all ones for a group the cull skipped) */
template <class R>
__device__ inline void mx_verify_group(const DevScene<R>& sc, const RayT<R>& ray, RayA<R> a_true, bool active, uint32_t g, uint32_t word) {
    for (uint32_t k_ = 0; k_ < 16; ++k_) {
        const uint32_t i_ = g * 16 + k_;
        if (i_ >= sc.n) break;
        bool big_ = false; /* left out of the filter on purpose: tested directly by every ray */
        for (uint32_t b_ = 0; b_ < sc.n_big; ++b_) big_ = big_ || sc.big_idx[b_] == i_;
        if (big_) continue;
        typename Vec4<R>::type sa_ = sc.A[i_];
        V3<R> C_ = {sa_.x, sa_.y, sa_.z};
        if (i_ >= sc.n_static) {
            typename Vec4<R>::type sb_ = sc.B[i_];
            C_ = center_at(C_, V3<R>{sb_.x, sb_.y, sb_.z}, ray.tm);
        }
        R hh_, dd_;
        disc(C_ - ray.o, ray.d, a_true, sa_.w, hh_, dd_);
        if (active && dd_ >= (R)0.0 && ((word >> (15 - k_)) & 1u)) atomicAdd(sc.dbg, 1ull);
    }
}
#endif

/* R = double (the parity-mode kernels): the same filter on the ray rounded to f32 - Dn is normalised in double first, so
 * direction and origin are each off by <= 2^-24 relatively, the f32 line passes within 5 eps (|M| + |O|) of the f64 one,
 * and 6 eps ((|M|+|O|)^2 + rho^2) replaces the 32 eps of the f32 derivation (an f64 discriminant needs ~1e-15) inside the
 * same 128 eps budget; the balls behind the operands are built around the f64 spheres (mirt_scene_create).  The narrow
 * phase is the exact f64 test. */
template <class R, class LDS, bool RES>
__device__ inline void trace_mx(const DevScene<R>& sc, LDS& L, uint32_t w, uint32_t lane, bool active, const RayT<R>& ray, R tmin,
                                Closest<R>& best, uint32_t& narrow_count, PhaseTimer& pt, uint32_t& tiles_done) {
    best.t = (R)INFINITY;
    best.idx = 0xFFFFFFFFu;
    const RayA<R> a_true = ray_a(ray.d);
    const uint32_t ntiles = (sc.n + 31) / 32;
    uint2(*ent2)[MIRT_WAVE] = reinterpret_cast<uint2(*)[MIRT_WAVE]>(L.ent[w]); /* one entry per tile: both 16-sphere halves */
    mirt_h8 bx0, bx1, by0, by1;
    mx_ray_operands(sc, ray, a_true, active, bx0, bx1, by0, by1);
    /* sphere operands: the block-resident LDS copy, or (big scenes) straight from L1/L2 - every wave of the chip reads the
     * same 1 KB per tile, one coalesced 16-byte load per lane, the next tile's already on its way while this one is tested
     * (the array is padded by one tile) */
    const mirt_h8* SM = RES ? (const mirt_h8*)L.SM : (const mirt_h8*)sc.SM;
    const uint32_t sub = lane >> 5;
    uint32_t t = 0;
    mirt_h8 pre0 = {}, pre1 = {};
    if (!RES && sc.mx_cull != 1u && sc.mx_cull != 3u) { pre0 = SM[lane]; pre1 = SM[64 + lane]; } /* no cull: tile 0 first (with the cull: the first needed tile, below) */
    constexpr bool BALANCED = RES && sizeof(R) == 4 && mx_pair_cap<LDS>::value > 0;
    uint32_t ring_head = 0, ring_tail = 0;
    /* the spheres the filter leaves out (far larger than the rest of the scene - the ground of the reference's scenes, which
     * every line meets): every ray tests them exactly, here, with no candidate list in between; the closest hit is the minimum
     * over all exact tests in any order (narrow_balanced) */
    if constexpr (BALANCED) {
        unsigned long long k0 = MIRT_KEY_NONE;
        if (active)
            for (uint32_t b = 0; b < sc.n_big; ++b) {
                const uint32_t i = sc.big_idx[b];
                const unsigned long long k = pair_key(sc, GeomLdsT<LDS>{L}, i, (uint32_t)L.oid[i], ray, a_true, tmin);
                k0 = k < k0 ? k : k0;
                ++narrow_count;
            }
        L.slot[w][lane] = k0;
    } else if (active) {
        for (uint32_t b = 0; b < sc.n_big; ++b) {
            if constexpr (LDS::GEOM_LDS) {
                if (RES) narrow_test(sc, GeomLdsT<LDS>{L}, sc.big_idx[b], ray, a_true, tmin, best);
                else narrow_test(sc, GeomGlobal<R>{sc}, sc.big_idx[b], ray, a_true, tmin, best);
            } else {
                narrow_test(sc, GeomGlobal<R>{sc}, sc.big_idx[b], ray, a_true, tmin, best);
            }
            ++narrow_count;
        }
    }
    /* ---- tile cull (resident scenes): which tiles can ANY ray of this wave touch?  The top tile behind the last one holds a
     * ball around each tile's balls; one pass of the same contraction over it gives every lane the tiles its ray may meet, the
     * wave ORs them, and the loop below skips the rest.  A skipped tile has no candidate for any lane (its ball test is as
     * conservative as the sphere tests, mx_ensure), so the entry lists - and every (id, t) - are what the full loop gives.
     * The lanes of a wave hold chunks of neighbouring pixels and their scattered rays start next to each other: with the
     * spheres in a spatial order most tiles drop out. */
    uint32_t need = 0xFFFFFFFFu;
    if (RES && sc.mx_cull) {
        need = mx_need_word(SM[ntiles * 128 + lane], SM[ntiles * 128 + 64 + lane], bx0, bx1, by0, by1);
        ++tiles_done; /* the top tile is a tile's worth of work too */
    }
    pt.stamp(PH_OTHER); /* phase-timing build: "other" = the ray operands and the top tile */
#ifdef MIRT_VERIFY_BROAD
    auto verify_group = [&](uint32_t g, uint32_t word) { mx_verify_group(sc, ray, a_true, active, g, word); };
#endif
    /* scenes that are not resident (every wave streams its tiles from L1/L2: f32 beyond 1024 spheres, f64 beyond 512, K1): one top
     * tile per 32 tiles behind the last tile (mx_cull 1: one, 3: several), one need word per top tile in the lanes of `needv0/1`,
     * and the loop below walks - and prefetches - only the tiles of this wave */
    const bool streamed = !RES && (sc.mx_cull == 1u || sc.mx_cull == 3u);
    uint32_t needv0 = 0xFFFFFFFFu, needv1 = 0xFFFFFFFFu; /* lane j of needv0: tiles 32 j .. 32 j + 31; of needv1: the same + 2048 */
    auto need_word = [&](uint32_t j) -> uint32_t { /* wave-uniform j: a scalar lane index */
        return (uint32_t)(j < 64u ? __builtin_amdgcn_readlane((int)needv0, (int)j) : __builtin_amdgcn_readlane((int)needv1, (int)(j - 64u)));
    };
    auto next_needed = [&](uint32_t from) -> uint32_t { /* the first tile >= from of this wave's, ntiles if there is none (wave-uniform) */
        uint32_t u = from;
        while (u < ntiles) {
            const uint32_t word = need_word(u >> 5) >> (u & 31u);
            if (word) { u += (uint32_t)__builtin_ctz(word); break; }
            u = (u | 31u) + 1u;
        }
        return u < ntiles ? u : ntiles;
    };
    if (streamed) {
        const uint32_t ntop = (ntiles + 31u) / 32u; /* <= 128 (mx_cull_mode: MIRT_MX_CULL_TILES) */
        bool top_in_lds = false;
        if constexpr (sizeof(R) == 4 && !RES) top_in_lds = LDS::top_tiles_resident(sc); /* copied by fill(): the f32 render kernel's operand area */
        if (top_in_lds) {
            const mirt_h8* TSM = (const mirt_h8*)L.SM;
            for (uint32_t j = 0; j < ntop; ++j) {
                const uint32_t wj = mx_need_word(TSM[j * 128 + lane], TSM[j * 128 + 64 + lane], bx0, bx1, by0, by1);
                needv0 = lane == j ? wj : needv0;
                needv1 = lane + 64u == j ? wj : needv1;
                ++tiles_done;
            }
        } else {
        mirt_h8 p0 = SM[ntiles * 128 + lane], p1 = SM[ntiles * 128 + 64 + lane];
        for (uint32_t j = 0; j < ntop; ++j) {
            const mirt_h8 a0 = p0, a1 = p1;
            if (j + 1u < ntop) { p0 = SM[(ntiles + j + 1u) * 128 + lane]; p1 = SM[(ntiles + j + 1u) * 128 + 64 + lane]; }
            const uint32_t wj = mx_need_word(a0, a1, bx0, bx1, by0, by1);
            needv0 = lane == j ? wj : needv0;
            needv1 = lane + 64u == j ? wj : needv1;
            ++tiles_done;
        }
        }
#ifdef MIRT_VERIFY_BROAD
        for (uint32_t s_ = 0; s_ < ntiles; ++s_)
            if (!((need_word(s_ >> 5) >> (s_ & 31u)) & 1u)) {
                verify_group(2 * s_, 0xFFFFu);
                verify_group(2 * s_ + 1, 0xFFFFu);
            }
#endif
        t = next_needed(0u);
        if (t < ntiles) { pre0 = SM[t * 128 + lane]; pre1 = SM[t * 128 + 64 + lane]; }
        pt.stamp(PH_OTHER);
    }
    /* mx_cull 2 (scenes of <= 16 tiles): row h of the top tile is the ball of HALF tile h (16 spheres), and the loop runs over
     * the half tiles the wave needs, two at a time: the rows of sub 0 of the A operand are fetched from the first, those of sub 1
     * from the second - a per-lane LDS address, nothing else - so a tile of which only one half is needed costs half a tile. */
    const bool packed = RES && sc.mx_cull == 2u;
    /* only rows that stand for a half tile: a lane marked "every record is a candidate" (origin out of range, NaN: c0 = +60 000)
     * also passes the "never" rows behind them (q = +0), and those have no spheres, not even dummy records, to meet */
    uint32_t rem = packed ? need & (0xFFFFFFFFu >> (32u - 2u * ntiles)) : 0u;
    while (packed ? rem != 0u : t < ntiles) { /* one pass unless some lane's entry list fills up */
        prio_broad();
        uint32_t cnt = 0;
        const uint32_t gbase = packed ? 0u : 2 * t;
        bool flush = false;
        if (packed) {
            /* The vector ALU's issue slots are what the kernel runs out of, so everything in this loop that is the same for all
             * lanes is kept on the scalar unit: the rows of MFMA sub-block 0 come from half tile ha, those of sub-block 1 from hb -
             * both wave-uniform - at (h >> 1) * 128 + (h & 1) * 4 records behind the lane's own constant place in a tile, which is
             * place(ha) + (this lane reads hb ? place(hb) - place(ha) : 0): one AND and one three-operand add per lane. */
            const uint32_t row = lane & 31u;
            const uint32_t lane_at = (lane & 32u) + (row & ~4u);
            uint32_t from_b = (row & 4u) ? 0xFFFFFFFFu : 0u, in_sub1 = sub ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(from_b), "+v"(in_sub1)); /* masks, not conditions: the compiler turns `mask & x` back into a move and a select otherwise */
            uint32_t in_pass = 0; /* tiles of this pass so far (wave-uniform): cnt <= in_pass, so the list cannot be full before it is */
            /* the loop below starts on a 64-byte boundary: without it the bench frame's K2 moved by 0.5 % (9.33 <-> 9.38 ms) with whatever
             * shifted the code in front of it by a few bytes - one more store in the kernel's prologue did */
            asm volatile(".p2align 6");
            while (rem != 0u && !flush) {
                const uint32_t ha = (uint32_t)__builtin_ctz(rem);
                rem ^= 1u << ha;
                const bool two = rem != 0u;
                const uint32_t hb = two ? (uint32_t)__builtin_ctz(rem) : ha;
                rem &= ~(two ? 1u << hb : 0u); /* (rem & (rem - 1) twice made the compiler take the borrow of rem - 1 through the vector ALU) */
                const uint32_t at_a = (ha >> 1) * 128u + ((ha & 1u) << 2), at_b = (hb >> 1) * 128u + ((hb & 1u) << 2);
                const uint32_t at = lane_at + at_a + (from_b & (at_b - at_a));
                uint32_t mx = ha + (in_sub1 & (hb - ha)), my = mx;
                mx_tile(SM[at], SM[at + 64], bx0, bx1, by0, by1, mx, my);
                my |= two ? 0u : 0xFFFFu; /* the odd one out: the rows of sub 1 were a second copy of ha */
#ifdef MIRT_VERIFY_BROAD
                verify_group(ha, mx);
                if (two) verify_group(hb, my);
#endif
                ent2[cnt][lane] = make_uint2(mx, my);
                cnt += ((mx & my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
                if (++in_pass > LDS::LDS_ENT_CAP / 2 - 1) flush = __ballot(cnt > LDS::LDS_ENT_CAP / 2 - 1) != 0ull;
                ++tiles_done;
            }
#ifdef MIRT_VERIFY_BROAD
            if (rem == 0u)
                for (uint32_t g_ = 0; g_ < 2 * ntiles; ++g_)
                    if (!((need >> g_) & 1u)) verify_group(g_, 0xFFFFu);
#endif
        } else if (streamed) { /* <= 4096 tiles: the 16-bit group numbers cannot run out within a pass */
            while (t < ntiles && !flush) {
                const mirt_h8 a0 = pre0, a1 = pre1;
                const uint32_t tn = next_needed(t + 1u);
                if (tn < ntiles) { pre0 = SM[tn * 128 + lane]; pre1 = SM[tn * 128 + 64 + lane]; }
                uint32_t mx = 2 * t - gbase + sub, my = mx;
                mx_tile(a0, a1, bx0, bx1, by0, by1, mx, my);
#ifdef MIRT_VERIFY_BROAD
                verify_group(2 * t, mx);
                verify_group(2 * t + 1, my);
#endif
                ent2[cnt][lane] = make_uint2(mx, my);
                cnt += ((mx & my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
                flush = __ballot(cnt > LDS::LDS_ENT_CAP / 2 - 1) != 0ull;
                ++tiles_done;
                t = tn;
            }
        } else {
        uint32_t tlimit = t + 0x8000u; /* group numbers in the entries are 16 bit, relative to gbase */
        tlimit = tlimit < ntiles ? tlimit : ntiles;
        for (; t < tlimit && !flush; ++t) {
            mirt_h8 a0, a1;
            if (RES && !((need >> t) & 1u)) { /* mx_cull 1: row t is the ball of tile t; wave-uniform: no ray of the wave can touch this tile */
#ifdef MIRT_VERIFY_BROAD
                verify_group(2 * t, 0xFFFFu); /* ... then no sphere of it may have an exact discriminant >= 0 */
                verify_group(2 * t + 1, 0xFFFFu);
#endif
                continue;
            }
            ++tiles_done;
            if (RES) {
                a0 = SM[t * 128 + lane];
                a1 = SM[t * 128 + 64 + lane];
            } else {
                a0 = pre0; a1 = pre1;
                pre0 = SM[(t + 1) * 128 + lane];
                pre1 = SM[(t + 1) * 128 + 64 + lane];
            }
            uint32_t mx = 2 * t - gbase + sub, my = mx;
            mx_tile(a0, a1, bx0, bx1, by0, by1, mx, my); /* mx: this lane's ray, spheres 32 t .. 32 t + 15; my: 32 t + 16 .. 32 t + 31 */
#ifdef MIRT_VERIFY_BROAD
            verify_group(2 * t, mx);
            verify_group(2 * t + 1, my);
#endif
            /* one 8-byte entry per tile (the list holds LDS_ENT_CAP / 2 of them): unconditional write, conditional advance
             * (cnt < LDS_ENT_CAP / 2 here: a full list flushes below) */
            ent2[cnt][lane] = make_uint2(mx, my);
            cnt += ((mx & my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
            flush = __ballot(cnt > LDS::LDS_ENT_CAP / 2 - 1) != 0ull; /* no room for another tile */
        }
        }
        prio_other();
        pt.stamp(PH_BROAD);
        MIRT_DIAG(sc, DG_PASS, active);
        if constexpr (BALANCED) {
            narrow_balanced(sc, L, w, lane, ent2, gbase, cnt, ray, a_true, tmin, ring_head, ring_tail, narrow_count);
            pt.stamp(PH_NARROW);
        } else
        /* ---- narrow phase: as in trace_cons ---- */
        {
            uint32_t k = 0, m = 0, ga = 0, gb = 0;
            for (;;) {
                if (m == 0 && k < cnt) {
                    const uint2 e = ent2[k][lane];
                    ++k;
                    m = (~e.x & 0xFFFFu) | (~e.y << 16); /* the two groups of the entry */
                    ga = (gbase + (e.x >> 16)) * MIRT_CGROUP + (MIRT_CGROUP - 1);
                    gb = (gbase + (e.y >> 16)) * MIRT_CGROUP + (2 * MIRT_CGROUP - 1);
                }
                if (__ballot(m != 0) == 0ull) break;
                MIRT_DIAG(sc, DG_NARROW_ROUND, m != 0);
                if (m != 0) {
                    const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                    m &= m - 1u;
                    const uint32_t idx = (bit >= 16u ? gb : ga) - bit;
                    if constexpr (LDS::GEOM_LDS) {
                        if (RES) narrow_test(sc, GeomLdsT<LDS>{L}, idx, ray, a_true, tmin, best);
                        else narrow_test(sc, GeomGlobal<R>{sc}, idx, ray, a_true, tmin, best);
                    } else {
                        narrow_test(sc, GeomGlobal<R>{sc}, idx, ray, a_true, tmin, best);
                    }
                    ++narrow_count;
                }
            }
            pt.stamp(PH_NARROW);
        }
    }
    /* after the last pass (or after none: a wave whose rays can touch no tile at all never enters the loop): run what is left in
     * the ring and collect this lane's result - the directly tested spheres' included */
    if constexpr (BALANCED) {
        narrow_drain(sc, L, w, lane, ray, a_true, tmin, ring_head, ring_tail);
        const unsigned long long ks = L.slot[w][lane];
        if (ks != MIRT_KEY_NONE) {
            best.t = __uint_as_float((uint32_t)(ks >> 32));
            best.idx = (uint32_t)ks & 0xFFFFu;
        }
        pt.stamp(PH_NARROW);
    }
}

/* ---- f32, scenes that do not fit the LDS: the sphere operands through a block-wide LDS ring ----------------
 * trace_mx's streaming loop has every wave fetch every tile for itself (2 KB per tile and wave through L1/L2: 239
 * SIMD-cycles per tile against 191 for a resident scene).  Here the 16 waves of the block walk the tiles in step: the
 * operand area of the LDS and the A / B areas behind it (96 KB) are a ring of two chunks of MIRT_RING_TILES tiles, each wave copies its 1 KB slabs of the
 * NEXT chunk with global_load_lds_dwordx4 (no VGPRs, lane-linear image = the operand layout) while all of them test the
 * current one, and one barrier per chunk hands the buffers over - 1/16 of the L1/L2 traffic.
 * The price is that the waves are no longer free-running: every wave of the block must make every call with the same
 * number of barriers (MIRT_RING chunks + 1, a function of the scene alone), so render_body keeps idle waves in the loop
 * until the whole block is out of work (T::LOCKSTEP), and a wave without a ray still stages and waits. */
#define MIRT_MX_CULL_TILES 4096u /* scenes beyond the LDS: the tile cull keeps one need word per 32 tiles in the lanes of two VGPRs (<= 128 top tiles = 131 072 spheres) */
#ifndef MIRT_RING_TILES
#define MIRT_RING_TILES 24u /* two chunks of 24 tiles = 96 KB: the operand area and the (unused, the scene is not resident) A / B areas behind it; 8 / 16 / 24 tiles: 119.5 / 116.0 / 114.0 ms on the 10 000-sphere frame */
#endif
template <class LDS>
__device__ inline void trace_mx_ring(const DevScene<float>& sc, LDS& L, uint32_t w, uint32_t lane, bool active, const RayT<float>& ray, float tmin,
                                     Closest<float>& best, uint32_t& narrow_count, PhaseTimer& pt, uint32_t& tiles_done) {
    static_assert(offsetof(LDS, A) == sizeof(L.SM) && offsetof(LDS, B) == offsetof(LDS, A) + sizeof(L.A), "SM, A, B must be contiguous: the ring runs over all three");
    static_assert((sizeof(L.SM) + sizeof(L.A) + sizeof(L.B)) / 16u >= 2u * MIRT_RING_TILES * 128u, "the ring must hold two chunks");
    static_assert((MIRT_RING_TILES * 2u) % MIRT_F32_WAVES == 0u, "whole slabs per wave");
    constexpr uint32_t SPW = MIRT_RING_TILES * 2u / MIRT_F32_WAVES; /* 1 KB slabs of a chunk per wave */
    best.t = INFINITY;
    best.idx = 0xFFFFFFFFu;
    const RayA<float> a_true = ray_a(ray.d);
    if (active)
        for (uint32_t b = 0; b < sc.n_big; ++b) { /* the spheres the filter leaves out: see trace_mx */
            narrow_test(sc, GeomGlobal<float>{sc}, sc.big_idx[b], ray, a_true, tmin, best);
            ++narrow_count;
        }
    const uint32_t ntiles = (sc.n + 31) / 32, nch = (ntiles + MIRT_RING_TILES - 1u) / MIRT_RING_TILES;
    const bool wave_live = __ballot(active) != 0ull; /* a wave without rays only stages and keeps the barriers */
    uint2(*ent2)[MIRT_WAVE] = reinterpret_cast<uint2(*)[MIRT_WAVE]>(L.ent[w]);
    mirt_h8 bx0 = {}, bx1 = {}, by0 = {}, by1 = {};
    if (wave_live) mx_ray_operands(sc, ray, a_true, active, bx0, bx1, by0, by1);
    /* slabs 2 w and 2 w + 1 of chunk c -> ring buffer c & 1 (the host pads the operand array by a whole chunk of zeros) */
    auto stage = [&](uint32_t c) {
        const uint4* src = sc.SM + ((size_t)c * MIRT_RING_TILES * 128u + (size_t)w * SPW * 64u + lane);
        uint4* dst = L.SM + ((c & 1u) * MIRT_RING_TILES * 128u + w * SPW * 64u);
#pragma unroll
        for (uint32_t k = 0; k < SPW; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 64u), (__attribute__((address_space(3))) void*)(dst + k * 64u), 16, 0, 0);
    };
    auto narrow = [&](uint32_t gbase, uint32_t cnt) {
        uint32_t k = 0, m = 0, gi = 0;
        for (;;) {
            if (m == 0 && k < cnt) {
                const uint2 e = ent2[k][lane];
                ++k;
                m = (~e.x & 0xFFFFu) | (~e.y << 16);
                gi = (gbase + (e.x >> 16)) * MIRT_CGROUP + (MIRT_CGROUP - 1);
            }
            if (__ballot(m != 0) == 0ull) break;
            if (m != 0) {
                const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
                m &= m - 1u;
                narrow_test(sc, GeomGlobal<float>{sc}, gi - bit + (bit >= 16u ? 32u : 0u), ray, a_true, tmin, best);
                ++narrow_count;
            }
        }
    };
    const uint32_t sub = lane >> 5;
    stage(0);
    __syncthreads(); /* waits for the copies (vmcnt) and for the other waves: also fences the previous call's readers */
    uint32_t t = 0, cnt = 0, gbase = 0;
    for (uint32_t c = 0; c < nch; ++c) {
        if (c + 1u < nch) stage(c + 1u);
        if (wave_live) {
            const mirt_h8* buf = (const mirt_h8*)(L.SM + (c & 1u) * MIRT_RING_TILES * 128u);
            const uint32_t t0 = c * MIRT_RING_TILES;
            uint32_t t_end = t0 + MIRT_RING_TILES;
            t_end = t_end < ntiles ? t_end : ntiles;
            while (t < t_end) {
                bool flush = false;
                for (; t < t_end && !flush; ++t) {
                    const mirt_h8 a0 = buf[(t - t0) * 128u + lane], a1 = buf[(t - t0) * 128u + 64u + lane];
                    uint32_t mx = 2 * t - gbase + sub, my = mx;
                    mx_tile(a0, a1, bx0, bx1, by0, by1, mx, my);
#ifdef MIRT_VERIFY_BROAD
                    mx_verify_group(sc, ray, a_true, active, 2 * t, mx);
                    mx_verify_group(sc, ray, a_true, active, 2 * t + 1, my);
#endif
                    ent2[cnt][lane] = make_uint2(mx, my);
                    cnt += ((mx & my & 0xFFFFu) != 0xFFFFu) ? 1u : 0u;
                    flush = __ballot(cnt > LDS::LDS_ENT_CAP / 2 - 1) != 0ull;
                    ++tiles_done;
                }
                /* a full list, or 0x7FFF tiles behind the pass's first one (group numbers in the entries are 16 bit) */
                if (flush || 2u * t - gbase >= 0xFFFEu) {
                    pt.stamp(PH_BROAD);
                    narrow(gbase, cnt);
                    pt.stamp(PH_NARROW);
                    cnt = 0;
                    gbase = 2 * t;
                }
            }
        }
        __syncthreads(); /* the next chunk has landed (vmcnt) and nobody reads this one any more */
    }
    pt.stamp(PH_BROAD);
    if (wave_live) narrow(gbase, cnt);
    pt.stamp(PH_NARROW);
}

/* ---- a few rays left in the wave: spheres across the lanes ------------------------------------------
 * Near the end of a launch a wave may be down to one or two live paths (a path through glass runs to the depth limit
 * while every other lane has finished); the tile loops above would still charge it the full price of 64 rays per
 * segment.  Here each live ray is broadcast in turn (v_readlane), every lane runs the exact test on spheres
 * lane, lane + 64, ... and the wave reduces (t, hit id) to the minimum - the same closest hit, since the exact test is
 * all the broad phases ever filter for.  ~8 rounds of one exact test per ray for the bench scene instead of 16 tiles of
 * matrix work: the tail of an 8-GPU band shrinks from ~1 ms to ~0.2 ms. */
#ifndef MIRT_FEW_LANES
#define MIRT_FEW_LANES 2u /* a wave alone on its SIMD (tools/lone_wave_probe.py): 0.9 us + 1.55 us per ray this way, ~5.1 us through the tile loop whatever the number of rays; 4: a wave of 4 rays 7.1 us per iteration instead of 5.1 */
#endif
template <class R, class G>
__device__ inline void trace_few(const DevScene<R>& sc, const G& geo, uint32_t lane, unsigned long long live, const RayT<R>& ray, R tmin, Closest<R>& best,
                                 uint32_t& narrow_count) {
    static_assert(sizeof(R) == 4, "f32 kernels only");
    best.t = INFINITY;
    best.idx = 0xFFFFFFFFu;
    while (live != 0ull) {
        const int L = __ffsll((long long)live) - 1;
        live &= live - 1ull;
        auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); };
        const RayT<float> r = {{bc(ray.o.x), bc(ray.o.y), bc(ray.o.z)}, {bc(ray.d.x), bc(ray.d.y), bc(ray.d.z)}, bc(ray.tm)};
        const RayA<float> a = ray_a(r.d);
        Closest<float> b = {INFINITY, 0xFFFFFFFFu};
        for (uint32_t i = lane; i < sc.n; i += MIRT_WAVE) narrow_test(sc, geo, i, r, a, tmin, b); /* dummy records never hit */
        /* minimum over the wave of (t, hit id): positive floats order like their bit patterns */
        uint32_t kt = __float_as_uint(b.t), kid = b.idx != 0xFFFFFFFFu ? sc.orig_id[b.idx] : 0xFFFFFFFFu, kidx = b.idx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t ot = (uint32_t)__shfl_xor((int)kt, o, 64), oid = (uint32_t)__shfl_xor((int)kid, o, 64), oidx = (uint32_t)__shfl_xor((int)kidx, o, 64);
            const bool take = ot < kt || (ot == kt && oid < kid);
            kt = take ? ot : kt;
            kid = take ? oid : kid;
            kidx = take ? oidx : kidx;
        }
        if ((int)lane == L) {
            best.t = __uint_as_float(kt);
            best.idx = kidx;
            narrow_count += (sc.n + MIRT_WAVE - 1) / MIRT_WAVE; /* charged to the ray's lane: exact tests per lane and ray */
        }
    }
}

/* minimum of a 64-bit key over the wave (every lane gets it): the DPP steps of wave_scan_incl with min for +; a lane a step
 * does not reach keeps its own value (old = the value itself) */
#define MIRT_MIN64_STEP(ctrl, rm)                                                                          \
    {                                                                                                      \
        const uint32_t ohi_ = (uint32_t)__builtin_amdgcn_update_dpp((int)hi_, (int)hi_, ctrl, rm, 0xF, false); \
        const uint32_t olo_ = (uint32_t)__builtin_amdgcn_update_dpp((int)lo_, (int)lo_, ctrl, rm, 0xF, false); \
        const bool take_ = ohi_ < hi_ || (ohi_ == hi_ && olo_ < lo_);                                          \
        hi_ = take_ ? ohi_ : hi_;                                                                          \
        lo_ = take_ ? olo_ : lo_;                                                                          \
    }
__device__ inline unsigned long long wave_min_u64(unsigned long long k) {
    uint32_t lo_ = (uint32_t)k, hi_ = (uint32_t)(k >> 32);
    MIRT_MIN64_STEP(0x111, 0xF) MIRT_MIN64_STEP(0x112, 0xF) MIRT_MIN64_STEP(0x114, 0xF) MIRT_MIN64_STEP(0x118, 0xF)
    MIRT_MIN64_STEP(0x142, 0xA) MIRT_MIN64_STEP(0x143, 0xC)
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi_, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo_, 63);
}
/* trace_few for the LDS layouts that hold the hit ids (the pair ring's: scenes of <= MIRT_MX_SMALL_RECORDS slots).  A wave in
 * this state runs alone on its SIMD - nothing hides a latency - and trace_few's rounds are one dependent chain each: load,
 * discriminant, and for whichever lane's sphere the line meets the square root and the divisions, round after round (1.75 us per
 * ray, tools/lone_wave_probe.py).  Here the discriminants of all of a lane's spheres (<= 8) come first, independent of each
 * other, then only the spheres that can contribute a root go through pair_key - the exact test of the balanced narrow phase, whose
 * 64-bit (t, hit id, slot) keys order like narrow_test's closest hit (see there) - and the wave takes the minimum. */
template <class LDS>
__device__ inline void trace_few_keys(const DevScene<float>& sc, const LDS& L, uint32_t lane, unsigned long long live, const RayT<float>& ray, float tmin,
                                      Closest<float>& best, uint32_t& narrow_count) {
    static_assert(LDS::PAIR_CAP > 0, "needs L.oid");
    const GeomLdsT<LDS> geo{L};
    best.t = INFINITY;
    best.idx = 0xFFFFFFFFu;
    while (live != 0ull) {
        const int Lr = __ffsll((long long)live) - 1;
        live &= live - 1ull;
        auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), Lr)); };
        const RayT<float> r = {{bc(ray.o.x), bc(ray.o.y), bc(ray.o.z)}, {bc(ray.d.x), bc(ray.d.y), bc(ray.d.z)}, bc(ray.tm)};
        const RayA<float> a = ray_a(r.d);
        uint32_t m = 0; /* bit k: sphere lane + 64 k passes the test pair_key starts with (same operations on the same values) */
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < MIRT_MX_SMALL_RECORDS / MIRT_WAVE; k0 += 4) /* four at a time: eight would spill */
#pragma unroll
        for (uint32_t k = k0; k < k0 + 4; ++k) {
            const uint32_t i = lane + k * MIRT_WAVE;
            const bool in = i < sc.n;
            const uint32_t ic = in ? i : 0u;
            const float4 sa = geo.a(ic), sb = geo.b(ic);
            V3<float> C = {sa.x, sa.y, sa.z};
            const V3<float> Cm = center_at(C, V3<float>{sb.x, sb.y, sb.z}, r.tm);
            if (ic >= sc.n_static) C = Cm;
            float h, d;
            disc(C - r.o, r.d, a, sa.w, h, d);
            if (in && d >= 0.0f && !(h < -h_normal_min(h) && d <= h * h && tmin >= 0.0f)) m |= 1u << k;
        }
        unsigned long long key = MIRT_KEY_NONE;
        while (__ballot(m != 0u) != 0ull) {
            if (m != 0u) {
                const uint32_t i = lane + ((uint32_t)__ffs((int)m) - 1u) * MIRT_WAVE;
                m &= m - 1u;
                const unsigned long long kk = pair_key(sc, geo, i, (uint32_t)L.oid[i], r, a, tmin);
                key = kk < key ? kk : key;
            }
        }
        key = wave_min_u64(key);
        if ((int)lane == Lr) {
            if (key != MIRT_KEY_NONE) {
                best.t = __uint_as_float((uint32_t)(key >> 32));
                best.idx = (uint32_t)key & 0xFFFFu;
            }
            narrow_count += (sc.n + MIRT_WAVE - 1) / MIRT_WAVE; /* as trace_few counts */
        }
    }
}

/* ---- f32, MIRT_FLAG_BVH: closest hit through the bounding-volume tree --------------------------------
 * Per-lane stack traversal (stack in LDS) of the tree built by mirt_scene_create; the leaves hold the same (M, w)
 * bounding records as the brute-force broad phase and are tested with the same expression, the survivors go through
 * the same exact narrow_test: the result is the brute-force result, sphere for sphere.
 *
 * Why a node may be skipped.  The leaf test is conservative for the exact test (trace_cons), and it passes only if the
 * real line comes within sqrt(rho_i^2 + G_i) <= rho_i + sqrt(G_i) of M_i, where G_i <= 240 eps (|M_i| + rho_i + |O|)^2
 * collects its margins and evaluation error, i.e. sqrt(G_i) <= s (|M_i| + rho_i + |O|) with s = MIRT_BVH_S = 4e-3
 * >= 15.5 sqrt(eps).  The exact root t of the sphere puts the point O + t D within the same distance of M_i (its error
 * is <= 3.3 sqrt(eps) max(|C - O|, |C|)).  So line and hit point lie inside the box of ball i widened by
 * s (|M_i| + rho_i) + s |O| per axis; a node's box is the union over its leaves, widened on the host by
 * s max_i(|M_i| + rho_i), and the ray contributes delta = s |O| by shifting its origin: the slab parameters are
 * (lo - (O + delta)) / D and (hi - (O - delta)) / D.  They carry a relative rounding error of 2 eps, hence the
 * factors (1 +- 1e-5) in the three tests
 *     tnear <= tfar    the line meets the widened box
 *     tfar  >= tmin    ... not wholly behind the origin
 *     tnear <= best.t  ... not wholly beyond the best hit (ties are not beyond) */
/* where the tree lives: HBM/L2, or (scenes of <= MIRT_LDS_RECORDS spheres) a copy in the block's LDS */
struct BvhGlobalStore {
    const DevScene<float>& sc;
    typedef GeomGlobal<float> Geom;
    __device__ inline void node(uint32_t i, float4& q0, float4& q1, float4& q2, float4& q3) const {
        const BvhNode& nd = sc.bvh[i];
        q0 = nd.q0; q1 = nd.q1; q2 = nd.q2; q3 = nd.q3;
    }
    __device__ inline float4 rec(uint32_t i) const { return sc.bvh_S[i]; }
    __device__ inline uint32_t idx(uint32_t i) const { return sc.bvh_idx[i]; }
    __device__ inline Geom geom() const { return Geom{sc}; }
};
#define MIRT_BVH_LDS_NODES 256
#define MIRT_BVH_LDS_STACK 16
struct BvhLds { /* 133 KB: one block of 16 waves per CU */
    BvhNode nodes[MIRT_BVH_LDS_NODES];
    float4 S[MIRT_LDS_RECORDS + 8]; /* leaf order */
    float4 A[MIRT_LDS_RECORDS], B[MIRT_LDS_RECORDS]; /* sorted-sphere order, as in HBM */
    uint16_t mat[MIRT_LDS_RECORDS], idx[MIRT_LDS_RECORDS];
    uint32_t stk[MIRT_F32_WAVES][MIRT_BVH_LDS_STACK][MIRT_WAVE];
    __device__ inline void fill(const DevScene<float>& sc) { /* the host launches this kernel only if everything fits */
        for (uint32_t i = threadIdx.x; i < sc.bvh_n_nodes; i += blockDim.x) nodes[i] = sc.bvh[i];
        for (uint32_t i = threadIdx.x; i < sc.bvh_n_recs + 8; i += blockDim.x) S[i] = sc.bvh_S[i];
        for (uint32_t i = threadIdx.x; i < sc.bvh_n_recs; i += blockDim.x) idx[i] = (uint16_t)sc.bvh_idx[i];
        for (uint32_t i = threadIdx.x; i < sc.n; i += blockDim.x) {
            A[i] = sc.A[i];
            B[i] = sc.B[i];
            mat[i] = (uint16_t)sc.material[i];
        }
        __syncthreads();
    }
};
struct GeomBvhLds {
    const BvhLds& L;
    __device__ inline float4 a(uint32_t i) const { return L.A[i]; }
    __device__ inline float4 b(uint32_t i) const { return L.B[i]; }
    __device__ inline uint32_t material(uint32_t i) const { return L.mat[i]; }
};
struct BvhLdsStore {
    const BvhLds& L;
    typedef GeomBvhLds Geom;
    __device__ inline void node(uint32_t i, float4& q0, float4& q1, float4& q2, float4& q3) const {
        const BvhNode& nd = L.nodes[i];
        q0 = nd.q0; q1 = nd.q1; q2 = nd.q2; q3 = nd.q3;
    }
    __device__ inline float4 rec(uint32_t i) const { return L.S[i]; }
    __device__ inline uint32_t idx(uint32_t i) const { return L.idx[i]; }
    __device__ inline Geom geom() const { return Geom{L}; }
};

/* A traversal can be suspended and resumed: its state is the node in hand, the stack depth (the stack itself lives in
 * LDS) and the best hit so far.  `keep` = the number of lanes of the wave that must still be traversing for the loop to
 * go on; 1 runs every lane to completion. */
struct BvhState {
    uint32_t cur, sp;
};
template <class Store>
__device__ inline void trace_bvh_resume(const DevScene<float>& sc, const Store& store, uint32_t (*stk)[MIRT_WAVE], uint32_t lane, const RayT<float>& ray,
                                        float tmin, Closest<float>& best, BvhState& st, uint32_t keep, uint32_t& narrow_count, PhaseTimer& pt) {
    const V3<float> O = ray.o;
    const RayA<float> a_true = ray_a(ray.d);
    const float inv = filter_rsq(a_true.a);
    const V3<float> Dn = {ray.d.x * inv, ray.d.y * inv, ray.d.z * inv};
    const float k1 = __builtin_fmaf(Dn.z, O.z, __builtin_fmaf(Dn.y, O.y, Dn.x * O.x));
    const float o2 = __builtin_fmaf(O.z, O.z, __builtin_fmaf(O.y, O.y, O.x * O.x));
    const float nk2 = -(o2 * MIRT_RAY_SHRINK);
    const V3<float> O2 = {O.x * 2.0f, O.y * 2.0f, O.z * 2.0f};
    /* slab constants: the per-ray widening delta = s |O| (rounded up), 1/D with |1/D| capped (D = 0 -> 1e30: a
     * parallel ray gets +-huge parameters of the right sign instead of inf - inf) */
    const float delta = filter_sqrt(o2) * (MIRT_BVH_S * 1.0001f) + 1e-30f;
    const V3<float> Op = {O.x + delta, O.y + delta, O.z + delta}, Om = {O.x - delta, O.y - delta, O.z - delta};
    auto cap_inv = [](float d) {
        const float r = 1.0f / d;
        return fabsf(r) <= 1e30f ? r : copysignf(1e30f, d); /* d = +-0 or denormal */
    };
    const V3<float> iD = {cap_inv(ray.d.x), cap_inv(ray.d.y), cap_inv(ray.d.z)};
    const float tmin_dn = tmin * 0.99999f;
    float best_up = best.t * 1.00001f;
    uint32_t cur = st.cur, sp = st.sp;
    /* "while-while": all lanes first walk inner nodes until each holds a leaf (or nothing), then the leaves are
     * processed together */
    for (;;) {
        const uint32_t live = (uint32_t)__popcll(__ballot(cur != MIRT_BVH_NONE));
        if (live == 0 || live < keep) break;
        while (__ballot(cur != MIRT_BVH_NONE && !(cur & 0x80000000u)) != 0ull) {
#ifdef MIRT_BVH_COUNTERS
            if (lane == 0) atomicAdd((unsigned long long*)sc.dbg + 6, 1ull);
            if (cur != MIRT_BVH_NONE && !(cur & 0x80000000u)) atomicAdd((unsigned long long*)sc.dbg + 5, 1ull);
#endif
            if (cur != MIRT_BVH_NONE && !(cur & 0x80000000u)) {
                float4 q0, q1, q2, q3;
                store.node(cur, q0, q1, q2, q3);
                /* a parameter pair per axis and child; which of the two is the entry depends on the sign of D */
                auto slab = [&](float lx, float ly, float lz, float hx, float hy, float hz, float& tn, float& tf) {
                    const float ax = (lx - Op.x) * iD.x, bx = (hx - Om.x) * iD.x;
                    const float ay = (ly - Op.y) * iD.y, by = (hy - Om.y) * iD.y;
                    const float az = (lz - Op.z) * iD.z, bz = (hz - Om.z) * iD.z;
                    tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
                    tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
                };
                float tn0, tf0, tn1, tf1;
                slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn0, tf0);
                slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn1, tf1);
                /* slack on the far side only matters where it is positive; tnear is compared after shrinking towards 0 */
                const float tf0u = tf0 * 1.00001f, tf1u = tf1 * 1.00001f;
                const float tn0d = tn0 > 0.0f ? tn0 * 0.99999f : tn0, tn1d = tn1 > 0.0f ? tn1 * 0.99999f : tn1;
                const uint32_t ref0 = __float_as_uint(q3.x), ref1 = __float_as_uint(q3.y);
                const bool p0 = tn0d <= tf0u && tf0u >= tmin_dn && tn0d <= best_up;
                const bool p1 = tn1d <= tf1u && tf1u >= tmin_dn && tn1d <= best_up;
                const bool swap = tn1 < tn0; /* nearer entry first */
                const uint32_t r_near = swap ? ref1 : ref0, r_far = swap ? ref0 : ref1;
                const bool p_near = swap ? p1 : p0, p_far = swap ? p0 : p1;
                if (p_near) {
                    cur = r_near;
                    if (p_far) stk[sp++][lane] = r_far; /* depth <= MIRT_BVH_STACK by construction */
                } else if (p_far) {
                    cur = r_far;
                } else {
                    cur = MIRT_BVH_NONE;
                    if (sp > 0) cur = stk[--sp][lane];
                }
            }
        }
#ifdef MIRT_BVH_COUNTERS
        if (lane == 0) atomicAdd((unsigned long long*)sc.dbg + 7, 1ull);
        if (cur != MIRT_BVH_NONE) atomicAdd((unsigned long long*)sc.dbg + 8, 1ull);
#endif
        if (cur != MIRT_BVH_NONE) { /* a leaf: the conservative record tests (fetched at once), then the exact tests */
            const uint32_t first = (cur >> 3) & 0x0FFFFFFFu, cnt = (cur & 7u) + 1u;
            float4 m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = store.rec(first + k); /* leaves hold <= 4 records; the array is padded */
            uint32_t msk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float h = __builtin_fmaf(Dn.z, m[k].z, __builtin_fmaf(Dn.y, m[k].y, __builtin_fmaf(Dn.x, m[k].x, -k1)));
                const float q = __builtin_fmaf(h, h, __builtin_fmaf(O2.z, m[k].z, __builtin_fmaf(O2.y, m[k].y, __builtin_fmaf(O2.x, m[k].x, nk2 - m[k].w))));
                if (q >= 0.0f && (uint32_t)k < cnt) msk |= 1u << k;
            }
            while (msk) {
                const uint32_t k = (uint32_t)__ffs((int)msk) - 1u;
                msk &= msk - 1u;
                narrow_test(sc, store.geom(), store.idx(first + k), ray, a_true, tmin, best);
                ++narrow_count;
            }
            best_up = best.t * 1.00001f;
            cur = MIRT_BVH_NONE;
            if (sp > 0) cur = stk[--sp][lane];
        }
    }
    st.cur = cur;
    st.sp = sp;
    pt.stamp(PH_BROAD);
}
template <class Store>
__device__ inline void trace_bvh(const DevScene<float>& sc, const Store& store, uint32_t (*stk)[MIRT_WAVE], uint32_t lane, bool active, const RayT<float>& ray,
                                 float tmin, Closest<float>& best, uint32_t& narrow_count, PhaseTimer& pt) {
    best.t = INFINITY;
    best.idx = 0xFFFFFFFFu;
    BvhState st{active ? sc.bvh_root : MIRT_BVH_NONE, 0u};
    trace_bvh_resume(sc, store, stk, lane, ray, tmin, best, st, 1u, narrow_count, pt);
}

/* ---- hit record of the winner: sphere.rs:102-107, hittable.rs:48-60 -------------------------------- */
template <class R> struct Hit {
    V3<R> p, n, on; /* on = outward normal (input of Sphere::uv) */
    bool front;
};
template <class R, class G> __device__ inline Hit<R> make_hit(const DevScene<R>& sc, const G& geo, const Closest<R>& c, const RayT<R>& ray) {
    typename Vec4<R>::type sa = geo.a(c.idx);
    typename Vec4<R>::type sb = geo.b(c.idx);
    V3<R> C = {sa.x, sa.y, sa.z};
    if (c.idx >= sc.n_static) C = center_at(C, V3<R>{sb.x, sb.y, sb.z}, ray.tm);
    Hit<R> h;
    if constexpr (sizeof(R) == 8) {
        h.p = ray.o + ray.d * c.t;   /* ray.rs:27-29 */
        h.on = (h.p - C) / sb.w;     /* sphere.rs:103 */
    } else { /* f32 spec: fused o + t d; B.w holds 1 / r (one division per sphere on the host instead of three per hit) */
        h.p = {__builtin_fmaf(ray.d.x, c.t, ray.o.x), __builtin_fmaf(ray.d.y, c.t, ray.o.y), __builtin_fmaf(ray.d.z, c.t, ray.o.z)};
        h.on = (h.p - C) * sb.w;
    }
    h.front = dot(ray.d, h.on) < (R)0.0;
    h.n = h.front ? h.on : -h.on;
    return h;
}

/* ---- closest-hit front ends used by the kernels: f64 = exact broad phase, f32 = conservative broad phase ---- */
#define MIRT_TINY_SCENE 8u /* sorted slots (padded to groups of 4) */
struct TracerF64 {
    typedef double Real;
    static constexpr bool RESUMABLE = false;
    const DevScene<double>& sc;
    WaveLds<double>& lds;
    __device__ inline void closest(uint32_t lane, bool active, const RayT<double>& ray, double tmin, Closest<double>& best, uint32_t& nc, PhaseTimer& pt) {
        if (sc.n_live <= MIRT_TINY_SCENE) { /* a handful of spheres: the exact test on each (see TracerF32::closest) */
            best.t = (double)INFINITY;
            best.idx = 0xFFFFFFFFu;
            const RayA<double> a = ray_a(ray.d);
            if (active) {
                for (uint32_t i = 0; i < sc.n_live; ++i) narrow_test(sc, GeomGlobal<double>{sc}, i, ray, a, tmin, best); /* dummy records never hit */
                nc += sc.n_live;
            }
            pt.stamp(PH_NARROW);
            return;
        }
        trace<double, true>(sc, lds, lane, active, ray, tmin, best, nc, pt);
    }
    __device__ inline Hit<double> hit(const Closest<double>& c, const RayT<double>& ray) const { return make_hit(sc, GeomGlobal<double>{sc}, c, ray); }
    __device__ inline uint32_t material(uint32_t idx) const { return sc.material[idx]; }
    __device__ inline const DevMaterial<double>& material_record(uint32_t idx) const { return sc.mats[material(idx)]; }
};
template <bool RES, bool ONLY_TINY = false> struct TracerF32 { /* ONLY_TINY: the kernel the host launches for scenes of <= MIRT_TINY_SCENE slots - nothing but the exact tests is compiled in */
    typedef float Real;
    static constexpr bool RESUMABLE = false;
    const DevScene<float>& sc;
    F32Lds& L;
    uint32_t w;
    __device__ inline void closest(uint32_t lane, bool active, const RayT<float>& ray, float tmin, Closest<float>& best, uint32_t& nc, PhaseTimer& pt) {
        if (ONLY_TINY || (RES && sc.n_live <= MIRT_TINY_SCENE)) {
            /* a handful of spheres (the textured presets have 1-3): the exact test on each of them, no lists, no broad
             * phase - 16 bounding records per group would be mostly padding */
            best.t = INFINITY;
            best.idx = 0xFFFFFFFFu;
            const RayA<float> a = ray_a(ray.d);
            if (active) {
                for (uint32_t i = 0; i < sc.n_live; ++i) narrow_test(sc, GeomLds{L}, i, ray, a, tmin, best); /* dummy records never hit */
                nc += sc.n_live;
            }
            pt.stamp(PH_NARROW);
            return;
        }
        /* (trace_few for the last rays of a wave, as in the matrix-pipe tracer: a lone wave's iteration 14.2 -> 2.8 us, the all-VALU
         * bench frame 27.07 -> 27.19 ms - measured and dropped) */
        trace_cons<RES>(sc, L, w, lane, active, ray, tmin, best, nc, pt);
    }
    __device__ inline Hit<float> hit(const Closest<float>& c, const RayT<float>& ray) const {
        if (RES) return make_hit(sc, GeomLds{L}, c, ray);
        return make_hit(sc, GeomGlobal<float>{sc}, c, ray);
    }
    __device__ inline uint32_t material(uint32_t idx) const { return RES ? (uint32_t)L.mat[idx] : sc.material[idx]; }
    /* by value: the record may come from the LDS copy or from HBM/L2 (two address spaces) */
    __device__ inline DevMaterial<float> material_record(uint32_t idx) const {
        const uint32_t m = material(idx);
        /* one generic pointer for both sources: the fields are fetched with flat_load, through the texture path also when the
         * record sits in LDS - which suits THIS tracer, whose broad phase keeps the LDS pipe busy with record broadcasts (with
         * ds_read here the all-VALU bench frame is 2.3 % slower, the textured 4K frame 0.8 %) */
        if (sc.n_materials <= MIRT_F32_LDS_MATS) return L.mats[m];
        return sc.mats[m];
    }
};

/* ML: the scene's material records sit in the block's LDS copy - a compile-time switch of the matrix-pipe tracers, one kernel body
 * per value (picked by a block-uniform branch in the kernel).  Written as `fits ? L.mats[m] : sc.mats[m]` (above) the two sources
 * become one generic pointer and every field is fetched with a flat_load; with the address space known the resident records
 * come through ds_read: bench frame K2 -1.4 %. */
template <bool RES, class LDS, bool ML> struct TracerF32Mx {
    typedef float Real;
    static constexpr bool RESUMABLE = false;
    static constexpr bool CONTINUES = !RES; /* scenes beyond the LDS: the paths the last waves are left with go to a second launch (render_body) */
    const DevScene<float>& sc;
    LDS& L;
    uint32_t w;
    uint32_t tiles = 0; /* tiles of 32 spheres this wave ran through the matrix pipe (wave-uniform): what the tile cull left */
    __device__ inline void closest(uint32_t lane, bool active, const RayT<float>& ray, float tmin, Closest<float>& best, uint32_t& nc, PhaseTimer& pt) {
        const unsigned long long live = __ballot(active);
        /* resident scenes only: trace_few runs n / 64 rounds of the exact test per live ray - 16 rounds from LDS at most - but
         * beyond the LDS every round is a dependent trip to L2 (10 000 spheres: 157 per ray, ~45 us per ray and bounce: the one
         * wave with a path inside a glass ball outlived all others by 6 ms of the 38 ms stress frame), while the culled tile
         * loop below costs such a wave its top tiles and the handful of tiles its few rays can touch */
        if (RES && (uint32_t)__popcll(live) <= MIRT_FEW_LANES) {
            if constexpr (mx_pair_cap<LDS>::value > 0 && LDS::N_RECORDS <= MIRT_MX_SMALL_RECORDS) trace_few_keys(sc, L, lane, live, ray, tmin, best, nc);
            else trace_few(sc, GeomLdsT<LDS>{L}, lane, live, ray, tmin, best, nc);
            pt.stamp(PH_NARROW);
        } else {
            trace_mx<float, LDS, RES>(sc, L, w, lane, active, ray, tmin, best, nc, pt, tiles);
        }
    }
    __device__ inline Hit<float> hit(const Closest<float>& c, const RayT<float>& ray) const {
        if (RES) return make_hit(sc, GeomLdsT<LDS>{L}, c, ray);
        return make_hit(sc, GeomGlobal<float>{sc}, c, ray);
    }
    __device__ inline uint32_t material(uint32_t idx) const { return RES ? (uint32_t)L.mat[idx] : sc.material[idx]; }
    /* by value: the record may come from the LDS copy or from HBM/L2 (two address spaces) */
    __device__ inline DevMaterial<float> material_record(uint32_t idx) const {
        const uint32_t m = material(idx);
        if constexpr (ML) return L.mats[m];
        else return sc.mats[m];
    }
};

/* scenes beyond the LDS: the block's waves in step over an LDS ring of operand chunks (trace_mx_ring) */
template <class LDS, bool ML> struct TracerF32MxRing {
    typedef float Real;
    static constexpr bool RESUMABLE = false;
    static constexpr bool LOCKSTEP = true; /* closest() holds block barriers: every wave of the block calls it the same number of times */
    const DevScene<float>& sc;
    LDS& L;
    uint32_t w;
    uint32_t tiles = 0;
    __device__ inline void closest(uint32_t lane, bool active, const RayT<float>& ray, float tmin, Closest<float>& best, uint32_t& nc, PhaseTimer& pt) {
        trace_mx_ring<LDS>(sc, L, w, lane, active, ray, tmin, best, nc, pt, tiles);
    }
    __device__ inline Hit<float> hit(const Closest<float>& c, const RayT<float>& ray) const { return make_hit(sc, GeomGlobal<float>{sc}, c, ray); }
    __device__ inline uint32_t material(uint32_t idx) const { return sc.material[idx]; }
    __device__ inline DevMaterial<float> material_record(uint32_t idx) const {
        const uint32_t m = material(idx);
        if constexpr (ML) return L.mats[m];
        else return sc.mats[m];
    }
};

template <bool RES> struct TracerF64Mx {
    typedef double Real;
    static constexpr bool RESUMABLE = false;
    const DevScene<double>& sc;
    MxLds64& L;
    uint32_t w;
    uint32_t tiles = 0;
    __device__ inline void closest(uint32_t lane, bool active, const RayT<double>& ray, double tmin, Closest<double>& best, uint32_t& nc, PhaseTimer& pt) {
        trace_mx<double, MxLds64, RES>(sc, L, w, lane, active, ray, tmin, best, nc, pt, tiles);
    }
    __device__ inline Hit<double> hit(const Closest<double>& c, const RayT<double>& ray) const { return make_hit(sc, GeomGlobal<double>{sc}, c, ray); }
    __device__ inline uint32_t material(uint32_t idx) const { return sc.material[idx]; }
    __device__ inline const DevMaterial<double>& material_record(uint32_t idx) const { return sc.mats[material(idx)]; }
};

template <class Store, bool DYN> struct TracerF32Bvh {
    typedef float Real;
    static constexpr bool RESUMABLE = DYN; /* render_body may leave a traversal half-way to refill finished lanes */
    const DevScene<float>& sc;
    Store store;
    uint32_t (*stk)[MIRT_WAVE];
    __device__ inline void closest(uint32_t lane, bool active, const RayT<float>& ray, float tmin, Closest<float>& best, uint32_t& nc, PhaseTimer& pt) {
        trace_bvh(sc, store, stk, lane, active, ray, tmin, best, nc, pt);
    }
    __device__ inline void resume(uint32_t lane, const RayT<float>& ray, float tmin, Closest<float>& best, BvhState& st, uint32_t keep, uint32_t& nc,
                                  PhaseTimer& pt) {
        trace_bvh_resume(sc, store, stk, lane, ray, tmin, best, st, keep, nc, pt);
    }
    __device__ inline Hit<float> hit(const Closest<float>& c, const RayT<float>& ray) const { return make_hit(sc, store.geom(), c, ray); }
    __device__ inline uint32_t material(uint32_t idx) const { return store.geom().material(idx); }
    __device__ inline const DevMaterial<float>& material_record(uint32_t idx) const { return sc.mats[material(idx)]; }
};

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
/* Lambertian attenuation = texture colour / reflectance (material.rs:78).  The constant colours come pre-divided from
 * mirt_scene_create (same IEEE quotient); a texel is divided here. */
template <class R> __device__ inline V3<R> lambertian_attenuation(const DevScene<R>& sc, const DevMaterial<R>& m, const Hit<R>& h) {
    if (m.tex_kind == MIRT_TEX_SOLID) return m.albedo; /* texture.rs:24-28 */
    if (m.tex_kind == MIRT_TEX_CHECKER) {              /* texture.rs:46-59 */
        MIRT_DIAG(sc, DG_CHECKER, true);
        uint32_t acc = (uint32_t)floor_to_i32(m_floor(h.p.x * m.inv_scale));
        acc += (uint32_t)floor_to_i32(m_floor(h.p.y * m.inv_scale));
        acc += (uint32_t)floor_to_i32(m_floor(h.p.z * m.inv_scale));
        return (((int32_t)acc) % 2 == 0) ? m.albedo : m.odd;
    }
    /* IMAGE: texture.rs:71-94 with (u,v) of sphere.rs:72-76 */
    if (m.img_h == 0) return V3<R>{(R)0.0, (R)1.0, (R)1.0} / m.reflectance;
    MIRT_DIAG(sc, DG_IMAGE, true);
    const R PI = (R)3.14159265358979323846264338327950288;
    const R FRAC_1_PI = (R)0.318309886183790671537767526745028724;
    R theta = -m_acos(h.on.y);
    R phi = -m_atan2(h.on.z, h.on.x) + PI;
    R u = (R)0.5 * phi * FRAC_1_PI;
    R v = theta * FRAC_1_PI;
    u = u < (R)0.0 ? (R)0.0 : (u > (R)1.0 ? (R)1.0 : u);
    v = m_fabs(v);
    v = v < (R)0.0 ? (R)0.0 : (v > (R)1.0 ? (R)1.0 : v);
    uint32_t i = to_u32_sat(u * (R)m.img_w);
    uint32_t j = to_u32_sat(v * (R)m.img_h);
    i = i < m.img_w - 1 ? i : m.img_w - 1;
    j = j < m.img_h - 1 ? j : m.img_h - 1;
    uint32_t px = sc.texels[m.img_offset + j * m.img_w + i];
    const V3<R> byte = {(R)(px & 0xFFu), (R)((px >> 8) & 0xFFu), (R)((px >> 16) & 0xFFu)};
    if constexpr (sizeof(R) == 8) return V3<R>{byte.x / (R)255.0, byte.y / (R)255.0, byte.z / (R)255.0} / m.reflectance; /* texture.rs:91-93, material.rs:78 */
    else return byte * m.inv_scale; /* f32 spec: inv_scale = (1 / 255) / reflectance, taken once on the host */
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

/* One scatter event.  Returns false when the path is absorbed (black for the whole path, camera.rs:279).
 * The bounce's Philox block is evaluated ONCE, before the material switch: a wave usually holds lanes of all three
 * materials, and divergent branches that each called the generator would run it up to three times. */
template <class R>
__device__ inline bool scatter(const DevScene<R>& sc, const DevMaterial<R>& m, const V3<R>& rin_d, const Hit<R>& hit, const Draws<R>& dr, V3<R>& att,
                               V3<R>& out_d) { /* the scattered ray starts at hit.p at the incoming ray's time (the caller's business): only its direction comes from here */
    R ua, ub, uc, ud; /* dims 0..3 of this bounce */
    dr.dims0123(ua, ub, uc, ud);
    if (m.kind == MIRT_MAT_LAMBERTIAN) { /* material.rs:59-84; dims: 0 = absorb test, 1..3 = vector */
        MIRT_DIAG(sc, DG_LAMBERT, true);
        if (!(ua < m.reflectance)) return false;
        V3<R> dir = hit.n + unit(V3<R>{ub, uc, ud}); /* positive-octant quirk of point3.rs:88-93 */
        const R e = (R)1e-8;
        if (m_fabs(dir.x) < e && m_fabs(dir.y) < e && m_fabs(dir.z) < e) dir = hit.n;
        out_d = dir;
        att = lambertian_attenuation(sc, m, hit);
        return true;
    }
    if (m.kind == MIRT_MAT_METAL) { /* material.rs:97-114; dims 0..2 = vector */
        MIRT_DIAG(sc, DG_METAL, true);
        MIRT_DIAG(sc, DG_FUZZ, m.has_fuzz != 0);
        V3<R> refl = reflect(rin_d, hit.n);
        if (m.has_fuzz) refl = unit(refl) + unit(V3<R>{ua, ub, uc}) * m.fuzz;
        out_d = refl;
        att = m.albedo;
        return !m.has_fuzz || dot(refl, hit.n) > (R)0.0;
    }
    /* DIELECTRIC material.rs:126-154; dim 0 (the reference short-circuits the draw away under total internal
     * reflection; with a counter RNG an unused draw changes nothing) */
    MIRT_DIAG(sc, DG_GLASS, true);
    att = {(R)1.0, (R)1.0, (R)1.0};
    R ri = hit.front ? (R)1.0 / m.ior : m.ior;
    V3<R> udir = unit(rin_d);
    R cos_theta = m_fmin(dot(-udir, hit.n), (R)1.0);
    R sin_theta = m_sqrt((R)1.0 - cos_theta * cos_theta);
    bool refl = ri * sin_theta > (R)1.0;
    if (!refl) refl = schlick(cos_theta, ri) > ua;
    out_d = refl ? reflect(udir, hit.n) : refract(udir, hit.n, ri);
    return true;
}

/* ---- camera rays: camera/camera.rs:209-230, 248-252, 261-264 --------------------------------------- */
template <class R> __device__ inline RayT<R> ray_for(const DevCamera<R>& c, uint32_t wn, uint32_t hn, const Draws<R>& dr) {
    R ua, ub, uc, ud, ue; /* dims 0,1 = anti-aliasing offsets, 2,3 = lens disk (r, theta), 4 = shutter time */
    dr.dims01234(ua, ub, uc, ud, ue);
    R ox = ua - (R)0.5, oy = ub - (R)0.5;
    V3<R> px = c.px00 + (c.du * ((R)wn + ox)) + (c.dv * ((R)hn + oy));
    V3<R> orig = c.lookfrom;
    if (!(c.defocus_angle <= (R)0.0)) {
        R r = m_sqrt(uc);
        R sn, cs;
        lens_sincos(ud, sn, cs); /* theta = 2 pi ud */
        R dx = r * cs, dy = r * sn;
        orig = c.lookfrom + (c.disk_u * dx) + (c.disk_v * dy);
    }
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

#ifdef MIRT_PHASE_TIMING
static __device__ unsigned long long g_prev_life_max; /* longest wave lifetime any launch has seen: the yardstick of the end-time histogram */
#endif

template <class R> __device__ inline R wave_sum_u(R v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

/* ---- K1 ------------------------------------------------------------------------------------------ */
template <class T>
__device__ inline void primary_hits_body(T& tr, const DevScene<typename T::Real>& sc, const DevCamera<typename T::Real>& cam, const DevPart& part,
                                         uint32_t wave_global, uint32_t lane, uint32_t* __restrict__ out_id, typename T::Real* __restrict__ out_t) {
    typedef typename T::Real R;
    PixelMap pm = map_pixel(wave_global, lane, cam.W, part);
    RayT<R> ray = primary_ray(cam, pm.lx, pm.gy);
    Closest<R> best;
    uint32_t narrow = 0;
    PhaseTimer pt;
    pt.start();
    tr.closest(lane, pm.valid, ray, (R)0.001, best, narrow, pt);
    if (pm.valid) {
        uint32_t row = part.compact ? pm.ly : pm.gy;
        size_t o = (size_t)row * cam.W + pm.lx;
        out_id[o] = best.idx != 0xFFFFFFFFu ? sc.orig_id[best.idx] : 0xFFFFFFFFu;
        if (out_t) out_t[o] = best.t;
    }
}
__global__ __launch_bounds__(MIRT_BLOCK) void primary_hits_kernel_f64(DevScene<double> sc, DevCamera<double> cam, DevPart part, uint32_t* __restrict__ out_id,
                                                                      double* __restrict__ out_t) {
    __shared__ WaveLds<double> lds_all[MIRT_WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF64 tr{sc, lds_all[w]};
    primary_hits_body(tr, sc, cam, part, blockIdx.x * MIRT_WAVES_PER_BLOCK + w, lane, out_id, out_t);
}
__global__ __launch_bounds__(MIRT_BLOCK) void primary_hits_kernel_f64_mx(DevScene<double> sc, DevCamera<double> cam, DevPart part, uint32_t* __restrict__ out_id,
                                                                         double* __restrict__ out_t) {
    __shared__ MxLds64 L;
    const bool res = L.fill(sc);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (res) {
        TracerF64Mx<true> tr{sc, L, w};
        primary_hits_body(tr, sc, cam, part, blockIdx.x * MIRT_WAVES_PER_BLOCK + w, lane, out_id, out_t);
    } else {
        TracerF64Mx<false> tr{sc, L, w};
        primary_hits_body(tr, sc, cam, part, blockIdx.x * MIRT_WAVES_PER_BLOCK + w, lane, out_id, out_t);
    }
}
__global__ __launch_bounds__(MIRT_F32_BLOCK) void primary_hits_kernel_f32(DevScene<float> sc, DevCamera<float> cam, DevPart part, uint32_t* __restrict__ out_id,
                                                                          float* __restrict__ out_t) {
    __shared__ F32Lds L;
    const bool res = L.fill(sc);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t wave_global = blockIdx.x * MIRT_F32_WAVES + w;
    if (res) {
        TracerF32<true> tr{sc, L, w};
        primary_hits_body(tr, sc, cam, part, wave_global, lane, out_id, out_t);
    } else {
        TracerF32<false> tr{sc, L, w};
        primary_hits_body(tr, sc, cam, part, wave_global, lane, out_id, out_t);
    }
}

__global__ __launch_bounds__(MIRT_F32_BLOCK) void primary_hits_kernel_f32_mx(DevScene<float> sc, DevCamera<float> cam, DevPart part, uint32_t* __restrict__ out_id,
                                                                             float* __restrict__ out_t) {
    __shared__ MxLds L;
    const bool res = L.fill(sc);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (res) {
        TracerF32Mx<true, MxLds, false> tr{sc, L, w};
        primary_hits_body(tr, sc, cam, part, blockIdx.x * MIRT_F32_WAVES + w, lane, out_id, out_t);
    } else {
        TracerF32Mx<false, MxLds, false> tr{sc, L, w};
        primary_hits_body(tr, sc, cam, part, blockIdx.x * MIRT_F32_WAVES + w, lane, out_id, out_t);
    }
}

__global__ __launch_bounds__(MIRT_BLOCK) void primary_hits_kernel_f32_bvh(DevScene<float> sc, DevCamera<float> cam, DevPart part, uint32_t* __restrict__ out_id,
                                                                          float* __restrict__ out_t) {
    __shared__ uint32_t stk[MIRT_WAVES_PER_BLOCK][MIRT_BVH_STACK][MIRT_WAVE];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF32Bvh<BvhGlobalStore, false> tr{sc, BvhGlobalStore{sc}, stk[w]};
    primary_hits_body(tr, sc, cam, part, blockIdx.x * MIRT_WAVES_PER_BLOCK + w, lane, out_id, out_t);
}

/* ---- K2: persistent waves pulling work items from a queue --------------------------------------------
 * Work item = (pixel of this rank's band, chunk of `chunk` consecutive samples).  A LANE owns an item: it
 * runs the item's samples one after the other (regenerating its next camera ray the moment a path ends,
 * camera.rs:242-245), sums their radiance in sample order and stores ONE partial sum.  A WAVE reserves
 * MIRT_QUEUE_BATCH items at a time with one atomic and deals them to its idle lanes, so lanes never wait
 * for a slow neighbour and the grid has no tail longer than one item.  Which lane runs which item depends
 * on timing, but an item's result does not (counter RNG keyed on pixel/sample/bounce), and K2b adds the
 * chunks of a pixel in chunk order: the image is bit-reproducible run to run and across GPU counts. */
#define MIRT_QUEUE_BATCH 256u
struct RenderParams {
    uint32_t n_items;    /* band pixels * chunks_per_pixel */
    uint32_t chunks;     /* chunks per pixel */
    uint32_t chunk;      /* samples per chunk */
    uint32_t chunk_first; /* first chunk of this call (progressive passes: mirt_render_pass_device), 0 otherwise */
    uint32_t norm_samples; /* samples per pixel accumulated once this call is done: the divisor of the normalised output */
    uint32_t nsamples;   /* samples per pixel (1 when anti-aliasing is off) */
    uint32_t k0, k1;     /* RNG key */
    uint32_t tail_items; /* queue tail served in batches of one item per lane (guided self-scheduling) */
    FastDiv d_row, d_pg, d_W, d_strip; /* divisions by W * slots per pixel, W << g_shift, image width and strip height when an item is decoded */
    uint32_t row_items;  /* queue slots per image row: W * (passes << g_shift) */
    uint32_t n_pixels;   /* band pixels */
    uint32_t g_shift;    /* queue order: 2^g_shift consecutive chunks of a pixel sit next to each other (see render_body) */
    uint32_t keep_num;   /* resumable tree traversal: go on while >= keep_num/64 of the traversing lanes are still at it (0: run to the end) */
    uint32_t att_lanes;  /* f64 kernels: lanes of the grid = stride of the per-lane attenuation stacks (0 in the f32 kernels) */
    uint32_t cont_mode;  /* tracers with T::CONTINUES: 0 = off, 1 = this launch hands the paths its last waves are left with to a second one, 2 = the second one (render_body) */
};

/* Order of the floating-point operations that build a pixel (camera.rs:242-246, 277):
 *   f64 (parity mode): the reference's.  color() multiplies a path's attenuations innermost first, a0 (a1 (... (a_k sky))),
 *     so the lane keeps the attenuations of the path in hand in `att_stack` ([bounce][rgb][lane of the grid], HBM) and folds
 *     them back to front when the path reaches the sky; with one sample per work item (the host's choice wherever the sums
 *     fit their budget) K2b then adds the samples of a pixel one after the other like the loop of camera.rs:242-245.
 *   f32 (throughput mode, the f32 spec of DESIGN.md): the running product thr = ((a0 a1) a2) ..., radiance = thr sky, and a
 *     pixel is the sum in chunk order of per-item sums; the oracle's f32 instantiation restates exactly this. */
/* T::LOCKSTEP, where a tracer declares it */
template <class T, class = void> struct tracer_lockstep { static constexpr bool value = false; };
template <class T> struct tracer_lockstep<T, decltype((void)T::LOCKSTEP)> { static constexpr bool value = T::LOCKSTEP; };
/* T::CONTINUES, where a tracer declares it */
template <class T, class = void> struct tracer_continues { static constexpr bool value = false; };
template <class T> struct tracer_continues<T, decltype((void)T::CONTINUES)> { static constexpr bool value = T::CONTINUES; };
#define MIRT_CONT_FIELDS 22u /* dwords of path state per handed-over lane */
#ifndef MIRT_CONT_AFTER
#define MIRT_CONT_AFTER 1u   /* iterations a wave goes on after it has nothing left to deal before it may hand its paths over */
#endif
#ifndef MIRT_CONT_LATE
#define MIRT_CONT_LATE 32u   /* ... and only once all but 1 / MIRT_CONT_LATE of the grid's waves have ended (10 000-sphere 4K frame: 4 / 8 / 16 / 32 / 64: 34.9 / 34.4 / 33.9 / 33.7 / 33.6 ms, 35.2 without; 0 / 1 / 2 iterations first: 35.7 / 35.0 / 35.0) */
#endif
#ifndef MIRT_CONT_MIN
#define MIRT_CONT_MIN 4u     /* ... if more than this many lanes still have one */
#endif
/* tr.tiles, where a tracer counts them */
template <class T, class = void> struct tracer_tiles { static __device__ inline uint32_t get(const T&) { return 0u; } };
template <class T> struct tracer_tiles<T, decltype((void)((T*)nullptr)->tiles)> { static __device__ inline uint32_t get(const T& t) { return t.tiles; } };

template <class T>
__device__ inline void render_body(T& tr, const DevScene<typename T::Real>& sc, const DevCamera<typename T::Real>& cam, const DevPart& part,
                                   const RenderParams& rp, typename T::Real* __restrict__ partial, DevStats* __restrict__ stats, uint32_t lane,
                                   typename T::Real* __restrict__ att_stack = nullptr) {
    typedef typename T::Real R;
    constexpr bool REF_ORDER = sizeof(R) == 8;
    const bool aa = cam.spp > 0;
    const size_t att_lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;

    /* wave-uniform: the reserved range of the queue */
    uint32_t q_next = 0, q_end = 0;
    bool exhausted = false;
    /* per lane */
    bool has_work = false, need_ray = false;
    uint32_t item = 0, pixel = 0, lx = 0, gy = 0, sample = 0, s_end = 0, depth_left = 0, bounce = 0;
    V3<R> acc = {(R)0.0, (R)0.0, (R)0.0}, thr = {(R)1.0, (R)1.0, (R)1.0};
    RayT<R> ray = {};
    uint32_t n_seg = 0, n_narrow = 0, n_iter = 0;
    /* resumable traversal (T::RESUMABLE): a lane's query may span several iterations of the loop below */
    bool traversing = false;
    BvhState trav = {MIRT_BVH_NONE, 0u};
    Closest<R> pbest = {(R)INFINITY, 0xFFFFFFFFu};
    PhaseTimer pt;
    pt.start();
#ifdef MIRT_PHASE_TIMING
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long drain_t0 = 0; /* when this wave first found the queue empty */
    unsigned long long drain_live = 0, drain_iters = 0, drain_c0 = 0; /* lanes with work at that moment; iterations after it */
#endif

    /* One loop iteration = one ray segment for every lane that has a ray, then ONE Philox evaluation in which a lane asks for what
     * it needs next - the uniforms of the scatter at the hit it just found, or those of the camera ray of the sample it is about
     * to start (its last ray went to the sky, or it has just been dealt an item) - never both.  (Rounds 1 and 2 generated camera
     * rays before the segment and scattered after it: two evaluations per iteration, since a wave holds lanes of both kinds.)  A
     * lane whose path ends INSIDE the scatter (absorbed, depth limit) has no uniforms for a camera ray in hand: it sits out the
     * next segment (need_ray) and starts its sample one iteration later - 0.2 % of the lane-iterations of the bench frame.  The
     * counters are those of the reference-order loop, so every sample sees the same uniforms and the image is the same bit for bit. */
    auto deal = [&]() {
        /* ---- deal work items to idle lanes ---------------------------------------------------------- */
        const unsigned long long idle = __ballot(!has_work);
        MIRT_DIAG(sc, DG_DEAL, !has_work && !(exhausted && q_next == q_end));
        if (idle != 0ull && !(exhausted && q_next == q_end)) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            uint32_t got = 0xFFFFFFFFu;
            uint32_t take = q_end - q_next;
            take = take < n_idle ? take : n_idle;
            if (!has_work && rank < take) got = q_next + rank;
            q_next += take;
            if (take < n_idle && !exhausted) { /* reserve the next batch: one atomic per wave per batch */
                /* guided: near the end of the queue (judged by the last head this wave saw) reserve one item per lane
                 * only, so that no wave sits on a backlog while others have run dry.  (Reserving exactly what the idle lanes
                 * take was tried: one atomic per wave and iteration on one address saturates at ~0.4 G/s - +7 % on the
                 * textured 4K frame when the last 3 % of the queue was handed out that way, and nothing gained on a band
                 * once the queue order below keeps a glass pixel's chunks out of one wave.) */
                const uint32_t batch = q_end + rp.tail_items >= rp.n_items ? MIRT_WAVE : MIRT_QUEUE_BATCH;
                uint32_t base = 0;
                MIRT_DIAG(sc, DG_ATOMIC, lane == 0);
                if (lane == 0) base = atomicAdd(&stats->queue_head, batch);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= rp.n_items) {
#ifdef MIRT_PHASE_TIMING
                    if (!exhausted) { drain_t0 = __builtin_amdgcn_s_memrealtime(); drain_c0 = __builtin_amdgcn_s_memtime(); drain_live = (unsigned long long)__popcll(__ballot(has_work)); }
#endif
                    exhausted = true;
                    q_next = q_end = 0;
                } else {
                    q_next = base;
                    q_end = base + batch < rp.n_items ? base + batch : rp.n_items;
                    uint32_t take2 = q_end - q_next;
                    take2 = take2 < n_idle - take ? take2 : n_idle - take;
                    if (!has_work && rank >= take && rank < take + take2) got = q_next + (rank - take);
                    q_next += take2;
                }
            }
            if (got != 0xFFFFFFFFu) {
                /* the queue runs through the band from its LAST pixel to its first: the sky gradient is the only light
                 * and cameras look roughly level, so the cheap one-segment paths (sky, top rows) end up in the tail
                 * of the launch and the long bounce chains at its start */
                /* ... row by row, and within a row in passes of G = 2^g_shift chunks: (row, pass, x, chunk within the pass).
                 * G trades two things: the 64 items a wave takes together should share pixels (rays of one pixel have the
                 * same candidates and the same tree walk: G = all chunks is 4 % faster per segment than G = 1 on the bench
                 * frame), but 64 chunks of ONE pixel on a glass ball are 64 paths to the depth limit that stall the wave and
                 * everything it has reserved (short launches - an 8-GPU band - want a small G).  Rows stay outermost so that
                 * ALL the work of the sky rows comes last, whatever G.  Slots beyond the last chunk (padding of the last
                 * pass) are skipped. */
                const uint32_t rev = rp.n_items - 1u - got;
                const uint32_t ly = fastdiv(rev, rp.d_row), rrow = rev - ly * rp.row_items;
                const uint32_t pass = fastdiv(rrow, rp.d_pg), rem = rrow - pass * (cam.W << rp.g_shift);
                lx = rem >> rp.g_shift;
                const uint32_t ch = (pass << rp.g_shift) + (rem & ((1u << rp.g_shift) - 1u)); /* chunk */
                item = rev; /* partial sums sit in queue order: the lanes of a wave store next to each other */
                gy = ly; /* part_global_row(part, ly) */
                if (part.part_count > 1) {
                    const uint32_t ls = fastdiv(ly, rp.d_strip);
                    gy = (ls * part.part_count + part.part_index) * part.strip_rows + (ly - ls * part.strip_rows);
                }
                pixel = gy * cam.W + lx; /* RNG key: the global pixel, independent of the partition */
                sample = (rp.chunk_first + ch) * rp.chunk;
                s_end = sample + rp.chunk < rp.nsamples ? sample + rp.chunk : rp.nsamples;
                acc = {(R)0.0, (R)0.0, (R)0.0};
                has_work = ch < rp.chunks;
                need_ray = has_work;
            }
        }
    };
    auto start_sample = [&](const Draws<R>& dr) { /* camera ray of (pixel, sample): camera.rs:209-230 / 248-252 */
        thr = {(R)1.0, (R)1.0, (R)1.0};
        depth_left = cam.depth;
        bounce = 0;
        ray = aa ? ray_for(cam, lx, gy, dr) : primary_ray(cam, lx, gy);
        need_ray = false;
    };
    /* ---- continuation pass (T::CONTINUES; scenes beyond the LDS) -----------------------------------------------
     * The end of such a launch is a few waves with a full complement of long paths each (the 64 items of a tail batch are
     * neighbouring pixels, all on one glass sphere: 100 iterations at the 16-40 us a wave of 64 rays takes, while the chip is
     * empty - DESIGN 9.5).  A wave that has nothing left to deal and still holds more than MIRT_CONT_MIN paths after
     * MIRT_CONT_AFTER more iterations writes their state behind the partial sums and ends; a second launch of the same kernel
     * (cont_mode 2) spreads all such paths evenly over its waves - a handful each, at the 8 us per iteration of a wave with few
     * rays.  A path goes on exactly where it stopped (same counters, same sums in the same order): the image is the same. */
    [[maybe_unused]] uint32_t drain_n = 0;
    [[maybe_unused]] R* const cont = partial + (((size_t)rp.n_items * 3u + 63u) & ~(size_t)63u);
    [[maybe_unused]] const uint32_t cont_cap = gridDim.x * blockDim.x;
    if constexpr (tracer_continues<T>::value) {
        if (rp.cont_mode == 2u) {
            const uint32_t count = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&stats->cont_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const uint32_t nw = gridDim.x * (blockDim.x / MIRT_WAVE), gw = blockIdx.x * (blockDim.x / MIRT_WAVE) + (threadIdx.x / MIRT_WAVE);
            uint32_t per = (count + nw - 1u) / nw; /* <= 64: at most one path per lane of the first launch, and the grids are the same */
            per = per < 1u ? 1u : per;
            const uint32_t slot = gw * per + lane;
            if (lane < per && slot < count && count <= cont_cap) {
                auto ld = [&](uint32_t f) { return cont[(size_t)f * cont_cap + slot]; };
                auto ldu = [&](uint32_t f) { return __float_as_uint(ld(f)); };
                item = ldu(0); pixel = ldu(1); lx = ldu(2); gy = ldu(3); sample = ldu(4); s_end = ldu(5); depth_left = ldu(6);
                bounce = ldu(7) & 0x7FFFFFFFu; need_ray = (ldu(7) >> 31) != 0u;
                acc = {ld(8), ld(9), ld(10)}; thr = {ld(11), ld(12), ld(13)};
                ray.o = {ld(14), ld(15), ld(16)}; ray.d = {ld(17), ld(18), ld(19)}; ray.tm = ld(20);
                has_work = true;
            }
            exhausted = true; /* this launch deals nothing */
        }
    }
    deal();
    pt.stamp(PH_QUEUE);
    {
        Draws<R> dr;
        MIRT_DIAG(sc, DG_RAYGEN, need_ray);
        if (need_ray) {
            if (aa) dr.fill(pixel, sample, 0u, rp.k0, rp.k1, true);
            start_sample(dr);
        }
    }
    pt.stamp(PH_RAYGEN);
    for (;;) {
#ifdef MIRT_PHASE_TIMING
        if (exhausted) ++drain_iters;
#endif
        if constexpr (tracer_continues<T>::value) {
            if (rp.cont_mode == 1u && exhausted && q_next == q_end && ++drain_n > MIRT_CONT_AFTER) {
                const unsigned long long lv = __ballot(has_work);
                const uint32_t nl = (uint32_t)__popcll(lv);
                /* ... and only a wave that is LATE: all but 1 / MIRT_CONT_LATE of the grid's waves have ended (where all waves end together - a
                 * depth limit of 1, a short launch - handing over would only move the last iteration or two into a second launch) */
                const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&stats->waves_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (nl > MIRT_CONT_MIN && MIRT_CONT_LATE * done >= (MIRT_CONT_LATE - 1u) * (gridDim.x * (blockDim.x / MIRT_WAVE))) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&stats->cont_count, nl);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (has_work) { /* base + nl <= cont_cap: a lane hands over at most once per launch */
                        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lv, 0u));
                        auto st = [&](uint32_t f, R v) { cont[(size_t)f * cont_cap + slot] = v; };
                        auto stu = [&](uint32_t f, uint32_t v) { st(f, __uint_as_float(v)); };
                        stu(0, item); stu(1, pixel); stu(2, lx); stu(3, gy); stu(4, sample); stu(5, s_end); stu(6, depth_left);
                        stu(7, bounce | (need_ray ? 0x80000000u : 0u));
                        st(8, acc.x); st(9, acc.y); st(10, acc.z); st(11, thr.x); st(12, thr.y); st(13, thr.z);
                        st(14, ray.o.x); st(15, ray.o.y); st(16, ray.o.z); st(17, ray.d.x); st(18, ray.d.y); st(19, ray.d.z); st(20, ray.tm);
                        has_work = false;
                        need_ray = false;
                    }
                }
            }
        }
        /* every lane's path ended inside the last scatter (a depth limit of 1 does that in every iteration): the wave's reserved
         * batch and the queue may still hold items - deal before deciding that the wave is done.  The lanes start their samples
         * after the coming (for them empty) segment like any lane that ends in a scatter. */
        if (__ballot(has_work) == 0ull) deal();
        if constexpr (tracer_lockstep<T>::value) {
            /* the waves of the block leave together: a wave that has run dry keeps calling closest() (for its barriers) */
            if (!__syncthreads_or(__ballot(has_work) != 0ull ? 1 : 0)) break;
            if (__ballot(has_work) != 0ull) ++n_iter;
        } else {
            if (__ballot(has_work) == 0ull) break;
            ++n_iter;
        }
#ifdef MIRT_ACTIVE_HIST
        {
            const uint32_t na_ = (uint32_t)__popcll(__ballot(has_work));
            if (lane == 0) atomicAdd((unsigned long long*)sc.dbg + (na_ <= 2 ? 5 : na_ <= 8 ? 6 : na_ <= 32 ? 7 : 8), 1ull);
        }
#endif
        pt.stamp(PH_QUEUE);
        /* ---- one ray segment for every lane of the wave that has a ray -------------------------------- */
        const bool has_ray = has_work && !need_ray;
        Closest<R> best;
        bool seg_done; /* this lane's closest-hit query completed in this iteration */
        if constexpr (T::RESUMABLE) {
            /* Lanes of a wave need very different numbers of tree steps (2-5x the mean for the slowest), and a step costs
             * the same whether 64 lanes take it or 3.  So a query is suspended as soon as a good share of the wave's
             * lanes have finished theirs: those are shaded, given their next ray (or item), and join the rest again. */
            if (has_ray && !traversing) { /* a fresh ray: from the camera or from the last scatter */
                trav = {sc.bvh_root, 0u};
                pbest = {(R)INFINITY, 0xFFFFFFFFu};
                traversing = true;
            }
            const uint32_t n_trav = (uint32_t)__popcll(__ballot(traversing));
            uint32_t keep = (n_trav * rp.keep_num) >> 6;
            keep = (keep < 1u || (exhausted && q_next == q_end)) ? 1u : keep;
            tr.resume(lane, ray, (R)0.001, pbest, trav, keep, n_narrow, pt);
            seg_done = traversing && trav.cur == MIRT_BVH_NONE;
            traversing = traversing && !seg_done;
            best = pbest;
        } else {
            /* no lane has a ray (all of them ended inside the last scatter): nothing to trace - except that a tracer whose waves
             * move in step must keep its barriers */
            best.t = (R)INFINITY;
            best.idx = 0xFFFFFFFFu;
            if (tracer_lockstep<T>::value || __ballot(has_ray) != 0ull) tr.closest(lane, has_ray, ray, (R)0.001, best, n_narrow, pt);
            seg_done = has_ray;
        }
        /* ---- rays that left the scene: sky (camera.rs:282-284), the sample is done --------------------- */
        auto sample_done = [&](const V3<R>& radiance) {
            acc = acc + radiance;
            ++sample;
            const bool more = sample < s_end; /* regenerate: this lane starts its next sample */
            MIRT_DIAG(sc, DG_STORE, !more);
            if (!more) {
                size_t o = (size_t)item * 3;
                partial[o] = acc.x;
                partial[o + 1] = acc.y;
                partial[o + 2] = acc.z;
            }
            /* plain assignments: a store to "one flag or the other" made the compiler keep both in scratch memory */
            need_ray = need_ray || more;
            has_work = has_work && more;
        };
        /* a hit with one bounce left: color(scattered, 0) is black whatever the scatter does (camera.rs:270, 277-279), so the path
         * ends here - without the scatter and without its uniforms */
        const bool hit_any = seg_done && best.idx != 0xFFFFFFFFu;
        const bool scat = hit_any && depth_left != 1u; /* a hit: this lane scatters below */
        if (seg_done) ++n_seg;
        MIRT_DIAG(sc, DG_HIT, scat);
        MIRT_DIAG(sc, DG_SKY, seg_done && !hit_any);
        if (hit_any && !scat) sample_done(V3<R>{(R)0.0, (R)0.0, (R)0.0});
        if (seg_done && !hit_any) {
            V3<R> ud = unit(ray.d);
            R a = (R)0.5 * (ud.y + (R)1.0);
            V3<R> sky = {(R)1.0 * ((R)1.0 - a) + (R)0.5 * a, (R)1.0 * ((R)1.0 - a) + (R)0.7 * a, (R)1.0 * ((R)1.0 - a) + (R)1.0 * a};
            V3<R> radiance;
            if constexpr (REF_ORDER) {
                radiance = sky;
                for (uint32_t b = bounce; b-- > 0u;) { /* att * color(scattered, depth - 1), innermost first (camera.rs:277) */
                    const size_t o = (size_t)b * 3u * rp.att_lanes + att_lane;
                    radiance = {att_stack[o] * radiance.x, att_stack[o + rp.att_lanes] * radiance.y, att_stack[o + 2u * (size_t)rp.att_lanes] * radiance.z};
                }
            } else {
                radiance = {thr.x * sky.x, thr.y * sky.y, thr.z * sky.z};
            }
            sample_done(radiance);
        }
        pt.stamp(PH_SHADE);
        deal(); /* lanes whose item is finished take the next one (and start its first sample below) */
        pt.stamp(PH_QUEUE);
        /* ---- the iteration's one Philox evaluation ------------------------------------------------------ */
        const bool starts = need_ray && !scat; /* need_ray of a lane that scatters is set below: that lane waits an iteration */
        Draws<R> dr;
        if (scat || (starts && aa)) dr.fill(pixel, sample, scat ? bounce + 1u : 0u, rp.k0, rp.k1, !scat);
        if (scat) {
            Hit<R> hit = tr.hit(best, ray);
            ++bounce;
            V3<R> att, nxt_d;
            bool done;
            /* the scattered ray (material.rs:81, 111, 151): from the hit point, at the incoming ray's time - origin and time are
             * settled here, for every lane that scatters (a path that ends in the scatter has no use for its ray any more), so that
             * only the direction travels through the material branches */
            const V3<R> in_d = ray.d;
            ray.o = hit.p;
            if (scatter(sc, tr.material_record(best.idx), in_d, hit, dr, att, nxt_d)) {
                if constexpr (REF_ORDER) {
                    const size_t o = (size_t)(bounce - 1u) * 3u * rp.att_lanes + att_lane;
                    att_stack[o] = att.x;
                    att_stack[o + rp.att_lanes] = att.y;
                    att_stack[o + 2u * (size_t)rp.att_lanes] = att.z;
                } else {
                    thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
                }
                ray.d = nxt_d;
                --depth_left;
                done = depth_left == 0; /* color(.., 0) = black */
            } else {
                done = true; /* absorbed: black for the whole path (camera.rs:279) */
            }
            if (done) sample_done(V3<R>{(R)0.0, (R)0.0, (R)0.0});
        }
        pt.stamp(PH_SHADE);
        MIRT_DIAG(sc, DG_RAYGEN, starts);
        if (starts) start_sample(dr);
        pt.stamp(PH_RAYGEN);
    }
#ifdef MIRT_PHASE_TIMING
    if (lane == 0 && rp.cont_mode != 2u) { /* (the continuation launch is left out of the diagnostic: its waves have no queue phase) */
        const unsigned long long wave_t1 = __builtin_amdgcn_s_memrealtime(); /* before the atomics below: 4096 waves ending together queue up on them */
        const unsigned long long wave_c1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 6; ++i) atomicAdd(&stats->phase_cycles[i], pt.acc[i]);
        atomicMax(&stats->inv_start_max, ~wave_t0);
        atomicMax(&stats->end_max, wave_t1);
        atomicAdd(&stats->life_sum, wave_t1 - wave_t0);
        atomicAdd(&stats->start_sum, wave_t0 & 0xFFFFFFFFFFull);
        atomicAdd(&stats->idle_sum, wave_t1 - drain_t0);
        {
            const unsigned long long d_ = wave_t1 - drain_t0; /* 100 MHz ticks: 5000 = 50 us */
            const int b_ = d_ < 5000 ? 0 : d_ < 10000 ? 1 : d_ < 20000 ? 2 : d_ < 40000 ? 3 : d_ < 80000 ? 4 : 5;
            atomicAdd(&stats->diag[40 + b_], 1ull);
            atomicAdd(&stats->diag[46], drain_live);
            atomicAdd(&stats->diag[47], drain_iters);
            atomicAdd(&stats->diag[39], wave_c1 - drain_c0);
#ifndef MIRT_BLOCK_COUNTS
            { /* how long before the longest-lived wave (of the previous, identical launch) this one ended: 25 us buckets */
                const unsigned long long life_ = wave_t1 - wave_t0, ref_ = g_prev_life_max;
                const unsigned long long gap_ = ref_ > life_ ? ref_ - life_ : 0ull;
                atomicAdd(&stats->diag[gap_ / 2500ull < 38ull ? gap_ / 2500ull : 38ull], 1ull);
                atomicMax(&g_prev_life_max, life_);
            }
#endif
        }
        atomicAdd(&stats->n_waves, 1ull);
    }
#endif
    if constexpr (tracer_continues<T>::value) {
        if (rp.cont_mode == 1u && lane == 0) atomicAdd(&stats->waves_done, 1u);
        /* a wave of the second launch that was handed nothing has nothing to add - and 4 096 waves ending at the same moment
         * would queue up on the counters below for 150 us */
        if (n_iter == 0u) return;
    }
    uint32_t s1 = wave_sum_u(n_seg), s2 = wave_sum_u(n_narrow);
    if (lane == 0) {
        atomicAdd(&stats->segments, (unsigned long long)s1);
        atomicAdd(&stats->narrow, (unsigned long long)s2);
        atomicAdd(&stats->wave_iters, (unsigned long long)n_iter);
        const uint32_t nt = tracer_tiles<T>::get(tr);
        if (nt) atomicAdd(&stats->mx_tiles, (unsigned long long)nt);
    }
}

/* Camera, partition and launch parameters are read once per loop iteration (item decode, ray generation) by code that sits
 * between the register-hungry phases.  As kernel arguments they live in SGPRs for the whole launch and most of them end up
 * spilled to VGPR lanes (62 SGPRs in render_kernel_f32_mx, one v_readlane per use: 275 of them in the kernel).  The render
 * kernels copy them to LDS once per block and the loop reads that copy (bench frame 15.62 -> 15.20 ms). */
template <class R> struct LaunchParams {
    DevCamera<R> cam;
    DevPart part;
    RenderParams rp;
};
/* thread 0 writes; the caller's next block barrier publishes */
template <class R>
__device__ inline void stage_params(LaunchParams<R>* P, const DevCamera<R>& cam, const DevPart& part, const RenderParams& rp) {
    if (threadIdx.x == 0) {
        P->cam = cam;
        P->part = part;
        P->rp = rp;
    }
}

template <int MIN_WAVES>
__global__ __launch_bounds__(MIRT_BLOCK, MIN_WAVES) void render_kernel_f64(DevScene<double> sc, DevCamera<double> cam, DevPart part, RenderParams rp,
                                                                            double* __restrict__ partial, DevStats* __restrict__ stats, double* __restrict__ att) {
    __shared__ WaveLds<double> lds_all[MIRT_WAVES_PER_BLOCK];
    __shared__ LaunchParams<double> P;
    stage_params(&P, cam, part, rp);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF64 tr{sc, lds_all[w]};
    /* measured: the LDS copy pays where an iteration is short (a handful of spheres: textured 4K f64 17.98 -> 17.56 ms), not in
     * the all-f64 loop over hundreds of spheres (115.3 -> 116.2 ms: the compiler keeps the copy in VGPRs there) */
    if (sc.n_live <= MIRT_TINY_SCENE) render_body(tr, sc, P.cam, P.part, P.rp, partial, stats, lane, att);
    else render_body(tr, sc, cam, part, rp, partial, stats, lane, att);
}
#ifndef MIRT_F64MX_BLOCKS
#define MIRT_F64MX_BLOCKS 3 /* 4-wave blocks per CU the register budget is set for: 168 VGPRs, 46-66 spilled; measured 2 (256 VGPRs, no spills) / 3 / 4 (128): bench frame 23.9 / 22.5 / 40.5 ms, 10 000 spheres 42.9 / 36.6 / 84.4 ms (tools/f64_blocks_ab.sh) */
#endif
/* parity mode with the broad phase on the matrix pipe: f32 filter, exact f64 narrow phase and shading.  RES: the operands are
 * resident (<= MIRT_MX64_RECORDS sorted slots), else every wave streams its tiles; the host picks (launch_render) */
template <bool RES>
__global__ __launch_bounds__(MIRT_BLOCK, MIRT_F64MX_BLOCKS) void render_kernel_f64_mx(DevScene<double> sc, DevCamera<double> cam, DevPart part, RenderParams rp,
                                                                      double* __restrict__ partial, DevStats* __restrict__ stats, double* __restrict__ att) {
    __shared__ MxLds64 L;
    __shared__ LaunchParams<double> P;
    stage_params(&P, cam, part, rp);
    const bool res = L.fill(sc); /* ends with the block barrier */
    if (res != RES) return;      /* never: the host launches <true> exactly for the scenes fill() makes resident */
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF64Mx<RES> tr{sc, L, w};
    render_body(tr, sc, P.cam, P.part, P.rp, partial, stats, lane, att);
}
/* MIRT_FLAG_BVH: latency-bound per-lane traversal, small blocks, as many waves as the registers allow */
__global__ __launch_bounds__(MIRT_BLOCK) void render_kernel_f32_bvh(DevScene<float> sc, DevCamera<float> cam, DevPart part, RenderParams rp,
                                                                    float* __restrict__ partial, DevStats* __restrict__ stats) {
    __shared__ uint32_t stk[MIRT_WAVES_PER_BLOCK][MIRT_BVH_STACK][MIRT_WAVE];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF32Bvh<BvhGlobalStore, true> tr{sc, BvhGlobalStore{sc}, stk[w]};
    /* no LDS copy of the parameters here (stage_params): it costs this kernel 42 VGPRs, i.e. two of its five waves per SIMD
     * (stress 10 000: 58.1 -> 66.4 ms) */
    render_body(tr, sc, cam, part, rp, partial, stats, lane);
}
/* the same for scenes whose tree, records and spheres fit the LDS: one block of 16 waves per CU, no HBM in the loop */
__global__ __launch_bounds__(MIRT_F32_BLOCK) void render_kernel_f32_bvh_lds(DevScene<float> sc, DevCamera<float> cam, DevPart part, RenderParams rp,
                                                                            float* __restrict__ partial, DevStats* __restrict__ stats) {
    __shared__ BvhLds L;
    __shared__ LaunchParams<float> P;
    stage_params(&P, cam, part, rp);
    L.fill(sc); /* ends with the block barrier */
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF32Bvh<BvhLdsStore, true> tr{sc, BvhLdsStore{L}, L.stk[w]};
    render_body(tr, sc, P.cam, P.part, P.rp, partial, stats, lane);
}
/* one block of 16 waves per CU: 4 waves per SIMD, <= 128 VGPRs */
__global__ __launch_bounds__(MIRT_F32_BLOCK, MIRT_F32_EU) void render_kernel_f32(DevScene<float> sc, DevCamera<float> cam, DevPart part, RenderParams rp,
                                                                    float* __restrict__ partial, DevStats* __restrict__ stats) {
    __shared__ F32Lds L;
    const bool res = L.fill(sc);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    /* no LDS copy of the parameters in this kernel (stage_params): 29.1 -> 29.5 ms on the bench frame, the copy ends up in
     * VGPRs the tile loop wants; scenes of a few spheres, where it pays, have a kernel of their own below */
    if (res) {
        TracerF32<true> tr{sc, L, w};
        render_body(tr, sc, cam, part, rp, partial, stats, lane);
    } else {
        TracerF32<false> tr{sc, L, w};
        render_body(tr, sc, cam, part, rp, partial, stats, lane);
    }
}
/* scenes of <= MIRT_TINY_SCENE sorted slots (chosen by the host): TracerF32's exact-test-on-every-sphere path with short loop
 * iterations, where reading the parameters from LDS instead of spilled SGPRs is worth 5 % (textured 4K 7.84 -> 7.41 ms) */
__global__ __launch_bounds__(MIRT_F32_BLOCK, MIRT_F32_EU) void render_kernel_f32_tiny(DevScene<float> sc, DevCamera<float> cam, DevPart part, RenderParams rp,
                                                                         float* __restrict__ partial, DevStats* __restrict__ stats) {
    __shared__ F32Lds L;
    __shared__ LaunchParams<float> P;
    stage_params(&P, cam, part, rp);
    L.fill(sc); /* ends with the block barrier; a scene this small is always resident */
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    TracerF32<true, true> tr{sc, L, w};
    render_body(tr, sc, P.cam, P.part, P.rp, partial, stats, lane);
}

/* the same with the broad phase on the matrix pipe (scenes whose coordinates fit the split-f16 operands); LDS = MxLdsSmall
 * for scenes of up to MIRT_MX_SMALL_RECORDS sorted slots, MxLds otherwise; RES: the scene is resident in the LDS layout, else
 * every wave streams its tiles from L1/L2 (tile cull, scenes beyond the LDS).  The host picks the instantiation (launch_render);
 * one tracer per kernel keeps the streaming loop's registers and code out of the resident kernels. */
template <class LDS, bool RES>
__global__ __launch_bounds__(MIRT_F32_BLOCK, MIRT_F32_EU) void render_kernel_f32_mx(DevScene<float> sc, DevCamera<float> cam, DevPart part, RenderParams rp,
                                                                       float* __restrict__ partial, DevStats* __restrict__ stats) {
    __shared__ LDS L;
    /* the LDS of this kernel is full: the parameter copy (stage_params) has its place inside the layout */
    static_assert(sizeof(LaunchParams<float>) <= sizeof(L.params), "MxLds::params too small");
    LaunchParams<float>* P = reinterpret_cast<LaunchParams<float>*>(L.params);
    stage_params(P, cam, part, rp);
    const bool res = L.fill(sc); /* ends with the block barrier */
    if (res != RES) return;      /* never: the host launches <.., true> exactly for the scenes fill() makes resident */
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (sc.n_materials <= LDS::N_MATS) {
        TracerF32Mx<RES, LDS, true> tr{sc, L, w};
        render_body(tr, sc, P->cam, P->part, P->rp, partial, stats, lane);
    } else {
        TracerF32Mx<RES, LDS, false> tr{sc, L, w};
        render_body(tr, sc, P->cam, P->part, P->rp, partial, stats, lane);
    }
}

/* the same for scenes of more than MIRT_LDS_RECORDS sorted slots: operands through the block-wide LDS ring */
__global__ __launch_bounds__(MIRT_F32_BLOCK, MIRT_F32_EU) void render_kernel_f32_mx_ring(DevScene<float> sc, DevCamera<float> cam, DevPart part, RenderParams rp,
                                                                            float* __restrict__ partial, DevStats* __restrict__ stats) {
    __shared__ MxLds L;
    LaunchParams<float>* P = reinterpret_cast<LaunchParams<float>*>(L.params);
    stage_params(P, cam, part, rp);
    (void)L.fill(sc); /* materials only (the scene is not resident); ends with the block barrier */
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (sc.n_materials <= MxLds::N_MATS) {
        TracerF32MxRing<MxLds, true> tr{sc, L, w};
        render_body(tr, sc, P->cam, P->part, P->rp, partial, stats, lane);
    } else {
        TracerF32MxRing<MxLds, false> tr{sc, L, w};
        render_body(tr, sc, P->cam, P->part, P->rp, partial, stats, lane);
    }
}

/* ---- K2b (+K3): add the chunks of a pixel in chunk order, scale by 1/spp (camera.rs:246), gamma + quantise
 *      (rgb.rs:20-36).  One wave per 64 consecutive pixels of a band row, one lane per pixel.  The partial sums sit in
 *      queue order - per row and pass, pixel by pixel, a pixel's chunks next to each other - so a lane that walked its
 *      own pixel's chunks would touch a different cache line than its neighbours with every load.  Instead the wave
 *      copies MIRT_RESOLVE_TC chunks of its 64 pixels at a time into LDS with consecutive lanes on consecutive words,
 *      and each lane adds its pixel's chunks from there, in chunk order as before (bit-identical sums). */
/* chunks of a pixel per stage, a template parameter picked by the host: 16 where a pass holds that many (bench frame, 32 chunks:
 * 0.234 -> 0.208 ms against 8; 32 at two waves per block: 0.314), 8 otherwise (the 4K frames have 8 chunks: with the larger tile
 * half empty and half as many blocks per CU their resolve went from 0.175 to 0.236 ms) */
#define MIRT_RESOLVE_TC_WIDE 16u
#define MIRT_RESOLVE_TC 8u
#ifndef MIRT_RESOLVE_WAVES
#define MIRT_RESOLVE_WAVES 4u
#endif
template <class R, uint32_t TC_>
__global__ __launch_bounds__(MIRT_RESOLVE_WAVES * MIRT_WAVE) void resolve_kernel(const R* __restrict__ partial, DevCamera<R> cam, DevPart part, RenderParams rp,
                                                                                   uint32_t n_pixels, R* __restrict__ out_lin, uint8_t* __restrict__ out_rgb8,
                                                                                   DevStats* __restrict__ stats, R* accum) {
    constexpr uint32_t TC = TC_, STRIDE = TC * 3u + 1u; /* odd stride: the lanes' rows start in different banks */
    __shared__ R tile_all[MIRT_RESOLVE_WAVES][MIRT_WAVE * STRIDE];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    R* tile = tile_all[w];
    const uint32_t segs = (cam.W + 63u) / 64u;
    const uint32_t wave_id = blockIdx.x * MIRT_RESOLVE_WAVES + w;
    const uint32_t ly = wave_id / segs, lx0 = (wave_id - ly * segs) * 64u;
    /* waves beyond the band take part in the barriers with no pixels (the trip counts below are the same for every wave) */
    const uint32_t nvalid = ly < part.local_rows ? (cam.W - lx0 < 64u ? cam.W - lx0 : 64u) : 0u;
    const bool valid = lane < nvalid;
    const uint32_t lx = lx0 + lane;
    const uint32_t row = (part.compact || !valid) ? ly : part_global_row(part, ly);
    const size_t o = ((size_t)row * cam.W + lx) * 3;
    /* progressive passes carry the running sum of the earlier chunks in `accum` (same layout as the output) and go on adding
     * in chunk order, so the sum after the last pass is the one-call sum bit for bit */
    V3<R> acc = {(R)0.0, (R)0.0, (R)0.0};
    if (valid && accum && rp.chunk_first != 0u) acc = {accum[o], accum[o + 1], accum[o + 2]};
    /* queue order (render_body): [row][pass][x][chunk within the pass]; summed in chunk order whatever the pass size */
    const uint32_t G = 1u << rp.g_shift;
    const R* p = partial + ((size_t)(nvalid ? ly : 0u) * rp.row_items + ((size_t)(nvalid ? lx0 : 0u) << rp.g_shift)) * 3;
    for (uint32_t c0 = 0; c0 < rp.chunks; c0 += G, p += ((size_t)cam.W << rp.g_shift) * 3) {
        const uint32_t n = rp.chunks - c0 < G ? rp.chunks - c0 : G;
        for (uint32_t c1 = 0; c1 < n; c1 += TC) {
            const uint32_t m = n - c1 < TC ? n - c1 : TC, m3 = m * 3u, count = nvalid * m3;
            __syncthreads(); /* the adds of the previous group are done with the tile */
            /* groups of 16, 8 or 4 f32 sums: 16-byte loads (M chunks of a pixel are 3 M / 4 aligned float4: slot numbers are
             * multiples of G >= 4 here, c1 of TC, and the workspace is a hipMalloc block) */
            auto copy16 = [&](auto mc) {
                constexpr uint32_t Q4 = decltype(mc)::value * 3u / 4u; /* float4 per pixel */
                for (uint32_t e = lane; e < nvalid * Q4; e += MIRT_WAVE) {
                    const uint32_t j = e / Q4, r = (e - j * Q4) * 4u;
                    const float4 v = *reinterpret_cast<const float4*>(p + ((size_t)j * G + c1) * 3 + r);
                    R* d = tile + j * STRIDE + r;
                    d[0] = (R)v.x; d[1] = (R)v.y; d[2] = (R)v.z; d[3] = (R)v.w;
                }
            };
            const bool wide = sizeof(R) == 4 && rp.g_shift >= 2;
            if (wide && TC >= 16u && m == 16u) copy16(std::integral_constant<uint32_t, 16u>{});
            else if (wide && m == 8u) copy16(std::integral_constant<uint32_t, 8u>{});
            else if (wide && m == 4u) copy16(std::integral_constant<uint32_t, 4u>{});
            else {
                for (uint32_t e = lane; e < count; e += MIRT_WAVE) {
                    const uint32_t j = m == TC ? e / (TC * 3u) : e / m3, r = e - j * m3;
                    tile[j * STRIDE + r] = p[((size_t)j * G + c1) * 3 + r];
                }
            }
            __syncthreads();
            if (valid)
                for (uint32_t k = 0; k < m; ++k) acc = acc + V3<R>{tile[lane * STRIDE + 3 * k], tile[lane * STRIDE + 3 * k + 1], tile[lane * STRIDE + 3 * k + 2]};
        }
    }
    if (!valid) return;
    (void)n_pixels;
    if (accum) {
        accum[o] = acc.x;
        accum[o + 1] = acc.y;
        accum[o + 2] = acc.z;
    }
    if (cam.spp > 0) acc = acc * ((R)1.0 / (R)rp.norm_samples); /* samples_scale, camera.rs:50,246 (norm_samples == spp unless this is an early pass) */
    if (out_lin) {
        out_lin[o] = acc.x;
        out_lin[o + 1] = acc.y;
        out_lin[o + 2] = acc.z;
    }
    uint32_t nan = 0;
    uint32_t r = quantize1(acc.x, nan), g = quantize1(acc.y, nan), b = quantize1(acc.z, nan);
    if (out_rgb8) {
        out_rgb8[o] = (uint8_t)r;
        out_rgb8[o + 1] = (uint8_t)g;
        out_rgb8[o + 2] = (uint8_t)b;
    }
    if (nan && stats) atomicAdd(&stats->nan_pixels, 1ull);
}

/* ---- K3 ------------------------------------------------------------------------------------------ */
template <class R> __global__ void quantize_kernel(const R* __restrict__ lin, size_t n, uint8_t* __restrict__ out, unsigned long long* nan_count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t nan = 0;
    out[i] = (uint8_t)quantize1(lin[i], nan);
    if (nan && nan_count) atomicAdd(nan_count, 1ull);
}

/* ---- measurement: the FP32 vector peak of the box (mirt_measure_fma_peak) --------------------------------------- */
__global__ __launch_bounds__(256) void fma_peak_kernel(float* out, float a, float b, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i);
    /* both constants in VGPRs: a VALU instruction with an SGPR source issues at half rate on gfx950 (bench/sgpr_rate.hip) */
    asm volatile("" : "+v"(a), "+v"(b));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 123.456f) out[0] = s; /* keeps the chains alive */
}

/* ---- K4: gathered bands -> frame.  One thread per unit of 16, 4 or 1 bytes: the widest that divides a row and the band
 *      stride and matches the pointers' alignment (chosen by the host). ------------------------------------------------ */
template <class U>
__global__ void deinterleave_kernel(const U* __restrict__ bands, size_t band_stride, uint32_t part_count, uint32_t strip_rows,
                                    size_t row_units, uint32_t height, U* __restrict__ frame) { /* band_stride and row_units in units of U */
    size_t total = row_units * height;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= total) return;
    uint32_t gy = (uint32_t)(i / row_units);
    size_t in_row = i - (size_t)gy * row_units;
    uint32_t strip = gy / strip_rows, within = gy - strip * strip_rows;
    uint32_t part = strip % part_count, ls = strip / part_count;
    /* rows of earlier local strips are all full strips (only the last global strip can be short) */
    size_t lr = (size_t)ls * strip_rows + within;
    frame[i] = bands[(size_t)part * band_stride + lr * row_units + in_row];
}

} /* namespace mirt */
#endif
