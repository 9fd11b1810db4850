/*
 * mirt_api.hip — the C ABI of include/mirt.h on top of the HIP kernels (mirt_kernels.h).
 * Host code only flattens, uploads, launches and times; there is no CPU rendering path in here.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "mirt_kernels.h"

using namespace mirt;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
void mirt_internal_set_error(const char* msg) { g_err = msg ? msg : ""; } /* for mirt_node.hip */
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail(MIRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

/* ---- device-resident scene ---------------------------------------------------------------------- */
template <class R> struct SceneBuffers {
    typename Vec4<R>::type *A = nullptr, *B = nullptr;
    DevMaterial<R>* mats = nullptr;
};
struct MxBall { double M[3]; double rho; bool real; bool big = false; /* real, but left out of the filter (tested directly by every ray): counts for the operand range only */ };
struct mirt_scene {
    int device = 0;
    uint32_t n = 0, n_static = 0;           /* padded counts (multiples of MIRT_GROUP) seen by the kernels */
    uint32_t n_live = 0;                    /* n before the padding to whole 32-sphere tiles */
    uint32_t true_n = 0, true_static = 0;   /* real sphere counts, for the statistics */
    uint32_t n_materials = 0;
    SceneBuffers<float> f32;
    SceneBuffers<double> f64;
    uint32_t *orig_id = nullptr, *material = nullptr, *texels = nullptr;
    float4* S = nullptr; /* f32 conservative broad-phase records */
    BvhNode* bvh = nullptr; /* bounding-ball tree (MIRT_FLAG_BVH) */
    float4* bvh_S = nullptr;
    uint32_t* bvh_idx = nullptr;
    uint32_t bvh_root = MIRT_BVH_NONE, bvh_n_nodes = 0, bvh_n_recs = 0, bvh_depth = 0;
    bool bvh_ok = false;
    /* matrix-pipe broad phase (trace_mx): the bounding balls on the host, and split-f16 sphere operands on the device for
     * every power-of-two scale a call has asked for so far (the scale depends on the camera position, see mx_pick) */
    struct MxSet {
        std::vector<MxBall> balls; /* one per sorted slot; empty when the scene does not qualify */
        double R = 0;              /* max (|M| + rho) */
        mutable std::vector<std::pair<int, uint4*>> cache;
    };
    MxSet mx32, mx64;             /* balls around the f32 spheres (what the f32 kernels test) and around the f64 ones */
    mutable uint4* SM = nullptr;  /* operands and scale of the call being launched */
    mutable float mx_scale = 1.f;
    uint32_t n_big = 0, big_idx[4] = {0, 0, 0, 0}; /* sorted slots left out of the matrix-pipe filter, tested directly (mirt_scene_create) */
    float mx_drift[3] = {0.f, 0.f, 0.f}; /* common part of the shutter motion taken out of the matrix-pipe filter's balls (mirt_scene_create) */
    mutable uint32_t mx_cull = 0; /* this call may skip tiles by the top tile (resident scenes, not MIRT_FLAG_ALL_PAIRS) */
    DevStats* stats = nullptr; /* one per scene: launches on one scene handle are ordered one after the other (ev_busy) */
    mutable std::vector<void*> allocs; /* everything freed with the scene (mutable: operand copies are made on first use) */
    /* render workspace (per-item partial sums; f64: per-lane attenuation stacks), grown on demand and kept until the scene
     * is destroyed */
    mutable void* partial = nullptr;
    mutable size_t partial_bytes = 0;
    mutable void* att = nullptr;
    mutable size_t att_bytes = 0;
    /* A handle owns ONE work queue, ONE set of counters and ONE workspace.  Every render launch on the handle records
     * ev_busy behind its kernels and the next launch - on whatever stream - waits for it first, so two calls on one handle
     * never run side by side (callers that want frames in flight use one handle per frame in flight).  `lock` serialises
     * host threads that share a handle: the _device entry points hold it for the duration of the enqueue, the host-buffer
     * ones (mirt_render, mirt_render_pass: they grow, clear and copy back the handle's staging frames) and mirt_render_wait
     * for the whole call.  Recursive, because the host-buffer calls go through the _device ones. */
    mutable hipEvent_t ev_busy = nullptr;
    mutable bool launched = false;
    mutable std::recursive_mutex lock;
    /* device frames behind the host-buffer entry points (mirt_render / mirt_render_pass), grown on demand like `partial` */
    mutable void* stage[3] = {nullptr, nullptr, nullptr};
    mutable size_t stage_bytes[3] = {0, 0, 0};
    mutable hipEvent_t ev0 = nullptr, ev1 = nullptr; /* timing events of render calls that ask for statistics (created with the scene) */
    mutable hipEvent_t ev_mid = nullptr;             /* between the render kernel and the resolve kernel (mirt_stats.resolve_ms) */
    /* MIRT_FLAG_DEFER_STATS: the counters land in pinned host memory behind the launch, mirt_render_wait reads them */
    mutable hipEvent_t ev2 = nullptr;
    mutable DevStats* host_stats = nullptr;
    mutable bool pending = false;
    mutable uint32_t pending_flags = 0;
    int n_cus = 256;
};

template <class T> static int upload(mirt_scene* s, const std::vector<T>& host, T** dev) {
    size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dev, bytes));
    s->allocs.push_back(*dev);
    if (!host.empty()) HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return MIRT_OK;
}


/* ---- bounding-volume tree (MIRT_FLAG_BVH) -------------------------------------------------------------
 * Leaves are runs of <= 4 bounding records (the same (M, w) records the brute-force broad phase tests, copied in leaf
 * order); an inner node holds the axis-aligned box of each child's bounding balls, widened by the constant part of
 * the rounding slack of the leaf test (trace_bvh has the derivation).  Built top-down with a binned surface-area
 * heuristic on the ball centres; when the remaining stack depth would not fit a balanced subtree, the split falls
 * back to the median, so the depth never exceeds MIRT_BVH_STACK - which is what the traversal stack holds. */
struct BallRec {
    double M[3];
    double rho;
    uint32_t idx; /* sorted sphere index */
};
struct BvhBox {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, A = 0; /* A = max (|M| + rho) */
    void add(const BallRec& b) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.M[a] - b.rho); hi[a] = std::max(hi[a], b.M[a] + b.rho); }
        A = std::max(A, std::sqrt(b.M[0] * b.M[0] + b.M[1] * b.M[1] + b.M[2] * b.M[2]) + b.rho);
    }
    void add(const BvhBox& o) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); }
        A = std::max(A, o.A);
    }
    double area() const {
        double d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        return d[0] < 0 ? 0.0 : 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};
struct BvhBuild {
    std::vector<BallRec>& b;
    std::vector<BvhNode> nodes;
    std::vector<float4> leafS;
    std::vector<uint32_t> leafIdx;
    const std::vector<float4>& S;
    uint32_t leaf_size;
    uint32_t depth_limit; /* stack entries the traversal has */
    uint32_t depth_max = 0;

    static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
    static float down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -INFINITY); return f; }
    static float up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, INFINITY); return f; }
    /* widened box of b[lo, hi) as the kernel sees it */
    void child_box(size_t lo, size_t hi, float o_lo[3], float o_hi[3]) const {
        BvhBox bx;
        for (size_t i = lo; i < hi; ++i) bx.add(b[i]);
        const double wd = (double)MIRT_BVH_S * bx.A * (1.0 + 1e-6);
        for (int a = 0; a < 3; ++a) { o_lo[a] = down(bx.lo[a] - wd); o_hi[a] = up(bx.hi[a] + wd); }
    }
    static uint32_t levels_for(size_t n, uint32_t leaf) { /* depth of a median-split subtree over n records */
        uint32_t l = 0;
        while (n > leaf) { n = (n + 1) / 2; ++l; }
        return l;
    }
    /* builds the subtree of b[lo, hi), returns its ref */
    uint32_t build(size_t lo, size_t hi, uint32_t depth) {
        depth_max = std::max(depth_max, depth);
        const size_t n = hi - lo;
        if (n <= leaf_size) {
            uint32_t first = (uint32_t)leafS.size();
            for (size_t i = lo; i < hi; ++i) { leafS.push_back(S[b[i].idx]); leafIdx.push_back(b[i].idx); }
            return 0x80000000u | (first << 3) | (uint32_t)(n - 1);
        }
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b[i].M[a]); mx[a] = std::max(mx[a], b[i].M[a]); }
        size_t mid = lo + n / 2;
        int ax = 0;
        for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
        bool median = depth + 1 + levels_for(n, leaf_size) >= depth_limit; /* no room left for an unbalanced split */
        if (!median) { /* binned SAH over the three axes */
            const int NB = 16;
            double best_cost = 1e300;
            int best_ax = -1, best_bin = 0;
            for (int a = 0; a < 3; ++a) {
                const double ext = mx[a] - mn[a];
                if (!(ext > 0)) continue;
                BvhBox bins[NB];
                size_t cnt[NB] = {0};
                for (size_t i = lo; i < hi; ++i) {
                    int k = (int)((b[i].M[a] - mn[a]) / ext * NB);
                    k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                    bins[k].add(b[i]);
                    ++cnt[k];
                }
                BvhBox left[NB], right[NB];
                size_t cl[NB], cr[NB];
                BvhBox acc;
                size_t c = 0;
                for (int k = 0; k < NB; ++k) { if (cnt[k]) acc.add(bins[k]); c += cnt[k]; left[k] = acc; cl[k] = c; }
                acc = BvhBox();
                c = 0;
                for (int k = NB - 1; k >= 0; --k) { if (cnt[k]) acc.add(bins[k]); c += cnt[k]; right[k] = acc; cr[k] = c; }
                for (int k = 0; k + 1 < NB; ++k) {
                    if (cl[k] == 0 || cr[k + 1] == 0) continue;
                    double cost = left[k].area() * (double)cl[k] + right[k + 1].area() * (double)cr[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_ax = a; best_bin = k; }
                }
            }
            if (best_ax < 0) {
                median = true; /* all centres coincide */
            } else {
                const double ext = mx[best_ax] - mn[best_ax], m0 = mn[best_ax];
                const int a = best_ax, kb = best_bin;
                auto it = std::partition(b.begin() + lo, b.begin() + hi, [=](const BallRec& x) {
                    int k = (int)((x.M[a] - m0) / ext * 16);
                    k = k < 0 ? 0 : (k >= 16 ? 15 : k);
                    return k <= kb;
                });
                mid = (size_t)(it - b.begin());
                if (mid == lo || mid == hi) median = true;
            }
        }
        if (median) {
            mid = lo + n / 2;
            std::nth_element(b.begin() + lo, b.begin() + mid, b.begin() + hi, [ax](const BallRec& x, const BallRec& y) {
                return x.M[ax] < y.M[ax] || (x.M[ax] == y.M[ax] && x.idx < y.idx);
            });
        }
        uint32_t me = (uint32_t)nodes.size();
        nodes.push_back(BvhNode{});
        uint32_t l = build(lo, mid, depth + 1), r = build(mid, hi, depth + 1);
        float l0[3], h0[3], l1[3], h1[3];
        child_box(lo, mid, l0, h0);
        child_box(mid, hi, l1, h1);
        BvhNode nd;
        nd.q0 = float4{l0[0], l0[1], l0[2], h0[0]};
        nd.q1 = float4{h0[1], h0[2], l1[0], l1[1]};
        nd.q2 = float4{l1[2], h1[0], h1[1], h1[2]};
        nd.q3 = float4{u2f(l), u2f(r), 0.f, 0.f};
        nodes[me] = nd;
        return me;
    }
};
static int build_bvh(mirt_scene* s, std::vector<BallRec>& balls, const std::vector<float4>& S) {
    s->bvh_ok = false;
    s->bvh_root = MIRT_BVH_NONE;
    if (balls.empty()) { s->bvh_ok = true; return MIRT_OK; } /* empty scene: every ray misses */
    if (balls.size() >= (1u << 27)) return MIRT_OK;          /* leaf refs hold 28 bits of record index: no tree, flag rejected */
    /* scenes that fit the LDS copy (render_kernel_f32_bvh_lds) get a tree for its shorter stack */
    const bool small = balls.size() <= MIRT_LDS_RECORDS;
    BvhBuild bb{balls, {}, {}, {}, S, 4, small ? (uint32_t)MIRT_BVH_LDS_STACK : (uint32_t)MIRT_BVH_STACK};
    uint32_t root = bb.build(0, balls.size(), 0);
    if (bb.depth_max > MIRT_BVH_STACK) return MIRT_OK; /* cannot happen below 2^27 records (median fallback), kept as a guard */
    for (int k = 0; k < 8; ++k) bb.leafS.push_back(float4{0.f, 0.f, 0.f, INFINITY}); /* leaf reads fetch four records at a time */
    int rc;
    if ((rc = upload(s, bb.nodes, &s->bvh)) || (rc = upload(s, bb.leafS, &s->bvh_S)) || (rc = upload(s, bb.leafIdx, &s->bvh_idx))) return rc;
    s->bvh_root = root;
    s->bvh_n_nodes = (uint32_t)bb.nodes.size();
    s->bvh_n_recs = (uint32_t)bb.leafIdx.size();
    s->bvh_depth = bb.depth_max;
    s->bvh_ok = true;
    return MIRT_OK;
}

template <class R>
static int upload_real(mirt_scene* s, const mirt_scene_desc* d, const std::vector<uint32_t>& order, const std::vector<DevImage>& imgs,
                       SceneBuffers<R>& out) {
    typedef typename Vec4<R>::type R4;
    std::vector<R4> A(order.size()), B(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        if (order[k] == 0xFFFFFFFFu) { /* padding record: r*r = -inf makes the discriminant -inf or NaN, never >= 0 */
            A[k].x = A[k].y = A[k].z = (R)0.0; A[k].w = -(R)INFINITY;
            B[k].x = B[k].y = B[k].z = (R)0.0; B[k].w = (R)1.0;
            continue;
        }
        const mirt_sphere& sp = d->spheres[order[k]];
        R c0[3] = {(R)sp.center0[0], (R)sp.center0[1], (R)sp.center0[2]};
        R c1[3] = {(R)sp.center1[0], (R)sp.center1[1], (R)sp.center1[2]};
        R r = (R)sp.radius;
        r = r > (R)0.0 ? r : (R)0.0; /* f64::max(r, 0.0), sphere.rs:31,50 */
        A[k].x = c0[0]; A[k].y = c0[1]; A[k].z = c0[2]; A[k].w = r * r;
        if (sp.moving) { B[k].x = c1[0] - c0[0]; B[k].y = c1[1] - c0[1]; B[k].z = c1[2] - c0[2]; }
        else { B[k].x = B[k].y = B[k].z = (R)0.0; }
        B[k].w = sizeof(R) == 8 ? r : (R)1.0 / r; /* make_hit: f64 divides by r (sphere.rs:103), the f32 spec multiplies by 1 / r */
    }
    std::vector<DevMaterial<R>> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const mirt_material& m = d->materials[i];
        DevMaterial<R>& o = mats[i];
        memset(&o, 0, sizeof(o));
        o.kind = m.kind; o.has_fuzz = m.has_fuzz; o.tex_kind = 0xFFFFFFFFu;
        o.reflectance = (R)m.reflectance; o.fuzz = (R)m.fuzz; o.ior = (R)m.ior;
        o.albedo = {(R)m.albedo[0], (R)m.albedo[1], (R)m.albedo[2]};
        if (m.kind == MIRT_MAT_LAMBERTIAN) { /* validated by the caller: texture and image indices are in range */
            const mirt_texture& t = d->textures[m.texture];
            o.tex_kind = t.kind;
            o.inv_scale = (R)t.inv_scale;
            if (t.kind != MIRT_TEX_IMAGE) {
                /* a surviving bounce carries colour / reflectance (material.rs:78): for constant colours the quotient is taken
                 * here, once, in the kernels' precision - the same IEEE division the kernel would repeat at every bounce */
                o.albedo = {(R)t.color[0] / o.reflectance, (R)t.color[1] / o.reflectance, (R)t.color[2] / o.reflectance};
                o.odd = {(R)t.odd[0] / o.reflectance, (R)t.odd[1] / o.reflectance, (R)t.odd[2] / o.reflectance};
            } else {
                o.img_w = imgs[t.image].w; o.img_h = imgs[t.image].h; o.img_offset = imgs[t.image].offset;
                if (sizeof(R) == 4) o.inv_scale = ((R)1.0 / (R)255.0) / o.reflectance; /* f32 spec: texel byte -> colour / reflectance in one product */
            }
        }
    }
    int rc;
    if ((rc = upload(s, A, &out.A)) || (rc = upload(s, B, &out.B)) || (rc = upload(s, mats, &out.mats))) return rc;
    return MIRT_OK;
}

/* ---- matrix-pipe broad phase (trace_mx): split-f16 sphere operands -----------------------------------------
 * Per tile of 32 sorted spheres 2 x 64 x 16 B, [tile][MFMA m][lane]: lane l carries K half (l / 32) of MFMA m for the
 * sphere in row (l % 32); row i holds sorted sphere 32 T + 16 ((i / 4) % 2) + (i % 4) + 4 (i / 8), so that the 16 rows a
 * result lane receives are 16 consecutive spheres.  The 32 K slots are listed in trace_mx (mirt_kernels.h): the six
 * products M_i M_j, the three M_i, 1 and w, each but the 1 as an f16 pair (hi, lo).  Values are in the scene scale 2^e
 * with max (|M| + rho) 2^e <= 64. */
static uint16_t f16_bits(_Float16 h) { uint16_t b; memcpy(&b, &h, 2); return b; }
static _Float16 f16_from_bits(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return h; }
static _Float16 f16_round_down(double x) { /* the largest f16 <= x (finite range) */
    _Float16 h = (_Float16)x;
    if ((double)h <= x) return h;
    uint16_t b = f16_bits(h);
    if ((b & 0x7FFFu) == 0) return f16_from_bits(0x8001u); /* below +-0: the smallest negative subnormal */
    return f16_from_bits((b & 0x8000u) ? (uint16_t)(b + 1) : (uint16_t)(b - 1));
}
/* Does the scene qualify at all?  (finite extents) */
static void mx_init(mirt_scene::MxSet& set, const std::vector<MxBall>& balls /* one per sorted slot */) {
    set.balls.clear();
    double R = 0;
    for (const MxBall& b : balls)
        if (b.real) {
            const double ext = std::sqrt(b.M[0] * b.M[0] + b.M[1] * b.M[1] + b.M[2] * b.M[2]) + b.rho;
            if (!(ext < 1e30)) return; /* NaN / inf / absurd extents: the VALU kernel copes, f16 does not */
            R = std::max(R, ext);
        }
    set.R = R;
    set.balls = balls;
}
/* The scale 2^e of a call: rays start at the lens or on a sphere, so with `reach` = the lens's largest distance from the
 * origin every ray has |O| <= max(R, reach); the operands fit the f16 range when R 2^e <= 64 and reach 2^e <= 180
 * (180^2 < MIRT_MX_K2MAX).  The absolute part of the margin (see trace_mx) is fixed in SCALED units, so a camera far
 * outside the scene, or a scene of tiny spheres around a huge one, makes it large against rho^2: where that is the
 * case for more than a quarter of the spheres they would be candidates for every ray passing anywhere near them, and
 * the call stays on the VALU kernel. */
static bool mx_pick(const mirt_scene* s, const mirt_scene::MxSet& set, const mirt_camera* cam, int* e_out) {
    if (set.balls.empty() && s->n != 0) return false;
    auto len = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    const double reach = len(cam->lookfrom) + len(cam->disk_u_r) + len(cam->disk_v_r);
    if (!(reach < 1e30)) return false;
    const double need = std::max(set.R, reach * (64.0 / 180.0));
    int e = 0;
    if (need > 0) {
        (void)std::frexp(64.0 / need, &e); /* 64 / need = m 2^e, m in [0.5, 1)  ->  2^(e-1) <= 64 / need */
        e -= 1;
    }
    if (e < -100 || e > 100) return false;
    const double sc = std::ldexp(1.0, e);
    size_t real = 0, coarse = 0;
    for (const MxBall& b : set.balls)
        if (b.real && !b.big) {
            const double Ml = std::sqrt(b.M[0] * b.M[0] + b.M[1] * b.M[1] + b.M[2] * b.M[2]) * sc, rho = b.rho * sc;
            ++real;
            if (2.0 * Ml / 2097152.0 + 1.6e-7 > 0.5 * rho * rho) ++coarse;
        }
    if (coarse * 4 > real) return false;
    *e_out = e;
    return true;
}
/* Operands for scale 2^e, built on first use and kept with the scene. */
static int mx_ensure(const mirt_scene* s, const mirt_scene::MxSet& set, int e) {
    s->mx_scale = (float)std::ldexp(1.0, e);
    for (const auto& c : set.cache)
        if (c.first == e) { s->SM = c.second; return MIRT_OK; }
    const std::vector<MxBall>& balls = set.balls;
    const double sc = std::ldexp(1.0, e);
    const double eps = 1.0 / 16777216.0, F = 1.0 / 2097152.0; /* 2^-24, 2^-21 */
    const size_t ntiles = (balls.size() + 31) / 32;
    /* top tiles (tile cull): one for a resident scene, one per 32 tiles for scenes of up to MIRT_MX_CULL_TILES tiles beyond the LDS */
    const size_t ntop = ntiles <= 32 ? 1 : ntiles <= MIRT_MX_CULL_TILES ? (ntiles + 31) / 32 : 0;
    std::vector<uint4> SM((ntiles + ntop + MIRT_RING_TILES + 1) * 128); /* + one tile: the streaming loop prefetches one ahead; + a chunk: trace_mx_ring stages whole chunks */
    auto split = [](double x, _Float16& h, _Float16& l) {
        h = (_Float16)x;
        l = (_Float16)(x - (double)h);
    };
    /* The top tile (slot `ntiles` of the operand array, resident scenes only: <= 32 tiles): row k holds a ball around ALL the
     * balls of tile k, tested by the same contraction with the same margins - if the exact discriminant of any sphere of
     * tile k can be >= 0 the real line meets that sphere's ball, hence the tile's, hence row k's test passes - so a tile whose
     * row no ray of the wave passes can be skipped whole (trace_mx).  A tile ball that leaves the operand range (|C| + R > 64
     * in scaled units) is marked "always" (w = -60000: q > 0 for every ray), an empty tile "never" (the dummy record). */
    std::vector<MxBall> top;
    if (ntop) {
        top.assign(32 * ntop, MxBall{{0, 0, 0}, 0, false});
        const size_t per = ntiles <= 16 ? 16 : 32; /* slots per row of the top tile: scenes of <= 16 tiles get a ball per HALF tile */
        for (size_t T = 0; T < (balls.size() + per - 1) / per; ++T) {
            double lo3[3] = {1e300, 1e300, 1e300}, hi3[3] = {-1e300, -1e300, -1e300};
            bool any = false;
            for (size_t i = per * T; i < std::min(balls.size(), per * T + per); ++i)
                if (balls[i].real && !balls[i].big) {
                    any = true;
                    for (int a = 0; a < 3; ++a) { lo3[a] = std::min(lo3[a], balls[i].M[a] - balls[i].rho); hi3[a] = std::max(hi3[a], balls[i].M[a] + balls[i].rho); }
                }
            if (!any) continue;
            MxBall t{{0.5 * (lo3[0] + hi3[0]), 0.5 * (lo3[1] + hi3[1]), 0.5 * (lo3[2] + hi3[2])}, 0, true};
            for (size_t i = per * T; i < std::min(balls.size(), per * T + per); ++i)
                if (balls[i].real && !balls[i].big) {
                    const double dx = balls[i].M[0] - t.M[0], dy = balls[i].M[1] - t.M[1], dz = balls[i].M[2] - t.M[2];
                    t.rho = std::max(t.rho, std::sqrt(dx * dx + dy * dy + dz * dz) + balls[i].rho);
                }
            t.rho = t.rho * (1.0 + 1e-9) + 1e-300;
            top[T] = t;
        }
    }
    for (size_t T = 0; T < ntiles + ntop; ++T)
        for (uint32_t row = 0; row < 32; ++row) {
            const bool is_top = T >= ntiles;
            const std::vector<MxBall>& src = is_top ? top : balls;
            const uint32_t sub = (row / 4) % 2, r = (row % 4) + 4 * (row / 8);
            const size_t idx = 32 * (is_top ? T - ntiles : T) + 16 * sub + r;
            _Float16 sv[32];
            for (int k = 0; k < 32; ++k) sv[k] = (_Float16)0.f;
            sv[27] = sv[28] = sv[29] = (_Float16)1.0f;
            sv[30] = (_Float16)60000.0f; /* dummy: q = c0 - 60000 < 0 for every admissible ray (c0 <= 0) */
            if (is_top && idx < src.size() && src[idx].real &&
                !((std::sqrt(src[idx].M[0] * src[idx].M[0] + src[idx].M[1] * src[idx].M[1] + src[idx].M[2] * src[idx].M[2]) + src[idx].rho) * sc <= 64.0)) {
                sv[30] = (_Float16)-50000.0f; /* out of the operand range: q = c0 + 50000 > 0 for every ray in range (c0 >= -36000) and for the
                                                 lanes marked "origin out of range" (c0 = +60000), < 0 for the lanes without a ray (c0 = -60000) */
            } else if (idx < src.size() && src[idx].real && !src[idx].big) {
                const MxBall& b = src[idx];
                const double m[3] = {b.M[0] * sc, b.M[1] * sc, b.M[2] * sc}; /* exact: power of two */
                const double M2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
                static const int pi[6] = {0, 1, 2, 0, 0, 1}, pj[6] = {0, 1, 2, 1, 2, 2}; /* xx yy zz xy xz yz */
                for (int k = 0; k < 9; ++k) {
                    _Float16 h, l;
                    split(k < 6 ? m[pi[k]] * m[pj[k]] : m[k - 6], h, l);
                    sv[3 * k] = h; sv[3 * k + 1] = l; sv[3 * k + 2] = h; /* against the ray's (hi, hi, lo) */
                }
                const double rho = b.rho * sc, Ml = std::sqrt(M2);
                const double w = M2 - rho * rho - (double)MIRT_MX_MARGIN_TEST * (256.0 * eps * (M2 + rho * rho) + F * Ml);
                sv[30] = (_Float16)w;
                sv[31] = f16_round_down(w - (double)sv[30]); /* wh + wl <= w */
            }
            for (uint32_t m = 0; m < 2; ++m)
                for (uint32_t half = 0; half < 2; ++half) memcpy(&SM[(T * 2 + m) * 64 + half * 32 + row], &sv[m * 16 + half * 8], 16);
        }
    uint4* dev = nullptr;
    HIP_TRY(hipMalloc((void**)&dev, SM.size() * sizeof(uint4)));
    s->allocs.push_back(dev);
    HIP_TRY(hipMemcpy(dev, SM.data(), SM.size() * sizeof(uint4), hipMemcpyHostToDevice));
    set.cache.emplace_back(e, dev);
    s->SM = dev;
    return MIRT_OK;
}

template <class R> static DevScene<R> dev_scene(const mirt_scene* s, const SceneBuffers<R>& b) {
    DevScene<R> d;
    d.A = b.A; d.B = b.B; d.orig_id = s->orig_id; d.material = s->material; d.mats = b.mats;
    d.texels = s->texels; d.n = s->n; d.n_static = s->n_static; d.n_live = s->n_live;
    d.S = sizeof(R) == 4 ? s->S : nullptr;
    d.n_materials = s->n_materials;
    d.bvh = s->bvh; d.bvh_S = s->bvh_S; d.bvh_idx = s->bvh_idx; d.bvh_root = s->bvh_root;
    d.bvh_n_nodes = s->bvh_n_nodes; d.bvh_n_recs = s->bvh_n_recs; d.bvh_depth = s->bvh_depth;
    d.SM = s->SM; d.mx_scale = s->mx_scale; d.mx_cull = s->mx_cull;
    for (int a = 0; a < 3; ++a) d.mx_drift[a] = s->mx_drift[a];
    d.n_big = s->n_big;
    for (int b = 0; b < 4; ++b) d.big_idx[b] = s->big_idx[b];
    d.dbg = s->stats ? &s->stats->broad_misses : nullptr;
    return d;
}
template <class R> static DevCamera<R> dev_camera(const mirt_camera* c) {
    auto v = [](const double* p) { return V3<R>{(R)p[0], (R)p[1], (R)p[2]}; };
    DevCamera<R> d;
    d.lookfrom = v(c->lookfrom); d.du = v(c->px_du); d.dv = v(c->px_dv); d.ul = v(c->vp_upper_left); d.px00 = v(c->px00_loc);
    d.disk_u = v(c->disk_u_r); d.disk_v = v(c->disk_v_r); d.defocus_angle = (R)c->defocus_angle;
    d.W = c->img_width; d.H = c->img_height; d.spp = c->samples_per_pixel; d.depth = c->max_bounce_depth;
    return d;
}

static int make_part(const mirt_camera* cam, const mirt_opts* o, DevPart* p) {
    p->strip_rows = o->strip_rows ? o->strip_rows : 8u;
    p->part_count = o->part_count ? o->part_count : 1u;
    p->part_index = o->part_index;
    if (p->part_index >= p->part_count) return fail(MIRT_ERR_INVALID, "mirt_opts: part_index >= part_count");
    p->local_rows = part_rows(cam->img_height, p->strip_rows, p->part_index, p->part_count);
    p->compact = (o->flags & MIRT_FLAG_COMPACT_ROWS) ? 1u : 0u;
    return MIRT_OK;
}
static int check_common(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o) {
    if (!s || !cam || !o) return fail(MIRT_ERR_INVALID, "null argument");
    if (o->precision != MIRT_F32 && o->precision != MIRT_F64) return fail(MIRT_ERR_INVALID, "mirt_opts.precision must be 32 or 64");
    if (o->_reserved[0] || o->_reserved[1]) return fail(MIRT_ERR_INVALID, "mirt_opts._reserved must be 0 (zero-initialise the struct)");
    if (o->bvh_keep > 64) return fail(MIRT_ERR_INVALID, "mirt_opts.bvh_keep is a share in 64ths (0 = automatic, 1..64)");
    if (cam->img_width == 0 || cam->img_height == 0) return fail(MIRT_ERR_INVALID, "empty image");
    if ((uint64_t)cam->img_width * cam->img_height > 0xFFFFFFFFull) return fail(MIRT_ERR_TOO_LARGE, "W*H must fit 32 bits (RNG pixel counter)");
    if (o->flags & MIRT_FLAG_BVH) {
        if (o->precision != MIRT_F32) return fail(MIRT_ERR_INVALID, "MIRT_FLAG_BVH is an f32-mode accelerator (precision must be MIRT_F32)");
        if (!s->bvh_ok) return fail(MIRT_ERR_TOO_LARGE, "scene too large for the bounding-ball tree (depth or record limit)");
    }
    return MIRT_OK;
}
/* Brute-force calls run their broad phase on the matrix pipe when the scene and the camera allow it (mx_pick); the f64
 * kernels use the balls built around the f64 spheres.  Decided by the arguments alone (no environment variable). */
static const mirt_scene::MxSet& mx_set(const mirt_scene* s, const mirt_opts* o) { return o->precision == MIRT_F64 ? s->mx64 : s->mx32; }
static bool use_mx(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, int* e) {
    if (o->flags & (MIRT_FLAG_BVH | MIRT_FLAG_VALU_BROAD)) return false;
    /* a handful of spheres is tested faster than the ray's operands are split and exchanged (3 spheres at 4K: 9.6 vs 10.1 ms) */
    if (s->true_n < 24 && !(o->flags & MIRT_FLAG_MX_SMALL)) return false;
    return mx_pick(s, mx_set(s, o), cam, e);
}
static uint32_t grid_for(uint32_t W, uint32_t local_rows, uint32_t waves_per_block) {
    uint32_t tiles = ((W + 7) / 8) * ((local_rows + 7) / 8);
    return (tiles + waves_per_block - 1) / waves_per_block;
}

/* ---- launches -------------------------------------------------------------------------------- */
static int launch_primary(const mirt_scene* s, const SceneBuffers<double>& b, const mirt_camera* cam, const DevPart& part, uint32_t* d_id, void* d_t,
                          hipStream_t st, bool mx = false) {
    uint32_t grid = grid_for(cam->img_width, part.local_rows, MIRT_WAVES_PER_BLOCK);
    if (grid == 0) return MIRT_OK;
    if (mx)
        hipLaunchKernelGGL(primary_hits_kernel_f64_mx, dim3(grid), dim3(MIRT_BLOCK), 0, st, dev_scene<double>(s, b), dev_camera<double>(cam), part, d_id, (double*)d_t);
    else
    hipLaunchKernelGGL(primary_hits_kernel_f64, dim3(grid), dim3(MIRT_BLOCK), 0, st, dev_scene<double>(s, b), dev_camera<double>(cam), part, d_id, (double*)d_t);
    HIP_TRY(hipGetLastError());
    return MIRT_OK;
}
static int launch_primary(const mirt_scene* s, const SceneBuffers<float>& b, const mirt_camera* cam, const DevPart& part, uint32_t* d_id, void* d_t,
                          hipStream_t st, bool bvh = false, bool mx = false) {
    uint32_t grid = grid_for(cam->img_width, part.local_rows, bvh ? MIRT_WAVES_PER_BLOCK : MIRT_F32_WAVES);
    if (grid == 0) return MIRT_OK;
    if (mx)
        hipLaunchKernelGGL(primary_hits_kernel_f32_mx, dim3(grid), dim3(MIRT_F32_BLOCK), 0, st, dev_scene<float>(s, b), dev_camera<float>(cam), part, d_id, (float*)d_t);
    else if (bvh)
        hipLaunchKernelGGL(primary_hits_kernel_f32_bvh, dim3(grid), dim3(MIRT_BLOCK), 0, st, dev_scene<float>(s, b), dev_camera<float>(cam), part, d_id, (float*)d_t);
    else
        hipLaunchKernelGGL(primary_hits_kernel_f32, dim3(grid), dim3(MIRT_F32_BLOCK), 0, st, dev_scene<float>(s, b), dev_camera<float>(cam), part, d_id, (float*)d_t);
    HIP_TRY(hipGetLastError());
    return MIRT_OK;
}
/* Samples per work item.  A lane runs the samples of its item back to back, so the launch cannot end before its
 * longest item has: samples of one pixel are correlated (all 64 samples of a pixel on a glass ball bounce to the depth
 * limit), and with 4 samples per item the tail of a 1/8-frame band (the 8-GPU share) was 2.5 ms against 1.5 ms with
 * one.  So: as few samples per item as the partial-sum workspace allows (12 B per item in f32).
 * n_pixels is the pixel count of the WHOLE frame, whatever part of it a call renders: a pixel's radiance is the sum, in
 * chunk order, of per-chunk sums, so its last bits depend on the chunk size - and the image must not depend on how many
 * GPUs share the frame (tests/test_gpu_parity.py::test_bench_frame_full_size_properties). */
/* f64 (parity mode) gets the larger budget: there the aim is one sample per item wherever it fits, because only then is a
 * pixel's sum the reference's sequential sum over its samples (camera.rs:242-246) bit for bit - 1080p x 64 spp needs 3.2 GB. */
#define MIRT_PARTIAL_BUDGET_F32 (1024ull << 20)
#define MIRT_PARTIAL_BUDGET_F64 (4096ull << 20)
static void choose_chunks(uint64_t n_pixels, uint32_t nsamples, size_t real_bytes, uint32_t min_chunk, uint32_t* chunk, uint32_t* chunks) {
    const uint64_t budget = real_bytes == 8 ? MIRT_PARTIAL_BUDGET_F64 : MIRT_PARTIAL_BUDGET_F32;
    uint64_t max_chunks = budget / (std::max<uint64_t>(n_pixels, 1) * 3 * real_bytes);
    if (max_chunks < 1) max_chunks = 1;
    uint32_t q = min_chunk ? min_chunk : 1u; /* mirt_opts.chunk_samples */
    while ((nsamples + q - 1) / q > max_chunks && q < 0x80000000u) q *= 2;
    *chunk = q;
    *chunks = (nsamples + q - 1) / q;
}

/* Device memory a render launch needs beyond its inputs and outputs: the counters and work-queue head, the per-item
 * partial sums, and (f64) the per-lane attenuation stacks.  The scene handle's own workspace grows on demand with
 * hipMalloc; the one behind mirt_render_device_static is a fixed slice of a file-scope device array and never grows. */
struct Workspace {
    DevStats* stats;
    /* growing (the scene handle's own): */
    void** partial; size_t* partial_bytes;
    void** att; size_t* att_bytes;
    /* fixed (mirt_render_device_static): everything behind the header of this device's static array */
    unsigned char* fixed_base; size_t fixed_cap;
};
static int ws_grow(void** buf, size_t* have, size_t need, const char* what) {
    if (need <= *have) return MIRT_OK;
    if (*buf) (void)hipFree(*buf); /* waits for the device: nothing in flight reads it any more */
    *buf = nullptr;
    *have = 0;
    if (hipMalloc(buf, need) != hipSuccess) { *buf = nullptr; return fail(MIRT_ERR_NOMEM, std::string("hipMalloc failed (") + what + ", " + std::to_string(need) + " bytes)"); }
    *have = need;
    return MIRT_OK;
}
#define MIRT_ATT_STACK_LIMIT (8192ull << 20)
/* launches of fewer work items per lane of the grid than this have no continuation pass (launch_render): the second launch costs
 * ~0.1 ms (10 000 spheres at 1080p x 4 spp, 32 items per lane: 4.0 -> 4.1 ms with it; at 4K x 16 spp, 253 per lane: 35.3 -> 33.2 ms) */
#define MIRT_CONT_ITEMS_PER_LANE 64ull

/* resident blocks per CU of every render kernel, asked once per process (not per call) */
static int occupancy_of(const void* fn, int block) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void*, int>, int>> cache;
    std::lock_guard<std::mutex> g(mu);
    for (const auto& c : cache) if (c.first.first == fn && c.first.second == block) return c.second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    cache.push_back({{fn, block}, per_cu});
    return per_cu;
}

/* Tile cull of the matrix-pipe kernels (trace_mx): 0 = every tile is evaluated (MIRT_FLAG_ALL_PAIRS; scenes beyond the LDS, whose
 * top tile does not exist; scenes of one or two tiles, where the extra tile costs more than it can save), 2 = the top tile holds
 * a ball per HALF tile (scenes of <= 16 tiles), 1 = a ball per tile, 3 = scenes beyond the LDS of up to MIRT_MX_CULL_TILES tiles:
 * one top tile per 32 tiles (trace_mx_ring) - exactly as mx_ensure lays the top tiles out */
static uint32_t mx_cull_mode(const mirt_scene* s, bool mx, const mirt_opts* o) {
    if (!mx || (o->flags & MIRT_FLAG_ALL_PAIRS) || s->n <= 64u) return 0u;
    if (s->n > MIRT_LDS_RECORDS) return s->n <= 32u * MIRT_MX_CULL_TILES ? 3u : 0u; /* one top tile per 32 tiles; every wave streams its own tiles (trace_mx, not resident); beyond: every tile, through the ring */
    return s->n <= 512u ? 2u : 1u;
}
/* the two render kernels behind one name */
static const void* render_fn(double, int bvh) { return bvh == 3 ? (const void*)render_kernel_f64_mx<true> : bvh == 7 ? (const void*)render_kernel_f64_mx<false> : (const void*)render_kernel_f64<2>; }
static const void* render_fn(float, int bvh) { return bvh == 4 ? (const void*)render_kernel_f32_tiny : bvh == 6 ? (const void*)render_kernel_f32_mx_ring : bvh == 7 ? (const void*)render_kernel_f32_mx<MxLds, false> : bvh == 5 ? (const void*)render_kernel_f32_mx<MxLdsSmall, true> : bvh == 3 ? (const void*)render_kernel_f32_mx<MxLds, true> : bvh == 2 ? (const void*)render_kernel_f32_bvh_lds : bvh ? (const void*)render_kernel_f32_bvh : (const void*)render_kernel_f32; }
static void launch_render_kernel(uint32_t grid, uint32_t block, int bvh, hipStream_t st, const DevScene<double>& sc, const DevCamera<double>& cam,
                                 const DevPart& part, const RenderParams& rp, double* partial, DevStats* stats, double* att) {
    if (bvh == 3) hipLaunchKernelGGL(render_kernel_f64_mx<true>, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats, att);
    else if (bvh == 7) hipLaunchKernelGGL(render_kernel_f64_mx<false>, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats, att);
    else hipLaunchKernelGGL(render_kernel_f64<2>, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats, att);
}
static void launch_render_kernel(uint32_t grid, uint32_t block, int bvh, hipStream_t st, const DevScene<float>& sc, const DevCamera<float>& cam,
                                 const DevPart& part, const RenderParams& rp, float* partial, DevStats* stats, float*) {
    if (bvh == 4) hipLaunchKernelGGL(render_kernel_f32_tiny, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else if (bvh == 6) hipLaunchKernelGGL(render_kernel_f32_mx_ring, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else if (bvh == 7) hipLaunchKernelGGL((render_kernel_f32_mx<MxLds, false>), dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else if (bvh == 5) hipLaunchKernelGGL((render_kernel_f32_mx<MxLdsSmall, true>), dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else if (bvh == 3) hipLaunchKernelGGL((render_kernel_f32_mx<MxLds, true>), dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else if (bvh == 2) hipLaunchKernelGGL(render_kernel_f32_bvh_lds, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else if (bvh) hipLaunchKernelGGL(render_kernel_f32_bvh, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
    else hipLaunchKernelGGL(render_kernel_f32, dim3(grid), dim3(block), 0, st, sc, cam, part, rp, partial, stats);
}

template <class R>
static int launch_render(const mirt_scene* s, const SceneBuffers<R>& b, const mirt_camera* cam, const mirt_opts* o, const DevPart& part, const Workspace& ws,
                         void* d_lin, uint8_t* d_rgb8, hipStream_t st, bool use_bvh, bool mx, uint32_t first_sample = 0, uint32_t sample_count = 0xFFFFFFFFu,
                         void* d_accum = nullptr, hipEvent_t ev_mid = nullptr) {
    /* closest-hit front end: 0 = brute force, VALU broad phase; 1 = tree in HBM/L2; 2 = tree, records and spheres resident in
     * LDS; 3 = brute force, broad phase on the matrix pipe; 4 = f32, a handful of spheres (render_kernel_f32_tiny);
     * 5 = 3 with the LDS layout for scenes of up to MIRT_MX_SMALL_RECORDS sorted slots (f32); 6 = 3 for scenes beyond the LDS,
     * every tile, operands through the block-wide ring (f32); 7 = 3 for scenes beyond the LDS with the tile cull: every wave
     * streams its own tiles (f32; the f64 kernel decides resident / streamed by itself) */
    const int bvh = mx ? (sizeof(R) == 4 ? (s->n <= MIRT_MX_SMALL_RECORDS ? 5 : s->n > MIRT_LDS_RECORDS ? (s->mx_cull == 3u ? 7 : 6) : 3) : (s->n <= MIRT_MX64_RECORDS ? 3 : 7)) : !use_bvh ? ((sizeof(R) == 4 && s->n_live <= MIRT_TINY_SCENE) ? 4 : 0) /* 4 = the f32 kernel for a handful of spheres */
                    : (s->bvh_n_nodes <= MIRT_BVH_LDS_NODES && s->bvh_n_recs <= MIRT_LDS_RECORDS && s->n <= MIRT_LDS_RECORDS &&
                       s->bvh_depth <= MIRT_BVH_LDS_STACK && s->n_materials <= 0x10000u && s->bvh_root != MIRT_BVH_NONE) ? 2 : 1;
    const uint32_t block = (sizeof(R) == 4 && bvh != 1) ? MIRT_F32_BLOCK : MIRT_BLOCK, waves_per_block = block / MIRT_WAVE;
    uint64_t n_pixels = (uint64_t)cam->img_width * part.local_rows;
    if (n_pixels == 0) return MIRT_OK;
    const uint64_t seed = o->seed;
    RenderParams rp;
    rp.nsamples = cam->samples_per_pixel ? cam->samples_per_pixel : 1u;
    choose_chunks((uint64_t)cam->img_width * cam->img_height, rp.nsamples, sizeof(R), o->chunk_samples, &rp.chunk, &rp.chunks);
    /* a progressive pass renders whole chunks [first / Q, (first + count) / Q) of the same decomposition the one-call render uses */
    rp.chunk_first = 0;
    rp.norm_samples = rp.nsamples;
    if (sample_count != 0xFFFFFFFFu) {
        if (!cam->samples_per_pixel) return fail(MIRT_ERR_INVALID, "progressive passes need anti-aliasing samples (samples_per_pixel > 0)");
        const uint64_t last = (uint64_t)first_sample + sample_count;
        if (sample_count == 0 || last > rp.nsamples) return fail(MIRT_ERR_INVALID, "pass: sample range outside [0, samples_per_pixel)");
        if (first_sample % rp.chunk != 0 || (last != rp.nsamples && last % rp.chunk != 0))
            return fail(MIRT_ERR_INVALID, "pass: boundaries must be multiples of the work-item size (" + std::to_string(rp.chunk) +
                                              " samples for this image; see mirt_pass_granularity)");
        if (first_sample != 0 && !d_accum) return fail(MIRT_ERR_INVALID, "pass: a later pass needs the accumulation buffer of the earlier ones");
        rp.chunk_first = first_sample / rp.chunk;
        rp.chunks = (uint32_t)((last + rp.chunk - 1) / rp.chunk) - rp.chunk_first;
        rp.norm_samples = (uint32_t)last;
    }
    /* queue order: passes of G = 2^g_shift chunks (render_body).  Long launches (>= 128 items per lane of the chip) keep all
     * chunks of a pixel together, short ones - an 8-GPU band of the bench frame has 63 - use passes of 4 (measured:
     * profiles/r01_microbench.txt); mirt_opts.group_chunks is the tuning knob (image-neutral) */
    uint32_t G = n_pixels * rp.chunks >= 128ull * 1024ull * (uint64_t)s->n_cus ? 64u : 4u;
    if (o->group_chunks) G = o->group_chunks;
    rp.g_shift = 0;
    while (rp.g_shift < 30 && (2u << rp.g_shift) <= G && (2u << rp.g_shift) <= rp.chunks) ++rp.g_shift;
    const uint64_t passes = (rp.chunks + (1u << rp.g_shift) - 1) >> rp.g_shift;
    if (n_pixels * (passes << rp.g_shift) > 0xFFFFFF00ull) return fail(MIRT_ERR_TOO_LARGE, "too many work items for one launch");
    rp.n_items = (uint32_t)(n_pixels * (passes << rp.g_shift)); /* queue slots; those beyond the last chunk are skipped */
    rp.k0 = (uint32_t)seed;
    rp.k1 = (uint32_t)(seed >> 32);
    rp.tail_items = 0;
    rp.n_pixels = (uint32_t)n_pixels;
    rp.row_items = (uint32_t)((uint64_t)cam->img_width * (passes << rp.g_shift));
    rp.d_row = make_fastdiv(rp.row_items);
    rp.d_pg = make_fastdiv(cam->img_width << rp.g_shift);
    rp.d_W = make_fastdiv(cam->img_width);
    rp.d_strip = make_fastdiv(part.strip_rows);
    rp.keep_num = 0;
    rp.att_lanes = 0;
    rp.cont_mode = 0;
    int rc;
    size_t need = (size_t)rp.n_items * 3 * sizeof(R); /* one slot per queue position */
    /* continuation pass (render_body, T::CONTINUES: the f32 kernel for scenes beyond the LDS): room for the state of one path per
     * lane of the grid behind the partial sums, where the launch is long enough to have a tail worth it and the room is there
     * (the static scratch may be too small: then the launch simply runs to its end as before) */
    bool cont = false;
    if (sizeof(R) == 4 && bvh == 7 && cam->max_bounce_depth != 0) {
        const uint64_t lanes = (uint64_t)s->n_cus * occupancy_of(render_fn(R(), bvh), (int)block) * block; /* the grid is at most this */
        const size_t with = ((((size_t)rp.n_items * 3 + 63) & ~(size_t)63) + (size_t)MIRT_CONT_FIELDS * lanes) * sizeof(R);
        if ((uint64_t)rp.n_items >= MIRT_CONT_ITEMS_PER_LANE * lanes && (!ws.fixed_base || with <= ws.fixed_cap)) { cont = true; need = with; }
    }
    R* partial = nullptr;
    if (ws.fixed_base) {
        if (need > ws.fixed_cap)
            return fail(MIRT_ERR_TOO_LARGE, "mirt_render_device_static: the per-item partial sums of this call need " + std::to_string(need) +
                                                " bytes, the static scratch has " + std::to_string(ws.fixed_cap) + " (MIRT_STATIC_SCRATCH_BYTES in mirt.h)");
        partial = (R*)ws.fixed_base;
    } else {
        if ((rc = ws_grow(ws.partial, ws.partial_bytes, need, "the per-item partial sums"))) return rc;
        partial = (R*)*ws.partial;
    }
    if (cam->max_bounce_depth == 0) {
        /* color(ray, 0) is black before any hit test (camera.rs:269-271): no segment is traced */
        HIP_TRY(hipMemsetAsync(partial, 0, need, st));
    } else {
        /* persistent grid: as many blocks as can be resident, every wave pulls items until the queue is empty */
        const int per_cu = occupancy_of(render_fn(R(), bvh), (int)block);
        uint64_t waves_needed = ((uint64_t)rp.n_items + 63) / 64;
        uint64_t grid = std::min<uint64_t>((uint64_t)s->n_cus * per_cu, (waves_needed + waves_per_block - 1) / waves_per_block);
        R* att = nullptr;
        if (sizeof(R) == 8) {
            /* parity mode multiplies a path's attenuations back to front like the reference's recursion (camera.rs:277): each
             * lane keeps the attenuations of the path in hand in a stack of its own, [bounce][rgb][lane of the grid] */
            const uint64_t per_block = (uint64_t)cam->max_bounce_depth * 3 * sizeof(R) * block;
            if (ws.fixed_base) {
                /* what the partial sums leave of the static array; the grid shrinks to the blocks whose stacks fit (any grid
                 * size renders the same image, a smaller one only takes longer) */
                const size_t used = (need + 255) & ~(size_t)255;
                const uint64_t fit = used < ws.fixed_cap ? (ws.fixed_cap - used) / per_block : 0;
                if (fit == 0)
                    return fail(MIRT_ERR_TOO_LARGE, "mirt_render_device_static: no room left in the static scratch for the f64 attenuation stacks (" +
                                                        std::to_string(per_block) + " bytes per block at max_bounce_depth " + std::to_string(cam->max_bounce_depth) + ")");
                grid = std::min<uint64_t>(grid, fit);
                att = (R*)(ws.fixed_base + used);
            } else {
                /* as above: a smaller grid renders the same image, so a very deep bounce limit shrinks the grid to the blocks whose
                 * stacks fit MIRT_ATT_STACK_LIMIT and only fails if not even one does (such depths rendered before the
                 * stacks existed) */
                if (per_block > MIRT_ATT_STACK_LIMIT)
                    return fail(MIRT_ERR_TOO_LARGE, "f64 parity mode keeps one attenuation per bounce and lane: max_bounce_depth " + std::to_string(cam->max_bounce_depth) +
                                                        " needs " + std::to_string(per_block) + " bytes for one block (limit 8 GiB)");
                grid = std::min<uint64_t>(grid, MIRT_ATT_STACK_LIMIT / per_block);
                const uint64_t att_need = per_block * grid;
                if ((rc = ws_grow(ws.att, ws.att_bytes, (size_t)att_need, "the f64 attenuation stacks"))) return rc;
                att = (R*)*ws.att;
            }
            rp.att_lanes = (uint32_t)(grid * block);
        }
        /* the last two full batches per wave are handed out one item per lane at a time */
        rp.tail_items = (uint32_t)std::min<uint64_t>(grid * waves_per_block * 2 * MIRT_QUEUE_BATCH, rp.n_items);
        if (bvh == 1 || bvh == 2) { /* suspend a traversal once this share (in 64ths) of the traversing lanes is all that is left (mirt_opts.bvh_keep) */
            rp.keep_num = bvh == 1 ? 32 : 16; /* measured: profiles/r01_microbench.txt */
            if (o->bvh_keep) rp.keep_num = o->bvh_keep;
        }
        if (cont && (uint64_t)rp.n_items >= MIRT_CONT_ITEMS_PER_LANE * grid * block) { /* (the grid is smaller than its bound only for launches too short to pass this) */
            rp.cont_mode = 1;
            launch_render_kernel((uint32_t)grid, block, bvh, st, dev_scene<R>(s, b), dev_camera<R>(cam), part, rp, partial, ws.stats, att);
            HIP_TRY(hipGetLastError());
            rp.cont_mode = 2; /* the same grid: the slots are spread over its waves */
        }
        launch_render_kernel((uint32_t)grid, block, bvh, st, dev_scene<R>(s, b), dev_camera<R>(cam), part, rp, partial, ws.stats, att);
        HIP_TRY(hipGetLastError());
    }
    if (ev_mid) HIP_TRY(hipEventRecord(ev_mid, st)); /* timed calls: where the render kernel ends and the resolve kernel begins */
    /* one wave per 64 pixels of a band row */
    const uint64_t rwaves = (uint64_t)part.local_rows * ((cam->img_width + 63u) / 64u);
    uint32_t rgrid = (uint32_t)((rwaves + MIRT_RESOLVE_WAVES - 1) / MIRT_RESOLVE_WAVES);
    /* chunks of a pixel staged at a time: 16 where the passes of the queue order hold that many, else 8 (resolve_kernel); f32 only -
     * in binary64 the wide tile is twice the LDS and the 4K x 16-chunk parity frame resolved in 3.35 ms instead of 1.45 */
    if (sizeof(R) == 4 && std::min<uint32_t>(rp.chunks, 1u << rp.g_shift) >= MIRT_RESOLVE_TC_WIDE)
        hipLaunchKernelGGL((resolve_kernel<R, MIRT_RESOLVE_TC_WIDE>), dim3(rgrid), dim3(MIRT_RESOLVE_WAVES * MIRT_WAVE), 0, st, (const R*)partial, dev_camera<R>(cam), part,
                           rp, (uint32_t)n_pixels, (R*)d_lin, d_rgb8, ws.stats, (R*)d_accum);
    else
        hipLaunchKernelGGL((resolve_kernel<R, MIRT_RESOLVE_TC>), dim3(rgrid), dim3(MIRT_RESOLVE_WAVES * MIRT_WAVE), 0, st, (const R*)partial, dev_camera<R>(cam), part, rp,
                           (uint32_t)n_pixels, (R*)d_lin, d_rgb8, ws.stats, (R*)d_accum);
    HIP_TRY(hipGetLastError());
    return MIRT_OK;
}

/* ---- the allocation-free entry point's memory (mirt_render_device_static) -----------------------------------------
 * A file-scope device array: every GPU the library runs on holds its own copy, at its own address, until the process ends.
 * In HOST code the array's name is only a stand-in for the symbol - never a device address - so the address is asked for
 * once per device (static_scratch_ptr, called by mirt_scene_create with that device current) and everything below uses
 * the pointer it returned.  Layout: [0, 4096) counters and work-queue head (DevStats), then the partial sums, then (f64)
 * the attenuation stacks. */
#define MIRT_STATIC_HEADER 4096u
static_assert(sizeof(DevStats) <= MIRT_STATIC_HEADER, "header of the static scratch too small");
static __device__ __align__(256) unsigned char mirt_static_scratch[MIRT_STATIC_SCRATCH_BYTES];
#define MIRT_MAX_DEVICES 64
static unsigned char* g_static_ptr[MIRT_MAX_DEVICES];
static std::mutex g_static_mu;
static int static_scratch_ptr(int device, unsigned char** out) { /* the calling thread's current device must be `device` */
    if (device < 0 || device >= MIRT_MAX_DEVICES) return fail(MIRT_ERR_NO_DEVICE, "device index beyond MIRT_MAX_DEVICES");
    std::lock_guard<std::mutex> g(g_static_mu);
    if (!g_static_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(mirt_static_scratch));
        if (e != hipSuccess || !p) return fail(MIRT_ERR_HIP, std::string("hipGetSymbolAddress(mirt_static_scratch): ") + hipGetErrorString(e));
        g_static_ptr[device] = (unsigned char*)p;
    }
    *out = g_static_ptr[device];
    return MIRT_OK;
}

/* k-d order of order[lo, hi) (indices into the description's spheres): split at the median of the axis along which the centres
 * (mid-shutter) spread most, with the cut placed on a tile boundary of the final array (slot numbers that are multiples of
 * 32; `lo` sits at slot `slot_lo`), until a range fits one tile.  Deterministic (ties by index). */
static void spatial_order(const mirt_scene_desc* d, std::vector<uint32_t>& order, size_t lo, size_t hi, size_t slot_lo) {
    if (hi - lo <= 1) return;
    const size_t slot_hi = slot_lo + (hi - lo);
    if (slot_lo / 16 == (slot_hi - 1) / 16) return; /* one half tile: the finest grain the top tile can tell apart */
    auto centre = [&](uint32_t i, int a) {
        const mirt_sphere& sp = d->spheres[i];
        return sp.moving ? 0.5 * (sp.center0[a] + sp.center1[a]) : sp.center0[a];
    };
    /* the axis of the widest spread - between the 10th and the 90th percentile of the centres, not between the extremes: one
     * huge ground sphere 1000 units below the rest must not make "height" the axis everything is split along */
    double spread[3] = {0, 0, 0};
    {
        std::vector<double> v(hi - lo);
        for (int a = 0; a < 3; ++a) {
            for (size_t k = lo; k < hi; ++k) { const double c = centre(order[k], a); v[k - lo] = c == c ? c : 0.0; }
            const size_t p10 = v.size() / 10, p90 = v.size() - 1 - v.size() / 10;
            std::nth_element(v.begin(), v.begin() + p10, v.end());
            const double lo_v = v[p10];
            std::nth_element(v.begin(), v.begin() + p90, v.end());
            spread[a] = v[p90] - lo_v;
        }
    }
    int ax = 0;
    for (int a = 1; a < 3; ++a) if (spread[a] > spread[ax]) ax = a;
    /* the tile boundary nearest to the middle of the range (inside one tile: the half-tile boundary) */
    const size_t mid_slot = (slot_lo + slot_hi) / 2;
    const size_t grain = slot_lo / 32 == (slot_hi - 1) / 32 ? 16 : 32;
    size_t cut_slot = (mid_slot + grain / 2) / grain * grain;
    if (cut_slot <= slot_lo) cut_slot += grain;
    if (cut_slot >= slot_hi) cut_slot -= grain;
    if (cut_slot <= slot_lo || cut_slot >= slot_hi) return;
    const size_t mid = lo + (cut_slot - slot_lo);
    std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](uint32_t x, uint32_t y) {
        const double cx = centre(x, ax), cy = centre(y, ax);
        return cx < cy || (cx == cy && x < y) || (cy != cy && cx == cx); /* NaN centres last */
    });
    spatial_order(d, order, lo, mid, slot_lo);
    spatial_order(d, order, mid, hi, cut_slot);
}

extern "C" {

int mirt_abi_version(void) { return (int)MIRT_ABI_VERSION; }
const char* mirt_last_error(void) { return g_err.c_str(); }
int mirt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
uint32_t mirt_part_rows(uint32_t height, uint32_t strip_rows, uint32_t part_index, uint32_t part_count) {
    return part_rows(height, strip_rows ? strip_rows : 8u, part_index, part_count);
}
int mirt_closest_hit_kind(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o) {
    int rc = check_common(s, cam, o);
    if (rc) return rc;
    if (o->flags & MIRT_FLAG_BVH) return 2;
    int e = 0;
    const bool mx = use_mx(s, cam, o, &e);
    if (o->precision == MIRT_F64) return mx ? 4 : 3;
    if (!mx && s->n_live <= MIRT_TINY_SCENE) return 5;
    return mx ? 1 : 0;
}
void mirt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

int mirt_scene_create(const mirt_scene_desc* d, int device, mirt_scene** out) {
    if (!d || !out) return fail(MIRT_ERR_INVALID, "null argument");
    *out = nullptr;
    if ((d->n_spheres && !d->spheres) || (d->n_materials && !d->materials) || (d->n_textures && !d->textures) || (d->n_images && !d->images))
        return fail(MIRT_ERR_INVALID, "scene description has a count without an array");
    for (uint32_t i = 0; i < d->n_spheres; ++i)
        if (d->spheres[i].material >= d->n_materials) return fail(MIRT_ERR_INVALID, "sphere references a material out of range");
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        if (d->materials[i].kind > MIRT_MAT_DIELECTRIC) return fail(MIRT_ERR_INVALID, "unknown material kind");
        if (d->materials[i].kind == MIRT_MAT_LAMBERTIAN && d->materials[i].texture >= d->n_textures)
            return fail(MIRT_ERR_INVALID, "material references a texture out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        if (d->textures[i].kind > MIRT_TEX_IMAGE) return fail(MIRT_ERR_INVALID, "unknown texture kind");
        if (d->textures[i].kind == MIRT_TEX_IMAGE && d->textures[i].image >= d->n_images)
            return fail(MIRT_ERR_INVALID, "texture references an image out of range");
    }
    for (uint32_t i = 0; i < d->n_images; ++i)
        if ((uint64_t)d->images[i].width * d->images[i].height != 0 && !d->images[i].rgb) return fail(MIRT_ERR_INVALID, "image without texels");
    int ndev = mirt_device_count();
    if (ndev <= 0) return fail(MIRT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(MIRT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    mirt_scene* s = new (std::nothrow) mirt_scene();
    if (!s) return fail(MIRT_ERR_NOMEM, "out of host memory");
    s->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cus = prop.multiProcessorCount;
    }
    if (d->n_spheres > 0xFFFFFF00u) { delete s; return fail(MIRT_ERR_TOO_LARGE, "too many spheres"); }
    /* statics first, each section padded to a whole broad-phase group.  Within a section the spheres are put in a SPATIAL
     * order (spatial_order below): the matrix-pipe kernels test a ball around each tile of 32 consecutive slots first and skip
     * the tiles no ray of the wave can touch, which pays only if a tile's spheres lie together.  The order is internal: hit ids
     * are the description's indices (orig_id), ties in t go to the lower hit id, so the image does not depend on it. */
    std::vector<uint32_t> order;
    order.reserve((size_t)d->n_spheres + MIRT_GROUP + 32);
    uint32_t n_stat = 0;
    bool neg_zero = false; /* a static centre component of -0.0 (see below) */
    for (uint32_t i = 0; i < d->n_spheres; ++i)
        if (!d->spheres[i].moving) {
            ++n_stat;
            for (int a = 0; a < 3; ++a) neg_zero = neg_zero || (d->spheres[i].center0[a] == 0.0 && std::signbit(d->spheres[i].center0[a]));
        }
    s->true_static = n_stat;
    /* Scenes that fit the LDS (where the tile cull works) and hold both kinds: ONE section, statics among the movings, so that
     * every tile is a compact patch of the scene instead of the statics' tiles spanning all of it.  The kernels then take every
     * sphere as moving with a zero displacement: c0 + tm * 0 is c0 exactly in both precisions (the one exception, a -0.0
     * component that the sum would turn into +0.0, keeps such a scene in two sections). */
    const bool one_section = n_stat > 0 && n_stat < d->n_spheres && d->n_spheres <= MIRT_LDS_RECORDS - 2 * MIRT_GROUP && !neg_zero;
    if (one_section) {
        for (uint32_t i = 0; i < d->n_spheres; ++i) order.push_back(i);
        s->n_static = 0;
        spatial_order(d, order, 0, order.size(), 0);
    } else {
        for (uint32_t i = 0; i < d->n_spheres; ++i)
            if (!d->spheres[i].moving) order.push_back(i);
        spatial_order(d, order, 0, order.size(), 0);
        while (order.size() % MIRT_GROUP) order.push_back(0xFFFFFFFFu);
        s->n_static = (uint32_t)order.size();
        for (uint32_t i = 0; i < d->n_spheres; ++i)
            if (d->spheres[i].moving) order.push_back(i);
        spatial_order(d, order, s->n_static, order.size(), s->n_static);
    }
    /* ... and the whole array to whole 32-sphere tiles of the matrix-pipe broad phase (a multiple of the 16-record groups
     * of trace_cons and of MIRT_GROUP): a lane whose operands say "every record of this tile is a candidate" (origin out of
     * range, NaN) then meets a dummy record in every slot up to the end of the last tile, in HBM and in the LDS copies */
    while (order.size() % MIRT_GROUP) order.push_back(0xFFFFFFFFu);
    s->n_live = (uint32_t)order.size();
    while (order.size() % 32u) order.push_back(0xFFFFFFFFu);
    s->n = (uint32_t)order.size();
    s->true_n = d->n_spheres;
    s->n_materials = d->n_materials;
    std::vector<uint32_t> mat(order.size());
    for (size_t k = 0; k < order.size(); ++k) mat[k] = order[k] == 0xFFFFFFFFu ? 0u : d->spheres[order[k]].material;

    /* images -> one RGBA8 pool (one dword per texel) */
    std::vector<DevImage> imgs(d->n_images);
    std::vector<uint32_t> texels;
    for (uint32_t i = 0; i < d->n_images; ++i) {
        const mirt_image& im = d->images[i];
        imgs[i].w = im.width; imgs[i].h = im.height; imgs[i].offset = (uint32_t)texels.size(); imgs[i]._pad = 0;
        size_t n = (size_t)im.width * im.height;
        if (texels.size() + n > 0xFFFFFFFFull) { mirt_scene_destroy(s); return fail(MIRT_ERR_TOO_LARGE, "texture pool exceeds 2^32 texels"); }
        for (size_t p = 0; p < n; ++p)
            texels.push_back((uint32_t)im.rgb[3 * p] | ((uint32_t)im.rgb[3 * p + 1] << 8) | ((uint32_t)im.rgb[3 * p + 2] << 16));
    }
    int rc;
    if ((rc = upload(s, order, &s->orig_id)) || (rc = upload(s, mat, &s->material)) ||
        (rc = upload(s, texels, &s->texels)) || (rc = upload_real<float>(s, d, order, imgs, s->f32)) || (rc = upload_real<double>(s, d, order, imgs, s->f64))) {
        mirt_scene_destroy(s);
        return rc;
    }
    {   /* conservative bounding records of trace_cons, from the f32 values the kernels see, in double */
        std::vector<float4> S((order.size() + MIRT_TILE - 1) / MIRT_TILE * MIRT_TILE + MIRT_TILE); /* whole tiles + one spare */
        for (size_t k = order.size(); k < S.size(); ++k) { S[k].x = S[k].y = S[k].z = 0.f; S[k].w = INFINITY; }
        const double eps = 1.0 / 16777216.0; /* 2^-24 */
        std::vector<BallRec> balls; /* real spheres only, for the bounding-ball tree */
        balls.reserve(d->n_spheres);
        std::vector<MxBall> mxb(order.size(), MxBall{{0, 0, 0}, 0, false}); /* every sorted slot, for the matrix-pipe operands */
        std::vector<MxBall> mxb64(order.size(), MxBall{{0, 0, 0}, 0, false});
        /* The drift of the matrix-pipe filter: a sphere's ball has to hold it for every shutter time, and in the scenes the
         * reference builds the moving spheres all move the same way (main.rs:170-173: up, by 0 .. 0.5) - their common motion
         * inflates every ball for nothing.  Take v = the midrange of the displacements (statics count as 0) out: the filter
         * tests the ray with its origin at O - tm v against balls around c0 + tm (dc - v), tm in [0, 1] - the same relative
         * geometry, radius r + |dc - v| / 2 instead of r + |dc| / 2 (a static sphere pays |v| / 2).  Kept only if it makes the
         * sum of the cross-sections smaller.  The exact tests and the vector-ALU / tree records know nothing of it. */
        {
            double lo3[3] = {0, 0, 0}, hi3[3] = {0, 0, 0};
            for (size_t k = 0; k < order.size(); ++k) {
                if (order[k] == 0xFFFFFFFFu) continue;
                const mirt_sphere& sp = d->spheres[order[k]];
                if (!sp.moving) continue;
                for (int a = 0; a < 3; ++a) {
                    const double dca = (double)((float)sp.center1[a] - (float)sp.center0[a]);
                    if (dca == dca && std::fabs(dca) < 1e30) { lo3[a] = std::min(lo3[a], dca); hi3[a] = std::max(hi3[a], dca); }
                }
            }
            float v[3];
            for (int a = 0; a < 3; ++a) v[a] = (float)(0.5 * (lo3[a] + hi3[a]));
            double with = 0, without = 0;
            for (size_t k = 0; k < order.size(); ++k) {
                if (order[k] == 0xFFFFFFFFu) continue;
                const mirt_sphere& sp = d->spheres[order[k]];
                double l0 = 0, l1 = 0;
                for (int a = 0; a < 3; ++a) {
                    const double dca = sp.moving ? (double)((float)sp.center1[a] - (float)sp.center0[a]) : 0.0;
                    l0 += dca * dca;
                    l1 += (dca - v[a]) * (dca - v[a]);
                }
                const double r = sp.radius > 0.0 ? sp.radius : 0.0;
                if (!(r < 1e30) || !(l0 < 1e60)) continue;
                without += (r + 0.5 * std::sqrt(l0)) * (r + 0.5 * std::sqrt(l0)) / (1.0 + r * r); /* relative growth of the cross-section, so that one huge sphere does not decide */
                with += (r + 0.5 * std::sqrt(l1)) * (r + 0.5 * std::sqrt(l1)) / (1.0 + r * r);
            }
            if (with < without) for (int a = 0; a < 3; ++a) s->mx_drift[a] = v[a];
        }
        const double vd[3] = {(double)s->mx_drift[0], (double)s->mx_drift[1], (double)s->mx_drift[2]};
        const double vl = std::sqrt(vd[0] * vd[0] + vd[1] * vd[1] + vd[2] * vd[2]);
        for (size_t k = 0; k < order.size(); ++k) {
            if (order[k] == 0xFFFFFFFFu) { S[k].x = S[k].y = S[k].z = 0.f; S[k].w = INFINITY; continue; }
            const mirt_sphere& sp = d->spheres[order[k]];
            double c0[3], dc[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                float f0 = (float)sp.center0[a], f1 = (float)sp.center1[a];
                c0[a] = f0;
                if (sp.moving) dc[a] = (double)(f1 - f0); /* the f32 difference the kernel uses */
            }
            float rf = (float)sp.radius;
            double r = rf > 0.f ? rf : 0.0;
            float Mf[3];
            double M2 = 0, dcl = 0;
            for (int a = 0; a < 3; ++a) { Mf[a] = (float)(c0[a] + 0.5 * dc[a]); dcl += dc[a] * dc[a]; }
            dcl = std::sqrt(dcl);
            double Ml = std::sqrt((double)Mf[0] * Mf[0] + (double)Mf[1] * Mf[1] + (double)Mf[2] * Mf[2]);
            M2 = Ml * Ml;
            /* ball radius: sphere + half the travel + rounding of M, of fma(tm, dc, c0) and of the travel itself */
            double rho = r + 0.5 * dcl + 8.0 * eps * (Ml + dcl + r);
            double w = M2 - rho * rho - 160.0 * eps * (M2 + rho * rho);
            float wf = (float)w;
            if ((double)wf > w) wf = std::nextafterf(wf, -INFINITY); /* round toward the conservative side */
            wf = std::nextafterf(wf, -INFINITY);
            S[k].x = Mf[0]; S[k].y = Mf[1]; S[k].z = Mf[2]; S[k].w = wf;
            balls.push_back(BallRec{{(double)Mf[0], (double)Mf[1], (double)Mf[2]}, rho, (uint32_t)k});
            {   /* the matrix-pipe filter's balls: around the motion relative to the drift, c0 + tm (dc - v) */
                float Mx[3];
                double rl = 0;
                for (int a = 0; a < 3; ++a) { Mx[a] = (float)(c0[a] + 0.5 * (dc[a] - vd[a])); rl += (dc[a] - vd[a]) * (dc[a] - vd[a]); }
                rl = std::sqrt(rl);
                const double Mxl = std::sqrt((double)Mx[0] * Mx[0] + (double)Mx[1] * Mx[1] + (double)Mx[2] * Mx[2]);
                /* sphere + half the relative travel + rounding of the centre, of fma(tm, dc, c0) and of the travel itself */
                mxb[k] = MxBall{{(double)Mx[0], (double)Mx[1], (double)Mx[2]}, r + 0.5 * rl + 8.0 * eps * (Mxl + dcl + vl + r), true};
                /* the same centre for the f64 kernels, whose sphere is the description's own: c0 + tm (c1 - c0) in double */
                double off2 = 0, dcl2 = 0, rl2 = 0;
                for (int a = 0; a < 3; ++a) {
                    const double dcd = sp.moving ? sp.center1[a] - sp.center0[a] : 0.0;
                    const double off = sp.center0[a] + 0.5 * (dcd - vd[a]) - (double)Mx[a];
                    off2 += off * off;
                    dcl2 += dcd * dcd;
                    rl2 += (dcd - vd[a]) * (dcd - vd[a]);
                }
                const double rd = sp.radius > 0.0 ? sp.radius : 0.0, dl = std::sqrt(dcl2);
                mxb64[k] = MxBall{{(double)Mx[0], (double)Mx[1], (double)Mx[2]}, rd + 0.5 * std::sqrt(rl2) + std::sqrt(off2) + 1e-12 * (Mxl + dl + vl + rd), true};
            }
        }
        if ((rc = upload(s, S, &s->S))) { mirt_scene_destroy(s); return rc; }
        if ((rc = build_bvh(s, balls, S))) { mirt_scene_destroy(s); return rc; }
        {   /* Spheres far larger than everything else - the ground of the reference's scenes (main.rs:155: radius 1000 under
             * spheres of 0.2) - are left out of the matrix-pipe filter: such a ball is met by nearly every line (one useless
             * candidate per ray) and keeps the tile it sits in needed by every wave.
             * Up to four of them, when leaving them out shrinks the scene's extent by 8 x or more; every ray tests them exactly
             * and directly (trace_mx); the filter's rows for them are dummies.  They keep their say in the RANGE of the operands:
             * rays start on them (a bounce off the ground at the horizon), and an origin out of range makes every record of
             * every tile a candidate (measured: scale 4 instead of 1/32 for the bench scene, frame 10.1 -> 15.2 ms). */
            std::vector<std::pair<double, size_t>> ext;
            for (size_t k = 0; k < mxb.size(); ++k)
                if (mxb[k].real) {
                    const double e32 = std::sqrt(mxb[k].M[0] * mxb[k].M[0] + mxb[k].M[1] * mxb[k].M[1] + mxb[k].M[2] * mxb[k].M[2]) + std::max(mxb[k].rho, mxb64[k].rho);
                    ext.push_back({e32 == e32 ? e32 : 1e300, k});
                }
            std::sort(ext.begin(), ext.end(), [](const std::pair<double, size_t>& x, const std::pair<double, size_t>& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
            uint32_t nb = 0;
            for (uint32_t k = 1; k <= 4 && k < ext.size(); ++k)
                if (ext[k - 1].first < 1e30 && ext[k - 1].first >= 8.0 * ext[k].first) nb = k;
            s->n_big = nb;
            for (uint32_t b = 0; b < nb; ++b) {
                s->big_idx[b] = (uint32_t)ext[b].second;
                mxb[ext[b].second].big = true; /* still real: rays start on them, so they keep their say in the operand range */
                mxb64[ext[b].second].big = true;
            }
        }
        mx_init(s->mx32, mxb);
        mx_init(s->mx64, mxb64);
    }
    std::vector<DevStats> st(1);
    memset(st.data(), 0, sizeof(DevStats));
    if ((rc = upload(s, st, &s->stats))) { mirt_scene_destroy(s); return rc; }
    /* everything a render call needs besides its workspace is made here, at set-up: events, the pinned landing place of
     * deferred statistics, this device's address of the static scratch array, the default operand set of the matrix-pipe
     * broad phase - so that mirt_render_device_static never has to allocate */
    {
        hipError_t e = hipEventCreate(&s->ev0);
        if (e == hipSuccess) e = hipEventCreate(&s->ev1);
        if (e == hipSuccess) e = hipEventCreate(&s->ev2);
        if (e == hipSuccess) e = hipEventCreate(&s->ev_mid);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_busy, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s->host_stats, sizeof(DevStats), hipHostMallocDefault);
        if (e != hipSuccess) { mirt_scene_destroy(s); return fail(MIRT_ERR_HIP, std::string("scene set-up: ") + hipGetErrorString(e)); }
        unsigned char* sp = nullptr;
        if ((rc = static_scratch_ptr(device, &sp))) { mirt_scene_destroy(s); return rc; }
        /* resident blocks per CU of every render kernel this scene can select: asked here, so that no render call - least of
         * all mirt_render_device_static - ever reaches hipOccupancyMaxActiveBlocksPerMultiprocessor or grows the cache */
        for (int k = 0; k <= 7; ++k) (void)occupancy_of(render_fn(float(), k), k == 1 ? (int)MIRT_BLOCK : (int)MIRT_F32_BLOCK);
        for (int k : {0, 3, 7}) (void)occupancy_of(render_fn(double(), k), (int)MIRT_BLOCK);
        /* operands for a camera inside or near the scene (the scale then follows from the scene's extent alone) */
        mirt_camera c0;
        memset(&c0, 0, sizeof(c0));
        for (const mirt_scene::MxSet* set : {&s->mx32, &s->mx64}) {
            int e2 = 0;
            if (s->true_n && mx_pick(s, *set, &c0, &e2) && (rc = mx_ensure(s, *set, e2))) { mirt_scene_destroy(s); return rc; }
        }
    }
    *out = s;
    return MIRT_OK;
}

void mirt_scene_destroy(mirt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->partial) (void)hipFree(s->partial);
    if (s->att) (void)hipFree(s->att);
    if (s->ev_busy) (void)hipEventDestroy(s->ev_busy);
    for (void* p : s->stage) if (p) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev2) (void)hipEventDestroy(s->ev2);
    if (s->ev_mid) (void)hipEventDestroy(s->ev_mid);
    if (s->host_stats) (void)hipHostFree(s->host_stats);
    delete s;
}

int mirt_primary_hits_device(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, uint32_t* d_id, void* d_t, void* stream) {
    int rc = check_common(s, cam, o);
    if (rc) return rc;
    if (!d_id) return fail(MIRT_ERR_INVALID, "id output is required");
    DevPart part;
    if ((rc = make_part(cam, o, &part))) return rc;
    std::lock_guard<std::recursive_mutex> guard(s->lock);
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    int mx_e = 0;
    const bool mx = use_mx(s, cam, o, &mx_e);
    if (mx && (rc = mx_ensure(s, mx_set(s, o), mx_e))) return rc;
    s->mx_cull = mx_cull_mode(s, mx, o);
    return o->precision == MIRT_F64 ? launch_primary(s, s->f64, cam, part, d_id, d_t, st, mx)
                                    : launch_primary(s, s->f32, cam, part, d_id, d_t, st, (o->flags & MIRT_FLAG_BVH) != 0, mx);
}

static int render_impl(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, uint32_t first_sample, uint32_t sample_count, void* d_accum,
                       void* d_lin, uint8_t* d_rgb8, void* stream, mirt_stats* stats, bool use_static);
int mirt_render_device(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, void* d_lin, uint8_t* d_rgb8, void* stream,
                       mirt_stats* stats) {
    return render_impl(s, cam, o, 0, 0xFFFFFFFFu, nullptr, d_lin, d_rgb8, stream, stats, false);
}
int mirt_render_pass_device(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, uint32_t first_sample, uint32_t sample_count,
                            void* d_accum, void* d_lin, uint8_t* d_rgb8, void* stream, mirt_stats* stats) {
    if (sample_count == 0xFFFFFFFFu) return fail(MIRT_ERR_INVALID, "pass: sample_count out of range");
    return render_impl(s, cam, o, first_sample, sample_count, d_accum, d_lin, d_rgb8, stream, stats, false);
}
uint32_t mirt_pass_granularity(const mirt_camera* cam, const mirt_opts* o) {
    if (!cam || !o || !cam->samples_per_pixel) return 0;
    DevPart part;
    if (make_part(cam, o, &part)) return 0;
    uint32_t chunk = 1, chunks = 1;
    choose_chunks((uint64_t)cam->img_width * cam->img_height, cam->samples_per_pixel, o->precision == MIRT_F64 ? 8 : 4, o->chunk_samples, &chunk, &chunks);
    return chunk;
}
/* diagnostic builds (-DMIRT_VERIFY_BROAD, -DMIRT_PHASE_TIMING, ...): what the kernels counted, on stderr */
static void print_diagnostics(const DevStats& h) {
    (void)h;
#ifdef MIRT_VERIFY_BROAD
    fprintf(stderr, "[mirt verify broad] exact candidates missed by the conservative broad phase: %llu\n", h.broad_misses);
#endif
#ifdef MIRT_PHASE_TIMING
    {
        static const char* names[6] = {"queue", "raygen", "broad", "narrow", "shade", "other"};
        unsigned long long tot = 0;
        for (int i = 0; i < 6; ++i) tot += h.phase_cycles[i];
        fprintf(stderr, "[mirt phase timing] wave-cycles:");
        for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h.phase_cycles[i] / (double)(tot ? tot : 1));
        fprintf(stderr, "  (total %.3g cycles, %.0f per wave-iteration)\n", (double)tot, (double)tot / (double)(h.wave_iters ? h.wave_iters : 1));
#ifdef MIRT_ACTIVE_HIST
        fprintf(stderr, "[mirt active-lane histogram] wave-iterations with 1-2 / 3-8 / 9-32 / 33-64 active lanes: %llu %llu %llu %llu of %llu\n", h.bvh_lane_steps,
                h.bvh_wave_steps, h.bvh_leaf_rounds, h.bvh_lane_leaves, h.wave_iters);
#else
        if (h.bvh_wave_steps)
            fprintf(stderr, "[mirt phase timing] tree: %.1f node steps per segment, %.1f wave-level steps and %.2f leaf rounds per wave-iteration, %.2f leaves per segment\n",
                    (double)h.bvh_lane_steps / (double)(h.segments ? h.segments : 1), (double)h.bvh_wave_steps / (double)(h.wave_iters ? h.wave_iters : 1),
                    (double)h.bvh_leaf_rounds / (double)(h.wave_iters ? h.wave_iters : 1), (double)h.bvh_lane_leaves / (double)(h.segments ? h.segments : 1));
#endif
        {
            static const char* dn[DG_N] = {"narrow round", "narrow: moving sphere", "narrow: roots (sqrt, near root)", "narrow: far root", "ray generation", "hit record + scatter",
                                           "lambertian", "metal", "metal fuzz", "dielectric", "image texture", "checker texture", "sky", "store partial sum",
                                           "deal items", "queue atomic", "-", "broad/narrow pass"};
            const double wi = (double)(h.wave_iters ? h.wave_iters : 1), sg = (double)(h.segments ? h.segments : 1);
            fprintf(stderr, "[mirt block counts] per wave-iteration: wave-level entries / lanes entering  (segments per wave-iteration %.2f)\n", sg / wi);
            for (int i = 0; i < DG_N; ++i)
                if (h.diag[i]) fprintf(stderr, "[mirt block counts]   %-32s %7.3f / %7.3f\n", dn[i], (double)h.diag[i] / wi, (double)h.diag[24 + i] / wi);
        }
        double span = (double)(h.end_max - ~h.inv_start_max) * 1e-5; /* 100 MHz ticks -> ms */
        double life = (double)h.life_sum / (double)(h.n_waves ? h.n_waves : 1) * 1e-5;
        if (h.cont_count) fprintf(stderr, "[mirt phase timing] paths handed to the continuation launch: %u (by the late waves of the first; this split and the times below are the first launch's)\n", h.cont_count);
        fprintf(stderr, "[mirt phase timing] %llu waves, first start -> last end %.3f ms, mean wave lifetime %.3f ms (%.1f%% of the span)\n",
                h.n_waves, span, life, 100.0 * life / (span > 0 ? span : 1));
        {
            const double nw = (double)(h.n_waves ? h.n_waves : 1);
            const double first = (double)((~h.inv_start_max) & 0xFFFFFFFFFFull);
            fprintf(stderr, "[mirt phase timing] mean start %.3f ms after the first wave's; mean end %.3f ms before the last wave's; mean time from 'queue empty' to the wave's end %.3f ms\n",
                    ((double)h.start_sum / nw - first) * 1e-5, span - ((double)h.start_sum / nw - first) * 1e-5 - life, (double)h.idle_sum / nw * 1e-5);
#ifndef MIRT_BLOCK_COUNTS
            fprintf(stderr, "[mirt phase timing] waves ending k x 25 us before the longest-lived one (of an earlier identical launch), k = 0..38+:");
            for (int k = 0; k < 39; ++k) fprintf(stderr, " %llu", h.diag[k]);
            fprintf(stderr, "\n");
#endif
            fprintf(stderr, "[mirt phase timing] drain per wave: < 50 us %llu, < 100 %llu, < 200 %llu, < 400 %llu, < 800 %llu, longer %llu; lanes with work at 'queue empty' %.1f, iterations after it %.1f, shader cycles after it %.0f\n",
                    h.diag[40], h.diag[41], h.diag[42], h.diag[43], h.diag[44], h.diag[45], (double)h.diag[46] / nw, (double)h.diag[47] / nw, (double)h.diag[39] / nw);
        }
    }
#endif
}
static void fill_stats(const mirt_scene* s, uint32_t flags, const DevStats& h, float ms, mirt_stats* stats, float resolve_ms = 0.f) {
    memset(stats, 0, sizeof(*stats));
    stats->segments = h.segments;
    stats->tree_segments = (flags & MIRT_FLAG_BVH) ? h.segments : 0;
    const uint64_t brute = h.segments - stats->tree_segments;
    stats->static_tests = brute * (uint64_t)s->true_static;
    stats->moving_tests = brute * (uint64_t)(s->true_n - s->true_static);
    stats->sphere_tests = stats->static_tests + stats->moving_tests;
    stats->narrow_tests = h.narrow;
    stats->wave_iterations = h.wave_iters;
    stats->nan_pixels = h.nan_pixels;
    stats->kernel_ms = ms;
    stats->mx_tiles = h.mx_tiles;
    stats->resolve_ms = (resolve_ms > 0.f && resolve_ms <= ms) ? resolve_ms : 0.0; /* a call that launched nothing leaves the mid event where an earlier call put it */
}
int mirt_render_wait(const mirt_scene* s, mirt_stats* stats) {
    if (!s) return fail(MIRT_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> guard(s->lock);
    if (!s->pending) return fail(MIRT_ERR_INVALID, "no deferred render outstanding on this scene handle");
    s->pending = false;
    hipError_t e = hipEventSynchronize(s->ev2);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, s->ev0, s->ev1);
    float rms = 0.f;
    if (e == hipSuccess && hipEventElapsedTime(&rms, s->ev_mid, s->ev1) != hipSuccess) rms = 0.f; /* max_bounce_depth 0: no mid event was recorded */
    if (e != hipSuccess) return fail(MIRT_ERR_HIP, std::string("render: ") + hipGetErrorString(e));
    if (stats) fill_stats(s, s->pending_flags, *s->host_stats, ms, stats, rms);
    print_diagnostics(*s->host_stats);
    return MIRT_OK;
}
/* mx operands of this call's scale: resident already, or (growing workspaces only) uploaded now */
static int mx_ready(const mirt_scene* s, const mirt_scene::MxSet& set, int e, bool may_allocate) {
    for (const auto& c : set.cache)
        if (c.first == e) return mx_ensure(s, set, e);
    if (!may_allocate)
        return fail(MIRT_ERR_INVALID, "mirt_render_device_static: the matrix-pipe operands for this camera's distance are not resident yet - make one "
                                      "mirt_render_device / mirt_primary_hits_device call with this camera at set-up, or pass MIRT_FLAG_VALU_BROAD");
    return mx_ensure(s, set, e);
}
static int render_impl(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, uint32_t first_sample, uint32_t sample_count, void* d_accum,
                       void* d_lin, uint8_t* d_rgb8, void* stream, mirt_stats* stats, bool use_static) {
    int rc = check_common(s, cam, o);
    if (rc) return rc;
    DevPart part;
    if ((rc = make_part(cam, o, &part))) return rc;
    std::lock_guard<std::recursive_mutex> guard(s->lock);
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool defer = (o->flags & MIRT_FLAG_DEFER_STATS) != 0;
    if (defer) stats = nullptr; /* they come from mirt_render_wait */
    if (s->pending) return fail(MIRT_ERR_INVALID, "a deferred render on this scene handle has not been collected (mirt_render_wait)");
    int mx_e = 0;
    const bool mx = use_mx(s, cam, o, &mx_e);
    if (mx && (rc = mx_ready(s, mx_set(s, o), mx_e, !use_static))) return rc; /* before the timed region: the first call at a new scale uploads operands */
    s->mx_cull = mx_cull_mode(s, mx, o); /* as mx_ensure lays the top tile out */
    /* where the launch keeps its counters, partial sums and attenuation stacks */
    Workspace ws{s->stats, &s->partial, &s->partial_bytes, &s->att, &s->att_bytes, nullptr, 0};
    if (use_static) {
        /* this device's copy of mirt_static_scratch, by the address taken at set-up (never the symbol's name): the header
         * holds the counters and the queue head, launch_render places the partial sums and the f64 stacks behind it and
         * refuses what does not fit; nothing is allocated */
        unsigned char* sp = nullptr;
        if ((rc = static_scratch_ptr(s->device, &sp))) return rc;
        ws = Workspace{(DevStats*)sp, nullptr, nullptr, nullptr, nullptr, sp + MIRT_STATIC_HEADER, (size_t)(MIRT_STATIC_SCRATCH_BYTES - MIRT_STATIC_HEADER)};
    } else if (s->launched) {
        /* the previous launch on this handle - on whatever stream - owns the queue head, the counters and the workspace until
         * it is done: order this one behind it */
        HIP_TRY(hipStreamWaitEvent(st, s->ev_busy, 0));
    }
    HIP_TRY(hipMemsetAsync(ws.stats, 0, sizeof(DevStats), st)); /* counters and the work-queue head */
    if (stats || defer) {
        e0 = s->ev0;
        e1 = s->ev1;
        HIP_TRY(hipEventRecord(e0, st));
    }
    rc = o->precision == MIRT_F64
             ? launch_render<double>(s, s->f64, cam, o, part, ws, d_lin, d_rgb8, st, false, mx, first_sample, sample_count, d_accum, e0 ? s->ev_mid : nullptr)
             : launch_render<float>(s, s->f32, cam, o, part, ws, d_lin, d_rgb8, st, (o->flags & MIRT_FLAG_BVH) != 0, mx, first_sample, sample_count, d_accum,
                                    e0 ? s->ev_mid : nullptr);
    if (!use_static) {
        /* also after a failed enqueue: whatever did get onto the stream still uses the handle's memory */
        if (hipEventRecord(s->ev_busy, st) == hipSuccess) s->launched = true;
    }
    if (defer && !rc) { /* nothing waits here: the counters follow the launch into pinned memory */
        HIP_TRY(hipEventRecord(e1, st));
        HIP_TRY(hipMemcpyAsync(s->host_stats, ws.stats, sizeof(DevStats), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(s->ev2, st));
        s->pending = true;
        s->pending_flags = o->flags;
    }
    if (stats) {
        if (!rc) {
            hipError_t e = hipEventRecord(e1, st);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            DevStats h;
            memset(&h, 0, sizeof(h));
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            float rms = 0.f;
            if (e == hipSuccess && hipEventElapsedTime(&rms, s->ev_mid, e1) != hipSuccess) rms = 0.f;
            if (e == hipSuccess) e = hipMemcpy(&h, ws.stats, sizeof(h), hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(MIRT_ERR_HIP, std::string("render: ") + hipGetErrorString(e));
            fill_stats(s, o->flags, h, ms, stats, rms);
            print_diagnostics(h);
        }
    }
    return rc;
}
int mirt_render_device_static(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, void* d_lin, uint8_t* d_rgb8, void* stream,
                              mirt_stats* stats) {
    return render_impl(s, cam, o, 0, 0xFFFFFFFFu, nullptr, d_lin, d_rgb8, stream, stats, true);
}

int mirt_primary_hits(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, uint32_t* id, void* t) {
    int rc = check_common(s, cam, o);
    if (rc) return rc;
    if (!id) return fail(MIRT_ERR_INVALID, "id output is required");
    if (o->flags & MIRT_FLAG_COMPACT_ROWS) return fail(MIRT_ERR_INVALID, "host-buffer calls write full frames; use the _device entry point for compact bands");
    if (o->flags & MIRT_FLAG_DEFER_STATS) return fail(MIRT_ERR_INVALID, "host-buffer calls return when the work is done; MIRT_FLAG_DEFER_STATS is for mirt_render_device / mirt_render_pass_device");
    HIP_TRY(hipSetDevice(s->device));
    size_t npx = (size_t)cam->img_width * cam->img_height;
    size_t rsz = o->precision == MIRT_F64 ? 8 : 4;
    uint32_t* d_id = nullptr;
    void* d_t = nullptr;
    HIP_TRY(hipMalloc((void**)&d_id, npx * 4));
    if (hipMalloc(&d_t, npx * rsz) != hipSuccess) { (void)hipFree(d_id); return fail(MIRT_ERR_NOMEM, "hipMalloc failed"); }
    /* rows outside this part keep MISS / +inf */
    rc = MIRT_OK;
    if (hipMemset(d_id, 0xFF, npx * 4) != hipSuccess) rc = fail(MIRT_ERR_HIP, "hipMemset failed");
    if (!rc && hipMemset(d_t, 0, npx * rsz) != hipSuccess) rc = fail(MIRT_ERR_HIP, "hipMemset failed");
    if (!rc) rc = mirt_primary_hits_device(s, cam, o, d_id, d_t, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(MIRT_ERR_HIP, "primary_hits kernel failed");
    if (!rc && hipMemcpy(id, d_id, npx * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    if (!rc && t && hipMemcpy(t, d_t, npx * rsz, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    (void)hipFree(d_id);
    (void)hipFree(d_t);
    return rc;
}

/* slot k of the scene's staging frames, at least `need` bytes (nullptr + error set on failure) */
static void* stage_frame(const mirt_scene* s, int k, size_t need) {
    if (need > s->stage_bytes[k]) {
        if (s->stage[k]) (void)hipFree(s->stage[k]);
        s->stage[k] = nullptr;
        s->stage_bytes[k] = 0;
        if (hipMalloc(&s->stage[k], need) != hipSuccess) { (void)fail(MIRT_ERR_NOMEM, "hipMalloc failed"); return nullptr; }
        s->stage_bytes[k] = need;
    }
    return s->stage[k];
}
int mirt_render(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, void* lin, uint8_t* rgb8, mirt_stats* stats) {
    int rc = check_common(s, cam, o);
    if (rc) return rc;
    if (o->flags & MIRT_FLAG_COMPACT_ROWS) return fail(MIRT_ERR_INVALID, "host-buffer calls write full frames; use the _device entry point for compact bands");
    if (o->flags & MIRT_FLAG_DEFER_STATS) return fail(MIRT_ERR_INVALID, "host-buffer calls return when the work is done; MIRT_FLAG_DEFER_STATS is for mirt_render_device / mirt_render_pass_device");
    std::lock_guard<std::recursive_mutex> guard(s->lock); /* the staging frames are the handle's: held until they are copied back */
    HIP_TRY(hipSetDevice(s->device));
    size_t n = (size_t)cam->img_width * cam->img_height * 3;
    size_t rsz = o->precision == MIRT_F64 ? 8 : 4;
    /* the device frames stay with the scene between calls (a hipMalloc + hipFree per frame cost more than the copy back) */
    void* d_lin = lin ? stage_frame(s, 0, n * rsz) : nullptr;
    uint8_t* d_rgb8 = rgb8 ? (uint8_t*)stage_frame(s, 1, n) : nullptr;
    if ((lin && !d_lin) || (rgb8 && !d_rgb8)) return MIRT_ERR_NOMEM;
    rc = MIRT_OK;
    /* rows another part owns read as zero */
    if (d_lin && hipMemsetAsync(d_lin, 0, n * rsz, nullptr) != hipSuccess) rc = fail(MIRT_ERR_HIP, "hipMemset failed");
    if (!rc && d_rgb8 && hipMemsetAsync(d_rgb8, 0, n, nullptr) != hipSuccess) rc = fail(MIRT_ERR_HIP, "hipMemset failed");
    mirt_stats local;
    if (!rc) rc = mirt_render_device(s, cam, o, d_lin, d_rgb8, nullptr, &local);
    if (!rc && lin && hipMemcpy(lin, d_lin, n * rsz, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    if (!rc && rgb8 && hipMemcpy(rgb8, d_rgb8, n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    if (!rc && stats) *stats = local;
    if (!rc && rgb8 && local.nan_pixels) return fail(MIRT_ERR_NAN, "a NaN reached the quantiser (the reference panics: rgb.rs:33)");
    return rc;
}

int mirt_render_pass(const mirt_scene* s, const mirt_camera* cam, const mirt_opts* o, uint32_t first_sample, uint32_t sample_count, void* accum,
                     void* lin, uint8_t* rgb8, mirt_stats* stats) {
    int rc = check_common(s, cam, o);
    if (rc) return rc;
    if (o->flags & MIRT_FLAG_COMPACT_ROWS) return fail(MIRT_ERR_INVALID, "host-buffer calls write full frames; use the _device entry point for compact bands");
    if (o->flags & MIRT_FLAG_DEFER_STATS) return fail(MIRT_ERR_INVALID, "host-buffer calls return when the work is done; MIRT_FLAG_DEFER_STATS is for mirt_render_device / mirt_render_pass_device");
    if (!accum) return fail(MIRT_ERR_INVALID, "pass: the accumulation buffer is required");
    std::lock_guard<std::recursive_mutex> guard(s->lock); /* as in mirt_render */
    HIP_TRY(hipSetDevice(s->device));
    const size_t n = (size_t)cam->img_width * cam->img_height * 3, rsz = o->precision == MIRT_F64 ? 8 : 4;
    void* d_acc = stage_frame(s, 2, n * rsz);
    void* d_lin = lin ? stage_frame(s, 0, n * rsz) : nullptr;
    uint8_t* d_rgb8 = rgb8 ? (uint8_t*)stage_frame(s, 1, n) : nullptr;
    if (!d_acc || (lin && !d_lin) || (rgb8 && !d_rgb8)) return MIRT_ERR_NOMEM;
    rc = MIRT_OK;
    /* rows outside the part stay as the caller's accumulation has them */
    if (hipMemcpy(d_acc, accum, n * rsz, hipMemcpyHostToDevice) != hipSuccess) rc = fail(MIRT_ERR_HIP, "upload of the accumulation buffer failed");
    if (!rc && d_lin && hipMemset(d_lin, 0, n * rsz) != hipSuccess) rc = fail(MIRT_ERR_HIP, "hipMemset failed");
    if (!rc && d_rgb8 && hipMemset(d_rgb8, 0, n) != hipSuccess) rc = fail(MIRT_ERR_HIP, "hipMemset failed");
    mirt_stats local;
    if (!rc) rc = mirt_render_pass_device(s, cam, o, first_sample, sample_count, d_acc, d_lin, d_rgb8, nullptr, &local);
    if (!rc && hipMemcpy(accum, d_acc, n * rsz, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    if (!rc && lin && hipMemcpy(lin, d_lin, n * rsz, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    if (!rc && rgb8 && hipMemcpy(rgb8, d_rgb8, n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MIRT_ERR_HIP, "copy back failed");
    if (!rc && stats) *stats = local;
    if (!rc && rgb8 && local.nan_pixels) return fail(MIRT_ERR_NAN, "a NaN reached the quantiser (the reference panics: rgb.rs:33)");
    return rc;
}

int mirt_measure_fma_peak(int device, double* tflops) {
    if (!tflops) return fail(MIRT_ERR_INVALID, "null argument");
    *tflops = 0.0;
    int ndev = mirt_device_count();
    if (device < 0 || device >= ndev) return fail(MIRT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256, iters = 4096, reps = 40; /* ~25 ms: the sustained clock, not a cold burst */
    const int blocks = cus * 8; /* 256 threads = one wave per SIMD; 8 blocks per CU */
    float* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipMalloc((void**)&out, 4));
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.f;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fma_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, out, 1.0001f, 0.5f, iters); /* warm-up: clocks, code upload */
        e = hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps && e == hipSuccess; ++r) hipLaunchKernelGGL(fma_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, out, 1.0001f, 0.5f, iters);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    if (e != hipSuccess || !(ms > 0.f)) return fail(MIRT_ERR_HIP, std::string("fma peak: ") + hipGetErrorString(e));
    *tflops = (double)reps * 2.0 * 16.0 * iters * (double)blocks * 256.0 / ((double)ms * 1e-3) / 1e12;
    return MIRT_OK;
}

int mirt_quantize_device(const void* d_linear, uint32_t precision, size_t n, uint8_t* d_out, void* stream, uint64_t* nan_count) {
    if (!d_linear || !d_out) return fail(MIRT_ERR_INVALID, "null argument");
    if (precision != MIRT_F32 && precision != MIRT_F64) return fail(MIRT_ERR_INVALID, "precision must be 32 or 64");
    if (n == 0) { if (nan_count) *nan_count = 0; return MIRT_OK; }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* d_nan = nullptr;
    if (nan_count) {
        HIP_TRY(hipMalloc((void**)&d_nan, 8));
        HIP_TRY(hipMemsetAsync(d_nan, 0, 8, st));
    }
    uint32_t grid = (uint32_t)((n + 255) / 256);
    if (precision == MIRT_F64) hipLaunchKernelGGL((quantize_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)d_linear, n, d_out, d_nan);
    else hipLaunchKernelGGL((quantize_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)d_linear, n, d_out, d_nan);
    HIP_TRY(hipGetLastError());
    if (nan_count) {
        unsigned long long h = 0;
        hipError_t e = hipMemcpyAsync(&h, d_nan, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_nan);
        if (e != hipSuccess) return fail(MIRT_ERR_HIP, hipGetErrorString(e));
        *nan_count = h;
    }
    return MIRT_OK;
}

int mirt_deinterleave_device(const void* d_bands, size_t band_stride_bytes, uint32_t part_count, uint32_t strip_rows, uint32_t width,
                             uint32_t height, uint32_t elem_bytes, void* d_frame, void* stream) {
    if (!d_bands || !d_frame || part_count == 0 || elem_bytes == 0) return fail(MIRT_ERR_INVALID, "bad argument");
    if (strip_rows == 0) strip_rows = 8;
    size_t total = (size_t)width * height * elem_bytes;
    if (total == 0) return MIRT_OK;
    size_t need = (size_t)part_rows(height, strip_rows, 0, part_count) * width * elem_bytes;
    if (band_stride_bytes < need) return fail(MIRT_ERR_INVALID, "band_stride_bytes smaller than the largest band");
    const size_t row_bytes = (size_t)width * elem_bytes;
    const uintptr_t mis = (uintptr_t)d_bands | (uintptr_t)d_frame | (uintptr_t)row_bytes | (uintptr_t)band_stride_bytes;
    const size_t unit = (mis % 16 == 0) ? 16 : (mis % 4 == 0) ? 4 : 1;
    size_t grid = (total / unit + 255) / 256;
    if (grid > 0x7FFFFFFFull) return fail(MIRT_ERR_TOO_LARGE, "frame too large");
    hipStream_t st = (hipStream_t)stream;
    if (unit == 16)
        hipLaunchKernelGGL(deinterleave_kernel<uint4>, dim3((uint32_t)grid), dim3(256), 0, st, (const uint4*)d_bands, band_stride_bytes / 16, part_count, strip_rows,
                           row_bytes / 16, height, (uint4*)d_frame);
    else if (unit == 4)
        hipLaunchKernelGGL(deinterleave_kernel<uint32_t>, dim3((uint32_t)grid), dim3(256), 0, st, (const uint32_t*)d_bands, band_stride_bytes / 4, part_count,
                           strip_rows, row_bytes / 4, height, (uint32_t*)d_frame);
    else
        hipLaunchKernelGGL(deinterleave_kernel<uint8_t>, dim3((uint32_t)grid), dim3(256), 0, st, (const uint8_t*)d_bands, band_stride_bytes, part_count, strip_rows,
                           row_bytes, height, (uint8_t*)d_frame);
    HIP_TRY(hipGetLastError());
    return MIRT_OK;
}

} /* extern "C" */
