/*
 * oracle.cpp — CPU restatement of the reference renderer's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (raytracer_amd/csrc, libmirt.so) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (ebriussenex/raytracer, Rust) cannot be built here (no cargo/rustc,
 * crates not vendored) and its own test-suite holds exactly one test, the AABB slab check
 * src/scene/aabb.rs:123-137, which is replayed by tests/test_oracle.py::test_reference_aabb_unit_test.
 * Nothing else in the reference pins this path, so every function below follows the Rust source
 * operation for operation instead (file:line cited at each function) and is cross-checked by an
 * independent numpy restatement (oracle/oracle_np.py).  Compile with -O2 -ffp-contract=off: Rust
 * never contracts a*b+c, so `double` here reproduces Rust f64 bit for bit for + - * / sqrt; the
 * libm calls (tan, acos, atan2, sin, cos) go to glibc, as Rust's f64 methods do on Linux.
 *
 * RNG: every reference stream is seeded from OS entropy (camera.rs:46,181,185, material.rs:67,107,142),
 * so no two reference runs agree; only the distributions are contractual.  The oracle draws from the
 * build's counter-based generator (Philox4x32-10, Salmon et al. SC'11, restated below from the paper)
 * keyed on (seed; pixel, sample, bounce, block) - see DESIGN.md "RNG contract".
 *
 * Two instantiations: Real = double (the reference's arithmetic) and Real = float (the build's
 * throughput mode; same operation sequence in binary32 except the discriminant, see Disc<float>).
 */
#include "../include/mirt.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace orc {

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Random123).  Published algorithm, restated; pinned by the Random123 KAT vectors in
 * tests/test_oracle.py.
 * ---------------------------------------------------------------------------------------------- */
static inline void philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    uint32_t k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < 10; ++r) {
        if (r > 0) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* RNG contract: dimension d of the stream (pixel, sample, bounce) - Philox counter (pixel, sample, bounce, block).
 *   f64: block = d / 2, 64-bit word pair d % 2 (x0|x1<<32 or x2|x3<<32), u = (word >> 11) * 2^-53
 *        (rand 0.9's StandardUniform for f64: 53 random bits in [0,1))
 *   f32: block = d / 4, 32-bit word d % 4,                              u = (word >> 8) * 2^-24
 * Blocks are computed lazily so that a draw the reference short-circuits away costs nothing here either. */
template <class R> struct Draws;
static inline float f32_layout_dim(uint32_t pixel, uint32_t sample, uint32_t bounce, uint64_t seed, uint32_t d);
template <> struct Draws<double> {
    uint32_t pixel, sample, bounce;
    uint64_t seed;
    /* f32_layout: the reference's f64 arithmetic fed with the uniforms of the f32 mode - dimension d is the binary32 value
     * Draws<float>::dim(d) widened (exactly) to double.  This is what ties the throughput mode to the reference's own
     * arithmetic "once the same seeded RNG stream is used" (oracle_render_ex, tests/test_gpu_f32_vs_reference.py). */
    bool f32_layout = false;
    double dim(uint32_t d) const {
        if (f32_layout) return (double)f32_layout_dim(pixel, sample, bounce, seed, d);
        uint32_t c[4] = {pixel, sample, bounce, d / 2};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t o[4];
        philox4x32_10(c, k, o);
        uint64_t u = (d % 2) ? ((uint64_t)o[2] | ((uint64_t)o[3] << 32)) : ((uint64_t)o[0] | ((uint64_t)o[1] << 32));
        return (double)(u >> 11) * 0x1.0p-53;
    }
};
template <> struct Draws<float> {
    uint32_t pixel, sample, bounce;
    uint64_t seed;
    /* dims 0..3: the top 24 bits of words 0..3 of block 0; dim 4: the 24 bits left over in the low bytes of words
     * 0..2 of the SAME block (a camera ray needs five draws and Philox is the dearest part of ray generation on the
     * GPU: one block instead of two); dims >= 5: word (d-5)%4 of block 1 + (d-5)/4 */
    float dim(uint32_t d) const {
        uint32_t c[4] = {pixel, sample, bounce, d < 5 ? 0u : 1u + (d - 5) / 4};
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t o[4];
        philox4x32_10(c, k, o);
        if (d == 4) return (float)((o[0] & 0xFFu) | ((o[1] & 0xFFu) << 8) | ((o[2] & 0xFFu) << 16)) * 0x1.0p-24f;
        return (float)(o[d < 5 ? d : (d - 5) % 4] >> 8) * 0x1.0p-24f;
    }
};
static inline float f32_layout_dim(uint32_t pixel, uint32_t sample, uint32_t bounce, uint64_t seed, uint32_t d) {
    return Draws<float>{pixel, sample, bounce, seed}.dim(d);
}

/* Path signature (test infrastructure, no reference counterpart): a hash of every DISCRETE decision the samples of a pixel
 * take - which sphere each segment hits and on which face, whether the material scatters, reflect or refract, checker
 * parity, texel index.  Two renders of one pixel whose signatures agree differ by rounding only; where they disagree some
 * sample's path branched differently.  Thread-local, reset per pixel by render_rows. */
static thread_local uint64_t tl_sig = 0;
static inline void sig_mix(uint64_t v) { tl_sig = (tl_sig ^ (v + 0x9E3779B97F4A7C15ull)) * 0x100000001B3ull; }

/* ------------------------------------------------------------------------------------------------
 * core/point3.rs, core/ray.rs, utils/interval.rs
 * ---------------------------------------------------------------------------------------------- */
template <class R> struct V3 {
    R x, y, z;
};
template <class R> static inline V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; } /* point3.rs:121-129 */
template <class R> static inline V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; } /* :131-139 */
template <class R> static inline V3<R> operator*(V3<R> a, R l) { return {a.x * l, a.y * l, a.z * l}; }           /* :141-149 */
template <class R> static inline V3<R> operator/(V3<R> a, R l) { return {a.x / l, a.y / l, a.z / l}; }           /* :151-159 true division */
template <class R> static inline V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }                         /* :161-169 */

/* point3.rs:48-58: scalar_prod; the `self == rhs` branch evaluates the same expression. */
template <class R> static inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> static inline R size(V3<R> a) { return std::sqrt(dot(a, a)); }                 /* :44-46 */
static inline V3<double> unit(V3<double> a) { return a / size(a); }                               /* :39-41 */
static inline V3<float> unit(V3<float> a) { return a * (1.0f / size(a)); } /* f32 spec: one division, three products */
template <class R> static inline V3<R> cross(V3<R> a, V3<R> b) {                                 /* :60-68 */
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> static inline bool near_zero(V3<R> a) {                                       /* :112-114 */
    const R e = (R)1e-8;
    return std::fabs(a.x) < e && std::fabs(a.y) < e && std::fabs(a.z) < e;
}
/* point3.rs:116-118 / material.rs:168-170: `*v - *n * 2.0 * v.scalar_prod(n)` = v - ((n*2)*dot). */
template <class R> static inline V3<R> reflect(V3<R> v, V3<R> n) { return v - (n * (R)2.0) * dot(v, n); }

template <class R> struct Ray { /* ray.rs:3-34 */
    V3<R> o, d;
    R tm;
    V3<R> at(R t) const { /* :27-29; f32 spec: fused */
        if constexpr (sizeof(R) == 8) return o + d * t;
        else return {std::fmaf(d.x, t, o.x), std::fmaf(d.y, t, o.y), std::fmaf(d.z, t, o.z)};
    }
};

template <class R> struct Interval { /* interval.rs */
    R min, max;
    bool surrounds(R v) const { return min < v && v < max; } /* :37-39 strict */
};

/* ------------------------------------------------------------------------------------------------
 * scene: flattened from mirt_scene_desc, values rounded once to Real
 * ---------------------------------------------------------------------------------------------- */
template <class R> struct Sphere {
    V3<R> c0, c1, dc; /* dc = c1 - c0 (the value sphere.rs:65-67 recomputes per call) */
    R r, r2;          /* r2 = r*r (sphere.rs:89 recomputes it per call)               */
    uint32_t mat;
    bool moving;
};
template <class R> struct Material {
    uint32_t kind, tex;
    R reflectance, fuzz, ior;
    bool has_fuzz;
    V3<R> albedo;
};
template <class R> struct Texture {
    uint32_t kind, image;
    V3<R> even, odd;
    R inv_scale;
};
struct Image {
    uint32_t w, h;
    std::vector<uint8_t> rgb;
};

template <class R> struct HitRec { /* hittable.rs:18-26 (+ the build's object id) */
    V3<R> p, n;
    R t, u, v;
    bool front; /* NormalFace::Outside */
    uint32_t id;
};

template <class R> static inline V3<R> cvt3(const double* p) { return {(R)p[0], (R)p[1], (R)p[2]}; }

/* The discriminant of sphere.rs:81-90.
 *   double: the reference's sequence, unfused.
 *   float : the build's f32 spec - the same quantity through the perpendicular from the centre to the line,
 *           d = a (r^2 - |oc - (h/a) D|^2) with h/a = h * (1/a): the textbook form cancels catastrophically in
 *           binary32 once |oc| >> r; products fused (fmaf) exactly as the HIP kernel does. */
template <class R> struct Disc;
template <> struct Disc<double> {
    static inline double a_of(V3<double> d) { return dot(d, d); } /* sphere.rs:87 */
    static inline V3<double> center(const Sphere<double>& s, double tm) { /* sphere.rs:61-70 */
        if (!s.moving) return s.c0;
        return {s.c0.x + tm * (s.c1.x - s.c0.x), s.c0.y + tm * (s.c1.y - s.c0.y), s.c0.z + tm * (s.c1.z - s.c0.z)};
    }
    static inline void hd(const Sphere<double>& s, V3<double> oc, V3<double> d, double a, double& h, double& disc) {
        h = dot(d, oc);                           /* :88 */
        double c = dot(oc, oc) - s.r * s.r;       /* :89 */
        disc = h * h - a * c;                     /* :90 */
    }
};
template <> struct Disc<float> {
    static inline float a_of(V3<float> d) { return std::fmaf(d.z, d.z, std::fmaf(d.y, d.y, d.x * d.x)); }
    static inline V3<float> center(const Sphere<float>& s, float tm) {
        if (!s.moving) return s.c0;
        return {std::fmaf(tm, s.dc.x, s.c0.x), std::fmaf(tm, s.dc.y, s.c0.y), std::fmaf(tm, s.dc.z, s.c0.z)};
    }
    static inline void hd(const Sphere<float>& s, V3<float> oc, V3<float> d, float a, float& h, float& disc) {
        h = std::fmaf(d.z, oc.z, std::fmaf(d.y, oc.y, d.x * oc.x));
        const float q = h * (1.0f / a);
        const float lx = std::fmaf(-q, d.x, oc.x), ly = std::fmaf(-q, d.y, oc.y), lz = std::fmaf(-q, d.z, oc.z);
        disc = a * (s.r2 - std::fmaf(lz, lz, std::fmaf(ly, ly, lx * lx)));
    }
};

/* acos / atan2 of Sphere::uv.  double: the math library's, as Rust's f64 methods.  float: the f32 spec's polynomial forms
 * (Abramowitz & Stegun 4.4.46 / 4.4.49), operation for operation as mirt_device.h::m_acos / m_atan2. */
static inline double uv_acos(double x) { return std::acos(x); }
static inline double uv_atan2(double y, double x) { return std::atan2(y, x); }
static inline float uv_acos(float x) {
    const float ax = std::fabs(x);
    float p = -0.0012624911f;
    p = std::fmaf(p, ax, 0.0066700901f);
    p = std::fmaf(p, ax, -0.0170881256f);
    p = std::fmaf(p, ax, 0.0308918810f);
    p = std::fmaf(p, ax, -0.0501743046f);
    p = std::fmaf(p, ax, 0.0889789874f);
    p = std::fmaf(p, ax, -0.2145988016f);
    p = std::fmaf(p, ax, 1.5707963050f);
    const float r = std::sqrt(1.0f - ax) * p;
    return x < 0.0f ? 3.14159265358979323846f - r : r;
}
static inline float uv_atan2(float y, float x) {
    const float ax = std::fabs(x), ay = std::fabs(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const float t = mx == 0.0f ? 0.0f : mn / mx;
    const float z = t * t;
    float q = 0.0028662257f;
    q = std::fmaf(q, z, -0.0161657367f);
    q = std::fmaf(q, z, 0.0429096138f);
    q = std::fmaf(q, z, -0.0752896400f);
    q = std::fmaf(q, z, 0.1065626393f);
    q = std::fmaf(q, z, -0.1420889944f);
    q = std::fmaf(q, z, 0.1999355085f);
    q = std::fmaf(q, z, -0.3333314528f);
    q = std::fmaf(q, z, 1.0f);
    float r = t * q;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return y < 0.0f ? -r : r;
}
/* sphere.rs:72-76 */
template <class R> static inline void sphere_uv(V3<R> p, R& u, R& v) {
    const R PI = (R)3.14159265358979323846264338327950288;
    const R FRAC_1_PI = (R)0.318309886183790671537767526745028724;
    R theta = -uv_acos(p.y);
    R phi = -uv_atan2(p.z, p.x) + PI;
    u = (R)0.5 * phi * FRAC_1_PI;
    v = theta * FRAC_1_PI;
}

/* sphere.rs:79-113 + hittable.rs:48-60 */
template <class R> static inline bool sphere_hit(const Sphere<R>& s, uint32_t id, const Ray<R>& ray, Interval<R> I, HitRec<R>& out) {
    V3<R> C = Disc<R>::center(s, ray.tm);
    V3<R> oc = C - ray.o;
    R a = Disc<R>::a_of(ray.d);
    R h, d;
    Disc<R>::hd(s, oc, ray.d, a, h, d);
    if (d < (R)0.0) return false;
    R sq = std::sqrt(d);
    /* f32 spec: the two roots through the ray's 1 / a, the normal through the sphere's 1 / r (products instead of divisions) */
    const R inv_a = (R)1.0 / a;
    R t = sizeof(R) == 8 ? (h - sq) / a : (h - sq) * inv_a;
    if (!I.surrounds(t)) t = sizeof(R) == 8 ? (h + sq) / a : (h + sq) * inv_a;
    if (!I.surrounds(t)) return false;
    V3<R> p = ray.at(t);
    V3<R> on = sizeof(R) == 8 ? (p - C) / s.r : (p - C) * ((R)1.0 / s.r);
    out.p = p;
    out.t = t;
    out.id = id;
    if (dot(ray.d, on) < (R)0.0) { /* hittable.rs:53-59 */
        out.front = true;
        out.n = on;
    } else {
        out.front = false;
        out.n = -on;
    }
    sphere_uv(on, out.u, out.v);
    return true;
}

/* ---- scene/aabb.rs + scene/bvh.rs (the reference's acceleration structure, kept to prove that its
 *      traversal gives the same closest hit as the brute-force loop the GPU runs) ---------------- */
template <class R> struct Aabb {
    R mn[3], mx[3];
};
template <class R> static inline Aabb<R> aabb_empty() { /* aabb.rs:16-20, interval.rs:7-10 */
    const R inf = std::numeric_limits<R>::infinity();
    return {{inf, inf, inf}, {-inf, -inf, -inf}};
}
template <class R> static inline Aabb<R> aabb_from_points(V3<R> a, V3<R> b) { /* aabb.rs:38-49 */
    R av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    Aabb<R> o;
    for (int i = 0; i < 3; ++i) {
        if (av[i] <= bv[i]) { o.mn[i] = av[i]; o.mx[i] = bv[i]; }
        else { o.mn[i] = bv[i]; o.mx[i] = av[i]; }
    }
    return o;
}
template <class R> static inline Aabb<R> aabb_merge(const Aabb<R>& l, const Aabb<R>& r) { /* aabb.rs:51-61, interval.rs:22-27 */
    Aabb<R> o;
    for (int i = 0; i < 3; ++i) {
        o.mn[i] = l.mn[i] <= r.mn[i] ? l.mn[i] : r.mn[i];
        o.mx[i] = l.mx[i] >= r.mx[i] ? l.mx[i] : r.mx[i];
    }
    return o;
}
/* aabb.rs:77-95: one axis, against a COPY of the incoming interval (the three axes are independent). */
template <class R> static inline bool aabb_slab(const Aabb<R>& b, const Ray<R>& r, int axis, Interval<R> I) {
    const R o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
    R adinv = (R)1.0 / d[axis]; /* f64::recip */
    R t0 = (b.mn[axis] - o[axis]) * adinv;
    R t1 = (b.mx[axis] - o[axis]) * adinv;
    R mint, maxt;
    if (t0 < t1) { mint = t0; maxt = t1; } else { mint = t1; maxt = t0; }
    if (mint > I.min) I.min = mint;
    if (maxt < I.max) I.max = maxt;
    return I.max > I.min;
}
template <class R> static inline bool aabb_hit(const Aabb<R>& b, const Ray<R>& r, Interval<R> I) { /* aabb.rs:71-75 */
    return aabb_slab(b, r, 0, I) && aabb_slab(b, r, 1, I) && aabb_slab(b, r, 2, I);
}
template <class R> static inline int aabb_longest_axis(const Aabb<R>& b) { /* aabb.rs:104-120 (quirk: never compares x with z) */
    R xs = b.mx[0] - b.mn[0], ys = b.mx[1] - b.mn[1], zs = b.mx[2] - b.mn[2];
    if (xs <= ys) return (ys <= zs) ? 2 : 1;
    return 0;
}
template <class R> static inline Aabb<R> sphere_bbox(const Sphere<R>& s, R r_unclamped) { /* sphere.rs:28-59 */
    V3<R> rv = {r_unclamped, r_unclamped, r_unclamped};
    Aabb<R> b1 = aabb_from_points(s.c0 - rv, s.c0 + rv);
    if (!s.moving) return b1;
    Aabb<R> b2 = aabb_from_points(s.c1 - rv, s.c1 + rv);
    return aabb_merge(b1, b2);
}

template <class R> struct Scene;

template <class R> struct BvhNode { /* bvh.rs:18-22; a child is another node (>=0 index) or a sphere */
    int left_node = -1, right_node = -1;
    uint32_t left_obj = 0, right_obj = 0;
    Aabb<R> box;
};

template <class R> struct Scene {
    std::vector<Sphere<R>> spheres;
    std::vector<Aabb<R>> boxes;
    std::vector<Material<R>> mats;
    std::vector<Texture<R>> texs;
    std::vector<Image> imgs;
    std::vector<BvhNode<R>> nodes;
    int root = -1;
    mutable std::atomic<uint64_t> segments{0};

    void load(const mirt_scene_desc* d) {
        for (uint32_t i = 0; i < d->n_spheres; ++i) {
            const mirt_sphere& s = d->spheres[i];
            Sphere<R> o;
            o.c0 = cvt3<R>(s.center0);
            o.moving = s.moving != 0;
            o.c1 = o.moving ? cvt3<R>(s.center1) : o.c0;
            o.dc = o.c1 - o.c0;
            R r = (R)s.radius;
            o.r = std::fmax(r, (R)0.0); /* sphere.rs:31,50 */
            o.r2 = o.r * o.r;
            o.mat = s.material;
            spheres.push_back(o);
            boxes.push_back(sphere_bbox(o, r));
        }
        for (uint32_t i = 0; i < d->n_materials; ++i) {
            const mirt_material& m = d->materials[i];
            Material<R> o;
            o.kind = m.kind; o.tex = m.texture;
            o.reflectance = (R)m.reflectance; o.fuzz = (R)m.fuzz; o.ior = (R)m.ior;
            o.has_fuzz = m.has_fuzz != 0;
            o.albedo = cvt3<R>(m.albedo);
            mats.push_back(o);
        }
        for (uint32_t i = 0; i < d->n_textures; ++i) {
            const mirt_texture& t = d->textures[i];
            Texture<R> o;
            o.kind = t.kind; o.image = t.image;
            o.even = cvt3<R>(t.color); o.odd = cvt3<R>(t.odd);
            o.inv_scale = (R)t.inv_scale;
            texs.push_back(o);
        }
        for (uint32_t i = 0; i < d->n_images; ++i) {
            Image im;
            im.w = d->images[i].width; im.h = d->images[i].height;
            im.rgb.assign(d->images[i].rgb, d->images[i].rgb + (size_t)im.w * im.h * 3);
            imgs.push_back(std::move(im));
        }
    }

    /* bvh.rs:28-58.  `objects` = insertion-order list of sphere ids, duplicates allowed (the preset
     * main.rs:225 adds the same ground sphere 22 times).  sort_unstable_by: order among equal keys is
     * unspecified in Rust; any order gives the same closest hit. */
    int build(uint32_t* objs, size_t n) {
        BvhNode<R> nd;
        Aabb<R> box = aabb_empty<R>();
        for (size_t i = 0; i < n; ++i) box = aabb_merge(box, boxes[objs[i]]);
        nd.box = box;
        int axis = aabb_longest_axis(box);
        if (n == 1) {
            nd.left_obj = nd.right_obj = objs[0];
        } else if (n == 2) {
            /* Ordering::Less | Equal -> (left,right); Greater -> swapped (bvh.rs:60-68) */
            if (boxes[objs[0]].mn[axis] <= boxes[objs[1]].mn[axis]) { nd.left_obj = objs[0]; nd.right_obj = objs[1]; }
            else { nd.left_obj = objs[1]; nd.right_obj = objs[0]; }
        } else {
            std::sort(objs, objs + n, [&](uint32_t l, uint32_t r) { return boxes[l].mn[axis] < boxes[r].mn[axis]; });
            size_t mid = n / 2;
            nd.left_node = build(objs, mid);
            nd.right_node = build(objs + mid, n - mid);
        }
        nodes.push_back(nd);
        return (int)nodes.size() - 1;
    }
    void build_bvh(const uint32_t* objects, size_t n) {
        std::vector<uint32_t> tmp(objects, objects + n);
        nodes.clear();
        root = build(tmp.data(), n);
    }

    bool child_hit(int node, uint32_t obj, const Ray<R>& r, Interval<R> I, HitRec<R>& out) const {
        if (node >= 0) return bvh_hit(node, r, I, out);
        return sphere_hit(spheres[obj], obj, r, I, out);
    }
    /* bvh.rs:78-92 */
    bool bvh_hit(int ni, const Ray<R>& r, Interval<R> I, HitRec<R>& out) const {
        const BvhNode<R>& nd = nodes[ni];
        if (!aabb_hit(nd.box, r, I)) return false;
        HitRec<R> hl, hr;
        bool left = child_hit(nd.left_node, nd.left_obj, r, I, hl);
        bool right = left ? child_hit(nd.right_node, nd.right_obj, r, Interval<R>{I.min, hl.t}, hr)
                          : child_hit(nd.right_node, nd.right_obj, r, I, hr);
        if (right) { out = hr; return true; } /* hit_right.or(hit_left) */
        if (left) { out = hl; return true; }
        return false;
    }
    /* The loop the GPU runs: every sphere in ascending id, interval shrinking to the best t so far.
     * Same closest hit as the tree walk; ties in t go to the lowest id. */
    bool brute_hit(const Ray<R>& r, Interval<R> I, HitRec<R>& out) const {
        bool any = false;
        HitRec<R> tmp;
        for (uint32_t i = 0; i < spheres.size(); ++i) {
            if (sphere_hit(spheres[i], i, r, I, tmp)) {
                any = true;
                I.max = tmp.t;
                out = tmp;
            }
        }
        return any;
    }
    bool hit(const Ray<R>& r, Interval<R> I, HitRec<R>& out, bool use_bvh) const { /* hittable.rs:92-98 */
        return use_bvh ? bvh_hit(root, r, I, out) : brute_hit(r, I, out);
    }
};

/* ---- scene/texture.rs ---------------------------------------------------------------------- */
template <class R> static inline V3<R> texture_color(const Scene<R>& sc, const Texture<R>& t, R u, R v, V3<R> p) {
    switch (t.kind) {
    case MIRT_TEX_SOLID: return t.even; /* texture.rs:24-28 */
    case MIRT_TEX_CHECKER: {            /* texture.rs:46-59 */
        int32_t acc = 0;
        const R e[3] = {p.x, p.y, p.z};
        for (int i = 0; i < 3; ++i) {
            R f = std::floor(e[i] * t.inv_scale);
            int32_t k; /* Rust `as i32`: saturating, NaN -> 0 */
            if (f != f) k = 0;
            else if (f >= (R)2147483647.0) k = INT32_MAX;
            else if (f <= (R)-2147483648.0) k = INT32_MIN;
            else k = (int32_t)f;
            acc = (int32_t)((uint32_t)acc + (uint32_t)k);
        }
        sig_mix(0x300u + (uint32_t)(acc % 2 == 0));
        return (acc % 2 == 0) ? t.even : t.odd;
    }
    default: { /* IMAGE texture.rs:71-94 */
        const Image& im = sc.imgs[t.image];
        if (im.h == 0) return {(R)0.0, (R)1.0, (R)1.0}; /* SOLID_CYAN_COLOR rgb.rs:11-13 */
        auto clamp01 = [](R x) { /* f64::clamp: NaN stays NaN */
            if (x < (R)0.0) return (R)0.0;
            if (x > (R)1.0) return (R)1.0;
            return x;
        };
        auto as_u32 = [](R x) -> uint32_t { /* Rust `as u32`: saturating, NaN -> 0 */
            if (!(x > (R)0.0)) return 0u;
            if (x >= (R)4294967295.0) return 0xFFFFFFFFu;
            return (uint32_t)x;
        };
        u = clamp01(u);
        v = clamp01(std::fabs(v));
        uint32_t i = std::min(as_u32(u * (R)im.w), im.w - 1);
        uint32_t j = std::min(as_u32(v * (R)im.h), im.h - 1);
        sig_mix(0x400u);
        sig_mix(((uint64_t)j << 32) | i);
        const uint8_t* px = &im.rgb[((size_t)j * im.w + i) * 3];
        if constexpr (sizeof(R) == 8) return {(R)px[0] / (R)255.0, (R)px[1] / (R)255.0, (R)px[2] / (R)255.0};
        else return {(R)px[0], (R)px[1], (R)px[2]}; /* f32 spec: bytes here, one product with (1/255)/reflectance in scatter() */
    }
    }
}

/* ---- scene/material.rs ---------------------------------------------------------------------- */
template <class R> static inline R powi5(R x) { /* f64::powi(x,5): x * ((x*x)*(x*x)) (llvm.powi / __powidf2) */
    R x2 = x * x;
    R x4 = x2 * x2;
    return x * x4;
}
template <class R> static inline R schlick(R cosv, R ri) { /* material.rs:173-177 */
    R r0 = ((R)1.0 - ri) / ((R)1.0 + ri);
    r0 = r0 * r0;
    return r0 + ((R)1.0 - r0) * powi5((R)1.0 - cosv);
}
template <class R> static inline V3<R> refract(V3<R> uv, V3<R> n, R e) { /* material.rs:157-166 */
    R cos_theta = std::fmin(dot(-uv, n), (R)1.0);
    V3<R> perp = (uv + n * cos_theta) * e;
    V3<R> par = n * (-std::sqrt(std::fabs((R)1.0 - dot(perp, perp))));
    return perp + par;
}
/* point3.rs:88-93: three U[0,1) draws -> positive octant only, then normalise (reference quirk). */
template <class R> static inline V3<R> random_unit_on_sphere(R x, R y, R z) { return unit(V3<R>{x, y, z}); }

/* RNG contract at bounce b (ctr.z = b >= 1), dimensions 0..3:
 *   Lambertian: test = dim 0, vector = dims (1, 2, 3).  Metal: vector = dims (0, 1, 2).  Dielectric: dim 0. */
template <class R>
static inline bool scatter(const Scene<R>& sc, const Material<R>& m, const Ray<R>& rin, const HitRec<R>& hr, const Draws<R>& dr,
                           V3<R>& att, Ray<R>& out) {
    switch (m.kind) {
    case MIRT_MAT_LAMBERTIAN: { /* material.rs:59-84 */
        R ua = dr.dim(0);
        sig_mix(0x100u + (uint32_t)(ua < m.reflectance));
        if (ua < m.reflectance) {
            R ub = dr.dim(1), uc = dr.dim(2), ud = dr.dim(3);
            V3<R> dir = hr.n + random_unit_on_sphere(ub, uc, ud);
            if (near_zero(dir)) dir = hr.n;
            out = {hr.p, dir, rin.tm};
            if (sizeof(R) == 4 && sc.texs[m.tex].kind == MIRT_TEX_IMAGE && sc.imgs[sc.texs[m.tex].image].h != 0)
                att = texture_color(sc, sc.texs[m.tex], hr.u, hr.v, hr.p) * (((R)1.0 / (R)255.0) / m.reflectance);
            else
                att = texture_color(sc, sc.texs[m.tex], hr.u, hr.v, hr.p) / m.reflectance;
            return true;
        }
        return false;
    }
    case MIRT_MAT_METAL: { /* material.rs:97-114 */
        V3<R> refl = reflect(rin.d, hr.n);
        if (m.has_fuzz) {
            R ua = dr.dim(0), ub = dr.dim(1), uc = dr.dim(2);
            refl = unit(refl) + random_unit_on_sphere(ua, ub, uc) * m.fuzz;
        }
        out = {hr.p, refl, rin.tm};
        att = m.albedo;
        sig_mix(0x110u + (uint32_t)(!m.has_fuzz || dot(out.d, hr.n) > (R)0.0));
        return !m.has_fuzz || dot(out.d, hr.n) > (R)0.0;
    }
    default: { /* DIELECTRIC material.rs:126-154 */
        att = {(R)1.0, (R)1.0, (R)1.0};
        R ri = hr.front ? (R)1.0 / m.ior : m.ior;
        V3<R> ud = unit(rin.d);
        R cos_theta = std::fmin(dot(-ud, hr.n), (R)1.0);
        R sin_theta = std::sqrt((R)1.0 - cos_theta * cos_theta);
        bool refl = ri * sin_theta > (R)1.0;
        if (!refl) { /* `||` short-circuits: the draw happens only when not total internal reflection */
            R ua = dr.dim(0);
            refl = schlick(cos_theta, ri) > ua;
        }
        sig_mix(0x120u + (uint32_t)refl);
        V3<R> dir = refl ? reflect(ud, hr.n) : refract(ud, hr.n, ri);
        out = {hr.p, dir, rin.tm};
        return true;
    }
    }
}

/* ---- camera/camera.rs ----------------------------------------------------------------------- */
template <class R> struct Cam {
    V3<R> lookfrom, du, dv, ul, px00, disk_u, disk_v;
    R defocus_angle;
    uint32_t W, H, spp, depth;
    explicit Cam(const mirt_camera& c)
        : lookfrom(cvt3<R>(c.lookfrom)), du(cvt3<R>(c.px_du)), dv(cvt3<R>(c.px_dv)), ul(cvt3<R>(c.vp_upper_left)),
          px00(cvt3<R>(c.px00_loc)), disk_u(cvt3<R>(c.disk_u_r)), disk_v(cvt3<R>(c.disk_v_r)), defocus_angle((R)c.defocus_angle),
          W(c.img_width), H(c.img_height), spp(c.samples_per_pixel), depth(c.max_bounce_depth) {}
};

/* closest-hit queries of this thread; folded into Scene::segments once per row (a shared atomic in the
 * recursion serialises the worker threads on one cache line) */
static thread_local uint64_t tl_segments = 0;

/* camera.rs:268-286, recursion kept (the product is evaluated innermost-first). */
template <class R>
static V3<R> color(const Scene<R>& sc, const Ray<R>& ray, uint32_t depth, Draws<R> dr, bool use_bvh) {
    if (depth == 0) return {(R)0.0, (R)0.0, (R)0.0};
    HitRec<R> rec;
    ++tl_segments;
    if (sc.hit(ray, Interval<R>{(R)0.001, std::numeric_limits<R>::infinity()}, rec, use_bvh)) {
        V3<R> att{(R)0.0, (R)0.0, (R)0.0};
        Ray<R> sca{};
        sig_mix(((uint64_t)rec.id << 1) | (uint64_t)rec.front);
        dr.bounce += 1;
        if (scatter(sc, sc.mats[sc.spheres[rec.id].mat], ray, rec, dr, att, sca)) {
            V3<R> c = color(sc, sca, depth - 1, dr, use_bvh);
            return {att.x * c.x, att.y * c.y, att.z * c.z};
        }
        return {(R)0.0, (R)0.0, (R)0.0};
    }
    sig_mix(0x7FFFFFFFFull); /* miss */
    V3<R> ud = unit(ray.d);
    R a = (R)0.5 * (ud.y + (R)1.0);
    /* ARgb(1,1,1)*(1-a) + ARgb(0.5,0.7,1.0)*a */
    return {(R)1.0 * ((R)1.0 - a) + (R)0.5 * a, (R)1.0 * ((R)1.0 - a) + (R)0.7 * a, (R)1.0 * ((R)1.0 - a) + (R)1.0 * a};
}

/* The f32 SPEC's pixel arithmetic (DESIGN.md "Two precisions"; no reference counterpart - the reference is f64 only and is
 * restated by color() above).  The throughput kernels carry the running product thr = ((a0 a1) a2) ... of a path's
 * attenuations and multiply the sky colour last, instead of the recursion's a0 (a1 (... sky)); same factors, another
 * association, so the last bits differ.  The f32 instantiation of this oracle can restate THAT order, which lets the
 * tests demand bit-equal f32 radiance from the HIP kernels; the f64 parity mode is held to color()'s order instead. */
template <class R>
static V3<R> color_running_product(const Scene<R>& sc, Ray<R> ray, uint32_t depth, Draws<R> dr, bool use_bvh) {
    V3<R> thr{(R)1.0, (R)1.0, (R)1.0};
    for (;;) {
        if (depth == 0) return {(R)0.0, (R)0.0, (R)0.0};
        HitRec<R> rec;
        ++tl_segments;
        if (!sc.hit(ray, Interval<R>{(R)0.001, std::numeric_limits<R>::infinity()}, rec, use_bvh)) break;
        V3<R> att{(R)0.0, (R)0.0, (R)0.0};
        Ray<R> sca{};
        sig_mix(((uint64_t)rec.id << 1) | (uint64_t)rec.front);
        dr.bounce += 1;
        if (!scatter(sc, sc.mats[sc.spheres[rec.id].mat], ray, rec, dr, att, sca)) return {(R)0.0, (R)0.0, (R)0.0};
        thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
        ray = sca;
        --depth;
    }
    sig_mix(0x7FFFFFFFFull); /* miss */
    V3<R> ud = unit(ray.d);
    R a = (R)0.5 * (ud.y + (R)1.0);
    V3<R> sky = {(R)1.0 * ((R)1.0 - a) + (R)0.5 * a, (R)1.0 * ((R)1.0 - a) + (R)0.7 * a, (R)1.0 * ((R)1.0 - a) + (R)1.0 * a};
    return {thr.x * sky.x, thr.y * sky.y, thr.z * sky.z};
}

/* point3.rs:107-109: theta = Uniform(0, 2 pi), cos / sin.  double: the reference's expression through libm.  float: the f32
 * spec's quarter-turn evaluation (DESIGN.md "Two precisions"; restated from the description in mirt_device.h, operation for
 * operation): x = 4u, k = rint(x), y = (x - k) pi/2, Cephes sinf/cosf kernels on [-pi/4, pi/4], quadrant k mod 4. */
static inline void lens_sincos(double u, double& s, double& c) {
    const double TAU = 6.28318530717958647692528676655900577;
    const double th = u * TAU;
    c = std::cos(th);
    s = std::sin(th);
}
static inline void lens_sincos(float u, float& s, float& c) {
    const float x = u * 4.0f;
    const float k = std::nearbyintf(x); /* round half to even, the default mode */
    const float y = (x - k) * 1.57079632679489661923f;
    const float z = y * y;
    const float sp = std::fmaf(std::fmaf(std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * y, y);
    const float cp = std::fmaf(std::fmaf(std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                               std::fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

/* camera.rs:209-230 with the RNG contract: bounce 0, dims 0,1 -> AA (w,h), dims 2,3 -> disk (r,theta), dim 4 -> time. */
template <class R> static inline Ray<R> ray_for(const Cam<R>& c, uint32_t wn, uint32_t hn, const Draws<R>& dr) {
    R ua = dr.dim(0), ub = dr.dim(1);
    R ox = ua - (R)0.5, oy = ub - (R)0.5; /* Uniform(-0.5, 0.5) */
    V3<R> px = c.px00 + (c.du * ((R)wn + ox)) + (c.dv * ((R)hn + oy));
    V3<R> orig = c.lookfrom;
    if (!(c.defocus_angle <= (R)0.0)) { /* camera.rs:220-224, 261-264; point3.rs:106-110 */
        R uc = dr.dim(2), ud = dr.dim(3);
        R r = std::sqrt(uc);
        R cs, sn;
        lens_sincos(ud, sn, cs);
        R dx = r * cs, dy = r * sn;
        orig = c.lookfrom + (c.disk_u * dx) + (c.disk_v * dy);
    }
    R ue = dr.dim(4);
    return {orig, px - orig, ue};
}
template <class R> static inline Ray<R> primary_ray(const Cam<R>& c, uint32_t wn, uint32_t hn) { /* camera.rs:248-252 */
    V3<R> pc = c.ul + (c.du * (R)wn) + (c.dv * (R)hn);
    return {c.lookfrom, pc - c.lookfrom, (R)0.0};
}

/* rgb.rs:20-36 + utils/math.rs:13-21.  Returns false on NaN (the reference panics). */
template <class R> static inline bool quantize(R x, uint8_t& out) {
    R g = x > (R)0.0 ? std::sqrt(x) : (R)0.0;
    R v = g * (R)254.999;
    if (v != v) { out = 0; return false; }
    R c = v < (R)0.0 ? (R)0.0 : (v > (R)255.0 ? (R)255.0 : v);
    out = (uint8_t)std::round(c); /* f64::round: half away from zero */
    return true;
}

/* spec_chunk == 0: the reference's pixel arithmetic (camera.rs:241-246, recursion order of camera.rs:277).
 * spec_chunk == Q > 0: the throughput kernels' - running product per path, and the pixel as the sum, in chunk order, of the
 * sums of Q consecutive samples (a work item of the HIP kernel; Q = mirt_pass_granularity). */
template <class R> static inline Draws<R> make_draws(uint32_t pix, uint32_t s, uint64_t seed, bool f32_layout);
template <> inline Draws<double> make_draws<double>(uint32_t pix, uint32_t s, uint64_t seed, bool f32_layout) {
    Draws<double> d{pix, s, 0, seed};
    d.f32_layout = f32_layout;
    return d;
}
template <> inline Draws<float> make_draws<float>(uint32_t pix, uint32_t s, uint64_t seed, bool) { return Draws<float>{pix, s, 0, seed}; }

struct Extra { /* oracle_render_ex: uniforms of the f32 layout in the f64 instantiation; per-pixel path signatures */
    bool f32_layout = false;
    uint64_t* sig = nullptr;
};

template <class R>
static int render_rows(const Scene<R>& sc, const Cam<R>& cam, uint64_t seed, bool use_bvh, uint32_t row0, uint32_t row1, R* lin,
                       uint8_t* rgb8, uint32_t spec_chunk = 0, Extra ex = Extra()) {
    int nan = 0;
    const uint64_t seg0 = tl_segments;
    for (uint32_t hn = row0; hn < row1; ++hn)
        for (uint32_t wn = 0; wn < cam.W; ++wn) {
            V3<R> px{(R)0.0, (R)0.0, (R)0.0};
            uint32_t pix = hn * cam.W + wn;
            tl_sig = 0;
            if (spec_chunk > 0) {
                const uint32_t ns = cam.spp > 0 ? cam.spp : 1u;
                for (uint32_t s0 = 0; s0 < ns; s0 += spec_chunk) {
                    V3<R> item{(R)0.0, (R)0.0, (R)0.0};
                    for (uint32_t s = s0; s < s0 + spec_chunk && s < ns; ++s) {
                        Draws<R> dr = make_draws<R>(pix, s, seed, ex.f32_layout);
                        Ray<R> r = cam.spp > 0 ? ray_for(cam, wn, hn, dr) : primary_ray(cam, wn, hn);
                        item = item + color_running_product(sc, r, cam.depth, dr, use_bvh);
                    }
                    px = px + item;
                }
                if (cam.spp > 0) px = px * ((R)1.0 / (R)cam.spp);
            } else if (cam.spp > 0) { /* camera.rs:241-246 */
                for (uint32_t s = 0; s < cam.spp; ++s) {
                    Draws<R> dr = make_draws<R>(pix, s, seed, ex.f32_layout);
                    Ray<R> r = ray_for(cam, wn, hn, dr);
                    px = px + color(sc, r, cam.depth, dr, use_bvh);
                }
                px = px * ((R)1.0 / (R)cam.spp);
            } else { /* camera.rs:247-254 */
                Draws<R> dr = make_draws<R>(pix, 0, seed, ex.f32_layout);
                px = color(sc, primary_ray(cam, wn, hn), cam.depth, dr, use_bvh);
            }
            size_t o = (size_t)pix * 3;
            if (ex.sig) ex.sig[pix] = tl_sig;
            if (lin) { lin[o] = px.x; lin[o + 1] = px.y; lin[o + 2] = px.z; }
            if (rgb8) {
                if (!quantize(px.x, rgb8[o])) ++nan;
                if (!quantize(px.y, rgb8[o + 1])) ++nan;
                if (!quantize(px.z, rgb8[o + 2])) ++nan;
            }
        }
    sc.segments.fetch_add(tl_segments - seg0, std::memory_order_relaxed);
    return nan;
}

template <class R>
static int render_t(const mirt_scene_desc* d, const mirt_camera* c, uint64_t seed, const uint32_t* objects, uint32_t n_objects,
                    int threads, uint32_t row_begin, uint32_t row_end, void* lin_v, uint8_t* rgb8, uint64_t* segments, uint32_t spec_chunk = 0,
                    Extra ex = Extra()) {
    Scene<R> sc;
    sc.load(d);
    bool use_bvh = objects != nullptr && n_objects > 0;
    if (use_bvh) sc.build_bvh(objects, n_objects);
    Cam<R> cam(*c);
    if (row_end == 0 || row_end > cam.H) row_end = cam.H;
    R* lin = (R*)lin_v;
    std::atomic<int> nan{0};
    if (threads <= 1) {
        nan = render_rows(sc, cam, seed, use_bvh, row_begin, row_end, lin, rgb8, spec_chunk, ex);
    } else {
        std::atomic<uint32_t> next{row_begin};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    uint32_t r = next.fetch_add(1);
                    if (r >= row_end) break;
                    nan += render_rows(sc, cam, seed, use_bvh, r, r + 1, lin, rgb8, spec_chunk, ex);
                }
            });
        for (auto& t : pool) t.join();
    }
    if (segments) *segments = sc.segments.load();
    return nan.load();
}

template <class R>
static int primary_hits_t(const mirt_scene_desc* d, const mirt_camera* c, const uint32_t* objects, uint32_t n_objects, uint32_t* id,
                          void* t_v) {
    Scene<R> sc;
    sc.load(d);
    bool use_bvh = objects != nullptr && n_objects > 0;
    if (use_bvh) sc.build_bvh(objects, n_objects);
    Cam<R> cam(*c);
    R* t = (R*)t_v;
    for (uint32_t hn = 0; hn < cam.H; ++hn)
        for (uint32_t wn = 0; wn < cam.W; ++wn) {
            HitRec<R> rec;
            size_t o = (size_t)hn * cam.W + wn;
            if (sc.hit(primary_ray(cam, wn, hn), Interval<R>{(R)0.001, std::numeric_limits<R>::infinity()}, rec, use_bvh)) {
                id[o] = rec.id;
                if (t) t[o] = rec.t;
            } else {
                id[o] = MIRT_MISS_ID;
                if (t) t[o] = std::numeric_limits<R>::infinity();
            }
        }
    return 0;
}

} /* namespace orc */

/* ================================================================================================
 * C interface for the ctypes test harness
 * ============================================================================================== */
extern "C" {

void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc::philox4x32_10(ctr, key, out); }

/* camera.rs:123-206.  Option arguments: pass has_* = 0 for None.  aa: -1 = None (-> 100), 0 = Some(0)
 * (anti-aliasing off), n = Some(n).  Returns 0, or -1 if the image height does not fit u32. */
int oracle_camera_build(const double* lookfrom, const double* lookat, const double* vup, uint32_t img_width, double ratio, int64_t aa,
                        int has_vfov, double vfov, int has_focus, double focus_dist, int has_defocus, double defocus_angle,
                        int has_depth, uint32_t max_depth, mirt_camera* out) {
    using V = orc::V3<double>;
    const double PI = 3.14159265358979323846264338327950288;
    V from = lookfrom ? V{lookfrom[0], lookfrom[1], lookfrom[2]} : V{0.0, 0.0, 0.0};
    V at = lookat ? V{lookat[0], lookat[1], lookat[2]} : V{0.0, 0.0, -1.0};
    V up = vup ? V{vup[0], vup[1], vup[2]} : V{0.0, 1.0, 0.0};
    if (!has_vfov) vfov = PI / 2.0;
    if (!has_focus) focus_dist = 1.0;
    if (!has_defocus) defocus_angle = 0.0;
    uint32_t spp = aa == 0 ? 0u : (aa < 0 ? 100u : (uint32_t)aa);
    if (!has_depth) max_depth = 10;

    uint32_t img_height;
    double hq = (double)img_width / ratio;
    if (hq < 1.0) img_height = 1;
    else {
        if (hq != hq || hq < 0.0 || hq > 4294967295.0) return -1; /* utils/math.rs:5-11 */
        img_height = (uint32_t)hq;                                /* truncating `as u32` */
    }
    double h = std::tan(vfov / 2.0);
    double vp_height = 2.0 * h * focus_dist;
    double vp_width = vp_height * ((double)img_width / (double)img_height);
    V w = orc::unit(from - at);
    V u = orc::cross(up, w); /* NOT normalised (camera.rs:164) */
    V v = orc::cross(w, u);
    V vp_v = (-v) * vp_height;
    V vp_u = u * vp_width;
    V px_du = vp_u / (double)img_width;
    V px_dv = vp_v / (double)img_height;
    V ul = from - (w * focus_dist) - (vp_u + vp_v) * 0.5;
    V px00 = ul + (px_dv + px_du) * 0.5;
    double defocus_radius = focus_dist * std::tan(defocus_angle * 0.5);
    V disk_u = u * defocus_radius, disk_v = v * defocus_radius;

    auto st = [](double* o, V a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; };
    std::memset(out, 0, sizeof(*out));
    st(out->lookfrom, from); st(out->px_du, px_du); st(out->px_dv, px_dv); st(out->vp_upper_left, ul); st(out->px00_loc, px00);
    st(out->disk_u_r, disk_u); st(out->disk_v_r, disk_v);
    out->defocus_angle = defocus_angle;
    out->img_width = img_width; out->img_height = img_height;
    out->samples_per_pixel = spp; out->max_bounce_depth = max_depth;
    return 0;
}

/* objects/n_objects: insertion-order object list (sphere ids, duplicates allowed) -> the reference's
 * BVH traversal is used; NULL/0 -> the brute-force loop. */
int oracle_primary_hits(const mirt_scene_desc* d, const mirt_camera* c, uint32_t precision, const uint32_t* objects, uint32_t n_objects,
                        uint32_t* id, void* t) {
    if (precision == 64) return orc::primary_hits_t<double>(d, c, objects, n_objects, id, t);
    return orc::primary_hits_t<float>(d, c, objects, n_objects, id, t);
}

/* Returns the number of NaN channels met by the quantiser (reference: panic). */
int oracle_render(const mirt_scene_desc* d, const mirt_camera* c, uint32_t precision, uint64_t seed, const uint32_t* objects,
                  uint32_t n_objects, int threads, uint32_t row_begin, uint32_t row_end, void* rgb_linear, uint8_t* rgb8,
                  uint64_t* segments) {
    if (precision == 64) return orc::render_t<double>(d, c, seed, objects, n_objects, threads, row_begin, row_end, rgb_linear, rgb8, segments);
    return orc::render_t<float>(d, c, seed, objects, n_objects, threads, row_begin, row_end, rgb_linear, rgb8, segments);
}

/* The same with the throughput kernels' pixel arithmetic (render_rows, spec_chunk = Q >= 1): what the f32 HIP kernels must
 * reproduce bit for bit on scenes whose paths meet no transcendental function. */
int oracle_render_spec(const mirt_scene_desc* d, const mirt_camera* c, uint32_t precision, uint64_t seed, const uint32_t* objects,
                       uint32_t n_objects, int threads, uint32_t row_begin, uint32_t row_end, uint32_t chunk, void* rgb_linear, uint8_t* rgb8,
                       uint64_t* segments) {
    if (chunk == 0) chunk = 1;
    if (precision == 64) return orc::render_t<double>(d, c, seed, objects, n_objects, threads, row_begin, row_end, rgb_linear, rgb8, segments, chunk);
    return orc::render_t<float>(d, c, seed, objects, n_objects, threads, row_begin, row_end, rgb_linear, rgb8, segments, chunk);
}

/* The general form.  precision 64 + draws 32: THE REFERENCE'S ARITHMETIC ON THE THROUGHPUT MODE'S RANDOM NUMBERS - Real = double,
 * every operation and its order as the Rust source has them (color()'s recursion, the textbook discriminant of
 * sphere.rs:83-90, true divisions, libm's tan/acos/atan2/sin/cos, the BVH walk when `objects` is given), with each uniform
 * the binary32 value the f32 mode draws for the same (pixel, sample, bounce, dimension), widened to double.  An f32 frame of
 * the HIP kernels compared with THIS frame differs by the rounding of binary32 arithmetic and by the f32 spec's own
 * departures only - not by Monte-Carlo noise.  draws 0 = the precision's own layout.  chunk 0 = reference
 * pixel arithmetic, Q >= 1 = the f32 spec's (render_rows).  sig (may be NULL): W*H path signatures (sig_mix). */
int oracle_render_ex(const mirt_scene_desc* d, const mirt_camera* c, uint32_t precision, uint32_t draws, uint64_t seed,
                     const uint32_t* objects, uint32_t n_objects, int threads, uint32_t row_begin, uint32_t row_end, uint32_t chunk,
                     void* rgb_linear, uint8_t* rgb8, uint64_t* segments, uint64_t* sig) {
    orc::Extra ex;
    ex.sig = sig;
    if (precision == 64) {
        ex.f32_layout = draws == 32;
        return orc::render_t<double>(d, c, seed, objects, n_objects, threads, row_begin, row_end, rgb_linear, rgb8, segments, chunk, ex);
    }
    if (draws == 64) return -1; /* binary32 arithmetic on 53-bit uniforms: not a mode anything here needs */
    return orc::render_t<float>(d, c, seed, objects, n_objects, threads, row_begin, row_end, rgb_linear, rgb8, segments, chunk, ex);
}

int oracle_quantize(const void* lin, uint32_t precision, size_t n, uint8_t* out) {
    int nan = 0;
    for (size_t i = 0; i < n; ++i) {
        bool ok = precision == 64 ? orc::quantize(((const double*)lin)[i], out[i]) : orc::quantize(((const float*)lin)[i], out[i]);
        if (!ok) ++nan;
    }
    return nan;
}

/* ---- per-function known-answer hooks (always f64: the reference's arithmetic) ---------------- */

/* out = {hit, t, px,py,pz, nx,ny,nz, front, u, v} */
int oracle_sphere_hit(const mirt_sphere* s, const double* o, const double* dir, double tm, double tmin, double tmax, double* out) {
    mirt_scene_desc d{};
    d.n_spheres = 1; d.spheres = s;
    orc::Scene<double> sc;
    sc.load(&d);
    orc::Ray<double> r{{o[0], o[1], o[2]}, {dir[0], dir[1], dir[2]}, tm};
    orc::HitRec<double> h{};
    bool hit = orc::sphere_hit(sc.spheres[0], 0, r, orc::Interval<double>{tmin, tmax}, h);
    out[0] = hit ? 1.0 : 0.0;
    if (hit) {
        out[1] = h.t; out[2] = h.p.x; out[3] = h.p.y; out[4] = h.p.z; out[5] = h.n.x; out[6] = h.n.y; out[7] = h.n.z;
        out[8] = h.front ? 1.0 : 0.0; out[9] = h.u; out[10] = h.v;
    }
    return hit ? 1 : 0;
}

/* The reference's only unit test (src/scene/aabb.rs:123-137) calls exactly this. */
int oracle_aabb_slab(const double* bmin, const double* bmax, const double* o, const double* dir, int axis, double tmin, double tmax) {
    orc::Aabb<double> b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
    orc::Ray<double> r{{o[0], o[1], o[2]}, {dir[0], dir[1], dir[2]}, 0.0};
    return orc::aabb_slab(b, r, axis, orc::Interval<double>{tmin, tmax}) ? 1 : 0;
}

/* Scatter with explicit uniforms replaced by the contract's draws for (seed,pixel,sample,bounce).
 * hit = {px,py,pz, nx,ny,nz, front, u, v}; out = {scattered?, ax,ay,az, ox,oy,oz, dx,dy,dz, tm} */
int oracle_scatter(const mirt_scene_desc* d, uint32_t material, const double* rin_o, const double* rin_d, double tm, const double* hit,
                   uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, double* out) {
    orc::Scene<double> sc;
    sc.load(d);
    orc::Ray<double> rin{{rin_o[0], rin_o[1], rin_o[2]}, {rin_d[0], rin_d[1], rin_d[2]}, tm};
    orc::HitRec<double> h{};
    h.p = {hit[0], hit[1], hit[2]}; h.n = {hit[3], hit[4], hit[5]}; h.front = hit[6] != 0.0; h.u = hit[7]; h.v = hit[8];
    orc::Draws<double> dr{pixel, sample, bounce, seed};
    orc::V3<double> att{0, 0, 0};
    orc::Ray<double> sca{};
    bool ok = orc::scatter(sc, sc.mats[material], rin, h, dr, att, sca);
    out[0] = ok ? 1.0 : 0.0;
    out[1] = att.x; out[2] = att.y; out[3] = att.z;
    out[4] = sca.o.x; out[5] = sca.o.y; out[6] = sca.o.z; out[7] = sca.d.x; out[8] = sca.d.y; out[9] = sca.d.z; out[10] = sca.tm;
    return ok ? 1 : 0;
}

int oracle_texture_color(const mirt_scene_desc* d, uint32_t texture, double u, double v, const double* p, double* rgb) {
    orc::Scene<double> sc;
    sc.load(d);
    orc::V3<double> c = orc::texture_color(sc, sc.texs[texture], u, v, orc::V3<double>{p[0], p[1], p[2]});
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return 0;
}

/* One uniform of the contract, for tests: dimension `dim` of (seed; pixel, sample, bounce). */
double oracle_draw_dim(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim, uint32_t precision) {
    if (precision == 64) return orc::Draws<double>{pixel, sample, bounce, seed}.dim(dim);
    return (double)orc::Draws<float>{pixel, sample, bounce, seed}.dim(dim);
}

} /* extern "C" */
