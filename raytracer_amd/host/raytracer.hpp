/*
 * raytracer.hpp — header-only C++17 host-side mirror of the reference's scene / camera interface, sitting on
 * the C ABI of include/mirt.h.  Product code: it never touches oracle/, and every pixel is computed in libmirt.so.
 *
 * The reference is a Rust crate; no Rust toolchain exists in this image, so the host side above the C ABI is
 * written in C++ with the reference's names, argument meaning and error behaviour
 * (all file:line are /root/reference/src):
 *   Camera::build / Camera::render               camera/camera.rs:123-206, 232-259
 *   Scene::{add, build_bvh}                      scene/hittable.rs:79-90
 *   Sphere::{new_static, new}                    scene/sphere.rs:28-59
 *   Lambertian::{new, with_texture}, Metal::new, Dielectric::new   scene/material.rs:35-57, 91-95, 120-124
 *   SolidColor::new, CheckerTexture::new, ImageTexture::new        scene/texture.rs:18-21, 36-44, 65-68
 * `Arc<dyn Trait>` becomes std::shared_ptr; `Option<T>` becomes std::optional<T>; `Result<_, E>` becomes an
 * exception of type E; the panic of hittable.rs:93-96 becomes std::logic_error with the same text.
 *
 * Compile the including translation unit with -ffp-contract=off: Camera::build must round like the Rust f64 code.
 */
#ifndef RAYTRACER_AMD_HOST_HPP
#define RAYTRACER_AMD_HOST_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mirt.h"

namespace raytracer {

/* core/point3.rs — only what Camera::build needs, same operation order */
struct Point {
    double x = 0.0, y = 0.0, z = 0.0;
    Point() = default;
    Point(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Point new_(double x, double y, double z) { return Point(x, y, z); }
    Point operator+(const Point& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Point operator-(const Point& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Point operator-() const { return {-x, -y, -z}; }
    Point operator*(double l) const { return {x * l, y * l, z * l}; }
    Point operator/(double l) const { return {x / l, y / l, z / l}; }
    double dot(const Point& o) const { return x * o.x + y * o.y + z * o.z; }
    Point cross(const Point& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    Point unit() const { return *this / std::sqrt(dot(*this)); }
};
using ARgb = Point; /* core/rgb.rs: linear colour triple */

struct MirtError : std::runtime_error {
    int code;
    MirtError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) {
    if (rc != MIRT_OK) throw MirtError(rc, std::string("mirt: ") + mirt_last_error());
}

/* ---- textures (scene/texture.rs) ---- */
struct RgbImage { /* image::RgbImage */
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
struct Texture {
    virtual ~Texture() = default;
};
struct SolidColor : Texture {
    ARgb albedo;
    explicit SolidColor(ARgb a) : albedo(a) {}
    static std::shared_ptr<Texture> new_(ARgb albedo) { return std::make_shared<SolidColor>(albedo); }
};
struct CheckerTexture : Texture {
    double inv_scale; /* scale.recip(), texture.rs:39 */
    ARgb even, odd;
    CheckerTexture(double scale, ARgb e, ARgb o) : inv_scale(1.0 / scale), even(e), odd(o) {}
    static std::shared_ptr<Texture> new_(double scale, ARgb even_color, ARgb odd_color) {
        return std::make_shared<CheckerTexture>(scale, even_color, odd_color);
    }
};
struct ImageTexture : Texture {
    std::shared_ptr<RgbImage> px_colors;
    explicit ImageTexture(std::shared_ptr<RgbImage> i) : px_colors(std::move(i)) {}
    static std::shared_ptr<Texture> new_(std::shared_ptr<RgbImage> img) { return std::make_shared<ImageTexture>(std::move(img)); }
};

/* ---- materials (scene/material.rs) ---- */
struct Material {
    virtual ~Material() = default;
};
struct Lambertian : Material {
    std::shared_ptr<Texture> texture;
    double reflectance;
    Lambertian(std::shared_ptr<Texture> t, double r) : texture(std::move(t)), reflectance(r) {}
    static std::shared_ptr<Material> new_(ARgb albedo, double reflectance) {
        return std::make_shared<Lambertian>(SolidColor::new_(albedo), reflectance);
    }
    static std::shared_ptr<Material> with_texture(std::shared_ptr<Texture> t, double reflectance) {
        return std::make_shared<Lambertian>(std::move(t), reflectance);
    }
};
struct Metal : Material {
    ARgb albedo;
    std::optional<double> fuzz;
    Metal(ARgb a, std::optional<double> f) : albedo(a), fuzz(f) {}
    static std::shared_ptr<Material> new_(ARgb albedo, std::optional<double> fuzz) { return std::make_shared<Metal>(albedo, fuzz); }
};
struct Dielectric : Material {
    double refraction_index;
    explicit Dielectric(double i) : refraction_index(i) {}
    static std::shared_ptr<Material> new_(double refraction_index) { return std::make_shared<Dielectric>(refraction_index); }
};

/* ---- hittables (scene/sphere.rs) ---- */
struct Hittable {
    virtual ~Hittable() = default;
};
struct Sphere : Hittable {
    double r;
    Point center;
    std::optional<Point> moving_center;
    std::shared_ptr<Material> mat;
    Sphere(double r_, Point c, std::optional<Point> mc, std::shared_ptr<Material> m)
        : r(r_), center(c), moving_center(mc), mat(std::move(m)) {}
    static std::shared_ptr<Hittable> new_static(double r, Point center, std::shared_ptr<Material> mat) {
        return std::make_shared<Sphere>(r, center, std::nullopt, std::move(mat));
    }
    static std::shared_ptr<Hittable> new_(double r, Point center, std::optional<Point> moving_center, std::shared_ptr<Material> mat) {
        return std::make_shared<Sphere>(r, center, moving_center, std::move(mat));
    }
};

/* ---- scene (scene/hittable.rs:72-99) ---- */
class Scene {
  public:
    Scene() = default;
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() {
        if (handle_) mirt_scene_destroy(handle_);
    }

    void add(const std::shared_ptr<Hittable>& object) { objects_.push_back(object); }

    /* Idempotent like hittable.rs:86-90.  Where the reference builds its BVH, this freezes the Arc graph into a
     * mirt_scene_desc (one hit id per distinct object, one slot per distinct material / texture / image) and
     * uploads it to HIP device `device`. */
    void build_bvh(int device = 0) {
        if (handle_) return;
        std::map<const void*, uint32_t> sid, mid, tid, iid;
        std::vector<mirt_sphere> sph;
        std::vector<mirt_material> mats;
        std::vector<mirt_texture> texs;
        std::vector<mirt_image> imgs;
        auto set3 = [](double* d, const Point& p) { d[0] = p.x; d[1] = p.y; d[2] = p.z; };
        auto tex_id = [&](const std::shared_ptr<Texture>& t) -> uint32_t {
            auto it = tid.find(t.get());
            if (it != tid.end()) return it->second;
            mirt_texture m{};
            if (auto* s = dynamic_cast<SolidColor*>(t.get())) {
                m.kind = MIRT_TEX_SOLID;
                set3(m.color, s->albedo);
            } else if (auto* c = dynamic_cast<CheckerTexture*>(t.get())) {
                m.kind = MIRT_TEX_CHECKER;
                set3(m.color, c->even);
                set3(m.odd, c->odd);
                m.inv_scale = c->inv_scale;
            } else if (auto* i = dynamic_cast<ImageTexture*>(t.get())) {
                m.kind = MIRT_TEX_IMAGE;
                auto ii = iid.find(i->px_colors.get());
                if (ii == iid.end()) {
                    ii = iid.emplace(i->px_colors.get(), (uint32_t)imgs.size()).first;
                    imgs.push_back({i->px_colors->width, i->px_colors->height, i->px_colors->rgb.data()});
                }
                m.image = ii->second;
            } else {
                throw std::invalid_argument("unknown texture type");
            }
            uint32_t id = (uint32_t)texs.size();
            tid.emplace(t.get(), id);
            texs.push_back(m);
            return id;
        };
        auto mat_id = [&](const std::shared_ptr<Material>& mt) -> uint32_t {
            auto it = mid.find(mt.get());
            if (it != mid.end()) return it->second;
            mirt_material m{};
            if (auto* l = dynamic_cast<Lambertian*>(mt.get())) {
                m.kind = MIRT_MAT_LAMBERTIAN;
                m.texture = tex_id(l->texture);
                m.reflectance = l->reflectance;
            } else if (auto* me = dynamic_cast<Metal*>(mt.get())) {
                m.kind = MIRT_MAT_METAL;
                set3(m.albedo, me->albedo);
                m.has_fuzz = me->fuzz.has_value();
                m.fuzz = me->fuzz.value_or(0.0);
            } else if (auto* d = dynamic_cast<Dielectric*>(mt.get())) {
                m.kind = MIRT_MAT_DIELECTRIC;
                m.ior = d->refraction_index;
            } else {
                throw std::invalid_argument("unknown material type");
            }
            uint32_t id = (uint32_t)mats.size();
            mid.emplace(mt.get(), id);
            mats.push_back(m);
            return id;
        };
        for (const auto& o : objects_) {
            if (sid.count(o.get())) continue; /* the same Arc added twice is one object (main.rs:225) */
            auto* s = dynamic_cast<Sphere*>(o.get());
            if (!s) throw std::invalid_argument("unknown hittable type");
            mirt_sphere m{};
            set3(m.center0, s->center);
            set3(m.center1, s->moving_center.value_or(s->center));
            m.radius = s->r;
            m.material = mat_id(s->mat);
            m.moving = s->moving_center.has_value();
            sid.emplace(o.get(), (uint32_t)sph.size());
            sph.push_back(m);
        }
        mirt_scene_desc d{};
        d.n_spheres = (uint32_t)sph.size();
        d.n_materials = (uint32_t)mats.size();
        d.n_textures = (uint32_t)texs.size();
        d.n_images = (uint32_t)imgs.size();
        d.spheres = sph.data();
        d.materials = mats.data();
        d.textures = texs.data();
        d.images = imgs.data();
        check(mirt_scene_create(&d, device, &handle_));
    }

    const mirt_scene* handle() const {
        if (!handle_) throw std::logic_error("expected bvh to exist when hit called"); /* hittable.rs:93-96 */
        return handle_;
    }
    size_t len() const { return objects_.size(); }

  private:
    std::vector<std::shared_ptr<Hittable>> objects_;
    mirt_scene* handle_ = nullptr;
};

/* ---- camera (camera/camera.rs) ---- */
struct InitError : std::runtime_error { /* camera.rs:26-29 */
    using std::runtime_error::runtime_error;
};
struct RenderError : std::runtime_error { /* camera.rs:21-24: WriteHeader / WritePx */
    using std::runtime_error::runtime_error;
};

inline std::optional<uint32_t> f64_to_u32(double v) { /* utils/math.rs:5-11 */
    if (std::isnan(v) || v < 0.0 || v > 4294967295.0) return std::nullopt;
    return (uint32_t)v;
}

class Camera {
  public:
    mirt_camera pod{};
    uint64_t seed = 0;                 /* key of the counter RNG (the reference seeds from the OS) */
    mirt_precision precision = MIRT_F64;
    bool use_bvh = false;              /* MIRT_FLAG_BVH: bounding-volume tree instead of the brute-force loop (MIRT_F32 only) */
    bool valu_broad = false;           /* MIRT_FLAG_VALU_BROAD: keep the f32 brute-force broad phase on the vector ALU */
    uint32_t chunk_samples = 0;        /* mirt_opts.chunk_samples: lower bound of the samples per work item (0 = the library's choice) */
    mirt_stats last_stats{};

    /* The ten positional Options of camera.rs:123-134 with the defaults of :135-149 and the arithmetic of :151-182. */
    static Camera build(std::optional<Point> lookfrom, std::optional<Point> lookat, std::optional<Point> vup, uint32_t img_width,
                        double ratio, std::optional<uint32_t> aa_samples_per_px, std::optional<double> vfov,
                        std::optional<double> focus_dist, std::optional<double> defocus_angle,
                        std::optional<uint32_t> max_bounce_depth) {
        const double PI = 3.14159265358979323846264338327950288;
        Point from = lookfrom.value_or(Point());
        Point at = lookat.value_or(Point(0.0, 0.0, -1.0));
        Point up = vup.value_or(Point(0.0, 1.0, 0.0));
        double fov = vfov.value_or(PI / 2.0);
        double fd = focus_dist.value_or(1.0);
        double da = defocus_angle.value_or(0.0);
        uint32_t spp = (aa_samples_per_px && *aa_samples_per_px == 0) ? 0u : aa_samples_per_px.value_or(100u);
        uint32_t depth = max_bounce_depth.value_or(10u);

        uint32_t img_height;
        if ((double)img_width / ratio < 1.0) {
            img_height = 1;
        } else {
            auto h = f64_to_u32((double)img_width / ratio);
            if (!h) throw std::overflow_error("img height in pixels should fit u32");
            img_height = *h;
        }
        double h = std::tan(fov / 2.0);
        double vp_height = 2.0 * h * fd;
        double vp_width = vp_height * ((double)img_width / (double)img_height);
        Point w = (from - at).unit();
        Point u = up.cross(w);
        Point v = w.cross(u);
        Point vp_v = -v * vp_height;
        Point vp_u = u * vp_width;
        Point px_du = vp_u / (double)img_width;
        Point px_dv = vp_v / (double)img_height;
        Point ul = from - (w * fd) - (vp_u + vp_v) * 0.5;
        Point px00 = ul + (px_dv + px_du) * 0.5;
        double defocus_radius = fd * std::tan(da * 0.5);

        Camera c;
        auto set3 = [](double* d, const Point& p) { d[0] = p.x; d[1] = p.y; d[2] = p.z; };
        set3(c.pod.lookfrom, from);
        set3(c.pod.px_du, px_du);
        set3(c.pod.px_dv, px_dv);
        set3(c.pod.vp_upper_left, ul);
        set3(c.pod.px00_loc, px00);
        set3(c.pod.disk_u_r, u * defocus_radius);
        set3(c.pod.disk_v_r, v * defocus_radius);
        c.pod.defocus_angle = da;
        c.pod.img_width = img_width;
        c.pod.img_height = img_height;
        c.pod.samples_per_pixel = spp;
        c.pod.max_bounce_depth = depth;
        return c;
    }

    /* the mirt_opts of this camera's calls; everything that can reach the image travels in here (no environment variable) */
    mirt_opts opts() const {
        mirt_opts o{};
        o.precision = precision;
        o.seed = seed;
        o.flags = (use_bvh ? MIRT_FLAG_BVH : 0u) | (valu_broad ? MIRT_FLAG_VALU_BROAD : 0u);
        o.chunk_samples = chunk_samples;
        return o;
    }
    /* samples per work item of this camera's image (mirt_pass_granularity): pass boundaries are multiples of it, and a
     * checkpoint is only valid for the value it was written with */
    uint32_t pass_granularity() const {
        mirt_opts o = opts();
        return mirt_pass_granularity(&pod, &o);
    }

    /* Camera::render without the text emission: quantised bytes, W*H*3. */
    std::vector<uint8_t> render_rgb8(const Scene& scene) {
        std::vector<uint8_t> px((size_t)pod.img_width * pod.img_height * 3);
        mirt_opts o = opts();
        check(mirt_render(scene.handle(), &pod, &o, nullptr, px.data(), &last_stats));
        return px;
    }

    /* Progressive rendering / checkpointing (no counterpart in the reference).  `accum` holds the running
     * radiance sums (W*H*3 floats or doubles, by `precision`) and `done` the samples in them; render_more adds up to
     * `count` samples per pixel (in whole work items) and returns the quantised image so far.  After the last sample the
     * image equals render_rgb8's, bit for bit. */
    struct Progress {
        std::vector<unsigned char> accum; /* raw bytes of the sums: what a checkpoint stores */
        uint32_t done = 0;
    };
    std::vector<uint8_t> render_more(const Scene& scene, Progress& pr, uint32_t count) {
        const size_t n = (size_t)pod.img_width * pod.img_height * 3, rsz = precision == MIRT_F64 ? 8 : 4;
        mirt_opts o = opts();
        const uint32_t g = mirt_pass_granularity(&pod, &o), spp = pod.samples_per_pixel;
        if (g == 0) throw RenderError("progressive rendering needs anti-aliasing samples");
        if (pr.accum.size() != n * rsz) {
            if (pr.done != 0) throw RenderError("checkpoint does not match the image size or precision");
            pr.accum.assign(n * rsz, 0);
        }
        if (pr.done >= spp) throw RenderError("all samples are already rendered");
        uint32_t last = pr.done + std::max(count, 1u);
        last = last >= spp ? spp : std::max((last / g) * g, pr.done + g); /* whole work items; the last pass takes the rest */
        last = std::min(last, spp);
        std::vector<uint8_t> px(n);
        check(mirt_render_pass(scene.handle(), &pod, &o, pr.done, last - pr.done, pr.accum.data(), nullptr, px.data(), &last_stats));
        pr.done = last;
        return px;
    }

    /* the "r g b\n" lines of the frame (rgb.rs:28-36 through Display): decimal digits from a 256-entry table, one buffer, one
     * write - per-pixel formatted output costs twenty times the render of a 1080p frame */
    static std::string ppm_body(const std::vector<uint8_t>& px) {
        struct Dec { char c[3]; uint8_t n; };
        static const std::vector<Dec> table = [] {
            std::vector<Dec> t(256);
            for (int v = 0; v < 256; ++v) {
                char tmp[4];
                int n = std::snprintf(tmp, sizeof tmp, "%d", v);
                t[v].n = (uint8_t)n;
                for (int k = 0; k < n; ++k) t[v].c[k] = tmp[k];
            }
            return t;
        }();
        std::string buf(px.size() * 4, '\0'); /* at most "255 " per byte */
        char* o = &buf[0];
        for (size_t i = 0; i + 2 < px.size(); i += 3) {
            for (int k = 0; k < 3; ++k) {
                const Dec& d = table[px[i + k]];
                for (uint8_t j = 0; j < d.n; ++j) *o++ = d.c[j];
                *o++ = k == 2 ? '\n' : ' ';
            }
        }
        buf.resize((size_t)(o - &buf[0]));
        return buf;
    }
    static void write_ppm(const std::vector<uint8_t>& px, uint32_t w, uint32_t h, std::ostream& out) {
        out << "P3\n" << w << ' ' << h << "\n255\n";
        if (!out) throw RenderError("WriteHeader");
        const std::string buf = ppm_body(px);
        out.write(buf.data(), (std::streamsize)buf.size());
        if (!out) throw RenderError("WritePx");
    }

    /* Camera::render (camera.rs:232-259): "P3\nW H\n255\n" then one "r g b\n" line per pixel. */
    void render(const Scene& scene, std::ostream& out) {
        out << "P3\n" << pod.img_width << ' ' << pod.img_height << "\n255\n";
        if (!out) throw RenderError("WriteHeader");
        const std::string buf = ppm_body(render_rgb8(scene));
        out.write(buf.data(), (std::streamsize)buf.size());
        if (!out) throw RenderError("WritePx");
    }

    /* hit-id / t maps of the deterministic branch (camera.rs:247-254) */
    void primary_hits(const Scene& scene, std::vector<uint32_t>& id, std::vector<double>& t) {
        size_t n = (size_t)pod.img_width * pod.img_height;
        id.resize(n);
        t.resize(n);
        mirt_opts o{};
        o.precision = MIRT_F64;
        check(mirt_primary_hits(scene.handle(), &pod, &o, id.data(), t.data()));
    }
};

} /* namespace raytracer */
#endif
