/*
 * presets.hpp — the reference's hard-coded scenes (/root/reference/src/main.rs) for the C++ host mirror.
 * Layouts that the reference draws from an OS-seeded RNG (bouncing_balls_scene, main.rs:199-221) are seeded here with
 * SplitMix64, draw for draw the same as raytracer_amd/presets.py, so that both front ends build identical scenes.
 */
#ifndef RAYTRACER_AMD_PRESETS_HPP
#define RAYTRACER_AMD_PRESETS_HPP

#include "jpeg_decode.hpp"
#include "raytracer.hpp"

namespace raytracer {
namespace presets {

static const char* MARS_TEXTURE = "mars_1k_color.jpg"; /* main.rs:25 */
static const char* EARTH_TEXTURE = "earthmap.jpg";     /* main.rs:26 */

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next_u64() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double u01() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
    double range(double lo, double hi) { return lo + (hi - lo) * u01(); }
};

/* image_loader.rs:5-8.  `dir` plays TEXTURES_PATH (main.rs:27); a pre-decoded "<stem>.ppm" next to it is accepted
 * when the .jpg itself is not there (this repository ships the two textures decoded, not the reference's files). */
inline std::shared_ptr<RgbImage> load_image_to_rgb(const std::string& dir, const std::string& name) {
    auto img = std::make_shared<RgbImage>();
    std::string path = dir + "/" + name;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (f) std::fclose(f);
    else path = dir + "/" + name.substr(0, name.rfind('.')) + ".ppm";
    raytracer::load_image_to_rgb(path, img->width, img->height, img->rgb);
    return img;
}

inline const double PI = 3.14159265358979323846264338327950288;
inline double radians(double deg) { return deg * (PI / 180.0); }

/* main.rs:30-56: the camera main() builds; `lookfrom` (0,0,12) for the mars scene, (13,2,3) for the bouncing balls */
inline Camera default_camera(uint32_t w, uint32_t spp, uint32_t depth, Point lookfrom = Point(0.0, 0.0, 12.0)) {
    return Camera::build(lookfrom, Point(0.0, 0.0, 0.0), Point(0.0, 1.0, 0.0), w, 16.0 / 9.0, spp, radians(25.0), 10.0, radians(0.4), depth);
}
/* the all-defaults camera (camera.rs:135-149) the three small scenes were written for: origin, looking down -z, 90 deg */
inline Camera plain_camera(uint32_t w, uint32_t spp, uint32_t depth) {
    return Camera::build(std::nullopt, std::nullopt, std::nullopt, w, 16.0 / 9.0, spp, std::nullopt, std::nullopt, std::nullopt, depth);
}

inline void mars_texture_scene(Scene& scene, const std::string& textures) { /* main.rs:136-149 */
    auto tex = ImageTexture::new_(load_image_to_rgb(textures, EARTH_TEXTURE)); /* NB: the "mars" scene loads the EARTH texture */
    scene.add(Sphere::new_static(1.8, Point(0.0, 0.0, 0.0), Lambertian::with_texture(tex, 1.0)));
}

inline void bouncing_balls_scene(Scene& scene, const std::string& textures, uint64_t seed) { /* main.rs:151-229 */
    SplitMix64 rng(seed);
    auto mars_material = Lambertian::with_texture(ImageTexture::new_(load_image_to_rgb(textures, MARS_TEXTURE)), 1.0);
    const Point near3(-4.0, 1.0, 0.0), near1(4.0, 1.0, 0.0), near2(0.0, 1.0, 0.0);
    auto mars_surface = Sphere::new_static(1.2, near1, mars_material);
    auto ground_material = Lambertian::with_texture(CheckerTexture::new_(0.32, ARgb(0.2, 0.3, 0.1), ARgb(0.9, 0.9, 0.9)), 1.0);
    auto ground_sphere = Sphere::new_static(1000.0, Point(0.0, -1000.0, 0.0), ground_material);
    auto glass_mat = Dielectric::new_(1.5);
    auto glass_sphere = Sphere::new_static(1.0, near2, glass_mat);
    auto metallic_sphere = Sphere::new_static(1.0, near3, Metal::new_(ARgb(0.7, 0.6, 0.5), std::nullopt));
    scene.add(glass_sphere);
    scene.add(mars_surface);
    scene.add(metallic_sphere);
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const double choose_mat = rng.u01();
            const double cx = (double)a + 0.9 * rng.u01();
            const double cz = (double)b + 0.9 * rng.u01();
            const Point center(cx, 0.2, cz);
            const double dx = center.x - 4.0, dy = center.y - 0.2, dz = center.z - 0.0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {
                std::optional<Point> center2;
                std::shared_ptr<Material> mat;
                if (choose_mat < 0.8) {
                    const double a1x = rng.u01(), a1y = rng.u01(), a1z = rng.u01();
                    const double a2x = rng.u01(), a2y = rng.u01(), a2z = rng.u01();
                    const double reflectance = rng.range(0.8, 1.0);
                    center2 = Point(center.x + 0.0, center.y + rng.range(0.0, 0.5), center.z + 0.0);
                    mat = Lambertian::new_(ARgb(a1x * a2x, a1y * a2y, a1z * a2z), reflectance);
                } else if (choose_mat < 0.95) {
                    const double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), bl = rng.range(0.5, 1.0);
                    const double fuzz = rng.range(0.0, 0.5);
                    mat = Metal::new_(ARgb(r, g, bl), fuzz);
                } else {
                    mat = glass_mat;
                }
                scene.add(Sphere::new_(0.2, center, center2, mat));
            }
        }
        scene.add(ground_sphere); /* the same Arc, 22 times (main.rs:225) */
    }
}

/* main.rs:74-121: the three small scenes main() builds and never renders */
inline void small_scene_materials(std::shared_ptr<Material>& ground, std::shared_ptr<Material>& center, std::shared_ptr<Material>& left,
                                  std::shared_ptr<Material>& bubble, std::shared_ptr<Material>& right) {
    ground = Lambertian::new_(ARgb(0.8, 0.8, 0.1), 1.0);
    center = Lambertian::new_(ARgb(0.7, 0.2, 0.8), 1.0);
    left = Dielectric::new_(1.5);
    bubble = Dielectric::new_(1.0 / 1.5);
    right = Metal::new_(ARgb(0.8, 0.6, 0.2), 0.5);
}
inline void scene2(Scene& scene) { /* main.rs:74-87 */
    const double r = std::cos(PI / 4.0);
    scene.add(Sphere::new_static(r, Point(-r, 0.0, -1.0), Lambertian::new_(ARgb(1.0, 1.0, 0.0), 1.0)));
    scene.add(Sphere::new_static(r, Point(r, 0.0, -1.0), Lambertian::new_(ARgb(0.0, 1.0, 1.0), 1.0)));
}
inline void scene(Scene& sc) { /* main.rs:95-107 */
    std::shared_ptr<Material> g, c, l, b, r;
    small_scene_materials(g, c, l, b, r);
    sc.add(Sphere::new_static(100.0, Point(0.0, -100.5, -1.0), g));
    sc.add(Sphere::new_static(0.5, Point(0.0, 0.0, -3.2), c));
    sc.add(Sphere::new_static(0.5, Point(-1.0, 0.0, -2.5), l));
    sc.add(Sphere::new_static(0.3, Point(-1.0, 0.0, -2.5), b));
    sc.add(Sphere::new_static(0.4, Point(1.0, 0.0, -1.0), r));
    sc.add(Sphere::new_static(0.4, Point(6.0, 0.0, -10.0), r));
    sc.add(Sphere::new_static(0.5, Point(-3.0, 0.0, -1.6), c));
}
inline void blur_scene(Scene& sc) { /* main.rs:109-121 */
    std::shared_ptr<Material> g, c, l, b, r;
    small_scene_materials(g, c, l, b, r);
    sc.add(Sphere::new_static(100.0, Point(0.0, -100.5, -1.0), g));
    sc.add(Sphere::new_static(0.5, Point(0.0, 0.0, -1.2), c));
    sc.add(Sphere::new_static(0.5, Point(-1.0, 0.0, -1.0), l));
    sc.add(Sphere::new_static(0.3, Point(-1.0, 0.0, -1.0), b));
    sc.add(Sphere::new_static(0.4, Point(1.0, 0.0, -1.0), r));
    sc.add(Sphere::new_static(0.4, Point(6.0, 0.0, -10.0), r));
    sc.add(Sphere::new_static(0.5, Point(-3.0, 0.0, -1.6), c));
}

} /* namespace presets */
} /* namespace raytracer */
#endif
