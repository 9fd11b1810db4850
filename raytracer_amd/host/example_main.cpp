/*
 * example_main.cpp — the shape of the reference's main.rs (build scene -> build_bvh -> Camera::build -> render to
 * stdout as PPM, main.rs:136-229) written against the C++ host mirror.  The scene is a small fixed one so that
 * tests/test_gpu_parity.py can build the same scene through the Python mirror and compare the bytes.
 *
 *   g++ -std=c++17 -O2 -ffp-contract=off example_main.cpp -L../csrc -lmirt -Wl,-rpath,'$ORIGIN/../csrc' -o example_main
 *   ./example_main [width] [spp] [depth] [seed] [32|64] > out.ppm
 *   ./example_main camera [width]      (CPU only: dumps Camera::build PODs)
 */
#include <cstdlib>
#include <iostream>

#include "raytracer.hpp"

using namespace raytracer;

static void dump_camera(const mirt_camera& c) {
    const double* groups[] = {c.lookfrom, c.px_du, c.px_dv, c.vp_upper_left, c.px00_loc, c.disk_u_r, c.disk_v_r};
    for (const double* g : groups) std::printf("%a %a %a\n", g[0], g[1], g[2]);
    std::printf("%a\n%u %u %u %u\n", c.defocus_angle, c.img_width, c.img_height, c.samples_per_pixel, c.max_bounce_depth);
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "camera") { /* no GPU needed: prints the POD of Camera::build as hex floats */
        uint32_t w = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 160;
        dump_camera(Camera::build(Point(13.0, 2.0, 3.0), Point(0.0, 0.0, 0.0), Point(0.0, 1.0, 0.0), w, 16.0 / 9.0, 8u,
                                  0.3490658503988659, 10.0, 0.010471975511965976, 10u).pod);
        dump_camera(Camera::build(std::nullopt, std::nullopt, std::nullopt, w, 1.5, std::nullopt, std::nullopt, std::nullopt,
                                  std::nullopt, std::nullopt).pod);
        return 0;
    }
    uint32_t width = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 160;
    uint32_t spp = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 8;
    uint32_t depth = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 10;
    uint64_t seed = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 7;
    int prec = argc > 5 ? std::atoi(argv[5]) : 64;
    try {
        Scene scene;
        auto checker = CheckerTexture::new_(0.32, ARgb(0.2, 0.3, 0.1), ARgb(0.9, 0.9, 0.9));
        auto ground = Sphere::new_static(1000.0, Point(0.0, -1000.0, 0.0), Lambertian::with_texture(checker, 0.9));
        scene.add(ground);
        scene.add(ground); /* same Arc twice: still one object */
        scene.add(Sphere::new_static(1.0, Point(0.0, 1.0, 0.0), Dielectric::new_(1.5)));
        scene.add(Sphere::new_static(1.0, Point(-4.0, 1.0, 0.0), Lambertian::new_(ARgb(0.4, 0.2, 0.1), 0.8)));
        scene.add(Sphere::new_static(1.0, Point(4.0, 1.0, 0.0), Metal::new_(ARgb(0.7, 0.6, 0.5), 0.1)));
        scene.add(Sphere::new_static(0.5, Point(2.0, 0.5, 2.0), Metal::new_(ARgb(0.8, 0.8, 0.9), std::nullopt)));
        scene.add(Sphere::new_(0.3, Point(-1.5, 0.3, 2.5), Point(-1.5, 0.6, 2.5), Lambertian::new_(ARgb(0.1, 0.2, 0.7), 1.0)));
        scene.build_bvh();

        Camera cam = Camera::build(Point(13.0, 2.0, 3.0), Point(0.0, 0.0, 0.0), Point(0.0, 1.0, 0.0), width, 16.0 / 9.0, spp,
                                   0.3490658503988659 /* 20 deg */, 10.0, 0.010471975511965976 /* 0.6 deg */, depth);
        cam.seed = seed;
        cam.precision = prec == 32 ? MIRT_F32 : MIRT_F64;
        cam.render(scene, std::cout);
        std::cerr << "segments " << cam.last_stats.segments << " kernel_ms " << cam.last_stats.kernel_ms << "\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
