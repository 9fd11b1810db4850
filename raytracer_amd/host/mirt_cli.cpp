/*
 * mirt_cli.cpp — command-line front end over the C++ host mirror: what the reference's main()
 * (/root/reference/src/main.rs:29-134) does with one hard-coded scene and camera, for all five of its scenes.
 *
 *   mirt_cli render --scene mars|bouncing_balls|scene|scene2|blur_scene [--width 400] [--spp 100] [--depth 50]
 *                   [--seed 1] [--layout-seed 1] [--precision 64|32] [--accel brute|valu|bvh] [--textures DIR] [--out FILE] [--passes N] [--checkpoint FILE] [--stop-after N] [--chunk Q]
 *        PPM P3 on stdout (or FILE), like the reference.  --spp 0 = the no-anti-aliasing branch.
 *   mirt_cli decode IN.jpg OUT.ppm      the image loader alone (binary PPM out)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <iostream>
#include <map>

#include "presets.hpp"

using namespace raytracer;

static int usage() {
    std::cerr << "usage: mirt_cli render --scene NAME [--width W] [--spp N] [--depth D] [--seed S] [--layout-seed S] [--precision 64|32]\n"
                 "                       [--accel brute|valu|bvh] [--textures DIR] [--out FILE] [--passes N] [--checkpoint FILE] [--stop-after N] [--chunk Q]\n"
                 "       mirt_cli decode IN.jpg OUT.ppm\n";
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    try {
        if (cmd == "decode") {
            if (argc != 4) return usage();
            uint32_t w, h;
            std::vector<uint8_t> rgb;
            load_image_to_rgb(argv[2], w, h, rgb);
            std::ofstream o(argv[3], std::ios::binary);
            o << "P6\n" << w << " " << h << "\n255\n";
            o.write((const char*)rgb.data(), (std::streamsize)rgb.size());
            return o ? 0 : 1;
        }
        if (cmd != "render") return usage();
        std::map<std::string, std::string> opt = {{"--scene", "mars"}, {"--width", "400"}, {"--spp", "100"}, {"--depth", "50"}, {"--seed", "1"},
                                                  {"--layout-seed", "1"}, {"--precision", "64"}, {"--accel", "brute"},
                                                  {"--textures", "./presets/textures"}, {"--out", ""},
                                                  {"--passes", "1"}, {"--checkpoint", ""}, {"--stop-after", "0"}, {"--chunk", "0"}};
        for (int i = 2; i < argc; i += 2) {
            if (i + 1 >= argc || !opt.count(argv[i])) return usage();
            opt[argv[i]] = argv[i + 1];
        }
        const uint32_t width = (uint32_t)std::atoi(opt["--width"].c_str()), spp = (uint32_t)std::atoi(opt["--spp"].c_str());
        const uint32_t depth = (uint32_t)std::atoi(opt["--depth"].c_str());
        const std::string name = opt["--scene"];
        Scene scene;
        Camera cam;
        if (name == "mars") {
            presets::mars_texture_scene(scene, opt["--textures"]);
            cam = presets::default_camera(width, spp, depth);
        } else if (name == "bouncing_balls") {
            presets::bouncing_balls_scene(scene, opt["--textures"], std::strtoull(opt["--layout-seed"].c_str(), nullptr, 10));
            cam = presets::default_camera(width, spp, depth, Point(13.0, 2.0, 3.0));
        } else if (name == "scene") {
            presets::scene(scene);
            cam = presets::plain_camera(width, spp, depth);
        } else if (name == "scene2") {
            presets::scene2(scene);
            cam = presets::plain_camera(width, spp, depth);
        } else if (name == "blur_scene") {
            presets::blur_scene(scene);
            cam = presets::plain_camera(width, spp, depth);
        } else {
            std::cerr << "unknown scene " << name << "\n";
            return 2;
        }
        scene.build_bvh();
        cam.seed = std::strtoull(opt["--seed"].c_str(), nullptr, 10);
        cam.precision = opt["--precision"] == "32" ? MIRT_F32 : MIRT_F64;
        cam.use_bvh = opt["--accel"] == "bvh";
        cam.valu_broad = opt["--accel"] == "valu";
        cam.chunk_samples = (uint32_t)std::max(0, std::atoi(opt["--chunk"].c_str())); /* mirt_opts.chunk_samples */
        const uint32_t passes = (uint32_t)std::max(1, std::atoi(opt["--passes"].c_str()));
        const uint32_t stop_after = (uint32_t)std::max(0, std::atoi(opt["--stop-after"].c_str()));
        if (passes > 1 || !opt["--checkpoint"].empty()) {
            /* progressive: `passes` roughly equal passes; after each one the image so far goes to --out (rewritten) and the
             * sums to --checkpoint; a run that finds a checkpoint of the same job continues it (--stop-after N ends the run
             * after N passes, e.g. to exercise resuming) */
            Camera::Progress pr;
            const std::string ck = opt["--checkpoint"];
            /* the samples per work item are part of the job: a pixel's sum is built from per-item sums, so sums written with one
             * item size cannot be continued with another (a checkpoint from a run with a different --chunk is refused) */
            const uint64_t job[9] = {0x4D495254434B5032ull /* "MIRTCKP2" */, cam.pod.img_width, cam.pod.img_height, cam.pod.samples_per_pixel,
                                     cam.pod.max_bounce_depth, cam.seed, (uint64_t)cam.precision, (uint64_t)scene.len(), (uint64_t)cam.pass_granularity()};
            if (!ck.empty()) {
                std::ifstream in(ck, std::ios::binary);
                if (in) {
                    uint64_t head[10];
                    in.read((char*)head, sizeof head);
                    if (!in || std::memcmp(head, job, sizeof job) != 0)
                        throw std::runtime_error("checkpoint " + ck + " belongs to another job (image, samples, seed, precision, scene or samples per work item differ)");
                    pr.done = (uint32_t)head[9];
                    pr.accum.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
                    std::cerr << "resuming " << ck << " at sample " << pr.done << " of " << spp << "\n";
                }
            }
            const uint32_t per = (spp + passes - 1) / passes;
            std::vector<uint8_t> px;
            uint32_t n_done = 0;
            while (pr.done < spp) {
                px = cam.render_more(scene, pr, per);
                ++n_done;
                if (!ck.empty()) {
                    std::ofstream o(ck + ".tmp", std::ios::binary);
                    const uint64_t done = pr.done;
                    o.write((const char*)job, sizeof job);
                    o.write((const char*)&done, 8);
                    o.write((const char*)pr.accum.data(), (std::streamsize)pr.accum.size());
                    o.close();
                    if (!o || std::rename((ck + ".tmp").c_str(), ck.c_str()) != 0) throw std::runtime_error("cannot write checkpoint " + ck);
                }
                if (!opt["--out"].empty()) {
                    std::ofstream o(opt["--out"], std::ios::binary);
                    Camera::write_ppm(px, cam.pod.img_width, cam.pod.img_height, o);
                }
                std::cerr << "pass " << n_done << ": " << pr.done << " / " << spp << " samples per pixel\n";
                if (stop_after && n_done >= stop_after) break;
            }
            if (opt["--out"].empty() && pr.done >= spp) Camera::write_ppm(px, cam.pod.img_width, cam.pod.img_height, std::cout);
        } else if (opt["--out"].empty()) {
            cam.render(scene, std::cout);
        } else {
            std::ofstream o(opt["--out"], std::ios::binary);
            cam.render(scene, o);
        }
        std::cerr << name << ": " << scene.len() << " objects, " << cam.pod.img_width << "x" << cam.pod.img_height << ", segments "
                  << cam.last_stats.segments << ", kernel " << cam.last_stats.kernel_ms << " ms\n";
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
