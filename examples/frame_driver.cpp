// frame_driver.cpp -- the reference's main() (Raytracer/Raytracer.cpp:448-557 scene set-up,
// :595-752 frame loop with OUTPUTFRAMES) with the per-pixel loop handed to the MI355X library:
// the scene is assembled from the same C++ classes, each frame does redoAABBs + KD build +
// flatten on the host, rtx_scene_create uploads it, rtx_render runs the Monte-Carlo passes,
// rtx_present + writePng stand in for the GL window and saveImage.
//
//   g++ -O2 -std=c++17 -I include -I raytracer_amd/host examples/frame_driver.cpp \
//       -L raytracer_amd/host -lrth_host -L raytracer_amd/csrc -lrtx_hip \
//       -Wl,-rpath,'$ORIGIN/../raytracer_amd/host:$ORIGIN/../raytracer_amd/csrc' -o examples/frame_driver
//   examples/frame_driver [frames=3] [width=640] [height=480] [passes=2] [out=out]
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <memory>
#include <string>
#include <vector>

#include "rtx.h"
#include "scene.hpp"

using namespace rth;

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 3;
    const uint32_t frameX = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 640;
    const uint32_t frameY = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 480;
    const uint32_t passes = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 2;
    const std::string outRoot = argc > 5 ? argv[5] : "out";
    const uint32_t fps = 30;                                            // Raytracer.cpp:568

    if (rtx_init(0) != RTX_OK) { std::fprintf(stderr, "%s\n", rtx_last_error()); return 1; }

    // the live scene of Raytracer.cpp:458-527
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Shape>> owned;
    auto mat = [&](Material* m) { materials.emplace_back(m); return m; };
    Material* white = mat(new Material("PlainWhiteTees"))->setColor(dvec3(1.0, 1.0, 1.0));
    Material* glass = mat(new Material("Glass"));
    Material* mirror = mat(new Material("Mirror"));
    FrameInfo fi;
    auto add = [&](Shape* s) { owned.emplace_back(s); fi.shapes.push_back(s); };
    const dvec3 up(0.0, 1.0, 0.0);
    Vertex a(dvec3(-1000.0, 0.0, -1000.0), up, dvec2(0.0, 0.0)), b(dvec3(1000.0, 0.0, -1000.0), up, dvec2(1.0, 0.0));
    Vertex c(dvec3(1000.0, 0.0, 1000.0), up, dvec2(1.0, 1.0)), d(dvec3(-1000.0, 0.0, 1000.0), up, dvec2(0.0, 1.0));
    add(new Triangle(a, c, b, white));
    add(new Triangle(a, d, c, white));
    add(new Sphere(dvec3(0.0, 2.5, 0.0), 5.0, mirror, noMovement));
    for (dvec3 p : {dvec3(-7.0, 1.5, 7.0), dvec3(-7.0, 1.5, -7.0), dvec3(7.0, 1.5, 7.0), dvec3(7.0, 1.5, -7.0)})
        add(new Sphere(p, 1.5, glass, noMovement));
    add(new Sphere(dvec3(0.0), 1.0, mat(new Material("PlainWhiteTees", dvec3(1.0, 0.0, 0.0))), circle0));
    add(new Sphere(dvec3(0.0), 1.0, mat(new Material("PlainWhiteTees", dvec3(0.0, 1.0, 0.0))), circle1));
    add(new Sphere(dvec3(0.0), 1.0, mat(new Material("PlainWhiteTees", dvec3(0.0, 0.0, 1.0))), circle2));

    // out/<UTC %F-%H-%M-%S>/ (Raytracer.cpp:582-590)
    time_t now;
    std::time(&now);
    char stamp[sizeof "####-##-##-##-##-##"];
    std::strftime(stamp, sizeof stamp, "%F-%H-%M-%S", std::gmtime(&now));
    mkdir(outRoot.c_str(), 0777);
    const std::string dir = outRoot + "/" + stamp;
    mkdir(dir.c_str(), 0777);

    FlatScene flat;
    std::vector<float> drawBuffer((size_t)frameX * frameY * 3);
    std::vector<uint8_t> rgb8(drawBuffer.size());
    for (int frameCounter = 0; frameCounter < frames; ++frameCounter) {
        auto t0 = std::chrono::steady_clock::now();
        fi.currentTime = double(frameCounter) / double(fps);            // Raytracer-Noah-Desktop.cpp:802-803
        AABB sceneBounding = redoAABBs(fi);                             // Raytracer.cpp:634
        flat.build(fi, sceneBounding);                                  // buildKDTree + flatten (:636)
        rtx_scene* scene = nullptr;
        if (rtx_scene_create(&flat.desc(), &scene) != RTX_OK) { std::fprintf(stderr, "%s\n", rtx_last_error()); return 1; }

        rtx_render_desc rd = {};
        rd.integrator = RTX_INTEGRATOR_PATH;
        rd.width = frameX; rd.height = frameY;
        rd.tile_w = frameX; rd.tile_h = frameY;
        rd.pixel_multisample_n = 2;
        rd.light_multisample_n = 1;
        rd.max_depth = 8;
        rd.bias = 1e-4;
        rd.frame = (uint32_t)frameCounter;
        rd.attenuation[0] = 1.0; rd.attenuation[1] = 0.09; rd.attenuation[2] = 0.032;
        rd.camera = makeCamera(dvec3(0.0, 35.0, 50.0), dvec3(0.0, 0.0, 0.0), 90.0);    // Raytracer.cpp:642-660
        std::fill(drawBuffer.begin(), drawBuffer.end(), 0.0f);          // clearBuffers
        rd.first_pass = 0; rd.passes = passes;                          // the loop at Raytracer.cpp:666
        rtx_stats stats;
        if (rtx_render(scene, &rd, drawBuffer.data(), &stats) != RTX_OK) { std::fprintf(stderr, "%s\n", rtx_last_error()); return 1; }
        if (rtx_present(scene, drawBuffer.data(), frameX, frameY, passes, RTX_TONEMAP_CBRT, 1, rgb8.data()) != RTX_OK) {
            std::fprintf(stderr, "%s\n", rtx_last_error());
            return 1;
        }
        rtx_scene_destroy(scene);
        const std::string file = dir + "/" + std::to_string(frameCounter) + ".png";       // Raytracer.cpp:748
        if (!writePng(file.c_str(), rgb8.data(), frameX, frameY)) { std::fprintf(stderr, "cannot write %s\n", file.c_str()); return 1; }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long rays = stats.rays_primary + stats.rays_secondary + stats.rays_shadow;
        std::printf("frame %d: %llu rays, GPU %.2f ms, whole frame %.1f ms -> %s\n", frameCounter, rays, stats.ms_total, ms, file.c_str());
    }
    return 0;
}
