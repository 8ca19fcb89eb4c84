// mrturntable -- many views of one scene: <scene.json> <out-prefix> <frames> [<degrees>].
//
// What the reference cannot do without starting over for every frame
// (mrraytracer.cpp:129-136 renders one frame per process; every stage call
// rebuilds its OpenCL objects, CLHelpers.hpp:79,119,155,196,242,281): the
// scene goes to the GPU once, the camera pans about its up axis from
// -degrees/2 to +degrees/2 (default 30) over the frames, and the frames are
// rendered through mrt_sequence_submit_host -- several in flight, each copied
// to page-locked host memory beside the frames that follow -- then written as
// <out-prefix>NNN.png.  With an odd number of frames the middle one is the
// scene's own view, byte for byte what mrraytracer writes.
//
// Environment: MRT_IN_FLIGHT=<1..4> (default 2), MRT_STATS=1 (one JSON line on stderr).
#include "mrt.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

namespace {

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4 || argc > 5) {
        std::cout << "mrturntable usage:" << std::endl << std::endl
                  << "./mrturntable <INPUT-JSON-SCENE-PATH> <OUTPUT-PNG-PREFIX> <FRAMES> [<DEGREES>]" << std::endl << std::endl;
        return -1;
    }
    const std::string scene_path = argv[1], prefix = argv[2];
    const int n_frames = std::atoi(argv[3]);
    const double degrees = argc == 5 ? std::atof(argv[4]) : 30.0;
    if (n_frames < 1 || n_frames > 999) {
        std::cerr << "error: the number of frames must be 1 .. 999" << std::endl;
        return -1;
    }

    char err[512];
    mrt_loaded_scene* loaded = nullptr;
    if (mrt_scene_load_file(scene_path.c_str(), &loaded, err, sizeof err) != MRT_OK) {
        std::cerr << "error loading scene \"" << scene_path << "\": " << err << std::endl;
        return 1;
    }
    const mrt_scene* s = mrt_scene_view(loaded);
    const mrt_scene_info* info = mrt_scene_info_view(loaded);

    // The views: the camera basis turned about v (the basis is built in double, as the loader
    // builds it, gfxraytrace.hpp:881-889, and cast to float at the end).
    std::vector<mrt_camera> views((size_t)n_frames);
    for (int k = 0; k < n_frames; ++k) {
        const double frac = n_frames > 1 ? (double)k / (n_frames - 1) - 0.5 : 0.0;
        const double a = frac * degrees * 3.14159265358979323846 / 180.0;
        const double ca = std::cos(a), sa = std::sin(a);
        mrt_camera& c = views[(size_t)k];
        for (int i = 0; i < 3; ++i) {
            c.eye[i] = s->eye[i];
            c.v[i] = s->v[i];
            // exactly the scene's own u and w at angle 0
            c.u[i] = a == 0.0 ? s->u[i] : (float)(ca * info->u[i] - sa * info->w[i]);
            c.w[i] = a == 0.0 ? s->w[i] : (float)(sa * info->u[i] + ca * info->w[i]);
        }
        c.xres = s->xres; c.yres = s->yres;
        c.left = s->left; c.right = s->right; c.top = s->top; c.bottom = s->bottom;
        c.projection = s->projection;
        c.focal = s->focal;
    }

    const char* env_flight = std::getenv("MRT_IN_FLIGHT");
    mrt_sequence_config qc = {-1, MRT_FLAG_NONE, env_flight ? (uint32_t)std::atoi(env_flight) : 0u};
    mrt_sequence* q = nullptr;
    int rc = mrt_sequence_create(&qc, &q);
    if (rc != MRT_OK) {
        std::cerr << "error: " << mrt_sequence_last_error(nullptr) << std::endl;
        mrt_scene_free(loaded);
        return 1;
    }
    auto fail = [&](const char* what) {
        std::cerr << "error: " << what << ": " << mrt_sequence_last_error(q) << std::endl;
        mrt_sequence_destroy(q);
        mrt_scene_free(loaded);
        return 1;
    };
    if (mrt_sequence_set_scene(q, s) != MRT_OK) return fail("scene");

    const size_t frame_bytes = (size_t)s->xres * s->yres * 4;
    std::vector<void*> frames((size_t)n_frames, nullptr);
    for (auto& f : frames)
        if (mrt_host_alloc(frame_bytes, &f) != MRT_OK) return fail("page-locked frame buffer");

    std::cout << "raytracing " << n_frames << " views...";
    std::cout.flush();
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n_frames; ++k)
        if (mrt_sequence_submit_host(q, &views[(size_t)k], (uint8_t*)frames[(size_t)k]) != MRT_OK) return fail("frame");
    if (mrt_sequence_wait(q) != MRT_OK) return fail("wait");
    const double render_ms = ms_since(t0);
    std::cout << "done" << std::endl;

    const auto t1 = std::chrono::steady_clock::now();
    for (int k = 0; k < n_frames; ++k) {
        char name[16];
        std::snprintf(name, sizeof name, "%03d.png", k);
        const std::string path = prefix + name;
        if (mrt_write_png_rgb(path.c_str(), (const uint8_t*)frames[(size_t)k], s->xres, s->yres) != MRT_OK) {
            std::cerr << "error: could not write PNG " << path << std::endl;
            return 1;
        }
    }
    const double png_ms = ms_since(t1);
    std::cout << "wrote " << prefix << "000.png .. " << n_frames - 1 << std::endl;

    if (std::getenv("MRT_STATS"))
        std::fprintf(stderr,
                     "{\"frames\": %d, \"xres\": %u, \"yres\": %u, \"in_flight\": %u, \"render_ms_per_frame\": %.4f, "
                     "\"mrays_per_s\": %.1f, \"png_ms_per_frame\": %.3f}\n",
                     n_frames, s->xres, s->yres, mrt_sequence_in_flight(q), render_ms / n_frames,
                     (double)s->xres * s->yres * n_frames / render_ms * 1e-3, png_ms / n_frames);
    for (void* f : frames) mrt_host_free(f);
    mrt_sequence_destroy(q);
    mrt_scene_free(loaded);
    return 0;
}
