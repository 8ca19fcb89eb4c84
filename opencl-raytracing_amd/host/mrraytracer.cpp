// mrraytracer -- command-line driver: <scene.json> in, <image.png> out.
//
// Same contract as the reference driver (mrraytracer.cpp:23-148): two
// positional arguments, the scene summary on stdout, "raytracing...done",
// "writing <path>...done", exit status -1 for bad usage, 1 for a scene or PNG
// error, 0 on success.  Rendering goes through the C ABI of libmrt_hip; there
// is no CPU fallback -- without a usable GPU the render step fails with the
// library's error text and exit status 1.
//
// MRT_STATS=1 in the environment adds one JSON line with timings on stderr.
#include "mrt.h"

#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

namespace {

void print_usage() {
    std::cout << "mrraytracer usage:" << std::endl << std::endl
              << "./mrraytracer <INPUT-JSON-SCENE-PATH> <OUTPUT-PNG-PATH>" << std::endl << std::endl;
}

// Vectors print as <x, y, z> with default ostream formatting (gfxalgebra.hpp:184-194).
struct V { const double* p; };
std::ostream& operator<<(std::ostream& os, const V& v) {
    return os << '<' << v.p[0] << ", " << v.p[1] << ", " << v.p[2] << '>';
}
// Colours print through std::to_string(float) (mrraytracer.cpp:53-59).
std::string rgb(const float* c) {
    return "<" + std::to_string(c[0]) + ", " + std::to_string(c[1]) + ", " + std::to_string(c[2]) + ">";
}

void print_scene(const mrt_scene& s, const mrt_scene_info& in) {
    const std::string indent("    "), indent2 = indent + indent;
    std::cout << "scene has:" << std::endl
              << indent << "camera"
              << " eye=" << V{in.eye} << " u=" << V{in.u} << " v=" << V{in.v} << " w=" << V{in.w} << std::endl
              << indent << "viewport"
              << " x_res=" << s.xres << " y_res=" << s.yres << " left=" << in.left << " top=" << in.top
              << " right=" << in.right << " bottom=" << in.bottom << std::endl;
    if (s.projection == MRT_PROJ_ORTHO)
        std::cout << indent << "projection is ORTHOGRAPHIC" << std::endl;
    else
        std::cout << indent << "projection is PERSPECTIVE with"
                  << " focal_length=" << in.focal << std::endl;
    if (s.shader == MRT_SHADER_FLAT) {
        std::cout << indent << "shader is FLAT" << std::endl;
    } else {
        std::cout << indent << "shader is BLINN-PHONG with" << std::endl
                  << indent2 << " ambient_coefficient=" << in.ka << std::endl
                  << indent2 << " ambient_color=" << rgb(s.ambient) << std::endl
                  << indent2 << " diffuse_coefficient=" << in.kd << std::endl
                  << indent2 << " specular_coefficient=" << in.ks << std::endl;
    }
    std::cout << indent << "background=" << rgb(s.background) << std::endl;
    std::cout << indent << "count of lights=" << s.n_lights << std::endl;
    for (uint32_t k = 0; k < s.n_lights; ++k) {
        const float col[3] = {s.light[3][k], s.light[4][k], s.light[5][k]};
        std::cout << indent2 << "location=" << V{in.light_pos + 3 * k} << " color=" << rgb(col)
                  << " intensity=" << in.light_intensity[k] << std::endl;
    }
    std::cout << indent << "count of objects=" << (s.n_spheres + s.n_tris) << std::endl;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 3) {
        print_usage();
        return -1;
    }
    const std::string input_path = argv[1], output_path = argv[2];
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    const auto t_start = clk::now();

    mrt_loaded_scene* loaded = nullptr;
    char err[512];
    if (mrt_scene_load_file(input_path.c_str(), &loaded, err, sizeof err) != MRT_OK) {
        std::cerr << "error loading scene \"" + input_path + "\": " << err << std::endl;
        return 1;
    }
    const mrt_scene* scene = mrt_scene_view(loaded);
    const double load_ms = ms_since(t_start);
    print_scene(*scene, *mrt_scene_info_view(loaded));

    std::cout << "raytracing..." << std::flush;
    mrt_ctx* ctx = nullptr;
    mrt_config cfg = {-1, MRT_FLAG_NONE};
    // MRT_FLAGS=<integer> selects one of the library's other kernels (mrt.h, mrt_flags); same pixels.
    if (const char* f = std::getenv("MRT_FLAGS")) cfg.flags = static_cast<uint32_t>(std::strtoul(f, nullptr, 0));
    const size_t n_bytes = static_cast<size_t>(scene->xres) * scene->yres * 4;
    mrt_stats stats;
    void* pinned = nullptr;
    const auto t_create = clk::now();
    int rc = mrt_create(&cfg, &ctx);
    if (rc == MRT_OK) rc = mrt_host_alloc(n_bytes, &pinned);
    const double create_ms = ms_since(t_create);       // HIP start-up, stream, page-locked frame buffer
    uint8_t* pixels = static_cast<uint8_t*>(pinned);
    if (rc == MRT_OK) rc = mrt_set_scene(ctx, scene);
    if (rc == MRT_OK) rc = mrt_render_rgba8(ctx, pixels, &stats);
    if (rc != MRT_OK) {
        std::cerr << std::endl << "error: render failed: " << mrt_last_error(ctx) << std::endl;
        mrt_host_free(pinned);
        mrt_destroy(ctx);
        mrt_scene_free(loaded);
        return 1;
    }
    std::cout << "done" << std::endl;

    std::cout << "writing " << output_path << "..." << std::flush;
    const auto t_png = clk::now();
    if (mrt_write_png_rgb(output_path.c_str(), pixels, scene->xres, scene->yres) != MRT_OK) {
        std::cerr << std::endl << "error: could not write PNG" << std::endl;
        mrt_host_free(pinned);
        mrt_destroy(ctx);
        mrt_scene_free(loaded);
        return 1;
    }
    std::cout << "done" << std::endl << std::endl;
    const double png_ms = ms_since(t_png);

    const char* want = std::getenv("MRT_STATS");
    if (want && want[0] == '1') {
        std::fprintf(stderr,
                     "{\"rays\": %llu, \"prim_tests\": %llu, \"load_ms\": %.3f, \"create_ms\": %.3f, "
                     "\"upload_ms\": %.4f, \"trace_ms\": %.4f, \"shade_ms\": %.4f, \"download_ms\": %.4f, "
                     "\"png_ms\": %.3f, \"total_ms\": %.3f, \"mrays_per_s\": %.2f}\n",
                     (unsigned long long)stats.rays, (unsigned long long)stats.prim_tests, load_ms, create_ms,
                     stats.upload_ms, stats.trace_ms, stats.shade_ms, stats.download_ms, png_ms, ms_since(t_start),
                     stats.kernel_ms > 0 ? stats.rays / stats.kernel_ms * 1e-3 : 0.0);
    }
    mrt_host_free(pinned);
    mrt_destroy(ctx);
    mrt_scene_free(loaded);
    return 0;
}
