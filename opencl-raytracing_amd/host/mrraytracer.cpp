// mrraytracer -- command-line driver: <scene.json> in, <image.png> out.
//
// Same contract as the reference driver (mrraytracer.cpp:23-148): two
// positional arguments, the scene summary on stdout, "raytracing...done",
// "writing <path>...done", exit status -1 for bad usage, 1 for a scene or PNG
// error, 0 on success.  Rendering goes through the C ABI of libmrt_hip; there
// is no CPU fallback -- without a usable GPU the render step fails with the
// library's error text and exit status 1.
//
// Environment (all optional; the two positional arguments are the whole contract):
//   MRT_STATS=1        one JSON line with timings on stderr (create_wait_ms: what of the GPU start-up,
//                      which runs on a thread of its own beside the reading of the scene, was left to wait for)
//   MRT_FLAGS=<int>    mrt_config.flags: one of the library's other brute-force kernels, same pixels
//   MRT_ACCEL=grid|cull|block   render through mrt_accel_* (screen-space grid, tile cull, block
//                      reject) instead of the brute-force context: same pixels; one GPU only
//   MRT_GPUS=<n|all>   render on n GPUs of this node at once (interleaved row bands, mrt_multi_*)
//   MRT_GPU_LIST=a,b,. the HIP ordinals of the ranks instead of 0 .. n-1; an ordinal may repeat
//   MRT_GATHER=rccl|peer|host   how the bands reach the host image (default: RCCL if it can)
//   MRT_REPEAT=<k>     render the frame k more times and report the fastest (with MRT_STATS)
//   HSA_ENABLE_SDMA=1  (a setting of the ROCm runtime) copies through the DMA engines; the driver
//                      sets it to 0 for itself unless the caller has set it
#include "mrt.h"

#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
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

    // What the environment asks for (a mistake in it is reported where the render would start).
    std::string env_error;
    uint32_t flags = MRT_FLAG_NONE;
    if (const char* f = std::getenv("MRT_FLAGS")) flags = static_cast<uint32_t>(std::strtoul(f, nullptr, 0));
    // How many GPUs, and which.
    std::vector<int32_t> gpu_list;
    if (const char* l = std::getenv("MRT_GPU_LIST")) {
        for (const char* q = l; *q;) {
            char* end = nullptr;
            const long v = std::strtol(q, &end, 10);
            if (end == q) break;
            gpu_list.push_back(static_cast<int32_t>(v));
            q = (*end == ',') ? end + 1 : end;
        }
    }
    uint32_t n_gpus = gpu_list.empty() ? 1u : static_cast<uint32_t>(gpu_list.size());
    bool multi = !gpu_list.empty();
    if (const char* g = std::getenv("MRT_GPUS")) {
        if (gpu_list.empty()) n_gpus = std::strcmp(g, "all") == 0 ? 0u : static_cast<uint32_t>(std::strtoul(g, nullptr, 10));
        multi = multi || n_gpus != 1;
    }
    uint32_t gather = MRT_GATHER_AUTO;
    if (const char* g = std::getenv("MRT_GATHER")) {
        if (!std::strcmp(g, "rccl")) gather = MRT_GATHER_RCCL;
        else if (!std::strcmp(g, "peer")) gather = MRT_GATHER_PEER;
        else if (!std::strcmp(g, "host")) gather = MRT_GATHER_HOST;
    }
    int repeat = 0;
    if (const char* r = std::getenv("MRT_REPEAT")) repeat = std::atoi(r);
    int accel_method = -1;
    if (const char* m = std::getenv("MRT_ACCEL")) {
        if (!std::strcmp(m, "grid")) accel_method = MRT_ACCEL_GRID;
        else if (!std::strcmp(m, "cull")) accel_method = MRT_ACCEL_TILE_CULL;
        else if (!std::strcmp(m, "block")) accel_method = MRT_ACCEL_BLOCK_REJECT;
        else env_error = "error: MRT_ACCEL must be grid, cull or block";
        if (env_error.empty() && multi) env_error = "error: MRT_ACCEL renders on one GPU; unset MRT_GPUS / MRT_GPU_LIST";
    }
    const bool accel = accel_method >= 0 && env_error.empty();

    // A process that renders one frame gains nothing from the DMA engines and pays for bringing
    // them up: the first copy in either direction costs 5-6 ms when it goes through one (teatime
    // at 400x400: 42 -> 32 ms in all, at 3840x2160 60 -> 47; the 1 GiB frame of the 16384^2 soup
    // leaves the GPU as fast through copy kernels).  Unless the caller has said otherwise, copies
    // run as kernels.  (A setting of the runtime for this process only; the library itself
    // leaves it to its host application.)
    // Only with one GPU: copies between GPUs are left as the runtime does them.
    if (!multi) setenv("HSA_ENABLE_SDMA", "0", 0);

    // The GPU side of the start-up (HIP runtime, a hardware queue, the context: 23 ms) runs on a
    // thread of its own while this one reads the scene file and prints the summary.
    mrt_ctx* ctx = nullptr;
    mrt_multi* mctx = nullptr;
    mrt_accel* actx = nullptr;
    int create_rc = MRT_OK;
    std::string create_why;
    double ctx_create_ms = 0.0;
    std::thread creator;
    if (env_error.empty())
        creator = std::thread([&]() {
            const auto t_create = clk::now();
            if (multi) {
                mrt_multi_config mc = {n_gpus, gpu_list.empty() ? nullptr : gpu_list.data(), flags, 0u, gather};
                create_rc = mrt_multi_create(&mc, &mctx);
                if (create_rc != MRT_OK) create_why = mrt_multi_last_error(nullptr);
            } else if (accel) {
                mrt_accel_config ac = {-1, static_cast<uint32_t>(accel_method), 0u};
                create_rc = mrt_accel_create(&ac, &actx);
                if (create_rc != MRT_OK) create_why = mrt_accel_last_error(nullptr);
            } else {
                // (the CLI never renders through the entry point that allocates nothing: its 64 MiB array stays off the GPU)
                mrt_config cfg = {-1, flags | MRT_FLAG_DEFER_NOALLOC_SETUP};
                create_rc = mrt_create(&cfg, &ctx);
                if (create_rc != MRT_OK) create_why = mrt_last_error(nullptr);     // (the text is kept per thread)
            }
            ctx_create_ms = ms_since(t_create);       // HIP start-up, streams
        });
    auto join_and_destroy = [&]() {
        if (creator.joinable()) creator.join();
        mrt_destroy(ctx);
        mrt_multi_destroy(mctx);
        mrt_accel_destroy(actx);
    };

    mrt_loaded_scene* loaded = nullptr;
    char err[512];
    if (mrt_scene_load_file(input_path.c_str(), &loaded, err, sizeof err) != MRT_OK) {
        std::cerr << "error loading scene \"" + input_path + "\": " << err << std::endl;
        join_and_destroy();
        return 1;
    }
    const mrt_scene* scene = mrt_scene_view(loaded);
    const double load_ms = ms_since(t_start);
    print_scene(*scene, *mrt_scene_info_view(loaded));

    std::cout << "raytracing..." << std::flush;
    if (!env_error.empty()) {
        std::cerr << std::endl << env_error << std::endl;
        mrt_scene_free(loaded);
        return 1;
    }

    const size_t n_bytes = static_cast<size_t>(scene->xres) * scene->yres * 4;
    mrt_stats stats{};
    mrt_multi_stats mstats{};
    mrt_accel_stats astats{};
    void* pinned = nullptr;
    double best_ms = 0.0;
    const auto t_wait = clk::now();
    creator.join();
    const double create_wait_ms = ms_since(t_wait);     // what of the start-up the scene file did not hide
    int rc = create_rc;
    const auto t_alloc = clk::now();
    if (rc == MRT_OK) rc = mrt_host_alloc(n_bytes, &pinned);
    const double create_ms = ctx_create_ms + ms_since(t_alloc);       // ... and the page-locked frame buffer
    uint8_t* pixels = static_cast<uint8_t*>(pinned);
    // The PNG is encoded while the frame arrives: the brute-force paths hand over rows as they
    // land in host memory (strips of one GPU's frame, bands of several GPUs'), and the encoder's
    // threads deflate every stripe of rows that is whole, under the copies and kernels still running.
    mrt_png_stream* png = nullptr;
    auto rows_ready = [](void* user, uint32_t row_begin, uint32_t row_end) {
        (void)mrt_png_stream_rows(static_cast<mrt_png_stream*>(user), row_begin, row_end);
    };
    auto render_first = [&]() {
        if (accel) {
            const int r = mrt_accel_render_rgba8(actx, pixels, &astats);
            stats = astats.frame;
            return r;
        }
        if (mrt_png_stream_begin(output_path.c_str(), pixels, scene->xres, scene->yres, &png) != MRT_OK) png = nullptr;
        mrt_rows_ready_fn fn = png ? static_cast<mrt_rows_ready_fn>(rows_ready) : nullptr;
        return multi ? mrt_multi_render_rgba8_rows(mctx, pixels, fn, png, &mstats)
                     : mrt_render_rgba8_rows(ctx, pixels, 0u, fn, png, &stats);
    };
    auto render_once = [&]() {
        if (multi) return mrt_multi_render_rgba8(mctx, pixels, &mstats);
        if (!accel) return mrt_render_rgba8(ctx, pixels, &stats);
        const int r = mrt_accel_render_rgba8(actx, pixels, &astats);
        stats = astats.frame;
        return r;
    };
    if (rc == MRT_OK) rc = multi ? mrt_multi_set_scene(mctx, scene) : accel ? mrt_accel_set_scene(actx, scene) : mrt_set_scene(ctx, scene);
    const auto t_first = clk::now();
    if (rc == MRT_OK) rc = render_first();
    const double first_frame_ms = ms_since(t_first);
    for (int k = 0; k < repeat && rc == MRT_OK; ++k) {
        const auto t_frame = clk::now();
        rc = render_once();
        const double ms = ms_since(t_frame);
        if (k == 0 || ms < best_ms) best_ms = ms;
    }
    auto cleanup = [&]() {
        if (png) (void)mrt_png_stream_end(png);        // (an error path: whatever it writes is not the frame)
        png = nullptr;
        mrt_host_free(pinned);
        mrt_destroy(ctx);
        mrt_multi_destroy(mctx);
        mrt_accel_destroy(actx);
        mrt_scene_free(loaded);
    };
    if (rc != MRT_OK) {
        const std::string why = create_rc != MRT_OK ? create_why
                                : multi ? mrt_multi_last_error(mctx) : accel ? mrt_accel_last_error(actx) : mrt_last_error(ctx);
        std::cerr << std::endl << "error: render failed: " << why << std::endl;
        cleanup();
        return 1;
    }
    std::cout << "done" << std::endl;

    std::cout << "writing " << output_path << "..." << std::flush;
    const auto t_png = clk::now();
    // With MRT_REPEAT the buffer has been rendered into again: the same bytes, but the stream's
    // rows were reported for the first frame, so the plain encoder writes the file.
    int png_rc;
    if (png && repeat == 0) {
        png_rc = mrt_png_stream_end(png);
        png = nullptr;
    } else {
        if (png) (void)mrt_png_stream_end(png);
        png = nullptr;
        png_rc = mrt_write_png_rgb(output_path.c_str(), pixels, scene->xres, scene->yres);
    }
    if (png_rc != MRT_OK) {
        std::cerr << std::endl << "error: could not write PNG" << std::endl;
        cleanup();
        return 1;
    }
    std::cout << "done" << std::endl << std::endl;
    const double png_ms = ms_since(t_png);

    const char* want = std::getenv("MRT_STATS");
    if (want && want[0] == '1') {
        const unsigned long long rays = static_cast<unsigned long long>(scene->xres) * scene->yres;
        if (multi) {
            static const char* const names[] = {"auto", "rccl", "peer", "host"};
            std::fprintf(stderr,
                         "{\"rays\": %llu, \"n_gpus\": %u, \"gather\": \"%s\", \"note\": \"%s\", \"load_ms\": %.3f, "
                         "\"first_frame_ms\": %.3f, \"create_ms\": %.3f, \"create_wait_ms\": %.3f, \"render_ms_max\": %.4f, \"render_ms_sum\": %.4f, \"frame_ms\": %.4f, "
                         "\"best_frame_ms\": %.4f, \"png_ms\": %.3f, \"total_ms\": %.3f, \"mrays_per_s\": %.2f}\n",
                         rays, mstats.n_gpus, names[mstats.gather & 3u], mrt_multi_note(mctx), load_ms, first_frame_ms, create_ms, create_wait_ms,
                         mstats.render_ms_max, mstats.render_ms_sum, mstats.total_ms, best_ms, png_ms,
                         ms_since(t_start), (repeat > 0 && best_ms > 0) ? rays / best_ms * 1e-3 : rays / mstats.total_ms * 1e-3);
        } else {
            std::fprintf(stderr,
                         "{\"rays\": %llu, \"prim_tests\": %llu, \"load_ms\": %.3f, \"first_frame_ms\": %.3f, \"create_ms\": %.3f, \"ctx_create_ms\": %.3f, \"create_wait_ms\": %.3f, \"host_alloc_ms\": %.3f, "
                         "\"upload_ms\": %.4f, \"trace_ms\": %.4f, \"shade_ms\": %.4f, \"download_ms\": %.4f, "
                         "\"best_frame_ms\": %.4f, \"png_ms\": %.3f, \"total_ms\": %.3f, \"mrays_per_s\": %.2f, "
                         "\"kernel\": %u, \"grid_build_ms\": %.4f, \"grid_cells\": %llu, \"grid_refs\": %llu}\n",
                         (unsigned long long)stats.rays, (unsigned long long)stats.prim_tests, load_ms, first_frame_ms, create_ms, ctx_create_ms, create_wait_ms, create_ms - ctx_create_ms,
                         stats.upload_ms, stats.trace_ms, stats.shade_ms, stats.download_ms, best_ms, png_ms,
                         ms_since(t_start), stats.kernel_ms > 0 ? stats.rays / stats.kernel_ms * 1e-3 : 0.0,
                         stats.kernel, astats.build_ms, (unsigned long long)astats.cells, (unsigned long long)astats.refs);
        }
    }
    cleanup();
    return 0;
}
