// Host-side robustness check, built with -fsanitize=address,undefined by
// tests/test_host_sanitizers.py: the JSON reader, the scene loader and the PNG
// writer are fed valid files, systematically truncated / mutated files and
// random bytes.  Any memory error or undefined behaviour aborts the run.
#include "mrt.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

static int try_load(const std::string& text) {
    mrt_loaded_scene* ls = nullptr;
    char err[256];
    int rc = mrt_scene_load_text(text.data(), text.size(), &ls, err, sizeof err);
    if (rc == MRT_OK) {
        const mrt_scene* s = mrt_scene_view(ls);
        volatile float sink = 0;
        for (uint32_t k = 0; k < s->n_tris; ++k) sink += s->tri[0][k] + s->tri[8][k];
        for (uint32_t k = 0; k < s->n_spheres; ++k) sink += s->sph[3][k];
        (void)sink;
        mrt_scene_free(ls);
    } else if (std::strlen(err) == 0) {
        std::printf("failure without a message\n");
        std::exit(2);
    }
    return rc;
}

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    std::ifstream f(argv[1]);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string good = ss.str();
    if (try_load(good) != MRT_OK) { std::printf("valid scene rejected\n"); return 1; }

    std::mt19937 rng(12345);
    int ok = 0, bad = 0;
    // every prefix
    for (size_t n = 0; n < good.size(); n += 1 + good.size() / 400) (try_load(good.substr(0, n)) == MRT_OK ? ok : bad)++;
    // point mutations
    const char alphabet[] = "{}[]\",:-+.eE0123456789 \n\\ntrufalse";
    for (int it = 0; it < 3000; ++it) {
        std::string t = good;
        int edits = 1 + rng() % 3;
        for (int e = 0; e < edits; ++e) {
            size_t pos = rng() % t.size();
            switch (rng() % 3) {
                case 0: t[pos] = alphabet[rng() % (sizeof alphabet - 1)]; break;
                case 1: t.erase(pos, 1 + rng() % 4); break;
                default: t.insert(pos, 1, alphabet[rng() % (sizeof alphabet - 1)]); break;
            }
        }
        (try_load(t) == MRT_OK ? ok : bad)++;
    }
    // random bytes and pathological nesting
    for (int it = 0; it < 300; ++it) {
        std::string t(1 + rng() % 200, 0);
        for (auto& c : t) c = static_cast<char>(rng());
        (try_load(t) == MRT_OK ? ok : bad)++;
    }
    try_load(std::string(100000, '['));
    try_load("{\"a\":" + std::string(5000, '[') + std::string(5000, ']') + "}");
    try_load("\"\\ud800\"");
    try_load("{\"x_resolution\": 99999999999999999999999999}");

    // PNG writer: odd sizes, several stripes
    for (auto wh : {std::pair<uint32_t, uint32_t>{1, 1}, {3, 2}, {257, 1}, {1, 4099}, {1200, 1500}}) {
        std::vector<uint8_t> img(static_cast<size_t>(wh.first) * wh.second * 4);
        for (auto& b : img) b = static_cast<uint8_t>(rng());
        std::string path = std::string(argv[2]) + "/t.png";
        if (mrt_write_png_rgb(path.c_str(), img.data(), wh.first, wh.second) != MRT_OK) { std::printf("png failed\n"); return 1; }
    }
    if (mrt_write_png_rgb("/nonexistent-dir/x.png", reinterpret_cast<const uint8_t*>("abcd"), 1, 1) == MRT_OK) return 1;
    std::printf("ok %d accepted %d rejected\n", ok, bad);
    return 0;
}
