// png_writer.cpp -- 8-bit RGB PNG from the renderer's RGBA8 frame.
//
// Output format is the one gfx::write_png produces through png++
// (gfxpng.hpp:63-86): colour type 2 (truecolour), bit depth 8, no alpha, no
// interlace.  Any valid deflate stream decodes to the same pixels, so the
// compression settings are this writer's own: the frame is cut into row
// stripes that are deflated independently on host threads (raw deflate, each
// stripe ended with a full flush so the pieces concatenate into one zlib
// stream), and the Adler-32 of the whole image is combined from per-stripe
// values.  A 16384x16384 frame is ~805 MB of scanlines, which is why this is
// not done on one thread.
#include "mrt.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {

void be32(uint8_t* p, uint32_t v) {
    p[0] = static_cast<uint8_t>(v >> 24); p[1] = static_cast<uint8_t>(v >> 16);
    p[2] = static_cast<uint8_t>(v >> 8);  p[3] = static_cast<uint8_t>(v);
}

bool write_chunk(FILE* f, const char type[4], const uint8_t* data, size_t len) {
    uint8_t hdr[8];
    be32(hdr, static_cast<uint32_t>(len));
    std::memcpy(hdr + 4, type, 4);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32_z(crc, data, len);
    uint8_t tail[4];
    be32(tail, static_cast<uint32_t>(crc));
    return std::fwrite(hdr, 1, 8, f) == 8 && (len == 0 || std::fwrite(data, 1, len, f) == len) &&
           std::fwrite(tail, 1, 4, f) == 4;
}

struct Stripe {
    uint32_t row_begin = 0, row_end = 0;
    std::vector<uint8_t> deflated;
    uLong adler = 1;
    size_t raw_len = 0;
    bool ok = false;
};

void deflate_stripe(const uint8_t* rgba, uint32_t xres, Stripe& st, bool last) {
    const size_t line = 1 + static_cast<size_t>(xres) * 3;
    const uint32_t rows = st.row_end - st.row_begin;
    std::vector<uint8_t> raw(line * rows);
    for (uint32_t r = 0; r < rows; ++r) {
        uint8_t* dst = raw.data() + line * r;
        const uint8_t* src = rgba + static_cast<size_t>(st.row_begin + r) * xres * 4;
        dst[0] = 0;  // filter type None
        for (uint32_t x = 0; x < xres; ++x) {
            dst[1 + 3 * x] = src[4 * x];
            dst[2 + 3 * x] = src[4 * x + 1];
            dst[3 + 3 * x] = src[4 * x + 2];
        }
    }
    st.raw_len = raw.size();
    st.adler = adler32_z(1L, raw.data(), raw.size());
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 3, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
    st.deflated.resize(deflateBound(&zs, static_cast<uLong>(raw.size())) + 16);
    zs.next_in = raw.data();
    zs.next_out = st.deflated.data();
    size_t in_left = raw.size(), out_left = st.deflated.size();
    int rc = Z_OK;
    const int final_flush = last ? Z_FINISH : Z_FULL_FLUSH;
    while (true) {
        zs.avail_in = static_cast<uInt>(std::min<size_t>(in_left, 1u << 30));
        zs.avail_out = static_cast<uInt>(std::min<size_t>(out_left, 1u << 30));
        const uInt gave_in = zs.avail_in, gave_out = zs.avail_out;
        const bool all_in = (in_left == gave_in);
        rc = deflate(&zs, all_in ? final_flush : Z_NO_FLUSH);
        in_left -= gave_in - zs.avail_in;
        out_left -= gave_out - zs.avail_out;
        if (rc == Z_STREAM_END) break;
        if (rc != Z_OK && rc != Z_BUF_ERROR) { deflateEnd(&zs); return; }
        if (all_in && !last && zs.avail_in == 0 && zs.avail_out != 0) break;  // full flush done
        if (out_left == 0) { deflateEnd(&zs); return; }
    }
    st.deflated.resize(st.deflated.size() - out_left);
    deflateEnd(&zs);
    st.ok = true;
}

}  // namespace

extern "C" int mrt_write_png_rgb(const char* path, const uint8_t* rgba, uint32_t xres, uint32_t yres) {
    if (!path || !rgba || xres == 0 || yres == 0) return MRT_ERR_INVALID;

    const size_t line = 1 + static_cast<size_t>(xres) * 3;
    // ~4 MB of scanlines per stripe, at least one row.
    uint32_t rows_per = static_cast<uint32_t>(std::max<size_t>(1, (4u << 20) / line));
    uint32_t n_stripes = (yres + rows_per - 1) / rows_per;
    std::vector<Stripe> stripes(n_stripes);
    for (uint32_t k = 0; k < n_stripes; ++k) {
        stripes[k].row_begin = k * rows_per;
        stripes[k].row_end = std::min(yres, (k + 1) * rows_per);
    }
    // At most 16 threads (a one-GPU share of a node's cores) unless MRT_PNG_THREADS says otherwise:
    // a caller who has the whole node (the frame of eight GPUs) can give the encoder more.
    unsigned hw = std::thread::hardware_concurrency();
    unsigned cap = 16u;
    if (const char* env = std::getenv("MRT_PNG_THREADS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 1024) cap = static_cast<unsigned>(v);
    }
    unsigned n_threads = std::max(1u, std::min<unsigned>(std::min(hw ? hw : 1u, cap), n_stripes));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; ++t) {
        pool.emplace_back([&, t]() {
            for (uint32_t k = t; k < n_stripes; k += n_threads)
                deflate_stripe(rgba, xres, stripes[k], k + 1 == n_stripes);
        });
    }
    for (auto& th : pool) th.join();
    for (const auto& st : stripes)
        if (!st.ok) return MRT_ERR_IO;

    FILE* f = std::fopen(path, "wb");
    if (!f) return MRT_ERR_IO;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    be32(ihdr, xres);
    be32(ihdr + 4, yres);
    ihdr[8] = 8;   // bit depth
    ihdr[9] = 2;   // colour type: truecolour (RGB)
    ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    ok = ok && write_chunk(f, "IHDR", ihdr, 13);

    // zlib header (deflate, 32K window, no preset dictionary), then the
    // stripes, then the Adler-32 of all scanlines.
    static const uint8_t zhdr[2] = {0x78, 0x5E};
    ok = ok && write_chunk(f, "IDAT", zhdr, 2);
    uLong adler = 1;
    bool first = true;
    for (const auto& st : stripes) {
        size_t off = 0;
        while (ok && off < st.deflated.size()) {   // keep every chunk below 2^31 bytes
            size_t n = std::min<size_t>(st.deflated.size() - off, 1u << 30);
            ok = write_chunk(f, "IDAT", st.deflated.data() + off, n);
            off += n;
        }
        adler = first ? st.adler : adler32_combine(adler, st.adler, static_cast<z_off_t>(st.raw_len));
        first = false;
    }
    uint8_t ad[4];
    be32(ad, static_cast<uint32_t>(adler));
    ok = ok && write_chunk(f, "IDAT", ad, 4);
    ok = ok && write_chunk(f, "IEND", nullptr, 0);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? MRT_OK : MRT_ERR_IO;
}
