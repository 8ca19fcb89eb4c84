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
// not done on one thread -- and why the encoder can start before the frame is
// complete (mrt_png_stream_*): a stripe is deflated as soon as its rows are in,
// while the GPUs still render and copy the rest.
#include "mrt.h"
#include "mrt_guard.hpp"

#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <string>
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

// A PNG being encoded while its pixels are still arriving: the stripes are those of
// mrt_write_png_rgb (so are the bytes of the file); a stripe is deflated by a pool thread as soon
// as all of its rows have been reported final.
struct mrt_png_stream {
    std::string path;
    const uint8_t* rgba = nullptr;
    uint32_t xres = 0, yres = 0, rows_per = 1;
    std::vector<Stripe> stripes;
    std::vector<uint32_t> rows_in;          // rows of each stripe reported so far
    std::deque<uint32_t> runnable;          // stripes whose rows are all in, not yet taken by a thread
    uint32_t done = 0;                      // stripes deflated (or failed)
    bool closing = false;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> pool;
};

namespace {

void stream_worker(mrt_png_stream* s) {
    std::unique_lock<std::mutex> lock(s->mu);
    for (;;) {
        s->cv_work.wait(lock, [&] { return s->closing || !s->runnable.empty(); });
        if (s->runnable.empty()) return;                        // closing and nothing left
        const uint32_t k = s->runnable.front();
        s->runnable.pop_front();
        lock.unlock();
        try {
            deflate_stripe(s->rgba, s->xres, s->stripes[k], k + 1 == (uint32_t)s->stripes.size());
        } catch (...) {                                         // out of memory: the stripe stays "not ok", the file is not written
            s->stripes[k].ok = false;
        }
        lock.lock();
        ++s->done;
        s->cv_done.notify_all();
    }
}

bool write_file(const mrt_png_stream& s) {
    FILE* f = std::fopen(s.path.c_str(), "wb");
    if (!f) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    uint8_t ihdr[13];
    be32(ihdr, s.xres);
    be32(ihdr + 4, s.yres);
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
    for (const auto& st : s.stripes) {
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
    return ok;
}

}  // namespace

extern "C" int mrt_png_stream_begin(const char* path, const uint8_t* rgba, uint32_t xres, uint32_t yres,
                                    mrt_png_stream** out) {
    if (!out) return MRT_ERR_INVALID;
    *out = nullptr;
    if (!path || !rgba || xres == 0 || yres == 0) return MRT_ERR_INVALID;
    mrt_png_stream* s = nullptr;
    try {
    s = new mrt_png_stream();
    s->path = path;
    s->rgba = rgba;
    s->xres = xres;
    s->yres = yres;
    const size_t line = 1 + static_cast<size_t>(xres) * 3;
    // ~4 MB of scanlines per stripe, at least one row.
    s->rows_per = static_cast<uint32_t>(std::max<size_t>(1, (4u << 20) / line));
    const uint32_t n_stripes = (yres + s->rows_per - 1) / s->rows_per;
    s->stripes.resize(n_stripes);
    s->rows_in.assign(n_stripes, 0u);
    for (uint32_t k = 0; k < n_stripes; ++k) {
        s->stripes[k].row_begin = k * s->rows_per;
        s->stripes[k].row_end = std::min(yres, (k + 1) * s->rows_per);
    }
    // At most 16 threads (a one-GPU share of a node's cores) unless MRT_PNG_THREADS says otherwise:
    // a caller who has the whole node (the frame of eight GPUs) can give the encoder more.
    unsigned hw = std::thread::hardware_concurrency();
    unsigned cap = 16u;
    if (const char* env = std::getenv("MRT_PNG_THREADS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 1 && v <= 1024) cap = static_cast<unsigned>(v);
    }
    const unsigned n_threads = std::max(1u, std::min<unsigned>(std::min(hw ? hw : 1u, cap), n_stripes));
    for (unsigned t = 0; t < n_threads; ++t) s->pool.emplace_back(stream_worker, s);
    } catch (...) {
        // no memory for the tables, or a thread that could not be started: the threads that did
        // start are told to leave, and there is no stream
        if (s) {
            {
                std::lock_guard<std::mutex> lock(s->mu);
                s->closing = true;
            }
            s->cv_work.notify_all();
            for (auto& th : s->pool) th.join();
            delete s;
        }
        return mrt::on_exception([](int, const char*) {});
    }
    *out = s;
    return MRT_OK;
}

extern "C" int mrt_png_stream_rows(mrt_png_stream* s, uint32_t row_begin, uint32_t row_end) try {
    if (!s || row_begin > row_end || row_end > s->yres) return MRT_ERR_INVALID;
    std::lock_guard<std::mutex> lock(s->mu);
    for (uint32_t r = row_begin; r < row_end;) {                            // a row reported twice: nothing is taken
        const uint32_t k = r / s->rows_per;
        const uint32_t upto = std::min(row_end, s->stripes[k].row_end);
        if (s->rows_in[k] + (upto - r) > s->stripes[k].row_end - s->stripes[k].row_begin) return MRT_ERR_INVALID;
        r = upto;
    }
    for (uint32_t r = row_begin; r < row_end;) {
        const uint32_t k = r / s->rows_per;
        const uint32_t upto = std::min(row_end, s->stripes[k].row_end);
        s->rows_in[k] += upto - r;
        const uint32_t rows = s->stripes[k].row_end - s->stripes[k].row_begin;
        if (s->rows_in[k] == rows) {
            s->runnable.push_back(k);
            s->cv_work.notify_one();
        }
        r = upto;
    }
    return MRT_OK;
} MRT_GUARD()

extern "C" int mrt_png_stream_end(mrt_png_stream* s) try {
    if (!s) return MRT_ERR_INVALID;
    int rc = MRT_OK;
    {
        std::unique_lock<std::mutex> lock(s->mu);
        uint32_t started = 0;
        for (size_t k = 0; k < s->stripes.size(); ++k)
            started += s->rows_in[k] == s->stripes[k].row_end - s->stripes[k].row_begin ? 1u : 0u;
        if (started != s->stripes.size()) rc = MRT_ERR_INVALID;             // rows were never reported: no file
        s->cv_done.wait(lock, [&] { return s->done == started; });
        s->closing = true;
        s->cv_work.notify_all();
    }
    for (auto& th : s->pool) th.join();
    if (rc == MRT_OK)
        for (const auto& st : s->stripes)
            if (!st.ok) rc = MRT_ERR_IO;
    if (rc == MRT_OK && !write_file(*s)) rc = MRT_ERR_IO;
    delete s;
    return rc;
} MRT_GUARD()

extern "C" int mrt_write_png_rgb(const char* path, const uint8_t* rgba, uint32_t xres, uint32_t yres) try {
    mrt_png_stream* s = nullptr;
    int rc = mrt_png_stream_begin(path, rgba, xres, yres, &s);
    if (rc != MRT_OK) return rc;
    rc = mrt_png_stream_rows(s, 0, yres);
    const int rc_end = mrt_png_stream_end(s);
    return rc != MRT_OK ? rc : rc_end;
} MRT_GUARD()
