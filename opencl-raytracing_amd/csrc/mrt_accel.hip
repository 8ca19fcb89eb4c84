// mrt_accel.hip -- screen-space acceleration behind its own entry points (mrt_accel_* in mrt.h).
//
// The reference's closest-hit loop visits every object for every pixel
// (kernel.cl:320-360) and so does the brute-force context (mrt_create).  An
// mrt_accel renders the same pixels while skipping, by construction, the pairs
// that cannot meet:
//
//   MRT_ACCEL_GRID          a uniform grid over the image, cells of 32x32
//                           pixels, built on the GPU once per view: every
//                           primitive is binned into the cells its prefilter
//                           region may touch (count -> exclusive scan -> fill),
//                           and grid_trace_kernel walks one cell's list per
//                           wave.  Primary rays of a pinhole or orthographic
//                           camera are affine in the pixel, so a grid over the
//                           pixels is a grid over the rays.
//   MRT_ACCEL_TILE_CULL     no structure: cull_trace_kernel tests every
//                           primitive against every 32x32 tile in every frame.
//   MRT_ACCEL_BLOCK_REJECT  trace_kernel with a 4x4 pixel block test ahead of
//                           the per-pixel funnel.
//
// The last two are kernels of an inner brute-force context created with
// internal flags; the grid adds the kernels and buffers of this file.  All three
// end in the reference arithmetic for the surviving pairs, merged by (t, id)
// keys, so ids, t and bytes equal the brute-force context's.
#include "mrt.h"
#include "mrt_guard.hpp"
#include "mrt_internal.hpp"
#include "mrt_kernels.hpp"
#include "mrt_tile_tests.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

namespace mrt {

template <int PROJ>
__global__ void grid_trace_kernel(const RenderParams p, const uint32_t* cell_start, const uint32_t* refs,
                                  uint32_t grid_tiles_x, float* t_out, int32_t* id_out, uint32_t pitch);

namespace {

constexpr int kBinWaves = 4;              // waves (= primitives) per block of the binning kernel
constexpr int kScanBlock = 256;           // threads per block of the scan kernels
constexpr int kScanPerThread = 16;        // consecutive cells per thread
constexpr int kScanTile = kScanBlock * kScanPerThread;   // 4096 cells per block

// Inclusive range of tiles [lo, hi] along one axis that pixels [a, b] (floats, any magnitude,
// a <= b) overlap, clipped to `tiles` tiles of `extent` pixels; false when there is none.
__device__ __forceinline__ bool tile_range(float a, float b, uint32_t extent, uint32_t tiles, uint32_t& lo, uint32_t& hi) {
    if (!(a == a) || !(b == b)) { lo = 0; hi = tiles - 1; return true; }       // NaN: every tile
    const float last = (float)(extent - 1);
    if (b < 0.0f || a > last) return false;
    const float ca = a < 0.0f ? 0.0f : a, cb = b > last ? last : b;
    lo = (uint32_t)ca / kWaveTile;
    hi = (uint32_t)cb / kWaveTile;
    if (hi > tiles - 1) hi = tiles - 1;
    return lo <= hi;
}

// The box of tiles [tx0, tx1] x [ty0, ty1] that bounds the prefilter region of object obj
// (spheres first, gfxraytrace.hpp:701-714), and what the tile test needs: e is the sphere's
// ellipse record or the triangle's bounding circle, rec the triangle's edge record.  False when
// the object cannot touch the image.
struct TileBox {
    uint32_t tx0, tx1, ty0, ty1;
    float4 e;
    const float4* rec;
    bool sphere;
};

__device__ __forceinline__ bool object_tile_box(const RenderParams& p, uint32_t obj, uint32_t tiles_x, uint32_t tiles_y,
                                                TileBox& b) {
    b.sphere = obj < p.n_sph;
    b.tx0 = 0; b.tx1 = tiles_x - 1; b.ty0 = 0; b.ty1 = tiles_y - 1;
    b.rec = nullptr;
    if (b.sphere) {
        // (x*sx + ox)^2 + (y*sy + oy)^2 <= 1 needs |x*sx + ox| <= 1: an interval of x, taken with
        // one pixel plus 1e-5 of its own size as slack (the tile test allows 1.000001, and both
        // it and the per-pixel test round)
        const float4 e = p.sph_filter[obj];
        b.e = e;
        if (e.x != 0.0f) {
            const float lo = (-1.0f - e.y) / e.x, hi = (1.0f - e.y) / e.x;
            const float pad = 1.0f + 1e-5f * (fabsf(lo) + fabsf(hi));
            if (!tile_range(fminf(lo, hi) - pad, fmaxf(lo, hi) + pad, p.xres, tiles_x, b.tx0, b.tx1)) return false;
        } else if (!(fabsf(e.y) <= 1.000001f)) {
            return false;
        }
        if (e.z != 0.0f) {
            const float lo = (-1.0f - e.w) / e.z, hi = (1.0f - e.w) / e.z;
            const float pad = 1.0f + 1e-5f * (fabsf(lo) + fabsf(hi));
            if (!tile_range(fminf(lo, hi) - pad, fmaxf(lo, hi) + pad, p.yres, tiles_y, b.ty0, b.ty1)) return false;
        } else if (!(fabsf(e.w) <= 1.000001f)) {
            return false;
        }
    } else {
        const uint32_t ti = obj - p.n_sph;
        const float4 e = p.tri_circle[ti];           // (xc, yc, R^2): 1e30 = every pixel, negative = none
        b.e = e;
        b.rec = p.tri_filter + (size_t)ti * 3;
        if (e.z < 0.0f) return false;
        if (e.z < 1e29f) {
            // the circle's bounding box, with a pixel and the rounding of centre -+ radius as slack
            const float r = sqrtf(e.z) * 1.00001f + 1.0f;
            const float px = 1e-6f * fabsf(e.x), py = 1e-6f * fabsf(e.y);
            if (!tile_range(e.x - r - px, e.x + r + px, p.xres, tiles_x, b.tx0, b.tx1)) return false;
            if (!tile_range(e.y - r - py, e.y + r + py, p.yres, tiles_y, b.ty0, b.ty1)) return false;
        }
    }
    return true;
}

// Tiles first, first + stride, ... of the box: the tile test, and for every tile that passes one
// more reference in its cell -- counted (FILL == false) or written at the cell's cursor, which
// starts at the cell's offset (FILL == true).  Both passes evaluate the same expressions on the
// same inputs, so they agree tile for tile.
template <bool FILL>
__device__ __forceinline__ void bin_tiles(const TileBox& b, uint32_t obj, uint32_t first, uint32_t stride,
                                          uint32_t tiles_x, uint32_t* __restrict__ cell_cursor,
                                          uint32_t* __restrict__ refs) {
    const uint32_t nx = b.tx1 - b.tx0 + 1, ny = b.ty1 - b.ty0 + 1;
    const uint32_t total = nx * ny;                  // at most 2048 x 2048
    for (uint32_t i = first; i < total; i += stride) {
        const uint32_t ry = i / nx, rx = i - ry * nx;
        const uint32_t tx = b.tx0 + rx, ty = b.ty0 + ry;
        const float txf = (float)(tx * kWaveTile), tyf = (float)(ty * kWaveTile);
        const bool pass = b.sphere ? tile_may_hit_sphere(b.e, txf, tyf) : tile_may_hit_triangle(b.e, b.rec, txf, tyf);
        if (pass) {
            const uint32_t at = atomicAdd(&cell_cursor[ty * tiles_x + tx], 1u);
            if (FILL) refs[at] = obj;
        }
    }
}

// Objects whose box holds more tiles than this are left to grid_bin_big_kernel: one wave walking
// a quarter of a million tiles (a triangle seen edge-on has a record every tile passes) would
// keep the whole build waiting.
constexpr uint32_t kBigBoxTiles = 2048;

// One wave per object; the lanes stride over the tiles of its box.  The counting pass also
// collects the objects with a big box in big_list.  n_refs / capacity: a fill enqueued before
// the host knows the reference count is skipped as a whole when the lists would not fit the
// buffer it was given.
template <bool FILL>
__global__ __launch_bounds__(64 * kBinWaves) void grid_bin_kernel(const RenderParams p, uint32_t tiles_x,
                                                                 uint32_t tiles_y, uint32_t* __restrict__ cell_cursor,
                                                                 uint32_t* __restrict__ refs,
                                                                 uint32_t* __restrict__ big_list,
                                                                 uint32_t* __restrict__ big_count,
                                                                 const unsigned long long* __restrict__ n_refs,
                                                                 unsigned long long capacity) {
    if (FILL && n_refs != nullptr && *n_refs > capacity) return;
    const uint32_t obj = blockIdx.x * kBinWaves + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (obj >= p.n_sph + p.n_tri) return;
    TileBox b;
    if (!object_tile_box(p, obj, tiles_x, tiles_y, b)) return;
    if ((b.tx1 - b.tx0 + 1) * (b.ty1 - b.ty0 + 1) > kBigBoxTiles) {
        if (!FILL && lane == 0) big_list[atomicAdd(big_count, 1u)] = obj;
        return;
    }
    bin_tiles<FILL>(b, obj, lane, 64, tiles_x, cell_cursor, refs);
}

// The objects with a big box, one after the other, each over all threads of the launch.
template <bool FILL>
__global__ __launch_bounds__(256) void grid_bin_big_kernel(const RenderParams p, uint32_t tiles_x, uint32_t tiles_y,
                                                          uint32_t* __restrict__ cell_cursor,
                                                          uint32_t* __restrict__ refs,
                                                          const uint32_t* __restrict__ big_list,
                                                          const uint32_t* __restrict__ big_count,
                                                          const unsigned long long* __restrict__ n_refs,
                                                          unsigned long long capacity) {
    if (FILL && n_refs != nullptr && *n_refs > capacity) return;
    const uint32_t n_big = *big_count;
    const uint32_t first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t k = 0; k < n_big; ++k) {
        const uint32_t obj = big_list[k];
        TileBox b;
        if (!object_tile_box(p, obj, tiles_x, tiles_y, b)) continue;
        bin_tiles<FILL>(b, obj, first, stride, tiles_x, cell_cursor, refs);
    }
}

// Exclusive scan of the per-cell counts in three small launches: sums of 4096-cell blocks, a
// scan of those sums by one block, then per block the offsets of its cells.  cell_start[c] gets
// the offset and the count is replaced by the same value, the cell's fill cursor;
// cell_start[n_cells] and *total get the number of references (a cell's count is at most the
// number of objects, below 2^31).
__global__ __launch_bounds__(kScanBlock) void scan_block_sums(const uint32_t* __restrict__ count, uint32_t n,
                                                             unsigned long long* __restrict__ block_sum) {
    __shared__ unsigned long long s_part[kScanBlock / 64];
    const uint32_t base = blockIdx.x * kScanTile;
    unsigned long long v = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t i = base + k * kScanBlock + threadIdx.x;
        v += i < n ? count[i] : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) t += s_part[w];
        block_sum[blockIdx.x] = t;
    }
}

// One block of 1024 threads: exclusive scan of up to 1024 block sums in place; the grand total
// goes to *total.  The sums are 64-bit so that the total is right whatever it is; the per-cell
// offsets scan_apply derives are 32-bit and only used once the host has seen that the total fits.
__global__ __launch_bounds__(1024) void scan_of_sums(unsigned long long* __restrict__ block_sum, uint32_t n_blocks,
                                                     unsigned long long* __restrict__ total) {
    __shared__ unsigned long long s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long mine = tid < n_blocks ? block_sum[tid] : 0ull;
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    if (tid < n_blocks) block_sum[tid] = before + incl - mine;
    if (tid == 1023) *total = before + incl;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply(uint32_t* __restrict__ count_then_cursor, uint32_t n,
                                                        const unsigned long long* __restrict__ block_offset,
                                                        const unsigned long long* __restrict__ total,
                                                        uint32_t* __restrict__ cell_start) {
    __shared__ uint32_t s_wave[kScanBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t first = blockIdx.x * kScanTile + tid * kScanPerThread;
    uint32_t v[kScanPerThread];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        v[k] = first + k < n ? count_then_cursor[first + k] : 0u;
        sum += v[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = (uint32_t)block_offset[blockIdx.x] + incl - sum;
    for (uint32_t w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        if (first + k < n) {
            cell_start[first + k] = run;
            count_then_cursor[first + k] = run;
        }
        run += v[k];
    }
    if (blockIdx.x == 0 && tid == 0) cell_start[n] = (uint32_t)*total;
}

}  // namespace
}  // namespace mrt

struct mrt_accel {
    int device = 0;
    uint32_t method = MRT_ACCEL_GRID;
    uint32_t flags = 0;
    mrt_ctx* base = nullptr;                 // scene, per-view records; the frame itself for the flag-driven methods
    std::string err;
    bool have_scene = false;

    // grid
    uint32_t tiles_x = 0, tiles_y = 0;
    uint32_t* d_cursor = nullptr;            // per cell: count, then fill cursor
    uint32_t* d_start = nullptr;             // per cell + 1: offset of its list in d_refs
    unsigned long long* d_block_sum = nullptr;   // 1024 scan partials + the total
    uint32_t* d_refs = nullptr;
    size_t cap_cells = 0, cap_refs = 0;
    uint32_t* d_big = nullptr;               // objects whose box is too big for one wave, and their count
    uint32_t* d_big_count = nullptr;
    size_t cap_big = 0;
    unsigned long long* h_total = nullptr;   // page-locked word the total is read back into
    uint64_t n_refs = 0;
    double build_ms = 0.0;
    bool build_pending = false;              // a build is enqueued whose reference count has not been looked at
    bool grid_valid = false;                 // the lists on the GPU are complete for the current view
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // build begin/end, frame begin/end
    hipEvent_t grid_ready = nullptr, frame_done = nullptr;
    bool frame_pending = false;
    hipStream_t last_stream = nullptr;       // the stream of the last GRID frame (frame_done is behind it)
    // frame buffers of the grid method (grown on demand)
    uint32_t* d_out = nullptr;
    size_t out_elems = 0;
    float* d_t = nullptr;
    int32_t* d_id = nullptr;
    size_t plane_elems = 0;
    unsigned long long* d_counter = nullptr;
};

namespace {

thread_local std::string g_accel_create_err;

int afail(mrt_accel* a, int code, const std::string& msg) {
    if (a) a->err = msg; else g_accel_create_err = msg;
    return code;
}

#define AHIP_TRY(a, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return afail((a), MRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct AccelDeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit AccelDeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
    }
    ~AccelDeviceGuard() {
        int now = -1;
        if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    }
};

inline uint32_t round_up_u32(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

int base_failed(mrt_accel* a, int rc) { return afail(a, rc, mrt_last_error(a->base)); }

// At most 2^28 (tile, primitive) references, 1 GiB of ids.
constexpr uint64_t kMaxRefs = 1ull << 28;
constexpr uint32_t kBigBlocks = 1024;        // blocks of grid_bin_big_kernel

// Completes a build enqueued by build_grid: waits for it, reads the reference count, and -- only
// when the lists did not fit the buffer the speculative fill had (or there was none yet) -- grows
// the buffer and fills again.  Called before a frame is enqueued, before the next build, and
// directly after the build in mrt_accel_set_scene.
int finish_build(mrt_accel* a) {
    if (!a->build_pending) return MRT_OK;
    a->build_pending = false;
    a->grid_valid = false;
    hipStream_t s = (hipStream_t)mrt_ctx_stream_(a->base);
    AHIP_TRY(a, hipEventSynchronize(a->ev[1]));
    const uint64_t total = *a->h_total;
    if (total > kMaxRefs)
        return afail(a, MRT_ERR_TOO_LARGE,
                     "the grid needs " + std::to_string(total) + " (tile, primitive) references, more than the " +
                         std::to_string(kMaxRefs) + " it may hold; use MRT_ACCEL_TILE_CULL or the brute-force context");
    if (total > a->cap_refs) {
        if (a->frame_pending) {                // a frame still in flight reads the buffer freed here
            AHIP_TRY(a, hipEventSynchronize(a->frame_done));
            a->frame_pending = false;
        }
        if (a->d_refs) { (void)hipFree(a->d_refs); a->d_refs = nullptr; }
        a->cap_refs = 0;
        const size_t want = (size_t)total + total / 4 + 1024;      // head room for the next view
        AHIP_TRY(a, hipMalloc((void**)&a->d_refs, want * sizeof(uint32_t)));
        a->cap_refs = want;
        // the guarded fill left the cursors at the cells' offsets
        const mrt::RenderParams p = *mrt_ctx_params_(a->base);
        const uint32_t n_obj = p.n_sph + p.n_tri;
        hipLaunchKernelGGL(mrt::grid_bin_kernel<true>, dim3((n_obj + mrt::kBinWaves - 1) / mrt::kBinWaves),
                           dim3(64 * mrt::kBinWaves), 0, s, p, a->tiles_x, a->tiles_y, a->d_cursor, a->d_refs,
                           a->d_big, a->d_big_count, (const unsigned long long*)nullptr, 0ull);
        AHIP_TRY(a, hipGetLastError());
        hipLaunchKernelGGL(mrt::grid_bin_big_kernel<true>, dim3(kBigBlocks), dim3(256), 0, s, p, a->tiles_x, a->tiles_y,
                           a->d_cursor, a->d_refs, (const uint32_t*)a->d_big, (const uint32_t*)a->d_big_count,
                           (const unsigned long long*)nullptr, 0ull);
        AHIP_TRY(a, hipGetLastError());
        AHIP_TRY(a, hipEventRecord(a->ev[1], s));
        AHIP_TRY(a, hipEventRecord(a->grid_ready, s));
        AHIP_TRY(a, hipEventSynchronize(a->ev[1]));
    }
    float ms = 0.f;
    AHIP_TRY(a, hipEventElapsedTime(&ms, a->ev[0], a->ev[1]));
    a->build_ms = ms;
    a->n_refs = total;
    a->grid_valid = true;
    return MRT_OK;
}

// Enqueues the build of the grid for the base context's current view on the base's stream, behind
// the record kernels mrt_set_scene / mrt_set_camera just enqueued there: count, scan, and a fill
// into the list buffer of the previous view, which the fill kernel itself skips when the new
// lists would not fit (finish_build then repeats it into a larger buffer).  Nothing here waits
// for the GPU unless an earlier build is still unfinished.
int build_grid(mrt_accel* a) {
    // An earlier build nobody rendered from is abandoned; the page-locked count word and the
    // events are reused, so it has to be over.
    if (a->build_pending) {
        AHIP_TRY(a, hipEventSynchronize(a->ev[1]));
        a->build_pending = false;
    }
    a->grid_valid = false;
    const mrt::RenderParams p = *mrt_ctx_params_(a->base);
    hipStream_t s = (hipStream_t)mrt_ctx_stream_(a->base);
    // a frame still in flight (on whatever stream) reads the lists this build replaces
    if (a->frame_pending) AHIP_TRY(a, hipStreamWaitEvent(s, a->frame_done, 0));
    a->tiles_x = (p.xres + mrt::kWaveTile - 1) / mrt::kWaveTile;
    a->tiles_y = (p.yres + mrt::kWaveTile - 1) / mrt::kWaveTile;
    const size_t cells = (size_t)a->tiles_x * a->tiles_y;
    const uint32_t n_obj = p.n_sph + p.n_tri;
    if (cells > a->cap_cells) {
        if (a->frame_pending) AHIP_TRY(a, hipEventSynchronize(a->frame_done));     // it reads what is freed here
        if (a->d_cursor) { (void)hipFree(a->d_cursor); a->d_cursor = nullptr; }
        if (a->d_start) { (void)hipFree(a->d_start); a->d_start = nullptr; }
        a->cap_cells = 0;
        AHIP_TRY(a, hipMalloc((void**)&a->d_cursor, cells * sizeof(uint32_t)));
        AHIP_TRY(a, hipMalloc((void**)&a->d_start, (cells + 1) * sizeof(uint32_t)));
        a->cap_cells = cells;
    }
    if (n_obj > a->cap_big) {
        if (a->d_big) { (void)hipFree(a->d_big); a->d_big = nullptr; }
        a->cap_big = 0;
        AHIP_TRY(a, hipMalloc((void**)&a->d_big, (size_t)n_obj * sizeof(uint32_t)));
        a->cap_big = n_obj;
    }
    const uint32_t n_blocks = (uint32_t)((cells + mrt::kScanTile - 1) / mrt::kScanTile);     // <= 1024 for 65536^2 pixels
    if (n_blocks > 1024) return afail(a, MRT_ERR_TOO_LARGE, "grid has more than 4 Mi cells");
    AHIP_TRY(a, hipEventRecord(a->ev[0], s));
    AHIP_TRY(a, hipMemsetAsync(a->d_cursor, 0, cells * sizeof(uint32_t), s));
    const dim3 bgrid((n_obj + mrt::kBinWaves - 1) / mrt::kBinWaves), bblock(64 * mrt::kBinWaves);
    unsigned long long* d_total = a->d_block_sum + 1024;
    AHIP_TRY(a, hipMemsetAsync(a->d_big_count, 0, sizeof(uint32_t), s));
    if (n_obj) {
        hipLaunchKernelGGL(mrt::grid_bin_kernel<false>, bgrid, bblock, 0, s, p, a->tiles_x, a->tiles_y, a->d_cursor,
                           (uint32_t*)nullptr, a->d_big, a->d_big_count, (const unsigned long long*)nullptr, 0ull);
        AHIP_TRY(a, hipGetLastError());
        hipLaunchKernelGGL(mrt::grid_bin_big_kernel<false>, dim3(kBigBlocks), dim3(256), 0, s, p, a->tiles_x, a->tiles_y,
                           a->d_cursor, (uint32_t*)nullptr, (const uint32_t*)a->d_big, (const uint32_t*)a->d_big_count,
                           (const unsigned long long*)nullptr, 0ull);
        AHIP_TRY(a, hipGetLastError());
    }
    hipLaunchKernelGGL(mrt::scan_block_sums, dim3(n_blocks), dim3(mrt::kScanBlock), 0, s, a->d_cursor, (uint32_t)cells,
                       a->d_block_sum);
    AHIP_TRY(a, hipGetLastError());
    hipLaunchKernelGGL(mrt::scan_of_sums, dim3(1), dim3(1024), 0, s, a->d_block_sum, n_blocks, d_total);
    AHIP_TRY(a, hipGetLastError());
    hipLaunchKernelGGL(mrt::scan_apply, dim3(n_blocks), dim3(mrt::kScanBlock), 0, s, a->d_cursor, (uint32_t)cells,
                       a->d_block_sum, d_total, a->d_start);
    AHIP_TRY(a, hipGetLastError());
    AHIP_TRY(a, hipMemcpyAsync(a->h_total, d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    if (n_obj && a->cap_refs) {
        hipLaunchKernelGGL(mrt::grid_bin_kernel<true>, bgrid, bblock, 0, s, p, a->tiles_x, a->tiles_y, a->d_cursor,
                           a->d_refs, a->d_big, a->d_big_count, (const unsigned long long*)d_total,
                           (unsigned long long)a->cap_refs);
        AHIP_TRY(a, hipGetLastError());
        hipLaunchKernelGGL(mrt::grid_bin_big_kernel<true>, dim3(kBigBlocks), dim3(256), 0, s, p, a->tiles_x, a->tiles_y,
                           a->d_cursor, a->d_refs, (const uint32_t*)a->d_big, (const uint32_t*)a->d_big_count,
                           (const unsigned long long*)d_total, (unsigned long long)a->cap_refs);
        AHIP_TRY(a, hipGetLastError());
    }
    AHIP_TRY(a, hipEventRecord(a->ev[1], s));
    AHIP_TRY(a, hipEventRecord(a->grid_ready, s));
    a->build_pending = true;
    return MRT_OK;
}

int ensure_out(mrt_accel* a, size_t elems) {
    if (elems <= a->out_elems) return MRT_OK;
    if (a->d_out) { (void)hipFree(a->d_out); a->d_out = nullptr; }
    a->out_elems = 0;
    AHIP_TRY(a, hipMalloc((void**)&a->d_out, elems * sizeof(uint32_t)));
    a->out_elems = elems;
    return MRT_OK;
}

int ensure_planes(mrt_accel* a, size_t elems) {
    if (elems <= a->plane_elems) return MRT_OK;
    if (a->d_t) { (void)hipFree(a->d_t); a->d_t = nullptr; }
    if (a->d_id) { (void)hipFree(a->d_id); a->d_id = nullptr; }
    a->plane_elems = 0;
    AHIP_TRY(a, hipMalloc((void**)&a->d_t, elems * sizeof(float)));
    AHIP_TRY(a, hipMalloc((void**)&a->d_id, elems * sizeof(int32_t)));
    a->plane_elems = elems;
    return MRT_OK;
}

// Enqueue one grid frame for this rank's rows on `stream`.
int launch_grid_frame(mrt_accel* a, uint32_t band_rows, uint32_t rank, uint32_t world, uint32_t* dev_out,
                      hipStream_t stream, bool timed, bool want_tid) {
    mrt::RenderParams p = *mrt_ctx_params_(a->base);
    p.band_rows = band_rows;
    p.rank = rank;
    p.world = world;
    p.local_rows = mrt_band_rows(p.yres, band_rows, rank, world);
    p.out = dev_out;
    p.flags = a->flags;
    p.exact_counter = (a->flags & MRT_FLAG_COUNT_EXACT) ? a->d_counter : nullptr;
    p.prim_slices = 1;
    p.tile_flags = nullptr;
    {
        const int rc = finish_build(a);
        if (rc != MRT_OK) return rc;
        if (!a->grid_valid) return afail(a, MRT_ERR_INVALID, "the grid of the current view was not built: " + a->err);
    }
    if (timed) AHIP_TRY(a, hipEventRecord(a->ev[2], stream));
    if (p.local_rows) {
        const bool consecutive = world == 1 || band_rows >= p.yres;
        if (!consecutive && band_rows % mrt::kWaveTile != 0)
            return afail(a, MRT_ERR_INVALID, "the grid needs bands of a multiple of 32 rows (a cell is a 32x32 pixel tile)");
        const uint32_t pitch = round_up_u32(p.xres, mrt::kBlockTileX);
        const uint32_t prow = round_up_u32(p.local_rows, mrt::kBlockTileY);
        float* d_t = nullptr;
        int32_t* d_id = nullptr;
        if (want_tid) {
            const int rc = ensure_planes(a, (size_t)pitch * prow);
            if (rc != MRT_OK) return rc;
            d_t = a->d_t;
            d_id = a->d_id;
        }
        p.fuse_shade = (p.xres % 4 == 0 && ((uintptr_t)dev_out & 15u) == 0) ? 2u : 1u;
        if (stream != (hipStream_t)mrt_ctx_stream_(a->base)) AHIP_TRY(a, hipStreamWaitEvent(stream, a->grid_ready, 0));
        // Frames of one handle share the counter and the (t, id) planes, and frame_done only
        // remembers the last one: a frame on another stream runs behind the one before it.
        if (a->frame_pending && stream != a->last_stream) AHIP_TRY(a, hipStreamWaitEvent(stream, a->frame_done, 0));
        if (p.exact_counter) AHIP_TRY(a, hipMemsetAsync(p.exact_counter, 0, sizeof(unsigned long long), stream));
        const dim3 grid(pitch / mrt::kBlockTileX, prow / mrt::kBlockTileY), block(mrt::kBlockThreads);
        if (mrt_ctx_projection_(a->base) == MRT_PROJ_PERSP)
            hipLaunchKernelGGL(mrt::grid_trace_kernel<MRT_PROJ_PERSP>, grid, block, 0, stream, p, a->d_start, a->d_refs,
                               a->tiles_x, d_t, d_id, pitch);
        else
            hipLaunchKernelGGL(mrt::grid_trace_kernel<MRT_PROJ_ORTHO>, grid, block, 0, stream, p, a->d_start, a->d_refs,
                               a->tiles_x, d_t, d_id, pitch);
        AHIP_TRY(a, hipGetLastError());
        AHIP_TRY(a, hipEventRecord(a->frame_done, stream));
        a->frame_pending = true;
        a->last_stream = stream;
    }
    if (timed) AHIP_TRY(a, hipEventRecord(a->ev[3], stream));
    return MRT_OK;
}

int fill_grid_stats(mrt_accel* a, mrt_accel_stats* st, uint32_t rows) {
    if (!st) return MRT_OK;
    const mrt::RenderParams& p = *mrt_ctx_params_(a->base);
    std::memset(st, 0, sizeof *st);
    float ms = 0.f;
    AHIP_TRY(a, hipEventElapsedTime(&ms, a->ev[2], a->ev[3]));
    st->frame.trace_ms = ms;
    st->frame.kernel_ms = ms;
    st->frame.rays = (uint64_t)rows * p.xres;
    st->frame.prim_tests = st->frame.rays * ((uint64_t)p.n_sph + p.n_tri);
    st->frame.kernel = MRT_KERNEL_GRID;
    st->frame.prim_slices = 1;
    if (a->flags & MRT_FLAG_COUNT_EXACT) {
        unsigned long long c = 0;
        AHIP_TRY(a, hipMemcpy(&c, a->d_counter, sizeof c, hipMemcpyDeviceToHost));
        st->frame.exact_tests = c;
    }
    st->build_ms = a->build_ms;
    st->cells = (uint64_t)a->tiles_x * a->tiles_y;
    st->refs = a->n_refs;
    return MRT_OK;
}

void wrap_stats(mrt_accel_stats* out, const mrt_stats& frame) {
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    out->frame = frame;
}

}  // namespace

extern "C" {

const char* mrt_accel_last_error(const mrt_accel* a) { return a ? a->err.c_str() : g_accel_create_err.c_str(); }

int mrt_accel_create(const mrt_accel_config* cfg, mrt_accel** out) try {
    if (!out) return afail(nullptr, MRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    const uint32_t method = cfg ? cfg->method : (uint32_t)MRT_ACCEL_GRID;
    const uint32_t flags = cfg ? cfg->flags : 0u;
    if (method > MRT_ACCEL_BLOCK_REJECT) return afail(nullptr, MRT_ERR_INVALID, "unknown acceleration method");
    if (flags & ~(uint32_t)MRT_FLAG_COUNT_EXACT)
        return afail(nullptr, MRT_ERR_INVALID, "mrt_accel_config.flags takes MRT_FLAG_COUNT_EXACT only");
    mrt_config inner;
    inner.device = cfg ? cfg->device : -1;
    inner.flags = flags | (method == MRT_ACCEL_TILE_CULL ? (uint32_t)MRT_IFLAG_SCREEN_CULL
                           : method == MRT_ACCEL_BLOCK_REJECT ? (uint32_t)MRT_IFLAG_BLOCK_REJECT : 0u);
    inner.flags |= MRT_FLAG_DEFER_NOALLOC_SETUP;         // no entry point of mrt_accel uses the fixed array
    mrt_ctx* base = nullptr;
    const int rc = mrt_create_internal_(&inner, &base);
    if (rc != MRT_OK) return afail(nullptr, rc, mrt_last_error(nullptr));
    mrt_accel* a = new mrt_accel();
    a->base = base;
    a->device = mrt_stage_device_(base);
    a->method = method;
    a->flags = flags;
    if (method == MRT_ACCEL_GRID) {
        AccelDeviceGuard guard(a->device);
        hipError_t e = guard.err;
        for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&a->ev[k]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a->grid_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a->frame_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void**)&a->d_block_sum, (1024 + 1) * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void**)&a->d_counter, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void**)&a->d_big_count, sizeof(uint32_t));
        if (e == hipSuccess) e = hipHostMalloc((void**)&a->h_total, sizeof(unsigned long long), hipHostMallocDefault);
        if (e != hipSuccess) {
            const std::string msg = std::string("mrt_accel_create: ") + hipGetErrorString(e);
            mrt_accel_destroy(a);
            return afail(nullptr, MRT_ERR_HIP, msg);
        }
    }
    *out = a;
    return MRT_OK;
} MRT_GUARD((void)afail(nullptr, code_, what_))

void mrt_accel_destroy(mrt_accel* a) {
    if (!a) return;
    {
        AccelDeviceGuard guard(a->device);
        (void)hipDeviceSynchronize();
        if (a->d_cursor) (void)hipFree(a->d_cursor);
        if (a->d_start) (void)hipFree(a->d_start);
        if (a->d_block_sum) (void)hipFree(a->d_block_sum);
        if (a->d_refs) (void)hipFree(a->d_refs);
        if (a->d_big) (void)hipFree(a->d_big);
        if (a->d_big_count) (void)hipFree(a->d_big_count);
        if (a->d_out) (void)hipFree(a->d_out);
        if (a->d_t) (void)hipFree(a->d_t);
        if (a->d_id) (void)hipFree(a->d_id);
        if (a->d_counter) (void)hipFree(a->d_counter);
        if (a->h_total) (void)hipHostFree(a->h_total);
        for (int k = 0; k < 4; ++k) if (a->ev[k]) (void)hipEventDestroy(a->ev[k]);
        if (a->grid_ready) (void)hipEventDestroy(a->grid_ready);
        if (a->frame_done) (void)hipEventDestroy(a->frame_done);
    }
    mrt_destroy(a->base);
    delete a;
}

uint32_t mrt_accel_method(const mrt_accel* a) { return a ? a->method : 0u; }

int mrt_accel_set_scene(mrt_accel* a, const mrt_scene* scene) try {
    if (!a) return MRT_ERR_INVALID;
    a->have_scene = false;
    int rc = mrt_set_scene(a->base, scene);
    if (rc != MRT_OK) return base_failed(a, rc);
    if (a->method == MRT_ACCEL_GRID) {
        AccelDeviceGuard guard(a->device);
        AHIP_TRY(a, guard.err);
        a->frame_pending = false;              // mrt_set_scene waited for the whole device
        if ((rc = build_grid(a)) != MRT_OK) return rc;
        if ((rc = finish_build(a)) != MRT_OK) return rc;       // this call is synchronous anyway
    }
    a->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)afail(a, code_, what_))

int mrt_accel_set_camera(mrt_accel* a, const mrt_camera* camera) try {
    if (!a) return MRT_ERR_INVALID;
    if (!a->have_scene) return afail(a, MRT_ERR_NO_SCENE, "mrt_accel_set_scene has not succeeded on this handle");
    a->have_scene = false;
    if (a->method == MRT_ACCEL_GRID && a->frame_pending) {
        // a grid frame still in flight on a caller's stream reads the per-view records that
        // mrt_set_camera is about to rebuild on the base context's stream
        AccelDeviceGuard guard(a->device);
        AHIP_TRY(a, guard.err);
        AHIP_TRY(a, hipStreamWaitEvent((hipStream_t)mrt_ctx_stream_(a->base), a->frame_done, 0));
    }
    int rc = mrt_set_camera(a->base, camera);
    if (rc != MRT_OK) {
        // a camera the base context refused outright (bad argument) has changed nothing
        a->have_scene = rc != MRT_ERR_HIP;
        return base_failed(a, rc);
    }
    if (a->method == MRT_ACCEL_GRID) {
        AccelDeviceGuard guard(a->device);
        AHIP_TRY(a, guard.err);
        if ((rc = build_grid(a)) != MRT_OK) return rc;
    }
    a->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)afail(a, code_, what_))

int mrt_accel_render_bands_device(mrt_accel* a, uint32_t band_rows, uint32_t rank, uint32_t world, void* dev_out,
                                  size_t dev_out_bytes, void* hip_stream, mrt_accel_stats* stats) try {
    if (!a) return MRT_ERR_INVALID;
    if (!a->have_scene) return afail(a, MRT_ERR_NO_SCENE, "mrt_accel_set_scene has not succeeded on this handle");
    if (a->method != MRT_ACCEL_GRID) {
        mrt_stats fs;
        const int rc = mrt_render_bands_device(a->base, band_rows, rank, world, dev_out, dev_out_bytes, hip_stream,
                                               stats ? &fs : nullptr);
        if (rc != MRT_OK) return base_failed(a, rc);
        if (stats) wrap_stats(stats, fs);
        return MRT_OK;
    }
    if (band_rows == 0 || world == 0 || rank >= world) return afail(a, MRT_ERR_INVALID, "bad band partition");
    const mrt::RenderParams& p = *mrt_ctx_params_(a->base);
    const uint32_t rows = mrt_band_rows(p.yres, band_rows, rank, world);
    const size_t need = (size_t)rows * p.xres * 4;
    if (need && !dev_out) return afail(a, MRT_ERR_INVALID, "dev_out is NULL");
    if (dev_out_bytes < need) return afail(a, MRT_ERR_INVALID, "dev_out too small for this rank's rows");
    if (((uintptr_t)dev_out & 3u) != 0) return afail(a, MRT_ERR_INVALID, "dev_out must be 4-byte aligned");
    AccelDeviceGuard guard(a->device);
    AHIP_TRY(a, guard.err);
    if (mrt_stream_is_capturing_(hip_stream))
        return afail(a, MRT_ERR_INVALID, "the stream is capturing a HIP graph: frames cannot be recorded into a graph (not supported)");
    int rc = launch_grid_frame(a, band_rows, rank, world, (uint32_t*)dev_out, (hipStream_t)hip_stream, stats != nullptr, false);
    if (rc != MRT_OK) return rc;
    if (stats) {
        AHIP_TRY(a, hipEventSynchronize(a->ev[3]));
        rc = fill_grid_stats(a, stats, rows);
    }
    return rc;
} MRT_GUARD((void)afail(a, code_, what_))

int mrt_accel_render_debug(mrt_accel* a, uint8_t* host_out, int32_t* ids, float* t, mrt_accel_stats* stats) try {
    if (!a) return MRT_ERR_INVALID;
    if (!a->have_scene) return afail(a, MRT_ERR_NO_SCENE, "mrt_accel_set_scene has not succeeded on this handle");
    if (!host_out && !ids && !t) return afail(a, MRT_ERR_INVALID, "no output requested");
    if (a->method != MRT_ACCEL_GRID) {
        mrt_stats fs;
        const int rc = mrt_render_debug(a->base, host_out, ids, t, &fs);
        if (rc != MRT_OK) return base_failed(a, rc);
        wrap_stats(stats, fs);
        return MRT_OK;
    }
    AccelDeviceGuard guard(a->device);
    AHIP_TRY(a, guard.err);
    const mrt::RenderParams& p = *mrt_ctx_params_(a->base);
    const uint32_t xres = p.xres, yres = p.yres;
    const size_t npx = (size_t)xres * yres;
    int rc = ensure_out(a, npx);
    if (rc != MRT_OK) return rc;
    hipStream_t s = (hipStream_t)mrt_ctx_stream_(a->base);
    const bool want_tid = ids != nullptr || t != nullptr;
    rc = launch_grid_frame(a, yres, 0, 1, a->d_out, s, true, want_tid);
    if (rc != MRT_OK) return rc;
    mrt_accel_stats local;
    AHIP_TRY(a, hipEventSynchronize(a->ev[3]));
    rc = fill_grid_stats(a, &local, yres);
    if (rc != MRT_OK) return rc;
    hipEvent_t d0, d1;
    AHIP_TRY(a, hipEventCreate(&d0));
    AHIP_TRY(a, hipEventCreate(&d1));
    (void)hipEventRecord(d0, s);
    hipError_t e = hipSuccess;
    if (host_out) e = hipMemcpyAsync(host_out, a->d_out, npx * 4, hipMemcpyDeviceToHost, s);
    const uint32_t pitch = round_up_u32(xres, mrt::kBlockTileX);
    if (e == hipSuccess && ids)
        e = hipMemcpy2DAsync(ids, (size_t)xres * 4, a->d_id, (size_t)pitch * 4, (size_t)xres * 4, yres, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && t)
        e = hipMemcpy2DAsync(t, (size_t)xres * 4, a->d_t, (size_t)pitch * 4, (size_t)xres * 4, yres, hipMemcpyDeviceToHost, s);
    (void)hipEventRecord(d1, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    float dms = 0.f;
    if (e == hipSuccess) (void)hipEventElapsedTime(&dms, d0, d1);
    (void)hipEventDestroy(d0);
    (void)hipEventDestroy(d1);
    if (e != hipSuccess) return afail(a, MRT_ERR_HIP, std::string("download: ") + hipGetErrorString(e));
    local.frame.download_ms = dms;
    if (stats) *stats = local;
    return MRT_OK;
} MRT_GUARD((void)afail(a, code_, what_))

int mrt_accel_render_rgba8(mrt_accel* a, uint8_t* host_out, mrt_accel_stats* stats) try {
    if (!a) return MRT_ERR_INVALID;
    if (!host_out) return afail(a, MRT_ERR_INVALID, "host_out is NULL");
    return mrt_accel_render_debug(a, host_out, nullptr, nullptr, stats);
} MRT_GUARD((void)afail(a, code_, what_))

}  // extern "C"
