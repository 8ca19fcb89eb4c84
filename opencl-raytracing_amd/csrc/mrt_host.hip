// mrt_host.hip -- HIP host layer behind the C ABI (include/mrt.h).
//
// Replaces CLHelpers.hpp: where each reference stage helper created a
// context, compiled kernel.cl, allocated buffers, copied in, launched with a
// fixed 160000 work-items and copied out (CLHelpers.hpp:76-322), a context
// here owns one stream and its device buffers for its whole life, the scene is
// uploaded once (mrt_set_scene), a change of view re-derives the per-view
// records on the GPU (mrt_set_camera, no copies, no waiting) and a frame is one
// or two kernel launches.
#include "mrt.h"
#include "mrt_guard.hpp"
#include "mrt_internal.hpp"
#include "mrt_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace mrt {
template <int PROJ, bool BLOCK_REJECT, int LX, int LY, bool SLICED>
__global__ void trace_kernel(const RenderParams p, unsigned long long* key_plane, float* t_out, int32_t* id_out,
                             uint32_t pitch);
template <int PROJ>
__global__ void cull_trace_kernel(const RenderParams p, float* t_out, int32_t* id_out, uint32_t pitch);
template <int PROJ>
__global__ void full_trace_kernel(const RenderParams p, const float4* full_rec, const uint32_t* full_ids,
                                  const uint32_t* class_counts, unsigned long long* key_plane, float* t_out,
                                  int32_t* id_out, uint32_t pitch);
template <int PROJ>
__global__ void prep_full_records(const RenderParams p, float4* rec, uint32_t* ids, uint32_t* counts);
__global__ void build_triangle_records(const ViewDerived w, const float4* tri_exact, uint32_t n_tri,
                                       uint32_t n_tri_padded, float4* tri_filter, float4* tri_circle);
__global__ void build_sphere_records(const ViewDerived w, const float4* sph, uint32_t n_sph, uint32_t n_sph_padded,
                                     float4* sph_filter);
template <int PROJ>
__global__ void shade_kernel(const RenderParams p, const float* t_in, const int32_t* id_in, uint32_t pitch);
template <int PROJ>
__global__ void shade_keys_kernel(const RenderParams p, unsigned long long* key_plane, float* t_out,
                                  int32_t* id_out, uint32_t pitch);
}  // namespace mrt

using mrt::RenderParams;

// Fixed working memory of mrt_render_bands_device_noalloc: the closest-hit key plane of one frame
// (MRT_NOALLOC_MAX_KEYS 64-bit keys, 64 MiB).  Every GPU the library runs on holds its own copy
// of the array at its own address until the process ends.  In host code the array's name is only
// a handle: the device address comes from hipGetSymbolAddress, once per device with that device
// current (scratch_address below), and that pointer -- never the name -- is what kernels, memsets
// and copies are given.
static __device__ __align__(256) unsigned long long scratch[MRT_NOALLOC_MAX_KEYS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned long long* g_scratch_ptr[kMaxDevices] = {};

// Device address of this device's copy of `scratch`; `device` must be the current device.
hipError_t scratch_address(int device, unsigned long long** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<unsigned long long*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct mrt_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    std::string err;

    bool have_scene = false;
    RenderParams params{};          // scene constants + device pointers
    int projection = 0;
    double upload_ms = 0.0;

    // The scene header (camera, viewport, counts; no plane pointers): the per-view records are
    // rebuilt on the GPU from the geometry already there when only the camera changes.
    mrt_scene h_scene{};
    hipEvent_t records_ready = nullptr;      // behind the last record build on ctx->stream
    bool records_pending = false;

    // scene buffers
    void* d_sph_filter = nullptr;
    void* d_tri_filter = nullptr;
    void* d_tri_circle = nullptr;
    size_t cap_sph_filter = 0, cap_tri_filter = 0, cap_tri_circle = 0;
    void* d_tri_exact = nullptr;
    size_t cap_tri_exact = 0, cap_sph = 0, cap_sph_mat = 0, cap_mats = 0, cap_lights = 0;
    // MRT_FLAG_FULL_TEST only: per-view records of full_trace_kernel, grouped by hit sign, the
    // triangle each record belongs to, and the two group sizes
    void* d_full_rec = nullptr;
    uint32_t* d_full_ids = nullptr;
    uint32_t* d_full_counts = nullptr;
    size_t cap_full_rec = 0;
    void* d_sph = nullptr;
    void* d_sph_mat = nullptr;
    void* d_mats = nullptr;
    void* d_lights = nullptr;
    // frame buffers (grown on demand)
    float* d_t = nullptr;
    int32_t* d_id = nullptr;
    unsigned long long* d_key = nullptr;     // trace_kernel's working plane: (t, id) keys under atomicMin
    size_t plane_elems = 0, key_elems = 0;
    // Every key of d_key is "no hit" (all ones): sliced frames keep it so (shade_keys_kernel puts
    // back what the frame wrote), which saves the memset; false after anything else touched it.
    bool keys_clean = false;
    uint32_t* d_tile_flags = nullptr;        // one word per trace_kernel wave tile of the padded plane (sized for 32x32 tiles)
    size_t tile_flag_elems = 0;
    uint32_t* d_out = nullptr;
    size_t out_elems = 0;
    unsigned long long* d_counter = nullptr;
    unsigned long long* scratch_keys = nullptr;   // this device's copy of `scratch` (not owned)

    // The last frame enqueued: which kernel it ran, and an event behind it so that a frame on
    // another stream can be ordered after it (the working planes above are shared by all frames).
    uint32_t last_kernel = MRT_KERNEL_TRACE, last_slices = 1;
    uint32_t last_pitch = 0;                 // pitch of the (t, id) planes the last frame wrote
    uint32_t frame_tag = 0;                  // tag of the last frame that used the tile flags
    // mrt_stage_*: device staging buffers that grow on demand and live as long as the ctx
    static constexpr int kStageSlots = 5;
    void* stage_buf[kStageSlots] = {};
    size_t stage_cap[kStageSlots] = {};
    double stage_kernel_ms = 0.0;            // device time of the kernels of the last mrt_stage_intersect
    uint32_t stage_t_switches = 0;           // ... and how many of its waves took up the fourth determinant
    void* stage_host = nullptr;              // page-locked host memory the stage calls' results land in before
    size_t stage_host_cap = 0;               // they are scattered into the caller's records; grows on demand
    hipEvent_t frame_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool frame_pending = false;
    // Tuning override of the automatic slice / layer count (0 = automatic): MRT_FORCE_SLICES in the
    // environment when the ctx is created, or mrt_set_slices_override_.
    uint32_t force_slices = 0;
    // mrt_render_rgba8_rows: a stream for the strip copies and two events per strip
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> strip_events;
};

namespace {

thread_local std::string g_create_err;

int fail(mrt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((ctx), MRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Makes a ctx's GPU the current device for the duration of an entry point and puts the caller's
// current device back afterwards (a process that drives several GPUs, or a framework with its own
// notion of the current device, must not find it changed).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
    }
    ~DeviceGuard() {
        int now = -1;
        if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ENTER_DEVICE(ctx)                          \
    DeviceGuard guard_((ctx)->device);              \
    HIP_TRY((ctx), guard_.err)

void free_dev(void*& p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

// Copies `src` into the device buffer `dst` of `cap` bytes, which is replaced only when the size
// changes: a scene of the same shape keeps its device addresses and costs no allocation.
template <typename T>
int upload(mrt_ctx* ctx, void*& dst, size_t& cap, const std::vector<T>& src) {
    size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    if (!dst || cap != bytes) {
        free_dev(dst);
        cap = 0;
        HIP_TRY(ctx, hipMalloc(&dst, bytes));
        cap = bytes;
    }
    if (!src.empty())
        HIP_TRY(ctx, hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return MRT_OK;
}

int check_scene(mrt_ctx* ctx, const mrt_scene* s) {
    if (!s) return fail(ctx, MRT_ERR_INVALID, "scene is NULL");
    if (s->xres == 0 || s->yres == 0) return fail(ctx, MRT_ERR_INVALID, "resolution must be positive");
    if (s->xres > 65536u || s->yres > 65536u) return fail(ctx, MRT_ERR_TOO_LARGE, "resolution above 65536");
    if (s->projection != MRT_PROJ_ORTHO && s->projection != MRT_PROJ_PERSP)
        return fail(ctx, MRT_ERR_INVALID, "unknown projection");
    if (s->shader != MRT_SHADER_FLAT && s->shader != MRT_SHADER_PHONG)
        return fail(ctx, MRT_ERR_INVALID, "unknown shader");
    if ((uint64_t)s->n_spheres + s->n_tris > 0x7FFFFFFFull) return fail(ctx, MRT_ERR_TOO_LARGE, "too many objects");
    if (s->n_spheres) {
        for (int k = 0; k < 4; ++k) if (!s->sph[k]) return fail(ctx, MRT_ERR_INVALID, "sphere plane is NULL");
        if (!s->sph_mat) return fail(ctx, MRT_ERR_INVALID, "sph_mat is NULL");
    }
    if (s->n_tris) {
        for (int k = 0; k < 9; ++k) if (!s->tri[k]) return fail(ctx, MRT_ERR_INVALID, "triangle plane is NULL");
        if (!s->tri_mat) return fail(ctx, MRT_ERR_INVALID, "tri_mat is NULL");
    }
    if (s->n_materials) for (int k = 0; k < 4; ++k) if (!s->mat[k]) return fail(ctx, MRT_ERR_INVALID, "material plane is NULL");
    if (s->n_lights) for (int k = 0; k < 6; ++k) if (!s->light[k]) return fail(ctx, MRT_ERR_INVALID, "light plane is NULL");
    for (uint32_t k = 0; k < s->n_spheres; ++k)
        if (s->sph_mat[k] >= s->n_materials) return fail(ctx, MRT_ERR_INVALID, "sphere material index out of range");
    for (uint32_t k = 0; k < s->n_tris; ++k)
        if (s->tri_mat[k] >= s->n_materials) return fail(ctx, MRT_ERR_INVALID, "triangle material index out of range");
    return MRT_OK;
}

// Working planes of a frame, grown on demand: the 64-bit key plane of trace_kernel, and the
// (t, id) planes that mrt_render_debug returns and cull_trace_kernel hands to shade_kernel.
int ensure_keys(mrt_ctx* ctx, size_t elems, hipStream_t stream) {
    if (elems <= ctx->key_elems) return MRT_OK;
    if (ctx->d_key) { (void)hipFree(ctx->d_key); ctx->d_key = nullptr; }
    ctx->key_elems = 0;
    ctx->keys_clean = false;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_key, elems * sizeof(unsigned long long)));
    ctx->key_elems = elems;
    ctx->keys_clean = false;
    const size_t tiles = elems / (mrt::TraceNarrow::tile_x * mrt::TraceNarrow::tile_y);      // the smaller of the two tile shapes
    if (tiles > ctx->tile_flag_elems) {
        if (ctx->d_tile_flags) { (void)hipFree(ctx->d_tile_flags); ctx->d_tile_flags = nullptr; }
        ctx->tile_flag_elems = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tile_flags, tiles * sizeof(uint32_t)));
        // No frame's tag.  On the stream of the frame that is about to use the new buffer: nothing
        // else knows it yet, and a later frame on another stream is ordered behind this one.  (Not
        // on the null stream: the first use of that one makes the runtime create its hardware
        // queue, 3 ms of a process's first frame.)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_tile_flags, 0, tiles * sizeof(uint32_t), stream));
        ctx->tile_flag_elems = tiles;
    }
    return MRT_OK;
}

int ensure_planes(mrt_ctx* ctx, size_t elems) {
    if (elems <= ctx->plane_elems) return MRT_OK;
    if (ctx->d_t) { (void)hipFree(ctx->d_t); ctx->d_t = nullptr; }
    if (ctx->d_id) { (void)hipFree(ctx->d_id); ctx->d_id = nullptr; }
    ctx->plane_elems = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_t, elems * sizeof(float)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_id, elems * sizeof(int32_t)));
    ctx->plane_elems = elems;
    return MRT_OK;
}

int ensure_out(mrt_ctx* ctx, size_t elems) {
    if (elems <= ctx->out_elems) return MRT_OK;
    if (ctx->d_out) { (void)hipFree(ctx->d_out); ctx->d_out = nullptr; }
    ctx->out_elems = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_out, elems * sizeof(uint32_t)));
    ctx->out_elems = elems;
    return MRT_OK;
}

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Whether `stream` is recording into a HIP graph (hipStreamBeginCapture) instead of executing.
bool stream_is_capturing(hipStream_t stream) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &status) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return status == hipStreamCaptureStatusActive;
}

// Where a frame's working planes come from: the ctx's own buffers, grown on demand, or
// planes the caller brings (no allocation; t / id may then be absent).
struct FramePlanes {
    unsigned long long* key = nullptr;   // caller-provided key plane, or nullptr = the ctx's own
    size_t key_capacity = 0;             // keys the caller's plane holds
    bool want_tid = false;               // write the (t, id) planes (ctx-owned)
    bool own = true;                     // the ctx may (re)allocate
};

// Enqueue the frame (one launch, two for the tile cull or a sliced trace) for this rank's rows
// on `stream`.
int launch_frame(mrt_ctx* ctx, uint32_t band_rows, uint32_t rank, uint32_t world, uint32_t* dev_out,
                 hipStream_t stream, bool timed, const FramePlanes& planes) {
    // A stream that is recording a HIP graph is refused: replaying a recorded frame ended in a GPU
    // memory fault every time it was tried (default and full-test kernels alike), so the call
    // fails cleanly here, before anything is enqueued, and the capture stays valid.
    const bool capturing = stream_is_capturing(stream);
#ifndef MRT_ALLOW_CAPTURE
    if (capturing)
        return fail(ctx, MRT_ERR_INVALID,
                    "the stream is capturing a HIP graph: frames cannot be recorded into a graph (not supported)");
#else
    // Diagnostic build only (tools/graph_frame_probe.cpp): the frame that allocates nothing may be
    // recorded, provided nothing of this context is pending on another stream; no event of the
    // context is recorded into the graph or waited on by it.
    if (capturing && (planes.own || timed))
        return fail(ctx, MRT_ERR_INVALID, "only mrt_render_bands_device_noalloc can be captured");
    if (capturing && ((ctx->frame_pending && stream != ctx->last_stream) || (ctx->records_pending && stream != ctx->stream)))
        return fail(ctx, MRT_ERR_INVALID, "work of this context is pending on another stream: mrt_synchronize first");
#endif
    RenderParams p = ctx->params;
    p.band_rows = band_rows;
    p.rank = rank;
    p.world = world;
    p.local_rows = mrt_band_rows(p.yres, band_rows, rank, world);
    p.out = dev_out;
    p.flags = ctx->flags;
    p.exact_counter = (ctx->flags & MRT_FLAG_COUNT_EXACT) ? ctx->d_counter : nullptr;
    if (p.local_rows == 0) {
        if (timed) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev[0], stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev[1], stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev[2], stream));
        }
        return MRT_OK;
    }
    // trace_kernel's lane block: 8x4 pixels (64x32 wave tiles) for frames of at least 3 Mpixel,
    // 4x4 (32x32 tiles) below, where four times more tiles matter more than 10 % fewer
    // instructions (teatime alone: 1080p 0.336 ms narrow, 0.360 wide; 1440p 0.570 narrow, 0.534
    // wide); MRT_FLAG_TRACE_NARROW / _WIDE override.  The block reject is 4x4 only.
    // (a slot of an mrt_sequence, whose frames run beside other frames: from 2 Mpixel, where the
    // smaller tile count no longer matters -- 1080p 0.313 -> 0.303 ms, 1440p 0.534 -> 0.501)
    bool wide = (uint64_t)p.xres * p.local_rows >= ((ctx->flags & MRT_IFLAG_IN_FLIGHT) ? 2000000u : 3000000u);
    if (ctx->flags & MRT_FLAG_TRACE_NARROW) wide = false;
    if (ctx->flags & MRT_FLAG_TRACE_WIDE) wide = true;
    if (ctx->flags & (MRT_IFLAG_BLOCK_REJECT | MRT_IFLAG_SCREEN_CULL | MRT_FLAG_FULL_TEST)) wide = false;
    const uint32_t block_x = wide ? mrt::TraceWide::block_x : mrt::TraceNarrow::block_x;
    const uint32_t block_y = wide ? mrt::TraceWide::block_y : mrt::TraceNarrow::block_y;
    const uint32_t tile_x = block_x / mrt::kBlockWavesX;
    const uint32_t pitch = round_up(p.xres, block_x);
    const uint32_t prow = round_up(p.local_rows, block_y);
    const size_t plane_elems = (size_t)pitch * prow;

    // A single band (one rank, or bands at least as tall as the image) is a run of consecutive
    // image rows whatever its height; only interleaved bands constrain the block kernels.
    const bool consecutive = world == 1 || band_rows >= p.yres;
    // The block reject needs a lane's four rows to be consecutive image rows.
    const bool block = (ctx->flags & MRT_IFLAG_BLOCK_REJECT) && (consecutive || band_rows % 4 == 0);
    const bool persp = ctx->projection == MRT_PROJ_PERSP;
    const dim3 tb(mrt::kBlockThreads);
    // The tile cull needs a wave tile's 32 rows to be consecutive image rows, and it has no
    // exact-only mode of its own.
    const bool full = (ctx->flags & MRT_FLAG_FULL_TEST) != 0;
    const bool cull = !full && (ctx->flags & MRT_IFLAG_SCREEN_CULL) && !(ctx->flags & MRT_FLAG_NO_PREFILTER) &&
                      (consecutive || band_rows % mrt::kWaveTile == 0);
    if (cull && !planes.own)
        return fail(ctx, MRT_ERR_INVALID, "the tile-cull kernel needs per-pixel planes of its own; not available without allocation");

    // full_trace_kernel in layers (see the kernel): a frame of few waves -- a 4K frame is four
    // rounds of them, which start and finish in step -- is cut into about 64 K waves, each layer
    // a share of the record chunks; the layers meet in the key plane.  ctx->force_slices overrides.
    const dim3 fgrid2(pitch / mrt::kBlockTileX, prow / mrt::kFullBlockH);      // blocks of 64 x 32 pixels
    uint32_t full_layers = 1;
    if (full) {
        const uint32_t waves = fgrid2.x * fgrid2.y * (mrt::kBlockThreads / 64);
        const uint32_t chunks = p.n_tri_padded / mrt::kChunkTris;
        full_layers = ctx->force_slices ? ctx->force_slices : (65536u + waves - 1) / waves;
        if (full_layers > chunks) full_layers = chunks;
        if (full_layers < 1) full_layers = 1;
        if (!planes.own && plane_elems > planes.key_capacity) full_layers = 1;    // no room for keys: one layer needs none
    }

    // Working planes.  trace_kernel (and full_trace_kernel in layers) keeps the closest hit of
    // every pixel in a 64-bit key plane; the tile cull hands (t, id) planes to shade_kernel.
    unsigned long long* d_key = nullptr;
    float* d_t = nullptr;
    int32_t* d_id = nullptr;
    if ((!full && !cull) || full_layers > 1) {
        if (planes.own) {
            const int rc = ensure_keys(ctx, plane_elems, stream);
            if (rc != MRT_OK) return rc;
            d_key = ctx->d_key;
        } else {
            if (plane_elems > planes.key_capacity)
                return fail(ctx, MRT_ERR_TOO_LARGE,
                            "this rank's rows, padded to whole block tiles, need " + std::to_string(plane_elems) +
                                " closest-hit keys; the fixed array holds " + std::to_string(planes.key_capacity));
            d_key = planes.key;
        }
    }
    if (planes.want_tid || cull) {
        const int rc = ensure_planes(ctx, plane_elems);
        if (rc != MRT_OK) return rc;
        d_t = ctx->d_t;
        d_id = ctx->d_id;
    }
    // The planes are shared by all frames of this ctx: a frame on another stream waits for the last one.
    if (ctx->frame_pending && stream != ctx->last_stream)
        HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->frame_done, 0));
    // The per-view records are built on the ctx's own stream.
    if (ctx->records_pending && stream != ctx->stream)
        HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->records_ready, 0));
    if (p.exact_counter) HIP_TRY(ctx, hipMemsetAsync(p.exact_counter, 0, sizeof(unsigned long long), stream));

    const dim3 tgrid(pitch / block_x, prow / block_y);
    const dim3 cgrid(pitch / mrt::kBlockTileX, prow / mrt::kBlockTileY);      // cull_trace_kernel: 64 x 64
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], stream));
    // trace_kernel and full_trace_kernel shade their own pixels in the epilogue (one launch per
    // frame); 16-byte stores need rows and buffer aligned to four pixels.  The tile cull runs at
    // low occupancy and leaves shading to shade_kernel.
    const uint32_t fused = (p.xres % 4 == 0 && ((uintptr_t)dev_out & 15u) == 0) ? 2u : 1u;
    p.fuse_shade = cull ? 0u : fused;
    // trace_kernel takes the triangle list in slices, one grid layer each; the (t, id) keys merge
    // them.  A frame with few tiles needs that to fill the GPU at all (a wave alone on its SIMD
    // issues at a fraction of the rate six reach together); a 4K frame still gains from several
    // times more, shorter waves, because the machine spends less of the frame draining.  Target:
    // about 64 K waves per launch, ten times what the GPU holds (measured in tile-major order,
    // teatime: 4K best at 13-19 slices, 1080p at 26, an eighth of the 8K frame flat from 16 to 43,
    // the 8K frame flat from 4 to 9).  A slice is a whole number of circle groups, at least
    // kMinSliceTris triangles (below that the per-wave set-up dominates).
    p.prim_slices = 1;
    p.slice_tris = p.n_tri_padded;
    if (!full && !cull) {
        constexpr uint32_t kMinSliceTris = 32;
        static_assert(kMinSliceTris % mrt::kGroupCircle == 0 && mrt::kChunkTris % kMinSliceTris == 0, "slices are whole circle groups");
        const uint32_t waves = tgrid.x * tgrid.y * (mrt::kBlockThreads / 64);
        const uint32_t units = p.n_tri_padded / kMinSliceTris;
        // Beside the frames of other contexts (a slot of an mrt_sequence) the GPU is full whatever
        // this frame's wave count, and what shorter waves cost in set-up is no longer paid back:
        // about 48 K waves measured best with two frames in flight (1080p 0.343 -> 0.313 ms per
        // frame, 4K 1.095 -> 1.063, 8K 4.09 -> 4.07).
        const uint32_t target = (ctx->flags & MRT_IFLAG_IN_FLIGHT) ? 49152u : 65536u;
        uint32_t want = ctx->force_slices ? ctx->force_slices : (target + waves - 1) / waves;
        const bool enough = want <= units;       // the list is long enough for the target
        if (want > units) want = units;
        if (want > 1) {
            uint32_t per = (units + want - 1) / want;
            // In tile-major order block b is slice b mod S of tile b / S and runs on XCD b mod 8:
            // with S a multiple of 4 an XCD would work through the same S / 4 or S / 8 slices of
            // every tile, whatever they cost (4K: 16 slices 1.111 ms, 13 and 15 1.097 ms).
            // Slightly longer slices avoid that (not for a forced count, nor where the list is
            // too short for the target anyway).
            while (!ctx->force_slices && enough && per < units && ((units + per - 1) / per) % 4 == 0 &&
                   (units + per - 1) / per >= mrt::kTileMajorSlices)
                ++per;
            p.slice_tris = per * kMinSliceTris;
            p.prim_slices = (units + per - 1) / per;
        }
    } else if (full) {
        p.prim_slices = full_layers;
    }
    p.tile_major = (full || cull) ? 0u : (MRT_TILE_MAJOR < 0 ? (p.prim_slices >= mrt::kTileMajorSlices ? 1u : 0u) : (uint32_t)(MRT_TILE_MAJOR != 0));
    if (tgrid.x > 65535u || tgrid.y > 65535u) p.tile_major = 0;     // the tile would not fit grid dimensions y and z
    const dim3 zgrid = p.tile_major ? dim3(p.prim_slices, tgrid.x, tgrid.y) : dim3(tgrid.x, tgrid.y, p.prim_slices);
    const dim3 fgrid(fgrid2.x, fgrid2.y, p.prim_slices);
    p.tile_flags = nullptr;
    p.tiles_x = pitch / tile_x;
    p.tile_x_log2 = wide ? 6u : 5u;
    p.tile_y_log2 = (wide && mrt::TraceWide::tile_y == 64) ? 6u : 5u;
    static_assert(mrt::TraceWide::tile_x == 64 && mrt::TraceNarrow::tile_x == 32 && mrt::TraceNarrow::tile_y == 32 &&
                      (mrt::TraceWide::tile_y == 32 || mrt::TraceWide::tile_y == 64), "tile sizes as shifts");
    const bool keeps_keys_clean = p.prim_slices > 1 && planes.own;
    if (keeps_keys_clean) {
        // The ctx's own key plane stays all "no hit" from frame to frame: shade_keys_kernel puts
        // back every key this frame writes.  Only the tile flags (4 bytes per 1024 pixels) are
        // cleared per frame.
        if (!ctx->keys_clean)
            HIP_TRY(ctx, hipMemsetAsync(d_key, 0xFF, ctx->key_elems * sizeof(unsigned long long), stream));
        ctx->keys_clean = false;               // until this frame's shade_keys_kernel is enqueued
        // The tile flags carry a tag per frame instead of being cleared per frame.
        p.tile_flags = ctx->d_tile_flags;
        if (++ctx->frame_tag == 0) {          // wrapped after 2^32 frames: start over from clean flags
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_tile_flags, 0, ctx->tile_flag_elems * sizeof(uint32_t), stream));
            ctx->frame_tag = 1;
        }
        p.frame_tag = ctx->frame_tag;
    } else if (p.prim_slices > 1) {
        HIP_TRY(ctx, hipMemsetAsync(d_key, 0xFF, plane_elems * sizeof(unsigned long long), stream));
    } else if (d_key != nullptr && planes.own) {
        ctx->keys_clean = false;               // a single-slice frame initialises and leaves its own keys
    }
    const bool sliced = p.prim_slices > 1;       // trace_kernel without its epilogue: shade_keys_kernel finishes the frame
    uint32_t kernel;
    if (full && persp) {
        kernel = MRT_KERNEL_FULL;
        hipLaunchKernelGGL((mrt::full_trace_kernel<MRT_PROJ_PERSP>), fgrid, tb, 0, stream, p,
                           (const float4*)ctx->d_full_rec, ctx->d_full_ids, ctx->d_full_counts, d_key, d_t, d_id, pitch);
    } else if (full) {
        kernel = MRT_KERNEL_FULL;
        hipLaunchKernelGGL((mrt::full_trace_kernel<MRT_PROJ_ORTHO>), fgrid, tb, 0, stream, p,
                           (const float4*)ctx->d_full_rec, ctx->d_full_ids, ctx->d_full_counts, d_key, d_t, d_id, pitch);
    } else if (cull && persp) {
        kernel = MRT_KERNEL_CULL;
        hipLaunchKernelGGL((mrt::cull_trace_kernel<MRT_PROJ_PERSP>), cgrid, tb, 0, stream, p, d_t, d_id, pitch);
    } else if (cull) {
        kernel = MRT_KERNEL_CULL;
        hipLaunchKernelGGL((mrt::cull_trace_kernel<MRT_PROJ_ORTHO>), cgrid, tb, 0, stream, p, d_t, d_id, pitch);
    } else if (persp && block) {
        kernel = MRT_KERNEL_TRACE_BLOCK;
        if (sliced) hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_PERSP, true, 4, 4, true>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
        else hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_PERSP, true, 4, 4, false>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
    } else if (persp && wide) {
        kernel = MRT_KERNEL_TRACE_WIDE;
        if (sliced) hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_PERSP, false, mrt::TraceWide::lane_x, mrt::TraceWide::lane_y, true>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
        else hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_PERSP, false, mrt::TraceWide::lane_x, mrt::TraceWide::lane_y, false>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
    } else if (persp) {
        kernel = MRT_KERNEL_TRACE;
        if (sliced) hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_PERSP, false, 4, 4, true>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
        else hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_PERSP, false, 4, 4, false>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
    } else if (block) {
        kernel = MRT_KERNEL_TRACE_BLOCK;
        if (sliced) hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_ORTHO, true, 4, 4, true>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
        else hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_ORTHO, true, 4, 4, false>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
    } else if (wide) {
        kernel = MRT_KERNEL_TRACE_WIDE;
        if (sliced) hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_ORTHO, false, mrt::TraceWide::lane_x, mrt::TraceWide::lane_y, true>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
        else hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_ORTHO, false, mrt::TraceWide::lane_x, mrt::TraceWide::lane_y, false>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
    } else {
        kernel = MRT_KERNEL_TRACE;
        if (sliced) hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_ORTHO, false, 4, 4, true>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
        else hipLaunchKernelGGL((mrt::trace_kernel<MRT_PROJ_ORTHO, false, 4, 4, false>), zgrid, tb, 0, stream, p, d_key, d_t, d_id, pitch);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], stream));
    if (p.prim_slices > 1) {
        // with tile flags the launch covers the padded plane: padding pixels hold keys too
        const dim3 sgrid(p.tile_flags ? pitch / 64 : (p.xres + 63) / 64, p.tile_flags ? prow / 4 : (p.local_rows + 3) / 4);
        if (persp)
            hipLaunchKernelGGL(mrt::shade_keys_kernel<MRT_PROJ_PERSP>, sgrid, dim3(256), 0, stream, p, d_key, d_t, d_id, pitch);
        else
            hipLaunchKernelGGL(mrt::shade_keys_kernel<MRT_PROJ_ORTHO>, sgrid, dim3(256), 0, stream, p, d_key, d_t, d_id, pitch);
        HIP_TRY(ctx, hipGetLastError());
        if (keeps_keys_clean) ctx->keys_clean = true;
    } else if (!p.fuse_shade) {
        const dim3 sgrid((p.xres + 63) / 64, (p.local_rows + 3) / 4);
        if (persp)
            hipLaunchKernelGGL(mrt::shade_kernel<MRT_PROJ_PERSP>, sgrid, dim3(256), 0, stream, p, d_t, d_id, pitch);
        else
            hipLaunchKernelGGL(mrt::shade_kernel<MRT_PROJ_ORTHO>, sgrid, dim3(256), 0, stream, p, d_t, d_id, pitch);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], stream));
    ctx->last_kernel = kernel;
    ctx->last_slices = p.prim_slices;
    ctx->last_pitch = pitch;
    // Every frame, whatever planes it worked in, reads the per-view records: mrt_set_camera orders
    // its rebuild behind this event.
    if (!capturing) {
        HIP_TRY(ctx, hipEventRecord(ctx->frame_done, stream));
        ctx->last_stream = stream;
        ctx->frame_pending = true;
    }
    return MRT_OK;
}

int fill_stats(mrt_ctx* ctx, mrt_stats* st, uint32_t rows) {
    if (!st) return MRT_OK;
    float a = 0.f, b = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->ev[1], ctx->ev[2]));
    st->upload_ms = ctx->upload_ms;
    st->trace_ms = a;
    st->shade_ms = b;
    st->kernel_ms = (double)a + (double)b;
    st->rays = (uint64_t)rows * ctx->params.xres;
    st->prim_tests = st->rays * ((uint64_t)ctx->params.n_sph + ctx->params.n_tri);
    st->exact_tests = 0;
    st->kernel = ctx->last_kernel;
    st->prim_slices = ctx->last_slices;
    if (ctx->flags & MRT_FLAG_COUNT_EXACT) {
        unsigned long long c = 0;
        HIP_TRY(ctx, hipMemcpy(&c, ctx->d_counter, sizeof c, hipMemcpyDeviceToHost));
        st->exact_tests = c;
    }
    return MRT_OK;
}

}  // namespace

// Shared with mrt_stages.hip (not part of the public ABI).
extern "C" int mrt_stage_set_error_(mrt_ctx* ctx, int code, const char* msg) { return fail(ctx, code, msg ? msg : ""); }
extern "C" int mrt_stage_device_(const mrt_ctx* ctx) { return ctx ? ctx->device : 0; }
// Staging buffer `slot` of the ctx with room for `bytes` (the ctx's GPU must be current); the
// previous contents are not kept when it grows.  nullptr + the ctx's error on failure.
extern "C" void* mrt_stage_buffer_(mrt_ctx* ctx, int slot, size_t bytes) {
    if (!ctx || slot < 0 || slot >= mrt_ctx::kStageSlots) return nullptr;
    if (bytes == 0) bytes = 1;
    if (bytes > ctx->stage_cap[slot]) {
        // an earlier stage call may still be reading the old buffer on the ctx's stream
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->stage_buf[slot]) (void)hipFree(ctx->stage_buf[slot]);
        ctx->stage_buf[slot] = nullptr;
        ctx->stage_cap[slot] = 0;
        const hipError_t e = hipMalloc(&ctx->stage_buf[slot], bytes);
        if (e != hipSuccess) {
            ctx->stage_buf[slot] = nullptr;
            fail(ctx, MRT_ERR_HIP, std::string("hipMalloc of a stage buffer: ") + hipGetErrorString(e));
            return nullptr;
        }
        ctx->stage_cap[slot] = bytes;
    }
    return ctx->stage_buf[slot];
}
// Page-locked host memory of the ctx with room for `bytes` (the ctx's GPU must be current): where a
// stage call's results arrive from the GPU -- at the link's rate, with no page of a fresh
// allocation to fault in -- before the host scatters them into the caller's records.  The stage
// calls block, so one buffer serves them all.  nullptr + the ctx's error on failure.
extern "C" void* mrt_stage_host_buffer_(mrt_ctx* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    if (bytes == 0) bytes = 1;
    if (bytes > ctx->stage_host_cap) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
        ctx->stage_host = nullptr;
        ctx->stage_host_cap = 0;
        const hipError_t e = hipHostMalloc(&ctx->stage_host, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            ctx->stage_host = nullptr;
            fail(ctx, MRT_ERR_HIP, std::string("hipHostMalloc of the stage calls' host buffer: ") + hipGetErrorString(e));
            return nullptr;
        }
        ctx->stage_host_cap = bytes;
    }
    return ctx->stage_host;
}
// The ctx's timing events (0, 1): the stage calls block, so they never overlap a timed frame.
extern "C" void* mrt_stage_event_(mrt_ctx* ctx, int which) { return (void*)ctx->ev[which & 1]; }
extern "C" void mrt_stage_set_kernel_ms_(mrt_ctx* ctx, double ms) { ctx->stage_kernel_ms = ms; }
extern "C" double mrt_stage_get_kernel_ms_(const mrt_ctx* ctx) { return ctx->stage_kernel_ms; }
// Tests and tuning scripts: waves of the last mrt_stage_intersect that switched to the filter with
// the fourth determinant.  Not part of the public ABI.
extern "C" void mrt_stage_set_t_switches_(mrt_ctx* ctx, uint32_t n) { ctx->stage_t_switches = n; }
extern "C" uint32_t mrt_stage_last_t_switches_(const mrt_ctx* ctx) { return ctx ? ctx->stage_t_switches : 0u; }
// Tests and tuning scripts: cut the primitive list of this ctx's frames into n slices / layers
// (clamped to what the frame allows), 0 = the automatic choice.  Not part of the public ABI.
extern "C" void mrt_set_slices_override_(mrt_ctx* ctx, uint32_t n) { if (ctx) ctx->force_slices = n > 65536u ? 65536u : n; }
// Shared with mrt_accel.hip (mrt_internal.hpp).
extern "C" int mrt_stream_is_capturing_(void* hip_stream) { return stream_is_capturing((hipStream_t)hip_stream) ? 1 : 0; }
extern "C" const mrt::RenderParams* mrt_ctx_params_(const mrt_ctx* ctx) { return &ctx->params; }
extern "C" void* mrt_ctx_stream_(const mrt_ctx* ctx) { return (void*)ctx->stream; }
extern "C" int mrt_ctx_projection_(const mrt_ctx* ctx) { return ctx->projection; }

extern "C" {

int mrt_abi_version(void) { return MRT_ABI_VERSION; }

const char* mrt_last_error(const mrt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mrt_create(const mrt_config* cfg, mrt_ctx** out) try {
    if (!out) return fail(nullptr, MRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (cfg && (cfg->flags & MRT_IFLAG_MASK))
        return fail(nullptr, MRT_ERR_INVALID,
                    "flag bits 2, 4 and 30 are reserved: screen-space culling is behind mrt_accel_create, not a flag of the brute-force context");
    if (cfg && (cfg->flags & MRT_FLAG_TRACE_NARROW) && (cfg->flags & MRT_FLAG_TRACE_WIDE))
        return fail(nullptr, MRT_ERR_INVALID, "MRT_FLAG_TRACE_NARROW and MRT_FLAG_TRACE_WIDE exclude each other");
    return mrt_create_internal_(cfg, out);
} MRT_GUARD((void)fail(nullptr, code_, what_))

int mrt_create_internal_(const mrt_config* cfg, mrt_ctx** out) try {
    if (!out) return fail(nullptr, MRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    // MRT_STATS=1: where the start-up of the first context of a process goes, one JSON line on stderr
    using clk = std::chrono::steady_clock;
    const char* want_stats = std::getenv("MRT_STATS");
    static bool first_context = true;                  // (the start-up is the first context's; later ones find it done)
    const bool stats = want_stats && want_stats[0] == '1' && first_context;
    first_context = false;
    const auto t0 = clk::now();
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    const double runtime_ms = ms_since(t0);            // the HIP runtime starts up here
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, MRT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= count) return fail(nullptr, MRT_ERR_NO_DEVICE, "device ordinal out of range");
    mrt_ctx* ctx = new mrt_ctx();
    struct Owner {                                     // an exception on the way out must not leak the ctx
        mrt_ctx* c;
        ~Owner() { if (c) mrt_destroy(c); }
    } owner{ctx};
    ctx->device = dev;
    ctx->flags = cfg ? cfg->flags : 0;
    if (const char* force = std::getenv("MRT_FORCE_SLICES")) {
        char* end = nullptr;
        const long v = std::strtol(force, &end, 10);
        if (end != force && *end == '\0' && v >= 1 && v <= 65536) ctx->force_slices = (uint32_t)v;
    }
    auto bail = [&](const char* what, hipError_t err) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(err);
        owner.c = nullptr;
        mrt_destroy(ctx);
        return fail(nullptr, MRT_ERR_HIP, msg);
    };
    DeviceGuard guard(dev);
    if (guard.err != hipSuccess) return bail("hipSetDevice", guard.err);
    const auto t1 = clk::now();
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    const double stream_ms = ms_since(t1);             // the device is opened, a hardware queue made
    for (int k = 0; k < 3; ++k)
        if ((e = hipEventCreate(&ctx->ev[k])) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->frame_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->records_ready, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    const auto t2 = clk::now();
    if ((e = hipMalloc((void**)&ctx->d_counter, sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    const double malloc_ms = ms_since(t2);
    const auto t3 = clk::now();
    // The fixed array of mrt_render_bands_device_noalloc is brought onto the GPU here, once per
    // device, unless the caller has said that it will not use that entry point (or cannot: the
    // contexts inside mrt_multi / mrt_sequence / mrt_accel handles); the entry point's first call
    // on the device does it then.
    if (!(ctx->flags & MRT_FLAG_DEFER_NOALLOC_SETUP))
        if ((e = scratch_address(dev, &ctx->scratch_keys)) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    const double code_object_ms = ms_since(t3);        // the code object that holds the 64 MiB array is loaded onto the GPU
    if (stats)
        std::fprintf(stderr, "{\"mrt_create\": {\"runtime_start_ms\": %.3f, \"first_stream_ms\": %.3f, \"first_malloc_ms\": %.3f, "
                             "\"code_object_load_ms\": %.3f, \"total_ms\": %.3f}}\n",
                     runtime_ms, stream_ms, malloc_ms, code_object_ms, ms_since(t0));
    owner.c = nullptr;
    *out = ctx;
    return MRT_OK;
} MRT_GUARD((void)fail(nullptr, code_, what_))

void mrt_destroy(mrt_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    // Frames may be in flight on the caller's streams too.
    (void)hipDeviceSynchronize();
    free_dev(ctx->d_sph_filter); free_dev(ctx->d_tri_filter); free_dev(ctx->d_tri_circle); free_dev(ctx->d_tri_exact); free_dev(ctx->d_full_rec); free_dev(ctx->d_sph);
    if (ctx->d_full_ids) (void)hipFree(ctx->d_full_ids);
    if (ctx->d_full_counts) (void)hipFree(ctx->d_full_counts);
    free_dev(ctx->d_sph_mat); free_dev(ctx->d_mats); free_dev(ctx->d_lights);
    if (ctx->d_t) (void)hipFree(ctx->d_t);
    if (ctx->d_id) (void)hipFree(ctx->d_id);
    if (ctx->d_key) (void)hipFree(ctx->d_key);
    if (ctx->d_tile_flags) (void)hipFree(ctx->d_tile_flags);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    for (int k = 0; k < mrt_ctx::kStageSlots; ++k) if (ctx->stage_buf[k]) (void)hipFree(ctx->stage_buf[k]);
    if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
    for (int k = 0; k < 3; ++k) if (ctx->ev[k]) (void)hipEventDestroy(ctx->ev[k]);
    for (hipEvent_t e : ctx->strip_events) if (e) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->frame_done) (void)hipEventDestroy(ctx->frame_done);
    if (ctx->records_ready) (void)hipEventDestroy(ctx->records_ready);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// (Re)allocates a device buffer when its size changes.
static int reserve(mrt_ctx* ctx, void*& dst, size_t& cap_bytes, size_t bytes) {
    if (!dst || cap_bytes != bytes) {
        free_dev(dst);
        cap_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&dst, bytes));
        cap_bytes = bytes;
    }
    return MRT_OK;
}

// Derives the per-view records from ctx->h_scene's header and the geometry already on the GPU
// (kernels of mrt_records.hip on ctx->stream, nothing is copied and nothing waits) and refreshes
// the camera part of the kernel parameters.  may_allocate: called from mrt_set_scene, where the
// record buffers may change size; mrt_set_camera finds them as mrt_set_scene left them.
static int rebuild_view(mrt_ctx* ctx, bool may_allocate) {
    const mrt_scene* s = &ctx->h_scene;
    RenderParams& p = ctx->params;
    const uint32_t n_tri_padded = round_up(s->n_tris ? s->n_tris : 1, mrt::kChunkTris);
    const uint32_t n_sph_padded = round_up(s->n_spheres ? s->n_spheres : 1, mrt::kChunkFloat4);
    // Records of full_trace_kernel: one chunk of slack, the kernel stages whole LDS chunks from
    // wherever a sign group starts (it never tests what lies beyond the group).
    const size_t n_full = (size_t)n_tri_padded + mrt::kChunkTris;
    const size_t full_bytes = n_full * mrt::kFullRecFloat4 * sizeof(float4);
    const bool full = (ctx->flags & MRT_FLAG_FULL_TEST) != 0;
    if (may_allocate) {
        int rc;
        if ((rc = reserve(ctx, ctx->d_sph_filter, ctx->cap_sph_filter, (size_t)n_sph_padded * sizeof(float4))) != MRT_OK) return rc;
        if ((rc = reserve(ctx, ctx->d_tri_filter, ctx->cap_tri_filter, (size_t)n_tri_padded * 3 * sizeof(float4))) != MRT_OK) return rc;
        if ((rc = reserve(ctx, ctx->d_tri_circle, ctx->cap_tri_circle, (size_t)n_tri_padded * sizeof(float4))) != MRT_OK) return rc;
        if (full) {
            if (!ctx->d_full_rec || ctx->cap_full_rec != full_bytes) {
                free_dev(ctx->d_full_rec);
                if (ctx->d_full_ids) { (void)hipFree(ctx->d_full_ids); ctx->d_full_ids = nullptr; }
                ctx->cap_full_rec = 0;
                HIP_TRY(ctx, hipMalloc(&ctx->d_full_rec, full_bytes));
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_full_ids, n_full * sizeof(uint32_t)));
                ctx->cap_full_rec = full_bytes;
            }
            if (!ctx->d_full_counts) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_full_counts, 2 * sizeof(uint32_t)));
        }
    }

    for (int k = 0; k < 3; ++k) { p.eye[k] = s->eye[k]; p.u[k] = s->u[k]; p.v[k] = s->v[k]; p.w[k] = s->w[k]; }
    p.left = s->left; p.right = s->right; p.top = s->top; p.bottom = s->bottom;
    // viewport::to_struct (gfxraytrace.hpp:736-740) stores the resolution as floats.
    p.xres_f = (float)s->xres; p.yres_f = (float)s->yres;
    p.focal = s->focal;
    {
        // Per component, the largest magnitude a primary-ray direction and origin can have
        // (persp_viewrays, kernel.cl:140-147: origin eye, direction -w f + u uv.x + v uv.y;
        // ortho_viewrays, :130-138: origin u uv.x + v uv.y, direction -w), with room for
        // rounding.  uv of the last pixel lies half a pixel beyond the viewport edge
        // (kernel.cl:126-127); one whole pixel is allowed for here.
        const double mux = std::fmax(std::fabs((double)s->left), std::fabs((double)s->right)) +
                           std::fabs((double)s->right - (double)s->left) / s->xres;
        const double muy = std::fmax(std::fabs((double)s->top), std::fabs((double)s->bottom)) +
                           std::fabs((double)s->top - (double)s->bottom) / s->yres;
        for (int k = 0; k < 3; ++k) {
            const double span = mux * std::fabs((double)s->u[k]) + muy * std::fabs((double)s->v[k]);
            const double dir = s->projection == MRT_PROJ_PERSP
                                   ? std::fabs((double)s->focal) * std::fabs((double)s->w[k]) + span
                                   : std::fabs((double)s->w[k]);
            const double org = s->projection == MRT_PROJ_PERSP ? std::fabs((double)s->eye[k]) : span;
            p.ray_abs[k] = (float)(dir * 1.01);
            p.org_abs[k] = (float)(org * 1.01);
        }
    }
    p.xres = s->xres; p.yres = s->yres;
    p.n_tri_padded = n_tri_padded;
    p.n_sph_padded = n_sph_padded;
    p.sph_filter = (const float4*)ctx->d_sph_filter;
    p.tri_filter = (const float4*)ctx->d_tri_filter;
    p.tri_circle = (const float4*)ctx->d_tri_circle;
    ctx->projection = s->projection;

    // The view as the record kernels see it.
    mrt::ViewDerived w{};
    for (int k = 0; k < 3; ++k) { w.eye[k] = s->eye[k]; w.u[k] = s->u[k]; w.v[k] = s->v[k]; w.w[k] = s->w[k]; }
    w.focal = s->focal;
    w.xres = s->xres; w.yres = s->yres;
    // uv(px) = u0 + u1*px for 0-based pixel index px (kernel.cl:126-127 with pos = px + 1);
    // (right-left) is a float subtraction in the reference.
    w.ux1 = (double)(float)(s->right - s->left) / (double)(float)s->xres;
    w.uy1 = (double)(float)(s->top - s->bottom) / (double)(float)s->yres;
    w.ux0 = (double)s->left + 1.5 * w.ux1;
    w.uy0 = (double)s->bottom + 1.5 * w.uy1;
    w.mux = std::fmax(std::fabs(w.ux0), std::fabs(w.ux0 + w.ux1 * (w.xres - 1)));
    w.muy = std::fmax(std::fabs(w.uy0), std::fabs(w.uy0 + w.uy1 * (w.yres - 1)));
    w.persp = s->projection == MRT_PROJ_PERSP;
    {
        auto dot3d = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
        const double tol = 1e-4;
        w.orthonormal = std::fabs(dot3d(w.u, w.u) - 1) < tol && std::fabs(dot3d(w.v, w.v) - 1) < tol &&
                        std::fabs(dot3d(w.w, w.w) - 1) < tol && std::fabs(dot3d(w.u, w.v)) < tol &&
                        std::fabs(dot3d(w.u, w.w)) < tol && std::fabs(dot3d(w.v, w.w)) < tol;
    }
    hipLaunchKernelGGL(mrt::build_triangle_records, dim3((n_tri_padded + 127) / 128), dim3(128), 0, ctx->stream, w,
                       (const float4*)ctx->d_tri_exact, s->n_tris, n_tri_padded, (float4*)ctx->d_tri_filter,
                       (float4*)ctx->d_tri_circle);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(mrt::build_sphere_records, dim3((n_sph_padded + 127) / 128), dim3(128), 0, ctx->stream, w,
                       (const float4*)ctx->d_sph, s->n_spheres, n_sph_padded, (float4*)ctx->d_sph_filter);
    HIP_TRY(ctx, hipGetLastError());

    if (full) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_full_counts, 0, 2 * sizeof(uint32_t), ctx->stream));
        if (s->n_tris) {
            const dim3 g((s->n_tris + 255) / 256), b(256);
            if (s->projection == MRT_PROJ_PERSP)
                hipLaunchKernelGGL(mrt::prep_full_records<MRT_PROJ_PERSP>, g, b, 0, ctx->stream, p,
                                   (float4*)ctx->d_full_rec, ctx->d_full_ids, ctx->d_full_counts);
            else
                hipLaunchKernelGGL(mrt::prep_full_records<MRT_PROJ_ORTHO>, g, b, 0, ctx->stream, p,
                                   (float4*)ctx->d_full_rec, ctx->d_full_ids, ctx->d_full_counts);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    // Frames on other streams wait for this event (launch_frame).
    HIP_TRY(ctx, hipEventRecord(ctx->records_ready, ctx->stream));
    ctx->records_pending = true;
    return MRT_OK;
}

int mrt_set_scene(mrt_ctx* ctx, const mrt_scene* s) try {
    if (!ctx) return MRT_ERR_INVALID;
    int rc = check_scene(ctx, s);
    if (rc != MRT_OK) return rc;
    ENTER_DEVICE(ctx);
    // Frames already enqueued on any stream still read the buffers this call replaces.
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->frame_pending = false;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    ctx->have_scene = false;

    // Host copy of the header; the plane pointers are the caller's and are not kept.
    ctx->h_scene = *s;
    for (int k = 0; k < 9; ++k) ctx->h_scene.tri[k] = nullptr;
    for (int k = 0; k < 4; ++k) { ctx->h_scene.sph[k] = nullptr; ctx->h_scene.mat[k] = nullptr; }
    for (int k = 0; k < 6; ++k) ctx->h_scene.light[k] = nullptr;
    ctx->h_scene.sph_mat = nullptr; ctx->h_scene.tri_mat = nullptr;

    std::vector<float4> exact((size_t)s->n_tris * 3);
    for (uint32_t k = 0; k < s->n_tris; ++k) {
        float mbits;
        uint32_t m = s->tri_mat[k];
        std::memcpy(&mbits, &m, 4);
        exact[3 * (size_t)k + 0] = make_float4(s->tri[0][k], s->tri[1][k], s->tri[2][k], mbits);
        exact[3 * (size_t)k + 1] = make_float4(s->tri[3][k], s->tri[4][k], s->tri[5][k], 0.f);
        exact[3 * (size_t)k + 2] = make_float4(s->tri[6][k], s->tri[7][k], s->tri[8][k], 0.f);
    }
    std::vector<float4> sph(s->n_spheres);
    std::vector<uint32_t> sph_mat(s->n_spheres);
    for (uint32_t k = 0; k < s->n_spheres; ++k) {
        sph[k] = make_float4(s->sph[0][k], s->sph[1][k], s->sph[2][k], s->sph[3][k]);
        sph_mat[k] = s->sph_mat[k];
    }
    std::vector<float4> mats(s->n_materials);
    for (uint32_t k = 0; k < s->n_materials; ++k)
        mats[k] = make_float4(s->mat[0][k], s->mat[1][k], s->mat[2][k], s->mat[3][k]);
    std::vector<float4> lights((size_t)s->n_lights * 2);
    for (uint32_t k = 0; k < s->n_lights; ++k) {
        lights[2 * (size_t)k] = make_float4(s->light[0][k], s->light[1][k], s->light[2][k], 0.f);
        lights[2 * (size_t)k + 1] = make_float4(s->light[3][k], s->light[4][k], s->light[5][k], 0.f);
    }

    if ((rc = upload(ctx, ctx->d_tri_exact, ctx->cap_tri_exact, exact)) != MRT_OK) return rc;
    if ((rc = upload(ctx, ctx->d_sph, ctx->cap_sph, sph)) != MRT_OK) return rc;
    if ((rc = upload(ctx, ctx->d_sph_mat, ctx->cap_sph_mat, sph_mat)) != MRT_OK) return rc;
    if ((rc = upload(ctx, ctx->d_mats, ctx->cap_mats, mats)) != MRT_OK) return rc;
    if ((rc = upload(ctx, ctx->d_lights, ctx->cap_lights, lights)) != MRT_OK) return rc;

    RenderParams& p = ctx->params;
    std::memset(&p, 0, sizeof p);
    for (int k = 0; k < 3; ++k) { p.amb[k] = s->ambient[k]; p.bg[k] = s->background[k]; }
    p.ka = s->ka; p.kd = s->kd; p.ks = s->ks;
    p.shader = s->shader;
    p.n_sph = s->n_spheres; p.n_tri = s->n_tris;
    p.n_lights = s->n_lights;
    p.tri_exact = (const float4*)ctx->d_tri_exact;
    p.sph = (const float4*)ctx->d_sph;
    p.sph_mat = (const uint32_t*)ctx->d_sph_mat;
    p.mats = (const float4*)ctx->d_mats;
    p.lights = (const float4*)ctx->d_lights;

    if ((rc = rebuild_view(ctx, true)) != MRT_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    // The staging vectors above are pageable and die with this call.
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[1]));
    ctx->records_pending = false;              // the record kernels ran before ev[1]
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->upload_ms = ms;
    ctx->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_set_camera(mrt_ctx* ctx, const mrt_camera* cam) try {
    if (!ctx) return MRT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, MRT_ERR_NO_SCENE, "mrt_set_scene has not succeeded on this context");
    if (!cam) return fail(ctx, MRT_ERR_INVALID, "camera is NULL");
    if (cam->xres == 0 || cam->yres == 0) return fail(ctx, MRT_ERR_INVALID, "resolution must be positive");
    if (cam->xres > 65536u || cam->yres > 65536u) return fail(ctx, MRT_ERR_TOO_LARGE, "resolution above 65536");
    if (cam->projection != MRT_PROJ_ORTHO && cam->projection != MRT_PROJ_PERSP)
        return fail(ctx, MRT_ERR_INVALID, "unknown projection");
    ENTER_DEVICE(ctx);
    // Asynchronous: the record kernels run on the ctx's stream behind the last frame enqueued
    // (which still reads the old records, on whatever stream), and the next frame waits for them.
    if (ctx->frame_pending && ctx->last_stream != ctx->stream)
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->frame_done, 0));
    mrt_scene& h = ctx->h_scene;
    for (int k = 0; k < 3; ++k) { h.eye[k] = cam->eye[k]; h.u[k] = cam->u[k]; h.v[k] = cam->v[k]; h.w[k] = cam->w[k]; }
    h.xres = cam->xres; h.yres = cam->yres;
    h.left = cam->left; h.right = cam->right; h.top = cam->top; h.bottom = cam->bottom;
    h.projection = cam->projection;
    h.focal = cam->focal;
    ctx->have_scene = false;
    int rc = rebuild_view(ctx, false);
    if (rc != MRT_OK) return rc;
    ctx->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_synchronize(mrt_ctx* ctx) try {
    if (!ctx) return MRT_ERR_INVALID;
    ENTER_DEVICE(ctx);
    if (ctx->records_pending) HIP_TRY(ctx, hipEventSynchronize(ctx->records_ready));
    ctx->records_pending = false;
    if (ctx->frame_pending) HIP_TRY(ctx, hipEventSynchronize(ctx->frame_done));
    ctx->frame_pending = false;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MRT_OK;
} MRT_GUARD((void)fail(ctx, code_, what_))

uint32_t mrt_band_rows(uint32_t yres, uint32_t band_rows, uint32_t rank, uint32_t world) {
    if (band_rows == 0 || world == 0 || rank >= world) return 0;
    const uint64_t n_bands = ((uint64_t)yres + band_rows - 1) / band_rows;
    if (n_bands <= rank) return 0;
    const uint64_t mine = (n_bands - rank + world - 1) / world;     // bands rank, rank+world, ...
    uint64_t rows = mine * band_rows;
    const uint64_t last_band = n_bands - 1;
    if (last_band % world == rank) rows -= n_bands * band_rows - yres;   // the short last band is mine
    return (uint32_t)rows;
}

uint32_t mrt_balanced_band_rows(uint32_t yres, uint32_t world) {
    if (yres == 0 || world <= 1) return 64;
    // the band height (a multiple of the 32 rows of a wave tile -- bands that cut tiles cost more
    // than they balance: 8K over two ranks 2.5 % slower with bands of 48 rows than of 64 -- and at
    // most 64) that leaves the busiest rank the fewest rows; among equals the tallest
    uint32_t best = 64, best_rows = 0xFFFFFFFFu;
    for (uint32_t b = 64; b >= 32; b -= 32) {
        uint32_t most = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t rows = mrt_band_rows(yres, b, r, world);
            if (rows > most) most = rows;
        }
        if (most < best_rows) { best_rows = most; best = b; }
    }
    return best;
}

int mrt_render_bands_device(mrt_ctx* ctx, uint32_t band_rows, uint32_t rank, uint32_t world,
                            void* dev_out, size_t dev_out_bytes, void* hip_stream, mrt_stats* stats) try {
    if (!ctx) return MRT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, MRT_ERR_NO_SCENE, "mrt_set_scene has not succeeded on this context");
    if (band_rows == 0 || world == 0 || rank >= world) return fail(ctx, MRT_ERR_INVALID, "bad band partition");
    const uint32_t rows = mrt_band_rows(ctx->params.yres, band_rows, rank, world);
    const size_t need = (size_t)rows * ctx->params.xres * 4;
    if (need && !dev_out) return fail(ctx, MRT_ERR_INVALID, "dev_out is NULL");
    if (dev_out_bytes < need) return fail(ctx, MRT_ERR_INVALID, "dev_out too small for this rank's rows");
    if (((uintptr_t)dev_out & 3u) != 0) return fail(ctx, MRT_ERR_INVALID, "dev_out must be 4-byte aligned");
    ENTER_DEVICE(ctx);
    hipStream_t stream = (hipStream_t)hip_stream;
    int rc = launch_frame(ctx, band_rows, rank, world, (uint32_t*)dev_out, stream, stats != nullptr, FramePlanes{});
    if (rc != MRT_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2]));
        rc = fill_stats(ctx, stats, rows);
    }
    return rc;
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_render_bands_device_noalloc(mrt_ctx* ctx, uint32_t band_rows, uint32_t rank, uint32_t world,
                                    void* dev_out, size_t dev_out_bytes, void* hip_stream) try {
    if (!ctx) return MRT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, MRT_ERR_NO_SCENE, "mrt_set_scene has not succeeded on this context");
    if (band_rows == 0 || world == 0 || rank >= world) return fail(ctx, MRT_ERR_INVALID, "bad band partition");
    const uint32_t rows = mrt_band_rows(ctx->params.yres, band_rows, rank, world);
    const size_t need = (size_t)rows * ctx->params.xres * 4;
    if (need && !dev_out) return fail(ctx, MRT_ERR_INVALID, "dev_out is NULL");
    if (dev_out_bytes < need) return fail(ctx, MRT_ERR_INVALID, "dev_out too small for this rank's rows");
    if (((uintptr_t)dev_out & 3u) != 0) return fail(ctx, MRT_ERR_INVALID, "dev_out must be 4-byte aligned");
    ENTER_DEVICE(ctx);
    // The key plane is this GPU's copy of the fixed array, shared by every such call on the GPU and
    // holding whatever the last one left: launch_frame clears the part this frame uses (a memset
    // for a sliced frame, the kernel's prologue otherwise) and never allocates in this mode.
    // (a context created with MRT_FLAG_DEFER_NOALLOC_SETUP: this device's copy of the array is
    // taken now, on the entry point's first call on the device)
    if (!ctx->scratch_keys) HIP_TRY(ctx, scratch_address(ctx->device, &ctx->scratch_keys));
    FramePlanes planes;
    planes.own = false;
    planes.key = ctx->scratch_keys;
    planes.key_capacity = MRT_NOALLOC_MAX_KEYS;
    return launch_frame(ctx, band_rows, rank, world, (uint32_t*)dev_out, (hipStream_t)hip_stream, false, planes);
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_render_debug(mrt_ctx* ctx, uint8_t* host_out, int32_t* ids, float* t, mrt_stats* stats) try {
    if (!ctx) return MRT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, MRT_ERR_NO_SCENE, "mrt_set_scene has not succeeded on this context");
    if (!host_out && !ids && !t) return fail(ctx, MRT_ERR_INVALID, "no output requested");
    ENTER_DEVICE(ctx);
    const uint32_t xres = ctx->params.xres, yres = ctx->params.yres;
    const size_t npx = (size_t)xres * yres;
    int rc = ensure_out(ctx, npx);
    if (rc != MRT_OK) return rc;
    FramePlanes planes;
    planes.want_tid = ids != nullptr || t != nullptr;
    rc = launch_frame(ctx, yres, 0, 1, ctx->d_out, ctx->stream, true, planes);
    if (rc != MRT_OK) return rc;
    mrt_stats local;
    std::memset(&local, 0, sizeof local);
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2]));
    rc = fill_stats(ctx, &local, yres);
    if (rc != MRT_OK) return rc;
    hipEvent_t d0, d1;
    HIP_TRY(ctx, hipEventCreate(&d0));
    HIP_TRY(ctx, hipEventCreate(&d1));
    (void)hipEventRecord(d0, ctx->stream);
    hipError_t e = hipSuccess;
    if (host_out) e = hipMemcpyAsync(host_out, ctx->d_out, npx * 4, hipMemcpyDeviceToHost, ctx->stream);
    const uint32_t pitch = ctx->last_pitch;                   // as launch_frame padded the planes
    if (e == hipSuccess && ids)
        e = hipMemcpy2DAsync(ids, (size_t)xres * 4, ctx->d_id, (size_t)pitch * 4, (size_t)xres * 4, yres, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && t)
        e = hipMemcpy2DAsync(t, (size_t)xres * 4, ctx->d_t, (size_t)pitch * 4, (size_t)xres * 4, yres, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipEventRecord(d1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    float dms = 0.f;
    if (e == hipSuccess) (void)hipEventElapsedTime(&dms, d0, d1);
    (void)hipEventDestroy(d0);
    (void)hipEventDestroy(d1);
    if (e != hipSuccess) return fail(ctx, MRT_ERR_HIP, std::string("download: ") + hipGetErrorString(e));
    local.download_ms = dms;
    if (stats) *stats = local;
    return MRT_OK;
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_render_rgba8_rows(mrt_ctx* ctx, uint8_t* host_out, uint32_t strip_rows, mrt_rows_ready_fn ready, void* user,
                          mrt_stats* stats) try {
    if (!ctx) return MRT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, MRT_ERR_NO_SCENE, "mrt_set_scene has not succeeded on this context");
    if (!host_out) return fail(ctx, MRT_ERR_INVALID, "host_out is NULL");
    ENTER_DEVICE(ctx);
    const uint32_t xres = ctx->params.xres, yres = ctx->params.yres;
    const size_t npx = (size_t)xres * yres;
    if (strip_rows == 0) {
        // Strips pay when the kernels run long enough to hide something behind them -- from about
        // 2^40 ray-primitive pairs (the 16384^2 soup has 2^42; a 3840x2160 frame of the teatime mesh,
        // 2^35, renders in a millisecond and loses more to seven small launches than its copy costs):
        // about eight strips then, whole block tiles each; one strip otherwise.
        const unsigned long long pairs = (unsigned long long)npx * ((unsigned long long)ctx->params.n_sph + ctx->params.n_tri);
        strip_rows = pairs < (1ull << 40) ? yres : round_up((yres + 7) / 8, 64);
    }
    const uint32_t n_strips = (yres + strip_rows - 1) / strip_rows;
    int rc = ensure_out(ctx, npx);
    if (rc != MRT_OK) return rc;
    // (a frame in one strip is copied on the ctx's own stream: a second stream costs a hardware
    // queue, milliseconds the first time, and has nothing to overlap)
    if (n_strips > 1 && !ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    hipStream_t copy_stream = n_strips > 1 ? ctx->copy_stream : ctx->stream;
    while (ctx->strip_events.size() < 2 * (size_t)n_strips) {
        hipEvent_t e = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->strip_events.push_back(e);
    }
    // Every strip is a frame of its own on the ctx's stream (band k of n_strips bands of strip_rows
    // rows belongs to "rank" k); its copy to the host runs on the copy stream, beside the strips
    // that follow, and the caller hears of it as soon as it has landed.
    if (n_strips > 1) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));    // (one strip: launch_frame times itself)
    for (uint32_t k = 0; k < n_strips; ++k) {
        const size_t off = (size_t)k * strip_rows * xres;
        const uint32_t rows = mrt_band_rows(yres, strip_rows, k, n_strips);
        rc = launch_frame(ctx, strip_rows, k, n_strips, ctx->d_out + off, ctx->stream, n_strips == 1, FramePlanes{});
        if (rc != MRT_OK) break;
        hipEvent_t rendered = ctx->strip_events[2 * k], copied = ctx->strip_events[2 * k + 1];
        hipError_t e = hipSuccess;
        if (copy_stream != ctx->stream) {
            e = hipEventRecord(rendered, ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(copy_stream, rendered, 0);
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(host_out + off * 4, ctx->d_out + off, (size_t)rows * xres * 4, hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(copied, copy_stream);
        if (e != hipSuccess) { rc = fail(ctx, MRT_ERR_HIP, std::string("strip copy: ") + hipGetErrorString(e)); break; }
    }
    if (rc != MRT_OK) {
        // nothing may still be writing host_out when the caller sees the error
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        return rc;
    }
    if (n_strips > 1) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    }
    for (uint32_t k = 0; k < n_strips; ++k) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->strip_events[2 * k + 1]));
        const uint32_t r0 = k * strip_rows;
        if (ready) ready(user, r0, r0 + mrt_band_rows(yres, strip_rows, k, n_strips));
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2]));
        rc = fill_stats(ctx, stats, yres);
    }
    return rc;
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_render_rgba8(mrt_ctx* ctx, uint8_t* host_out, mrt_stats* stats) try {
    if (!ctx) return MRT_ERR_INVALID;
    if (!host_out) return fail(ctx, MRT_ERR_INVALID, "host_out is NULL");
    return mrt_render_debug(ctx, host_out, nullptr, nullptr, stats);
} MRT_GUARD((void)fail(ctx, code_, what_))

int mrt_host_alloc(size_t bytes, void** out) try {
    if (!out) return MRT_ERR_INVALID;
    *out = nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, MRT_ERR_HIP, "hipHostMalloc failed");
    }
    *out = p;
    return MRT_OK;
} MRT_GUARD((void)fail(nullptr, code_, what_))

void mrt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int mrt_bands_to_image(const uint8_t* gathered, size_t stride_bytes, uint32_t xres, uint32_t yres,
                       uint32_t band_rows, uint32_t world, uint8_t* image_out) {
    if (!gathered || !image_out || band_rows == 0 || world == 0 || xres == 0) return MRT_ERR_INVALID;
    const size_t line = (size_t)xres * 4;
    for (uint32_t r = 0; r < world; ++r)
        if ((size_t)mrt_band_rows(yres, band_rows, r, world) * line > stride_bytes) return MRT_ERR_INVALID;
    for (uint32_t y = 0; y < yres; ++y) {
        const uint32_t band = y / band_rows, within = y % band_rows;
        const uint32_t rank = band % world, local_band = band / world;
        const size_t lr = (size_t)local_band * band_rows + within;
        std::memcpy(image_out + (size_t)y * line, gathered + (size_t)rank * stride_bytes + lr * line, line);
    }
    return MRT_OK;
}

}  // extern "C"
