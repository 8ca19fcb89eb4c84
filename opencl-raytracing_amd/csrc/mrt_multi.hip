// mrt_multi.hip -- one frame over the GPUs of a node, from one process (mrt_multi_* in mrt.h).
//
// The reference renders on one device it picks by index (CLHelpers.hpp:28-45)
// and assembles the image on the host (gfxraytrace.hpp:872-875).  Here the
// image is cut into interleaved row bands (mrt_band_rows), every GPU holds the
// whole scene and renders its own bands on its own stream, all GPUs at once,
// and the bands are brought together in one of three ways:
//
//   RCCL   ncclGather to rank 0 inside one ncclGroupStart/End (one communicator
//          per GPU from ncclCommInitAll), then rank 0 copies band by band into
//          the caller's image;
//   PEER   hipMemcpyPeerAsync of every rank's buffer into rank 0's gather
//          buffer, then as above;
//   HOST   every GPU copies its own bands straight into the caller's image over
//          its own PCIe link -- no gather at all.
//
// A band is a run of whole image rows, contiguous both in the rank's buffer and
// in the image, so the copies to the host de-interleave as they go: there is no
// pass over the image on the host.  RCCL is loaded with dlopen the first time
// it is asked for, so that the library has no load-time dependency on it.
#include "mrt.h"
#include "mrt_guard.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGather) gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;             // why it is not available
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.why = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.handle, n); if (!p && api.why.empty()) api.why = std::string("missing symbol ") + n; return p; };
        api.comm_init_all = (decltype(api.comm_init_all))sym("ncclCommInitAll");
        api.comm_destroy = (decltype(api.comm_destroy))sym("ncclCommDestroy");
        api.group_start = (decltype(api.group_start))sym("ncclGroupStart");
        api.group_end = (decltype(api.group_end))sym("ncclGroupEnd");
        api.gather = (decltype(api.gather))sym("ncclGather");
        api.error_string = (decltype(api.error_string))sym("ncclGetErrorString");
        api.ok = api.why.empty();
    });
    return api;
}

thread_local std::string g_multi_create_err;

struct Rank {
    int device = 0;
    mrt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t begin = nullptr, rendered = nullptr, delivered = nullptr;
    uint8_t* local = nullptr;        // this rank's bands, on its GPU (rank 0 with a gather: slot 0 of `gathered`)
    bool owns_local = false;
    ncclComm_t comm = nullptr;
};

// The event behind the copy of one band to the host, and the GPU it lives on.
struct BandEvent {
    hipEvent_t event = nullptr;
    int device = 0;
};

}  // namespace

struct mrt_multi {
    std::vector<Rank> ranks;
    uint32_t band_rows = 64;
    uint32_t band_rows_asked = 0;            // what the config said: 0 = balanced for every frame height
    std::vector<BandEvent> band_events;      // mrt_multi_render_rgba8_rows: behind the copy of band b
    uint32_t gather = MRT_GATHER_PEER;
    uint32_t flags = 0;
    std::string err, note;
    bool have_scene = false;
    uint32_t xres = 0, yres = 0;
    size_t stride = 0;               // bytes per rank slot: the most rows any rank owns
    size_t cap_stride = 0;           // what the buffers were allocated for
    uint8_t* gathered = nullptr;     // on rank 0's GPU: world slots of `stride` bytes (RCCL, PEER)
    uint32_t n_prims = 0;
};

namespace {

int mfail(mrt_multi* m, int code, const std::string& msg) {
    if (m) m->err = msg; else g_multi_create_err = msg;
    return code;
}

#define MHIP(m, expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return mfail((m), MRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct CurrentDevice {
    int prev = -1;
    CurrentDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~CurrentDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void release_buffers(mrt_multi* m) {
    for (Rank& r : m->ranks) {
        if (r.local && r.owns_local) { (void)hipSetDevice(r.device); (void)hipFree(r.local); }
        r.local = nullptr;
        r.owns_local = false;
    }
    if (m->gathered) { (void)hipSetDevice(m->ranks[0].device); (void)hipFree(m->gathered); m->gathered = nullptr; }
    m->cap_stride = 0;
}

// Buffers for the current frame size: one slot per rank on its own GPU, and, for the gathering
// transports, world slots on rank 0's GPU (whose first slot is rank 0's own buffer).
int ensure_buffers(mrt_multi* m) {
    const uint32_t world = (uint32_t)m->ranks.size();
    if (!m->band_rows_asked) m->band_rows = mrt_balanced_band_rows(m->yres, world);
    uint32_t most = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const uint32_t rows = mrt_band_rows(m->yres, m->band_rows, r, world);
        if (rows > most) most = rows;
    }
    m->stride = (size_t)most * m->xres * 4;
    if (m->stride <= m->cap_stride) return MRT_OK;
    for (Rank& r : m->ranks) {             // nothing of the old size may still be in flight
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipStreamSynchronize(r.stream));
    }
    release_buffers(m);
    const bool gathers = m->gather != MRT_GATHER_HOST;
    if (gathers) {
        MHIP(m, hipSetDevice(m->ranks[0].device));
        MHIP(m, hipMalloc((void**)&m->gathered, m->stride * world));
    }
    for (uint32_t r = 0; r < world; ++r) {
        Rank& k = m->ranks[r];
        if (gathers && r == 0) { k.local = m->gathered; k.owns_local = false; continue; }
        MHIP(m, hipSetDevice(k.device));
        MHIP(m, hipMalloc((void**)&k.local, m->stride));
        k.owns_local = true;
    }
    m->cap_stride = m->stride;
    return MRT_OK;
}

}  // namespace

extern "C" {

const char* mrt_multi_last_error(const mrt_multi* m) { return m ? m->err.c_str() : g_multi_create_err.c_str(); }

void mrt_multi_destroy(mrt_multi* m) {
    if (!m) return;
    CurrentDevice keep;
    for (Rank& r : m->ranks) {
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
    }
    for (Rank& r : m->ranks)
        if (r.comm) { (void)hipSetDevice(r.device); (void)rccl().comm_destroy(r.comm); r.comm = nullptr; }
    for (BandEvent& be : m->band_events)
        if (be.event) { (void)hipSetDevice(be.device); (void)hipEventDestroy(be.event); }
    if (!m->ranks.empty()) release_buffers(m);
    for (Rank& r : m->ranks) {
        (void)hipSetDevice(r.device);
        if (r.ctx) mrt_destroy(r.ctx);
        if (r.begin) (void)hipEventDestroy(r.begin);
        if (r.rendered) (void)hipEventDestroy(r.rendered);
        if (r.delivered) (void)hipEventDestroy(r.delivered);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    delete m;
}

int mrt_multi_create(const mrt_multi_config* cfg, mrt_multi** out) try {
    if (!out) return mfail(nullptr, MRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return mfail(nullptr, MRT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    uint32_t world = cfg ? cfg->n_gpus : 0;
    if (world == 0) world = (uint32_t)count;
    if (world > 64) return mfail(nullptr, MRT_ERR_TOO_LARGE, "more than 64 ranks");
    if (cfg && cfg->gather > MRT_GATHER_HOST) return mfail(nullptr, MRT_ERR_INVALID, "unknown gather transport");
    std::vector<int> devs(world);
    bool repeats = false;
    for (uint32_t r = 0; r < world; ++r) {
        devs[r] = (cfg && cfg->devices) ? cfg->devices[r] : (int)r;
        if (devs[r] < 0 || devs[r] >= count) return mfail(nullptr, MRT_ERR_NO_DEVICE, "device ordinal out of range");
        for (uint32_t q = 0; q < r; ++q) repeats |= devs[q] == devs[r];
    }
    CurrentDevice keep;
    mrt_multi* m = new mrt_multi();
    m->band_rows_asked = cfg ? cfg->band_rows : 0u;
    m->band_rows = m->band_rows_asked ? m->band_rows_asked : 64u;
    m->flags = cfg ? cfg->flags : 0u;
    m->ranks.resize(world);
    auto bail = [&](int code, const std::string& msg) {
        mrt_multi_destroy(m);
        return mfail(nullptr, code, msg);
    };
    for (uint32_t r = 0; r < world; ++r) {
        Rank& k = m->ranks[r];
        k.device = devs[r];
        mrt_config c = {devs[r], m->flags | MRT_FLAG_DEFER_NOALLOC_SETUP};    // no entry point of mrt_multi uses the fixed array
        const int rc = mrt_create(&c, &k.ctx);
        if (rc != MRT_OK) return bail(rc, std::string("rank ") + std::to_string(r) + ": " + mrt_last_error(nullptr));
        if ((e = hipSetDevice(k.device)) != hipSuccess) return bail(MRT_ERR_HIP, hipGetErrorString(e));
        if ((e = hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking)) != hipSuccess) return bail(MRT_ERR_HIP, hipGetErrorString(e));
        if ((e = hipEventCreate(&k.begin)) != hipSuccess || (e = hipEventCreate(&k.rendered)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&k.delivered, hipEventDisableTiming)) != hipSuccess)
            return bail(MRT_ERR_HIP, hipGetErrorString(e));
    }
    // Direct copies between the GPUs where the hardware allows (errors only mean "already on" or
    // "not possible": hipMemcpyPeerAsync then goes through the host).
    for (uint32_t r = 1; r < world; ++r) {
        if (devs[r] == devs[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devs[r], devs[0]) == hipSuccess && can) {
            (void)hipSetDevice(devs[r]);
            (void)hipDeviceEnablePeerAccess(devs[0], 0);
            (void)hipSetDevice(devs[0]);
            (void)hipDeviceEnablePeerAccess(devs[r], 0);
        }
        (void)hipGetLastError();
    }
    // The transport.
    uint32_t want = cfg ? cfg->gather : (uint32_t)MRT_GATHER_AUTO;
    if (want == MRT_GATHER_RCCL || (want == MRT_GATHER_AUTO && world > 1 && !repeats)) {
        std::string why;
        RcclApi& api = rccl();
        if (repeats) why = "a GPU appears more than once in the device list";
        else if (!api.ok) why = api.why;
        else {
            std::vector<ncclComm_t> comms(world, nullptr);
            const ncclResult_t rc = api.comm_init_all(comms.data(), (int)world, devs.data());
            if (rc != ncclSuccess) why = std::string("ncclCommInitAll: ") + api.error_string(rc);
            else for (uint32_t r = 0; r < world; ++r) m->ranks[r].comm = comms[r];
        }
        if (why.empty()) m->gather = MRT_GATHER_RCCL;
        else if (want == MRT_GATHER_RCCL) return bail(MRT_ERR_HIP, "RCCL gather not available: " + why);
        else { m->gather = MRT_GATHER_PEER; m->note = "RCCL not used (" + why + "); peer copies instead"; }
    } else {
        m->gather = (want == MRT_GATHER_AUTO) ? (uint32_t)MRT_GATHER_PEER : want;
    }
    *out = m;
    return MRT_OK;
} MRT_GUARD((void)mfail(nullptr, code_, what_))

uint32_t mrt_multi_world(const mrt_multi* m) { return m ? (uint32_t)m->ranks.size() : 0u; }

int mrt_multi_set_scene(mrt_multi* m, const mrt_scene* scene) try {
    if (!m) return MRT_ERR_INVALID;
    if (!scene) return mfail(m, MRT_ERR_INVALID, "scene is NULL");
    m->have_scene = false;
    for (size_t r = 0; r < m->ranks.size(); ++r) {
        const int rc = mrt_set_scene(m->ranks[r].ctx, scene);
        if (rc != MRT_OK) return mfail(m, rc, "rank " + std::to_string(r) + ": " + mrt_last_error(m->ranks[r].ctx));
    }
    m->xres = scene->xres; m->yres = scene->yres;
    m->n_prims = scene->n_spheres + scene->n_tris;
    CurrentDevice keep;
    const int rc = ensure_buffers(m);
    if (rc != MRT_OK) return rc;
    m->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)mfail(m, code_, what_))

int mrt_multi_set_camera(mrt_multi* m, const mrt_camera* camera) try {
    if (!m) return MRT_ERR_INVALID;
    if (!m->have_scene) return mfail(m, MRT_ERR_NO_SCENE, "mrt_multi_set_scene has not succeeded");
    if (!camera) return mfail(m, MRT_ERR_INVALID, "camera is NULL");
    m->have_scene = false;
    for (size_t r = 0; r < m->ranks.size(); ++r) {
        const int rc = mrt_set_camera(m->ranks[r].ctx, camera);
        if (rc != MRT_OK) {
            // a camera rank 0 refused outright (bad argument) has changed nothing anywhere
            m->have_scene = r == 0 && rc != MRT_ERR_HIP;
            return mfail(m, rc, "rank " + std::to_string(r) + ": " + mrt_last_error(m->ranks[r].ctx));
        }
    }
    m->xres = camera->xres; m->yres = camera->yres;
    CurrentDevice keep;
    const int rc = ensure_buffers(m);
    if (rc != MRT_OK) return rc;
    m->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)mfail(m, code_, what_))

int mrt_multi_render_rgba8(mrt_multi* m, uint8_t* host_out, mrt_multi_stats* stats) try {
    return mrt_multi_render_rgba8_rows(m, host_out, nullptr, nullptr, stats);
} MRT_GUARD((void)mfail(m, code_, what_))

int mrt_multi_render_rgba8_rows(mrt_multi* m, uint8_t* host_out, mrt_rows_ready_fn ready, void* user,
                                mrt_multi_stats* stats) try {
    if (!m) return MRT_ERR_INVALID;
    if (!m->have_scene) return mfail(m, MRT_ERR_NO_SCENE, "mrt_multi_set_scene has not succeeded");
    if (!host_out) return mfail(m, MRT_ERR_INVALID, "host_out is NULL");
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    CurrentDevice keep;
    const uint32_t world = (uint32_t)m->ranks.size();
    const size_t line = (size_t)m->xres * 4;
    const size_t band_bytes = (size_t)m->band_rows * line;
    const uint32_t n_bands = (m->yres + m->band_rows - 1) / m->band_rows;

    // Steps 1-3 only enqueue; whatever they return, every rank's stream is drained before this call
    // does (a failure half-way must not leave copies into host_out running behind the caller's back).
    auto enqueue = [&]() -> int {
    // 1. every GPU renders its bands, all at once
    for (uint32_t r = 0; r < world; ++r) {
        Rank& k = m->ranks[r];
        MHIP(m, hipSetDevice(k.device));
        MHIP(m, hipEventRecord(k.begin, k.stream));
        const int rc = mrt_render_bands_device(k.ctx, m->band_rows, r, world, k.local, m->stride, k.stream, nullptr);
        if (rc != MRT_OK) return mfail(m, rc, "rank " + std::to_string(r) + ": " + mrt_last_error(k.ctx));
        MHIP(m, hipEventRecord(k.rendered, k.stream));
    }
    // 2. bring the bands to rank 0 (RCCL, PEER)
    Rank& root = m->ranks[0];
    if (m->gather == MRT_GATHER_RCCL) {
        RcclApi& api = rccl();
        ncclResult_t rc = api.group_start();
        hipError_t he = hipSuccess;
        for (uint32_t r = 0; r < world && rc == ncclSuccess && he == hipSuccess; ++r) {
            Rank& k = m->ranks[r];
            he = hipSetDevice(k.device);
            // rank 0 sends from slot 0 of the buffer it receives into: in place
            if (he == hipSuccess)
                rc = api.gather(k.local, r == 0 ? m->gathered : nullptr, m->stride, ncclUint8, 0, k.comm, k.stream);
        }
        const ncclResult_t rc2 = api.group_end();          // the group is closed whatever happened inside
        if (he != hipSuccess) return mfail(m, MRT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
        if (rc != ncclSuccess || rc2 != ncclSuccess)
            return mfail(m, MRT_ERR_HIP, std::string("ncclGather: ") + api.error_string(rc != ncclSuccess ? rc : rc2));
    } else if (m->gather == MRT_GATHER_PEER) {
        for (uint32_t r = 1; r < world; ++r) {
            Rank& k = m->ranks[r];
            const size_t bytes = (size_t)mrt_band_rows(m->yres, m->band_rows, r, world) * line;
            MHIP(m, hipSetDevice(k.device));
            if (bytes)
                MHIP(m, hipMemcpyPeerAsync(m->gathered + (size_t)r * m->stride, root.device, k.local, k.device, bytes, k.stream));
            MHIP(m, hipEventRecord(k.delivered, k.stream));
            MHIP(m, hipSetDevice(root.device));
            MHIP(m, hipStreamWaitEvent(root.stream, k.delivered, 0));
        }
    }
    // 3. to the caller's image, band by band (a band is contiguous on both sides)
    for (uint32_t b = 0; b < n_bands; ++b) {
        const uint32_t r = b % world;
        const size_t local_off = (size_t)(b / world) * band_bytes;
        const size_t row0 = (size_t)b * m->band_rows;
        const size_t bytes = ((row0 + m->band_rows <= m->yres) ? m->band_rows : (m->yres - row0)) * line;
        Rank& from = m->gather == MRT_GATHER_HOST ? m->ranks[r] : root;
        MHIP(m, hipSetDevice(from.device));
        if (m->gather == MRT_GATHER_HOST)
            MHIP(m, hipMemcpyAsync(host_out + row0 * line, from.local + local_off, bytes, hipMemcpyDeviceToHost, from.stream));
        else
            MHIP(m, hipMemcpyAsync(host_out + row0 * line, m->gathered + (size_t)r * m->stride + local_off, bytes,
                                   hipMemcpyDeviceToHost, root.stream));
        if (ready) {
            // one event per band, on the GPU that copies it (made on first use, kept with the handle)
            if (m->band_events.size() <= b) m->band_events.resize((size_t)b + 1);
            BandEvent& be = m->band_events[b];
            if (be.event && be.device != from.device) { (void)hipEventDestroy(be.event); be.event = nullptr; }
            if (!be.event) {
                MHIP(m, hipEventCreateWithFlags(&be.event, hipEventDisableTiming));
                be.device = from.device;
            }
            MHIP(m, hipEventRecord(be.event, from.stream));
        }
    }
    if (ready) {
        for (uint32_t b = 0; b < n_bands; ++b) {
            MHIP(m, hipSetDevice(m->band_events[b].device));
            MHIP(m, hipEventSynchronize(m->band_events[b].event));
            const uint32_t row0 = b * m->band_rows;
            ready(user, row0, row0 + m->band_rows <= m->yres ? row0 + m->band_rows : m->yres);
        }
    }
    return MRT_OK;
    };
    const int enq = enqueue();
    if (enq != MRT_OK) {
        const std::string why = m->err;                     // the first failure is the one reported
        for (uint32_t r = 0; r < world; ++r)
            if (hipSetDevice(m->ranks[r].device) == hipSuccess) (void)hipStreamSynchronize(m->ranks[r].stream);
        (void)hipGetLastError();
        return mfail(m, enq, why);
    }
    for (uint32_t r = 0; r < world; ++r) {
        MHIP(m, hipSetDevice(m->ranks[r].device));
        MHIP(m, hipStreamSynchronize(m->ranks[r].stream));
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (uint32_t r = 0; r < world; ++r) {
            float ms = 0.f;
            MHIP(m, hipSetDevice(m->ranks[r].device));
            MHIP(m, hipEventElapsedTime(&ms, m->ranks[r].begin, m->ranks[r].rendered));
            if (ms > stats->render_ms_max) stats->render_ms_max = ms;
            stats->render_ms_sum += ms;
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        stats->rays = (uint64_t)m->xres * m->yres;
        stats->prim_tests = stats->rays * m->n_prims;
        stats->n_gpus = world;
        stats->gather = m->gather;
    }
    return MRT_OK;
} MRT_GUARD((void)mfail(m, code_, what_))

uint32_t mrt_multi_gather(const mrt_multi* m) { return m ? m->gather : 0u; }
const char* mrt_multi_note(const mrt_multi* m) { return m ? m->note.c_str() : ""; }

}  // extern "C"
