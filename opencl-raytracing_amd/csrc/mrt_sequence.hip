// mrt_sequence.hip -- many frames of one scene on one GPU, several in flight (mrt_sequence_* in mrt.h).
//
// The reference renders one frame per process and sets up its OpenCL context,
// program and buffers anew in every stage call (CLHelpers.hpp:79,119,155,196,
// 242,281).  A sequence is the opposite end: in_flight slots, each an mrt_ctx
// (scene copy, per-view records, working planes) with a stream of its own, and
// frames handed to the slots in rotation.  Nothing here touches the GPU
// directly except the slots' streams; the frames are mrt_set_camera +
// mrt_render_bands_device on the slot's context.
#include "mrt.h"
#include "mrt_guard.hpp"
#include "mrt_internal.hpp"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace {

thread_local std::string g_sequence_create_err;

struct Slot {
    mrt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    uint64_t view_serial = 0;        // which of the sequence's views the slot's records are built for
};

// A device frame buffer of mrt_sequence_submit_host and the events around its use.
struct HostLane {
    uint8_t* dev = nullptr;
    size_t cap = 0;
    hipEvent_t rendered = nullptr;   // the frame is in `dev`
    hipEvent_t copied = nullptr;     // `dev` has been copied out and may be overwritten
    bool used = false;
};

struct KeepDevice {
    int prev = -1;
    KeepDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~KeepDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct mrt_sequence {
    int device = 0;
    std::vector<Slot> slots;
    std::string err;
    bool have_scene = false;
    mrt_camera view{};               // the view of the last frame submitted (of the scene at first)
    uint64_t view_serial = 0;
    uint64_t next = 0;               // frames submitted so far
    // mrt_sequence_submit_host: 2 x in_flight device buffers, taken in rotation, and the stream
    // the copies to the host run on
    std::vector<HostLane> lanes;
    hipStream_t copy_stream = nullptr;
    uint64_t host_next = 0;
};

namespace {

int qfail(mrt_sequence* q, int code, const std::string& msg) {
    if (q) q->err = msg; else g_sequence_create_err = msg;
    return code;
}

}  // namespace

extern "C" {

const char* mrt_sequence_last_error(const mrt_sequence* q) { return q ? q->err.c_str() : g_sequence_create_err.c_str(); }

uint32_t mrt_sequence_in_flight(const mrt_sequence* q) { return q ? (uint32_t)q->slots.size() : 0u; }

void mrt_sequence_destroy(mrt_sequence* q) {
    if (!q) return;
    KeepDevice keep;
    (void)hipSetDevice(q->device);
    for (Slot& s : q->slots)
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (q->copy_stream) (void)hipStreamSynchronize(q->copy_stream);
    for (HostLane& l : q->lanes) {
        if (l.dev) (void)hipFree(l.dev);
        if (l.rendered) (void)hipEventDestroy(l.rendered);
        if (l.copied) (void)hipEventDestroy(l.copied);
    }
    if (q->copy_stream) (void)hipStreamDestroy(q->copy_stream);
    for (Slot& s : q->slots) {
        if (s.ctx) mrt_destroy(s.ctx);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete q;
}

int mrt_sequence_create(const mrt_sequence_config* cfg, mrt_sequence** out) try {
    if (!out) return qfail(nullptr, MRT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    uint32_t n = cfg ? cfg->in_flight : 0u;
    if (n == 0) n = 2;
    if (n > 4) return qfail(nullptr, MRT_ERR_TOO_LARGE, "at most 4 frames in flight");
    KeepDevice keep;
    mrt_sequence* q = new mrt_sequence();
    q->slots.resize(n);
    if (cfg && (cfg->flags & MRT_IFLAG_MASK)) {
        mrt_sequence_destroy(q);
        return qfail(nullptr, MRT_ERR_INVALID, "reserved flag bits set in mrt_sequence_config.flags");
    }
    if (cfg && (cfg->flags & MRT_FLAG_TRACE_NARROW) && (cfg->flags & MRT_FLAG_TRACE_WIDE)) {
        mrt_sequence_destroy(q);
        return qfail(nullptr, MRT_ERR_INVALID, "MRT_FLAG_TRACE_NARROW and MRT_FLAG_TRACE_WIDE exclude each other");
    }
    for (uint32_t k = 0; k < n; ++k) {
        mrt_config c = {cfg ? cfg->device : -1, cfg ? cfg->flags : 0u};
        if (k > 0) c.device = q->device;                 // "-1" must mean the same GPU for every slot
        if (n > 1) c.flags |= MRT_IFLAG_IN_FLIGHT;       // slices sized for a GPU that other frames keep full
        c.flags |= MRT_FLAG_DEFER_NOALLOC_SETUP;         // no entry point of mrt_sequence uses the fixed array
        const int rc = mrt_create_internal_(&c, &q->slots[k].ctx);
        if (rc != MRT_OK) {
            const std::string msg = mrt_last_error(nullptr);
            mrt_sequence_destroy(q);
            return qfail(nullptr, rc, msg);
        }
        if (k == 0) {
            // the ordinal slot 0 settled on
            int dev = c.device;
            if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
            q->device = dev;
        }
    }
    // The slots' streams are made one right after the other: the runtime deals its few hardware
    // queues out to streams in the order they are made, and two slots that shared a queue would
    // run their frames one behind the other.
    hipError_t e = hipSetDevice(q->device);
    for (uint32_t k = 0; k < n && e == hipSuccess; ++k)
        e = hipStreamCreateWithFlags(&q->slots[k].stream, hipStreamNonBlocking);
    // ... and the stream mrt_sequence_submit_host copies on right behind them, for the same reason
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&q->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        const std::string msg = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        mrt_sequence_destroy(q);
        return qfail(nullptr, MRT_ERR_HIP, msg);
    }
    *out = q;
    return MRT_OK;
} MRT_GUARD((void)qfail(nullptr, code_, what_))

int mrt_sequence_wait(mrt_sequence* q) try {
    if (!q) return MRT_ERR_INVALID;
    KeepDevice keep;
    hipError_t e = hipSetDevice(q->device);
    for (Slot& s : q->slots)
        if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
    if (e == hipSuccess && q->copy_stream) e = hipStreamSynchronize(q->copy_stream);
    if (e != hipSuccess) return qfail(q, MRT_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return MRT_OK;
} MRT_GUARD((void)qfail(q, code_, what_))

int mrt_sequence_set_scene(mrt_sequence* q, const mrt_scene* scene) try {
    if (!q) return MRT_ERR_INVALID;
    if (!scene) return qfail(q, MRT_ERR_INVALID, "scene is NULL");
    q->have_scene = false;
    int rc = mrt_sequence_wait(q);
    if (rc != MRT_OK) return rc;
    for (size_t k = 0; k < q->slots.size(); ++k) {
        rc = mrt_set_scene(q->slots[k].ctx, scene);
        if (rc != MRT_OK) return qfail(q, rc, "slot " + std::to_string(k) + ": " + mrt_last_error(q->slots[k].ctx));
        q->slots[k].view_serial = 0;
    }
    mrt_camera& v = q->view;
    for (int k = 0; k < 3; ++k) { v.eye[k] = scene->eye[k]; v.u[k] = scene->u[k]; v.v[k] = scene->v[k]; v.w[k] = scene->w[k]; }
    v.xres = scene->xres; v.yres = scene->yres;
    v.left = scene->left; v.right = scene->right; v.top = scene->top; v.bottom = scene->bottom;
    v.projection = scene->projection;
    v.focal = scene->focal;
    q->view_serial = 0;
    q->have_scene = true;
    return MRT_OK;
} MRT_GUARD((void)qfail(q, code_, what_))

// The slot of the next frame, with `view` (NULL = as before) taken: a new view counts only once
// a slot has accepted it.
static int next_slot(mrt_sequence* q, const mrt_camera* view, Slot** out) {
    Slot& s = q->slots[q->next % q->slots.size()];
    const mrt_camera& want = view ? *view : q->view;
    const uint64_t want_serial = q->view_serial + (view ? 1u : 0u);
    if (s.view_serial != want_serial) {
        // asynchronous: the slot's records are rebuilt behind its last frame, its next frame waits for them
        const int rc = mrt_set_camera(s.ctx, &want);
        if (rc != MRT_OK) return qfail(q, rc, mrt_last_error(s.ctx));
        s.view_serial = want_serial;
    }
    if (view) {
        q->view = *view;
        q->view_serial = want_serial;
    }
    *out = &s;
    return MRT_OK;
}

int mrt_sequence_submit(mrt_sequence* q, const mrt_camera* view, uint32_t band_rows, uint32_t rank,
                        uint32_t world, void* dev_out, size_t dev_out_bytes, void** frame_stream) try {
    if (!q) return MRT_ERR_INVALID;
    if (frame_stream) *frame_stream = nullptr;
    if (!q->have_scene) return qfail(q, MRT_ERR_NO_SCENE, "mrt_sequence_set_scene has not succeeded");
    Slot* s = nullptr;
    int rc = next_slot(q, view, &s);
    if (rc != MRT_OK) return rc;
    rc = mrt_render_bands_device(s->ctx, band_rows, rank, world, dev_out, dev_out_bytes, s->stream, nullptr);
    if (rc != MRT_OK) return qfail(q, rc, mrt_last_error(s->ctx));
    if (frame_stream) *frame_stream = (void*)s->stream;
    ++q->next;
    return MRT_OK;
} MRT_GUARD((void)qfail(q, code_, what_))

int mrt_sequence_submit_host(mrt_sequence* q, const mrt_camera* view, uint8_t* host_out) try {
    if (!q) return MRT_ERR_INVALID;
    if (!q->have_scene) return qfail(q, MRT_ERR_NO_SCENE, "mrt_sequence_set_scene has not succeeded");
    if (!host_out) return qfail(q, MRT_ERR_INVALID, "host_out is NULL");
    KeepDevice keep;
#define QHIP(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return qfail(q, MRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    QHIP(hipSetDevice(q->device));
    if (q->lanes.empty()) {
        // built aside and adopted only when every event exists: a failure leaves no half-made lanes
        std::vector<HostLane> lanes(2 * q->slots.size());
        hipError_t made = hipSuccess;
        for (HostLane& l : lanes) {
            if (made == hipSuccess) made = hipEventCreateWithFlags(&l.rendered, hipEventDisableTiming);
            if (made == hipSuccess) made = hipEventCreateWithFlags(&l.copied, hipEventDisableTiming);
        }
        if (made != hipSuccess) {
            for (HostLane& l : lanes) {
                if (l.rendered) (void)hipEventDestroy(l.rendered);
                if (l.copied) (void)hipEventDestroy(l.copied);
            }
            return qfail(q, MRT_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(made));
        }
        q->lanes.swap(lanes);
    }
    Slot* s = nullptr;
    int rc = next_slot(q, view, &s);
    if (rc != MRT_OK) return rc;
    const size_t bytes = (size_t)q->view.xres * q->view.yres * 4;
    HostLane& l = q->lanes[q->host_next % q->lanes.size()];
    if (bytes > l.cap) {
        if (l.used) QHIP(hipEventSynchronize(l.copied));       // the copy out of the buffer freed here
        if (l.dev) { (void)hipFree(l.dev); l.dev = nullptr; }
        l.cap = 0;
        l.used = false;
        QHIP(hipMalloc((void**)&l.dev, bytes));
        l.cap = bytes;
    }
    // the frame may not overwrite a buffer whose last frame is still on its way to the host
    if (l.used) QHIP(hipStreamWaitEvent(s->stream, l.copied, 0));
    rc = mrt_render_bands_device(s->ctx, q->view.yres, 0, 1, l.dev, bytes, s->stream, nullptr);
    if (rc != MRT_OK) return qfail(q, rc, mrt_last_error(s->ctx));
    QHIP(hipEventRecord(l.rendered, s->stream));
    QHIP(hipStreamWaitEvent(q->copy_stream, l.rendered, 0));
    QHIP(hipMemcpyAsync(host_out, l.dev, bytes, hipMemcpyDeviceToHost, q->copy_stream));
    QHIP(hipEventRecord(l.copied, q->copy_stream));
#undef QHIP
    l.used = true;
    ++q->host_next;
    ++q->next;
    return MRT_OK;
} MRT_GUARD((void)qfail(q, code_, what_))

}  // extern "C"
