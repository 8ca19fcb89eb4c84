// mrt_internal.hpp -- what the translation units of libmrt_hip share beyond the public ABI.
#pragma once

#include "mrt.h"
#include "mrt_kernels.hpp"

// Context flags that are not part of the public ABI: the screen-space culling
// kernels of a brute-force context, reachable only through mrt_accel_* (which
// creates its inner context with mrt_create_internal_), and the mark of a
// context that is one slot of an mrt_sequence.  mrt_create rejects them.
enum : uint32_t {
    MRT_IFLAG_BLOCK_REJECT = 1u << 2,   // trace_kernel preceded by the 4x4 block reject
    MRT_IFLAG_SCREEN_CULL = 1u << 4,    // cull_trace_kernel: 32x32 tile cull
    MRT_IFLAG_IN_FLIGHT = 1u << 30,     // frames of other contexts run beside this one's: fewer, longer slices
    MRT_IFLAG_MASK = MRT_IFLAG_BLOCK_REJECT | MRT_IFLAG_SCREEN_CULL | MRT_IFLAG_IN_FLIGHT
};

extern "C" {
// mrt_create that also takes the MRT_IFLAG_* bits.
int mrt_create_internal_(const mrt_config* cfg, mrt_ctx** out);
// The kernel parameters of the ctx's current scene and view (device pointers included), its own
// stream (the per-view records are built there), its GPU and its projection.
const mrt::RenderParams* mrt_ctx_params_(const mrt_ctx* ctx);
void* mrt_ctx_stream_(const mrt_ctx* ctx);
int mrt_ctx_projection_(const mrt_ctx* ctx);
int mrt_stage_device_(const mrt_ctx* ctx);
// Whether the hipStream_t is recording into a HIP graph (such streams are refused by every entry point).
int mrt_stream_is_capturing_(void* hip_stream);
}
