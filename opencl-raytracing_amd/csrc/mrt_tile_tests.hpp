// mrt_tile_tests.hpp -- can a primitive touch any pixel of a 32x32 pixel tile?
//
// Conservative tests of a whole tile against the per-view prefilter records
// (mrt_records.hip): true for every tile that holds a pixel the per-pixel tests
// of the render kernels can pass, so nothing the reference arithmetic would
// accept is lost by skipping a tile that fails.  Shared by the tile-cull kernel
// (which runs them for every tile and primitive of every frame) and by the
// builder of the screen-space grid (which runs them once per view and keeps the
// answers as per-tile primitive lists).  The reference has no counterpart: its
// loop visits every object for every pixel (kernel.cl:320-360).
#pragma once

#include <hip/hip_runtime.h>

#include "mrt_kernels.hpp"

namespace mrt {

// Sphere record (sx, ox, sy, oy): pixel (x, y) passes when (x*sx + ox)^2 + (y*sy + oy)^2 <= 1.
// Each factor is monotone along its axis, so its value nearest zero over the tile's columns
// (rows) is the median of its two end values and zero.  (txf, tyf): the tile's first pixel.
__device__ __forceinline__ bool tile_may_hit_sphere(float4 e, float txf, float tyf) {
    const float span = (float)(kWaveTile - 1);
    const float du_lo = __builtin_fmaf(txf, e.x, e.y), dv_lo = __builtin_fmaf(tyf, e.z, e.w);
    const float du = __builtin_amdgcn_fmed3f(du_lo, __builtin_fmaf(span, e.x, du_lo), 0.0f);
    const float dv = __builtin_amdgcn_fmed3f(dv_lo, __builtin_fmaf(span, e.z, dv_lo), 0.0f);
    // 1e-6: the per-pixel evaluation rounds differently from this one.
    return __builtin_fmaf(du, du, dv * dv) <= 1.000001f;
}

// Triangle: the bounding circle (xc, yc, R^2) against the tile's nearest point, then each of the
// three edge functions A x + B y + C >= 0 of the record rec[0..2] at the tile corner where it is
// largest.
__device__ __forceinline__ bool tile_may_hit_triangle(float4 c, const float4* __restrict__ rec, float txf, float tyf) {
    const float span = (float)(kWaveTile - 1);
    const float cu = __builtin_amdgcn_fmed3f(txf - c.x, txf + span - c.x, 0.0f);
    const float cv = __builtin_amdgcn_fmed3f(tyf - c.y, tyf + span - c.y, 0.0f);
    bool pass = __builtin_fmaf(cu, cu, __builtin_fmaf(cv, cv, -c.z)) <= 0.0f;
    if (pass) {
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        const float a[3] = {q0.x, q0.w, q1.z}, b[3] = {q0.y, q1.x, q1.w}, cc[3] = {q0.z, q1.y, q2.x};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float best_x = a[k] > 0.0f ? txf + span : txf;
            const float best_y = b[k] > 0.0f ? tyf + span : tyf;
            // slack 1e-3 px for the different rounding of the per-pixel evaluation
            pass = pass && (__builtin_fmaf(a[k], best_x, __builtin_fmaf(b[k], best_y, cc[k])) >= -1e-3f);
        }
    }
    return pass;
}

}  // namespace mrt
