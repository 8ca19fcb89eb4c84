// mrt_kernels.hip -- CDNA4 (gfx950) device code of the fused render path.
//
// Two launches per frame replace the reference's four stage kernels and the
// host round trips between them (gfxraytrace.hpp:830-878):
//
//   trace_kernel  primary-ray generation + brute-force closest hit over every
//                 sphere and triangle (kernels uv / *_viewrays / intersections,
//                 kernel.cl:123-379).  Writes (t, object id) per pixel.
//   shade_kernel  rebuilds the winning hit record and shades it (flat_shader /
//                 phong_shader, kernel.cl:381-445), converts to bytes
//                 (gfximage.hpp:46-51) and writes coalesced RGBA8.
//
// trace_kernel is the hot loop: pixels x primitives tests, no acceleration
// structure.  Per (pixel, triangle) it evaluates three affine edge functions
// of the pixel coordinate (primary rays of a pinhole or orthographic camera
// are affine in the pixel, so the numerators of Cramer's rule are too) and
// rejects the pair when any is below a per-edge error margin.  The few pairs
// that survive run the reference arithmetic (mrt_device.hpp), in object
// order, so the accepted hits, their t and the winner are those of the
// reference loop.  Spheres always run the reference discriminant.
#include "mrt.h"
#include "mrt_device.hpp"
#include "mrt_kernels.hpp"

#include <float.h>

#pragma clang fp contract(off)

namespace mrt {

namespace {

// Image row of packed local row lr for this rank (interleaved bands).
__device__ __forceinline__ uint32_t image_row(const RenderParams& p, uint32_t lr) {
    uint32_t band = lr / p.band_rows;
    uint32_t within = lr - band * p.band_rows;
    return (band * p.world + p.rank) * p.band_rows + within;
}

template <int PROJ>
struct RaySetup {
    // Column and row partial sums of the ray, associated exactly as the
    // reference adds them: persp d = ((-w*f) + u*uvx) + v*uvy,
    // ortho o = (u*uvx) + (v*uvy).
    F3 col[kLanePx];
    F3 row[kLanePx];
    F3 fixed;   // persp: origin (eye); ortho: direction (-w)

    __device__ __forceinline__ void ray(int i, int j, F3& o, F3& d) const {
        if (PROJ == MRT_PROJ_PERSP) {
            o = fixed;
            d = add3(col[i], row[j]);
        } else {
            o = add3(col[i], row[j]);
            d = fixed;
        }
    }
};

}  // namespace

template <int PROJ>
__global__ __launch_bounds__(kBlockThreads) void trace_kernel(const RenderParams p,
                                                             float* __restrict__ t_out,
                                                             int32_t* __restrict__ id_out,
                                                             uint32_t pitch) {
    __shared__ float4 s_rec[kChunkTris * 3];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t x0 = blockIdx.x * kBlockTileX + (wave & 1) * kWaveTile + (lane & 7) * kLanePx;
    const uint32_t lr0 = blockIdx.y * kBlockTileY + (wave >> 1) * kWaveTile + (lane >> 3) * kLanePx;
    // Wave-uniform: does this wave's 32x32 tile touch the image at all?
    const uint32_t wave_x0 = blockIdx.x * kBlockTileX + (wave & 1) * kWaveTile;
    const uint32_t wave_lr0 = blockIdx.y * kBlockTileY + (wave >> 1) * kWaveTile;
    const bool wave_active = (wave_x0 < p.xres) && (wave_lr0 < p.local_rows);

    const F3 eye = mk3(p.eye[0], p.eye[1], p.eye[2]);
    const F3 cu = mk3(p.u[0], p.u[1], p.u[2]);
    const F3 cv = mk3(p.v[0], p.v[1], p.v[2]);
    const F3 cw = mk3(p.w[0], p.w[1], p.w[2]);

    // Pixel coordinates as floats for the prefilter, and the exact rays.
    const float xf = (float)x0;
    float yf[kLanePx];
    RaySetup<PROJ> rs;
    if (PROJ == MRT_PROJ_PERSP) rs.fixed = eye; else rs.fixed = mk3(-cw.x, -cw.y, -cw.z);
    const F3 negwf = mk3(-cw.x * p.focal, -cw.y * p.focal, -cw.z * p.focal);
#pragma unroll
    for (int i = 0; i < kLanePx; ++i) {
        // gfxraytrace.hpp:850-852: pixel grid is (x + 1, y + 1) as floats.
        float ux = uv_component(p.left, p.right, (float)(x0 + i) + 1.0f, p.xres_f);
        if (PROJ == MRT_PROJ_PERSP) rs.col[i] = add3(negwf, scale3(cu, ux));
        else rs.col[i] = scale3(cu, ux);
    }
#pragma unroll
    for (int j = 0; j < kLanePx; ++j) {
        uint32_t lr = lr0 + j;
        uint32_t y = image_row(p, lr < p.local_rows ? lr : (p.local_rows ? p.local_rows - 1 : 0));
        yf[j] = (float)y;
        float uy = uv_component(p.bottom, p.top, (float)y + 1.0f, p.yres_f);
        rs.row[j] = scale3(cv, uy);
    }

    // scene::intersect, gfxraytrace.hpp:812-813.
    const float t_lower = 0.0f;
    float best_t[kLanePx * kLanePx];
    int32_t best_id[kLanePx * kLanePx];
#pragma unroll
    for (int r = 0; r < kLanePx * kLanePx; ++r) { best_t[r] = FLT_MAX; best_id[r] = -1; }

    unsigned long long n_exact = 0;
    const bool count_exact = p.exact_counter != nullptr;

    // ---- spheres: object ids 0 .. n_sph-1 (gfxraytrace.hpp:701-706) ----
    if (wave_active) {
        for (uint32_t s = 0; s < p.n_sph; ++s) {
            const float4 sp = p.sph[s];
            const F3 center = mk3(sp.x, sp.y, sp.z);
#pragma unroll
            for (int j = 0; j < kLanePx; ++j) {
#pragma unroll
                for (int i = 0; i < kLanePx; ++i) {
                    const int r = j * kLanePx + i;
                    F3 o, d;
                    rs.ray(i, j, o, d);
                    float a, b;
                    float disc = sphere_disc(center, sp.w, o, d, a, b);
                    if (!(disc < 0)) {
                        float t = -b;
                        if (disc > 0) t -= sqrtf(disc);
                        t /= a;
                        // circle_intersect's range test against the current
                        // bound (kernel.cl:268,287) and the loop's own
                        // acceptance (kernel.cl:344) are the same predicate.
                        if (t < best_t[r] && t > t_lower) { best_t[r] = t; best_id[r] = (int32_t)s; }
                        if (count_exact) ++n_exact;
                    }
                }
            }
        }
    }

    // ---- triangles: object ids n_sph .. n_sph+n_tri-1 ----
    const bool no_prefilter = (p.flags & 1u) != 0;
    for (uint32_t base = 0; base < p.n_tri_padded; base += kChunkTris) {
        __syncthreads();
        {
            const float4* src = p.tri_filter + (size_t)base * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) s_rec[tid + k * kBlockThreads] = src[tid + k * kBlockThreads];
        }
        __syncthreads();
        if (!wave_active) continue;

        for (int g = 0; g < kChunkTris; g += kGroupTris) {
            float acc = -1.0f;
#pragma unroll
            for (int tt = 0; tt < kGroupTris; ++tt) {
                const float4 e0 = s_rec[(g + tt) * 3 + 0];
                const float4 e1 = s_rec[(g + tt) * 3 + 1];
                const float4 e2 = s_rec[(g + tt) * 3 + 2];
                const float c0 = __builtin_fmaf(e0.x, xf, e0.z);
                const float c1 = __builtin_fmaf(e1.x, xf, e1.z);
                const float c2 = __builtin_fmaf(e2.x, xf, e2.z);
#pragma unroll
                for (int j = 0; j < kLanePx; ++j) {
                    float v0 = __builtin_fmaf(e0.y, yf[j], c0);
                    float v1 = __builtin_fmaf(e1.y, yf[j], c1);
                    float v2 = __builtin_fmaf(e2.y, yf[j], c2);
#pragma unroll
                    for (int i = 0; i < kLanePx; ++i) {
                        float m = __builtin_fminf(__builtin_fminf(v0, v1), v2);
                        acc = __builtin_fmaxf(acc, m);
                        v0 += e0.x; v1 += e1.x; v2 += e2.x;
                    }
                }
            }
            const bool pass = (acc >= 0.0f) || no_prefilter;
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;

            // Rare path: find the (pixel, triangle) pairs and run the exact test.
            for (int tt = 0; tt < kGroupTris; ++tt) {
                const uint32_t ti = base + g + tt;
                if (ti >= p.n_tri) break;
                const float4 e0 = s_rec[(g + tt) * 3 + 0];
                const float4 e1 = s_rec[(g + tt) * 3 + 1];
                const float4 e2 = s_rec[(g + tt) * 3 + 2];
                const float c0 = __builtin_fmaf(e0.x, xf, e0.z);
                const float c1 = __builtin_fmaf(e1.x, xf, e1.z);
                const float c2 = __builtin_fmaf(e2.x, xf, e2.z);
                uint32_t mask = 0;
#pragma unroll
                for (int j = 0; j < kLanePx; ++j) {
                    float v0 = __builtin_fmaf(e0.y, yf[j], c0);
                    float v1 = __builtin_fmaf(e1.y, yf[j], c1);
                    float v2 = __builtin_fmaf(e2.y, yf[j], c2);
#pragma unroll
                    for (int i = 0; i < kLanePx; ++i) {
                        float m = __builtin_fminf(__builtin_fminf(v0, v1), v2);
                        if (m >= 0.0f) mask |= 1u << (j * kLanePx + i);
                        v0 += e0.x; v1 += e1.x; v2 += e2.x;
                    }
                }
                if (no_prefilter) mask = 0xFFFFu;
                if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) continue;

                const float4 va = p.tri_exact[(size_t)ti * 3 + 0];
                const float4 vb = p.tri_exact[(size_t)ti * 3 + 1];
                const float4 vc = p.tri_exact[(size_t)ti * 3 + 2];
                const F3 A = mk3(va.x, va.y, va.z), B = mk3(vb.x, vb.y, vb.z), C = mk3(vc.x, vc.y, vc.z);
#pragma unroll
                for (int j = 0; j < kLanePx; ++j) {
#pragma unroll
                    for (int i = 0; i < kLanePx; ++i) {
                        const int r = j * kLanePx + i;
                        if (mask & (1u << r)) {
                            F3 o, d;
                            rs.ray(i, j, o, d);
                            float t;
                            // tri_intersect with the current bound (kernel.cl:324),
                            // then the loop's strict acceptance (kernel.cl:326).
                            if (tri_exact(A, B, C, o, d, best_t[r], t_lower, t) && t > t_lower &&
                                t < best_t[r]) {
                                best_t[r] = t;
                                best_id[r] = (int32_t)(p.n_sph + ti);
                            }
                            if (count_exact) ++n_exact;
                        }
                    }
                }
            }
        }
    }

    if (wave_active) {
        // The (t, id) planes are padded to whole block tiles, so no bounds
        // checks: each lane writes four aligned 16-byte row segments per plane.
#pragma unroll
        for (int j = 0; j < kLanePx; ++j) {
            const size_t off = (size_t)(lr0 + j) * pitch + x0;
            float4 tv;
            int4 iv;
            tv.x = best_id[j * 4 + 0] < 0 ? -1.0f : best_t[j * 4 + 0];
            tv.y = best_id[j * 4 + 1] < 0 ? -1.0f : best_t[j * 4 + 1];
            tv.z = best_id[j * 4 + 2] < 0 ? -1.0f : best_t[j * 4 + 2];
            tv.w = best_id[j * 4 + 3] < 0 ? -1.0f : best_t[j * 4 + 3];
            iv.x = best_id[j * 4 + 0]; iv.y = best_id[j * 4 + 1];
            iv.z = best_id[j * 4 + 2]; iv.w = best_id[j * 4 + 3];
            *reinterpret_cast<float4*>(t_out + off) = tv;
            *reinterpret_cast<int4*>(id_out + off) = iv;
        }
        if (count_exact && n_exact) atomicAdd(p.exact_counter, n_exact);
    }
}

// One thread per pixel; consecutive lanes take consecutive pixels of a row so
// each wave writes 256 contiguous bytes of RGBA8.
template <int PROJ>
__global__ __launch_bounds__(256) void shade_kernel(const RenderParams p,
                                                   const float* __restrict__ t_in,
                                                   const int32_t* __restrict__ id_in,
                                                   uint32_t pitch) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t lr = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.xres || lr >= p.local_rows) return;
    const uint32_t y = image_row(p, lr);
    const size_t src = (size_t)lr * pitch + x;
    const int32_t id = id_in[src];
    const float t = t_in[src];

    float r, g, b;
    if (id < 0) {
        // kernel.cl:384-389 / :441-444.
        r = p.bg[0]; g = p.bg[1]; b = p.bg[2];
    } else {
        const F3 eye = mk3(p.eye[0], p.eye[1], p.eye[2]);
        const F3 cu = mk3(p.u[0], p.u[1], p.u[2]);
        const F3 cv = mk3(p.v[0], p.v[1], p.v[2]);
        const F3 cw = mk3(p.w[0], p.w[1], p.w[2]);
        const float ux = uv_component(p.left, p.right, (float)x + 1.0f, p.xres_f);
        const float uy = uv_component(p.bottom, p.top, (float)y + 1.0f, p.yres_f);
        F3 o, d;
        if (PROJ == MRT_PROJ_PERSP) persp_ray(eye, cu, cv, cw, p.focal, ux, uy, o, d);
        else ortho_ray(cu, cv, cw, ux, uy, o, d);
        // kernel.cl:224 / :270 / :289: location = origin + direction * t.
        const F3 loc = add3(o, scale3(d, t));
        F3 n;
        uint32_t mat;
        if ((uint32_t)id < p.n_sph) {
            const float4 sp = p.sph[id];
            const F3 center = mk3(sp.x, sp.y, sp.z);
            float a, bq;
            const float disc = sphere_disc(center, sp.w, o, d, a, bq);
            // kernel.cl:271 (two roots) vs :290 (tangent).
            if (disc > 0) n = normalize3(div3(sub3(loc, center), sp.w));
            else n = normalize3(sub3(loc, center));
            mat = p.sph_mat[id];
        } else {
            const uint32_t ti = (uint32_t)id - p.n_sph;
            const float4 va = p.tri_exact[(size_t)ti * 3 + 0];
            const float4 vb = p.tri_exact[(size_t)ti * 3 + 1];
            const float4 vc = p.tri_exact[(size_t)ti * 3 + 2];
            const F3 A = mk3(va.x, va.y, va.z);
            // kernel.cl:225-227.
            n = normalize3(cross3(sub3(A, mk3(vb.x, vb.y, vb.z)), sub3(A, mk3(vc.x, vc.y, vc.z))));
            mat = __float_as_uint(va.w);
        }
        const float4 m = p.mats[mat];
        const F3 color = mk3(m.x, m.y, m.z);
        if (p.shader == MRT_SHADER_FLAT) {
            r = color.x; g = color.y; b = color.z;           // kernel.cl:392-394
        } else {
            r = p.ka * p.amb[0];                             // kernel.cl:404-406
            g = p.ka * p.amb[1];
            b = p.ka * p.amb[2];
            for (uint32_t l = 0; l < p.n_lights; ++l) {
                const float4 lp = p.lights[2 * l], lc = p.lights[2 * l + 1];
                phong_light(loc, n, color, m.w, eye, mk3(lp.x, lp.y, lp.z), mk3(lc.x, lc.y, lc.z),
                            p.kd, p.ks, r, g, b);
            }
            r = clamp01(r); g = clamp01(g); b = clamp01(b);
        }
    }
    const uint32_t px = (to_byte(r) & 0xFFu) | ((to_byte(g) & 0xFFu) << 8) |
                        ((to_byte(b) & 0xFFu) << 16) | 0xFF000000u;
    p.out[(size_t)lr * p.xres + x] = px;
}

template __global__ void trace_kernel<MRT_PROJ_ORTHO>(const RenderParams, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP>(const RenderParams, float*, int32_t*, uint32_t);
template __global__ void shade_kernel<MRT_PROJ_ORTHO>(const RenderParams, const float*, const int32_t*, uint32_t);
template __global__ void shade_kernel<MRT_PROJ_PERSP>(const RenderParams, const float*, const int32_t*, uint32_t);

}  // namespace mrt
