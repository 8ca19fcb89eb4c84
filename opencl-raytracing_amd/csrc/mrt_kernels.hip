// mrt_kernels.hip -- CDNA4 (gfx950) device code of the fused render path.
//
// One launch per frame replaces the reference's four stage kernels and the
// host round trips between them (gfxraytrace.hpp:830-878): primary-ray
// generation + brute-force closest hit over every sphere and triangle (kernels
// uv / *_viewrays / intersections, kernel.cl:123-379), then, in the same
// kernel's epilogue, shade_pixel rebuilds the winning hit record, shades it
// (flat_shader / phong_shader, kernel.cl:381-445), converts to bytes
// (gfximage.hpp:46-51) and writes coalesced RGBA8.  (t, object id) per pixel
// go to two planes only when the caller asks for them (t_out != nullptr) and
// for shade_kernel, the separate shading launch cull_trace_kernel uses.
//
// Three kernels implement the closest-hit loop, selected by mrt_config.flags;
// all give the same winners, t and bytes:
//
//   trace_kernel       the default.  Per (pixel, triangle) it tests a bounding
//                      circle, then three affine edge functions of the pixel
//                      coordinate (primary rays of a pinhole or orthographic
//                      camera are affine in the pixel, so the numerators of
//                      Cramer's rule are too) and rejects the pair when one is
//                      below a per-edge error margin.  The few pairs that
//                      survive run the reference arithmetic (mrt_device.hpp),
//                      in object order, so the accepted hits, their t and the
//                      winner are those of the reference loop.
//   full_trace_kernel  MRT_FLAG_FULL_TEST: the determinants of the 3x3 solve
//                      (fused) for every pair, no rejection in screen space,
//                      the reference arithmetic for the survivors.  The rays
//                      of a frame share origin or direction and the per-view
//                      records fold that half in: not a general-ray kernel
//                      (that one is general_intersect_kernel, mrt_stages.hip).
//   cull_trace_kernel  not a brute-force kernel (mrt_accel_*): lanes first test
//                      64 primitives each against the whole wave tile.
//   grid_trace_kernel  not a brute-force kernel either (mrt_accel_*): walks the
//                      per-tile primitive lists of a grid built per view.
#include "mrt.h"
#include "mrt_device.hpp"
#include "mrt_kernels.hpp"
#include "mrt_tile_tests.hpp"

#include <float.h>
#include <type_traits>

#pragma clang fp contract(off)

namespace mrt {

namespace {

// A constant held in a vector register.  VALU instructions that read a scalar
// register or a literal issue at half rate on this chip, so the constants of
// the inner loop are made opaque to the compiler once, outside it.
__device__ __forceinline__ float in_vgpr(float x) {
    asm("" : "+v"(x));
    return x;
}

#ifdef MRT_TIMELINE
// Diagnostic build only (tools/wave_timeline.py): every wave of trace_kernel records when it
// started, when it had its first chunk of records in LDS and when it ended (the 100 MHz clock all
// XCDs share), and where it ran (HW_ID and XCC_ID registers).
constexpr uint32_t kTimelineMax = 1u << 18;
struct TimelineEntry {
    unsigned long long t_begin, t_staged, t_end;
    uint32_t hw_id, xcc_id;
    uint32_t block_x, block_y, block_z, wave;
    unsigned long long cycles;       // shader clock ticks between begin and end
    unsigned long long t_setup, t_sync, t_loaded;   // before the first barrier, after it, first chunk written to LDS
    unsigned long long pad;
};
__device__ TimelineEntry g_timeline[kTimelineMax];
__device__ uint32_t g_timeline_count;
// Each mark is stored at once by lane 0 (a slot per wave of the grid, no atomics: a shared counter
// costs the frame 0.5 ms), so that the kernel keeps no timestamps in registers.
#define MRT_TL_SLOT (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (kBlockThreads / 64) + (threadIdx.x >> 6))
#define MRT_TL_MARK(field)                                                                  \
    do {                                                                                    \
        const uint32_t tl_slot = MRT_TL_SLOT;                                               \
        if ((threadIdx.x & 63) == 0 && tl_slot < kTimelineMax) g_timeline[tl_slot].field = wall_clock64(); \
    } while (0)
#else
#define MRT_TL_MARK(field) do { } while (0)
#endif

// Image row of packed local row lr for this rank (interleaved bands).
__device__ __forceinline__ uint32_t image_row(const RenderParams& p, uint32_t lr) {
    if (p.world == 1) return lr;              // one rank: its rows are the image's (wave-uniform branch)
    uint32_t band = lr / p.band_rows;
    uint32_t within = lr - band * p.band_rows;
    return (band * p.world + p.rank) * p.band_rows + within;
}

// The exact primary ray of a pixel from its two uv components.
template <int PROJ>
__device__ __forceinline__ void pixel_ray(const RenderParams& p, float ux, float uy, F3& o, F3& d) {
    const F3 cu = mk3(p.u[0], p.u[1], p.u[2]);
    const F3 cv = mk3(p.v[0], p.v[1], p.v[2]);
    const F3 cw = mk3(p.w[0], p.w[1], p.w[2]);
    if (PROJ == MRT_PROJ_PERSP) persp_ray(mk3(p.eye[0], p.eye[1], p.eye[2]), cu, cv, cw, p.focal, ux, uy, o, d);
    else ortho_ray(cu, cv, cw, ux, uy, o, d);
}

// Cooperative copy of one chunk of prefilter records into LDS.
__device__ __forceinline__ void stage_chunk(float4* dst, const float4* __restrict__ src, int tid) {
#pragma unroll
    for (int k = 0; k < kChunkFloat4 / kBlockThreads; ++k)
        dst[tid + k * kBlockThreads] = src[tid + k * kBlockThreads];
}
// ... of its first n float4 only (the last chunk of a slice that is not a whole number of chunks).
__device__ __forceinline__ void stage_chunk(float4* dst, const float4* __restrict__ src, int tid, uint32_t n) {
#pragma unroll
    for (int k = 0; k < kChunkFloat4 / kBlockThreads; ++k)
        if ((uint32_t)(tid + k * kBlockThreads) < n) dst[tid + k * kBlockThreads] = src[tid + k * kBlockThreads];
}

// The colour of pixel (x, y) given its closest hit (t, object id; id < 0: no hit): rebuilds
// the hit record tri_intersect / circle_intersect would have returned, shades it (flat_shader /
// phong_shader, kernel.cl:381-445) and converts to bytes (gfximage.hpp:46-51).  RGBA8, A = 255.
template <int PROJ>
__device__ __forceinline__ uint32_t shade_pixel(const RenderParams& p, uint32_t x, uint32_t y, float t, int32_t id) {
    float r, g, b;
    if (id < 0) {
        // kernel.cl:384-389 / :441-444.
        r = p.bg[0]; g = p.bg[1]; b = p.bg[2];
    } else {
        const F3 eye = mk3(p.eye[0], p.eye[1], p.eye[2]);
        const float ux = uv_component(p.left, p.right, (float)x + 1.0f, p.xres_f);
        const float uy = uv_component(p.bottom, p.top, (float)y + 1.0f, p.yres_f);
        F3 o, d;
        pixel_ray<PROJ>(p, ux, uy, o, d);
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
    return (to_byte(r) & 0xFFu) | ((to_byte(g) & 0xFFu) << 8) | ((to_byte(b) & 0xFFu) << 16) | 0xFF000000u;
}

// Shades one lane row segment of four pixels starting at (x0, local row lr) and writes it to
// the RGBA8 output: one aligned 16-byte store when p.fuse_shade == 2, single pixels otherwise.
template <int PROJ>
__device__ __forceinline__ void shade_row4(const RenderParams& p, uint32_t x0, uint32_t lr, float4 tv, int4 iv) {
    if (!(lr < p.local_rows && x0 < p.xres)) return;
    const uint32_t y = image_row(p, lr);
    const float tt[4] = {tv.x, tv.y, tv.z, tv.w};
    const int32_t ii[4] = {iv.x, iv.y, iv.z, iv.w};
    uint32_t px[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) px[i] = (x0 + i < p.xres) ? shade_pixel<PROJ>(p, x0 + i, y, tt[i], ii[i]) : 0u;
    uint32_t* dst = p.out + (size_t)lr * p.xres + x0;
    if (p.fuse_shade == 2) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(px[0], px[1], px[2], px[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (x0 + i < p.xres) dst[i] = px[i];
    }
}

// The closest hit of a pixel as one 64-bit key, (t bits << 32) | object id.  Accepted hits have
// 0 < t < FLT_MAX, so keys order like (t, id), and the smallest key over all hits of a pixel is
// the hit the reference's loop ends with: the smallest t, and among equal t the lowest object id
// (strict "<" in object order, kernel.cl:326,344).  That holds in whatever order the pairs are
// evaluated, which lets the kernels below hand pixels to whichever lane is free.
constexpr unsigned long long kNoHitKey = ((unsigned long long)0x7F7FFFFFu << 32) | 0xFFFFFFFFull;   // (FLT_MAX, -1)

// The reference arithmetic for one primitive over a list of pixels of the wave's tile, TW pixels
// wide (entry = TW * row + column), 64 pixels at a time whichever lane owns them.  tile_ux / tile_uy
// are the uv components of the tile's columns and rows; keys points at the tile's first pixel
// in a key array with `stride` keys per row (LDS or global memory).
template <int PROJ, bool TRI, int TW = kWaveTile>
__device__ __forceinline__ void exact_list(const RenderParams& p, uint32_t prim, uint32_t total, int lane,
                                           const uint16_t* __restrict__ list, const float* __restrict__ tile_ux,
                                           const float* __restrict__ tile_uy, unsigned long long* __restrict__ keys,
                                           uint32_t stride, unsigned long long& n_exact, bool& recorded) {
    F3 A, B, C;
    float radius = 0.0f;
    if (TRI) {
        const float4 va = p.tri_exact[(size_t)prim * 3 + 0];
        const float4 vb = p.tri_exact[(size_t)prim * 3 + 1];
        const float4 vc = p.tri_exact[(size_t)prim * 3 + 2];
        A = mk3(va.x, va.y, va.z); B = mk3(vb.x, vb.y, vb.z); C = mk3(vc.x, vc.y, vc.z);
    } else {
        const float4 sp = p.sph[prim];
        A = mk3(sp.x, sp.y, sp.z); B = A; C = A;
        radius = sp.w;
    }
    const uint32_t obj_id = TRI ? p.n_sph + prim : prim;          // spheres first, gfxraytrace.hpp:701-714
    for (uint32_t b = 0; b < total; b += 64) {
        const uint32_t idx = b + (uint32_t)lane;
        bool rec = false;
        if (idx < total) {
            const uint32_t px = list[idx];
            F3 o, d;
            pixel_ray<PROJ>(p, tile_ux[px % (uint32_t)TW], tile_uy[px / (uint32_t)TW], o, d);
            float t;
            bool hit, two;
            // Bounds of scene::intersect (gfxraytrace.hpp:812-813); the running bound of the
            // reference's loop only rejects what the minimum rejects too.
            if (TRI) hit = tri_exact(A, B, C, o, d, FLT_MAX, 0.0f, t);
            else hit = sphere_exact(A, radius, o, d, FLT_MAX, 0.0f, t, two);
            if (hit && t > 0.0f && t < FLT_MAX) {
                atomicMin(&keys[(size_t)(px / (uint32_t)TW) * stride + (px % (uint32_t)TW)],
                          ((unsigned long long)__float_as_uint(t) << 32) | obj_id);
                rec = true;
            }
            ++n_exact;
        }
        // wave-uniform, so it lives in a scalar register
        recorded = recorded || __builtin_amdgcn_ballot_w64(rec) != 0;
    }
}

// Lists the pixels of `mask` (bit r = row r/LX, column r%LX of the lane's LX-wide block at
// (bx, by) inside the tile, TW pixels wide) in `list`; returns how many pixels the whole wave listed.  LDS operations of
// one wave complete in order and the arrays belong to this wave alone, so wave-level fences are
// all the synchronisation needed.
template <int LX = kLanePx, int TW = kWaveTile, typename MASK = uint32_t>
__device__ __forceinline__ uint32_t list_pixels(MASK mask, uint32_t bx, uint32_t by, int lane,
                                                uint16_t* __restrict__ list, uint32_t* __restrict__ count) {
    const uint32_t c = (uint32_t)__builtin_popcountll((unsigned long long)mask);
    uint32_t at = 0;
    if (c) at = atomicAdd(count, c);
    while (mask) {
        const uint32_t r = (uint32_t)__builtin_ctzll((unsigned long long)mask);
        mask &= mask - 1;
        list[at++] = (uint16_t)((by + r / (uint32_t)LX) * TW + bx + r % (uint32_t)LX);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t total = *count;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) *count = 0;
    return (uint32_t)__builtin_amdgcn_readfirstlane(total);
}

}  // namespace

template <int PROJ, bool BLOCK_REJECT, int LX, int LY, bool SLICED>
__global__ __launch_bounds__(kBlockThreads, MRT_TRACE_WAVES_PER_SIMD) void trace_kernel(const RenderParams p,
                                                             unsigned long long* __restrict__ key_plane,
                                                             float* __restrict__ t_out,
                                                             int32_t* __restrict__ id_out,
                                                             uint32_t pitch) {
    using Shape = TraceShape<LX, LY>;
    constexpr int kTrLaneX = Shape::lane_x, kTrLaneY = Shape::lane_y;
    constexpr int kTrTileX = Shape::tile_x, kTrTileY = Shape::tile_y;
    constexpr int kTrBlockX = Shape::block_x, kTrBlockY = Shape::block_y;
    static_assert(!BLOCK_REJECT || (LX == 4 && LY == 4), "the block-origin constants of the records are made for 4x4 blocks");
    __shared__ float4 s_rec[kChunkFloat4];
    __shared__ float4 s_circ[kChunkTris];
    // per wave: the pixels one primitive still has to be tested on, and the tile's uv components
    // (the pixel list holds 16 pixels per lane: a larger lane block is listed and tested in parts
    // of 16 / LX rows, which keeps the kernel's LDS at six blocks per CU for either shape)
    constexpr int kListRows = 16 / kTrLaneX, kListParts = kTrLaneY / kListRows;
    static_assert(kListRows >= 1 && kTrLaneY % kListRows == 0, "whole rows per part");
    __shared__ uint16_t s_list[kBlockThreads / 64][64 * 16];
    __shared__ float s_ux[kBlockThreads / 64][kTrTileX], s_uy[kBlockThreads / 64][kTrTileY];
    // one bit per pixel of the lane's block, bit j * kTrLaneX + i = row j, column i
    using Mask = typename std::conditional<(kTrLaneX * kTrLaneY > 32), unsigned long long, uint32_t>::type;
    constexpr Mask kAllPixels = (Mask)(~(Mask)0) >> (8 * sizeof(Mask) - kTrLaneX * kTrLaneY);
    __shared__ uint32_t s_count[kBlockThreads / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#ifdef MRT_TIMELINE
    MRT_TL_MARK(t_begin);
    if (lane == 0 && MRT_TL_SLOT < kTimelineMax) g_timeline[MRT_TL_SLOT].cycles = clock64();
#endif
    // Launch order (see mrt_kernels.hpp): which of blockIdx's components is the slice.
    const uint32_t blk_x = p.tile_major ? blockIdx.y : blockIdx.x;
    const uint32_t blk_y = p.tile_major ? blockIdx.z : blockIdx.y;
    const uint32_t blk_slice = p.tile_major ? blockIdx.x : blockIdx.z;
    const uint32_t x0 = blk_x * kTrBlockX + (wave & 1) * kTrTileX + (lane & 7) * kTrLaneX;
    const uint32_t lr0 = blk_y * kTrBlockY + (wave >> 1) * kTrTileY + (lane >> 3) * kTrLaneY;
    // Wave-uniform: does this wave's tile touch the image at all?
    const uint32_t wave_x0 = blk_x * kTrBlockX + (wave & 1) * kTrTileX;
    const uint32_t wave_lr0 = blk_y * kTrBlockY + (wave >> 1) * kTrTileY;
    const bool wave_active = (wave_x0 < p.xres) && (wave_lr0 < p.local_rows);

    // Pixel coordinates as floats: all the prefilters need.
    const float xf = (float)x0;
    float yf[kTrLaneY];
#pragma unroll
    for (int j = 0; j < kTrLaneY; ++j) {
        const uint32_t lr = lr0 + j;
        const uint32_t last_row = p.local_rows ? p.local_rows - 1 : 0;
        yf[j] = (float)image_row(p, lr < last_row ? lr : last_row);
    }

    // Closest hit so far, one key per pixel in global memory: (FLT_MAX, -1), scene::intersect's
    // upper bound (gfxraytrace.hpp:813) and "no object".  The planes are padded to whole block
    // tiles, so no bounds checks.  Also the uv components of the tile's 32 columns and 32 rows
    // (gfxraytrace.hpp:850-852: the pixel grid is (x + 1, y + 1)) for the exact step.
    const uint32_t bx = (lane & 7) * kTrLaneX, by = (lane >> 3) * kTrLaneY;
    uint16_t* list = s_list[wave];
    float* tile_ux = s_ux[wave];
    float* tile_uy = s_uy[wave];
    uint32_t* list_count = &s_count[wave];
    unsigned long long* tile_keys = key_plane + (size_t)wave_lr0 * pitch + wave_x0;
    // Small frames do not have enough tiles to fill the GPU, so the host may cut the triangle list
    // into slices of p.slice_tris triangles, one grid layer each (the keys merge them: their minimum does
    // not depend on who found a hit).  It then initialises the key plane itself, before the
    // launch, and shades in shade_keys_kernel, after it; a single slice does both here.
    // SLICED is a template parameter because the epilogue (shading) is what costs registers: built
    // without it the kernel spills nothing and uses no scratch memory at all.
    constexpr bool single_slice = !SLICED;
    if (wave_active) {
        if (single_slice) {
#pragma unroll
            for (int j = 0; j < kTrLaneY; ++j) {
                ulonglong2* row = reinterpret_cast<ulonglong2*>(key_plane + (size_t)(lr0 + j) * pitch + x0);
#pragma unroll
                for (int i = 0; i < kTrLaneX / 2; ++i) row[i] = make_ulonglong2(kNoHitKey, kNoHitKey);
            }
        }
        const uint32_t last_row = p.local_rows ? p.local_rows - 1 : 0;
        for (uint32_t k = (uint32_t)lane; k < (uint32_t)(kTrTileX + kTrTileY); k += 64) {
            if (k < (uint32_t)kTrTileX) {
                tile_ux[k] = uv_component(p.left, p.right, (float)(wave_x0 + k) + 1.0f, p.xres_f);
            } else {
                const uint32_t lr = wave_lr0 + (k - kTrTileX);
                tile_uy[k - kTrTileX] = uv_component(p.bottom, p.top, (float)image_row(p, lr < last_row ? lr : last_row) + 1.0f, p.yres_f);
            }
        }
        if (lane == 0) *list_count = 0;
    }
    // (the first __syncthreads() of the chunk loops below orders these writes before their use)
    MRT_TL_MARK(t_setup);

    unsigned long long n_exact = 0;
    bool recorded = false;           // this wave recorded a hit in the key plane
    const bool no_prefilter = (p.flags & MRT_FLAG_NO_PREFILTER) != 0;
    const bool use_circle = (p.flags & (MRT_FLAG_NO_PREFILTER | MRT_FLAG_NO_CIRCLE)) == 0;

    // ---- spheres: object ids 0 .. n_sph-1 ----
    // Record (sx, ox, sy, oy): the pixel can only hit if
    // (x*sx + ox)^2 + (y*sy + oy)^2 <= 1, an ellipse around the sphere's
    // projected bounding box grown by the error margin.
    const uint32_t sph_end = blk_slice == 0 ? p.n_sph_padded : 0u;       // spheres go with the first slice
    for (uint32_t base = 0; base < sph_end; base += kChunkFloat4) {
        __syncthreads();
        stage_chunk(s_rec, p.sph_filter + base, tid);
        __syncthreads();
        if (!wave_active) continue;
        for (int g = 0; g < kChunkFloat4; g += kGroupSph) {
            if (base + g >= p.n_sph) break;
            float acc = 1.0f;
#pragma unroll
            for (int tt = 0; tt < kGroupSph; ++tt) {
                const float4 e = s_rec[g + tt];
                float dvm[kTrLaneY];
#pragma unroll
                for (int j = 0; j < kTrLaneY; ++j) {
                    const float dv = __builtin_fmaf(yf[j], e.z, e.w);
                    dvm[j] = __builtin_fmaf(dv, dv, -1.0f);
                }
                float du = __builtin_fmaf(xf, e.x, e.y);
#pragma unroll
                for (int i = 0; i < kTrLaneX; ++i) {
#pragma unroll
                    for (int j = 0; j < kTrLaneY; ++j) acc = __builtin_fminf(acc, __builtin_fmaf(du, du, dvm[j]));
                    du += e.x;
                }
            }
            const bool pass = (acc <= 0.0f) || no_prefilter;
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;

            for (int tt = 0; tt < kGroupSph; ++tt) {
                const uint32_t si = base + g + tt;
                if (si >= p.n_sph) break;
                const float4 e = s_rec[g + tt];
                Mask mask = 0;
                float du = __builtin_fmaf(xf, e.x, e.y);
#pragma unroll
                for (int i = 0; i < kTrLaneX; ++i) {
#pragma unroll
                    for (int j = 0; j < kTrLaneY; ++j) {
                        const float dv = __builtin_fmaf(yf[j], e.z, e.w);
                        if (__builtin_fmaf(du, du, __builtin_fmaf(dv, dv, -1.0f)) <= 0.0f) mask |= (Mask)1 << (j * kTrLaneX + i);
                    }
                    du += e.x;
                }
                if (no_prefilter) mask = kAllPixels;
                if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) continue;
#pragma unroll
                for (int part = 0; part < kListParts; ++part) {
                    const uint32_t bits = (uint32_t)(mask >> (part * 16)) & 0xFFFFu;
                    if (kListParts > 1 && __builtin_amdgcn_ballot_w64(bits != 0) == 0) continue;
                    const uint32_t total = list_pixels<kTrLaneX, kTrTileX>(bits, bx, by + part * kListRows, lane, list, list_count);
                    exact_list<PROJ, false, kTrTileX>(p, si, total, lane, list, tile_ux, tile_uy, tile_keys, pitch, n_exact, recorded);
                }
            }
        }
    }

    // ---- triangles: object ids n_sph .. n_sph+n_tri-1 ----
    // Record: 12 floats (A0,B0,C0,A1 | B1,C1,A2,B2 | C2,-,-,-); edge k passes at
    // pixel (x, y) when A_k*x + B_k*y + C_k >= 0.
    const uint32_t tri_begin = single_slice ? 0u : blk_slice * p.slice_tris;
    // A slice is a whole number of circle groups (the host sees to that), not of chunks.
    const uint32_t tri_end = single_slice ? p.n_tri_padded
                                          : (tri_begin + p.slice_tris < p.n_tri_padded ? tri_begin + p.slice_tris : p.n_tri_padded);
    const uint32_t tri_stop = tri_end < p.n_tri ? tri_end : p.n_tri;     // the first triangle this wave does not test
    for (uint32_t base = tri_begin; base < tri_end; base += kChunkTris) {
        const uint32_t staged = tri_end - base < (uint32_t)kChunkTris ? tri_end - base : (uint32_t)kChunkTris;
        __syncthreads();
#ifdef MRT_TIMELINE
        if (base == tri_begin) MRT_TL_MARK(t_sync);
#endif
        stage_chunk(s_rec, p.tri_filter + (size_t)base * 3, tid, staged * 3);
#pragma unroll
        for (int k = 0; k < kChunkTris / kBlockThreads; ++k)
            if ((uint32_t)(tid + k * kBlockThreads) < staged) s_circ[tid + k * kBlockThreads] = p.tri_circle[base + tid + k * kBlockThreads];
#ifdef MRT_TIMELINE
        if (base == tri_begin) MRT_TL_MARK(t_loaded);
#endif
        __syncthreads();
#ifdef MRT_TIMELINE
        if (base == tri_begin) MRT_TL_MARK(t_staged);
#endif
        if (!wave_active) continue;

        for (int gc = 0; gc < kChunkTris; gc += kGroupCircle) {
        if (base + gc >= tri_stop) break;
        if (use_circle && BLOCK_REJECT) {
            // Opt-in block form of the circle test: the point of this lane's 4x4 block
            // nearest to the centre, per axis, is the centre's coordinate clamped to the
            // block's range; if even that point is outside the circle, every pixel is.
            float accc = 1.0f;
#pragma unroll
            for (int tt = 0; tt < kGroupCircle; ++tt) {
                const float4 c = s_circ[gc + tt];
                const float du0 = xf - c.x, dv0 = yf[0] - c.y;
                const float du = __builtin_amdgcn_fmed3f(du0, du0 + (float)(kTrLaneX - 1), 0.0f);
                const float dv = __builtin_amdgcn_fmed3f(dv0, dv0 + (float)(kTrLaneY - 1), 0.0f);
                accc = __builtin_fminf(accc, __builtin_fmaf(du, du, __builtin_fmaf(dv, dv, -c.z)));
            }
            if (__builtin_amdgcn_ballot_w64(accc <= 0.0f) == 0) continue;
        } else if (use_circle) {
            // First rejection, per pixel: squared distance to the centre of a circle
            // that contains everything the three edge functions can pass, (xc, yc, R^2).
            float accc = 1.0f;
#pragma unroll
            for (int tt = 0; tt < kGroupCircle; ++tt) {
                const float4 c = s_circ[gc + tt];
                float dvm[kTrLaneY];
#pragma unroll
                for (int j = 0; j < kTrLaneY; ++j) {
                    const float dv = yf[j] - c.y;
                    dvm[j] = __builtin_fmaf(dv, dv, -c.z);
                }
                const float du0 = xf - c.x;
#pragma unroll
                for (int i = 0; i < kTrLaneX; ++i) {
                    const float du = du0 + (float)i;
#pragma unroll
                    for (int j = 0; j < kTrLaneY; ++j) accc = __builtin_fminf(accc, __builtin_fmaf(du, du, dvm[j]));
                }
            }
            if (__builtin_amdgcn_ballot_w64(accc <= 0.0f) == 0) continue;
        }
        for (int g = gc; g < gc + kGroupCircle; g += kGroupTris) {
            if (base + g >= tri_stop) break;
            if (BLOCK_REJECT) {
                // mrt_accel_* only (method MRT_ACCEL_BLOCK_REJECT): each edge function is monotone in x and
                // in y, so its maximum over this lane's 4x4 block is its value at one
                // corner; the record carries C'_k = C_k + 3*max(A_k,0) + 3*max(B_k,0), which
                // moves that corner to the block origin.  If some edge is negative at its
                // best corner, no pixel of the block passes and the per-pixel test is skipped.
                float accb = -1.0f;
#pragma unroll
                for (int tt = 0; tt < kGroupTris; ++tt) {
                    const float4 q0 = s_rec[(g + tt) * 3 + 0];
                    const float4 q1 = s_rec[(g + tt) * 3 + 1];
                    const float4 q2 = s_rec[(g + tt) * 3 + 2];
                    const float m0 = __builtin_fmaf(q0.x, xf, __builtin_fmaf(q0.y, yf[0], q2.y));
                    const float m1 = __builtin_fmaf(q0.w, xf, __builtin_fmaf(q1.x, yf[0], q2.z));
                    const float m2 = __builtin_fmaf(q1.z, xf, __builtin_fmaf(q1.w, yf[0], q2.w));
                    accb = __builtin_fmaxf(accb, __builtin_fminf(__builtin_fminf(m0, m1), m2));
                }
                if (__builtin_amdgcn_ballot_w64((accb >= 0.0f) || no_prefilter) == 0) continue;
            }
            float acc = -1.0f;
#pragma unroll
            for (int tt = 0; tt < kGroupTris; ++tt) {
                const float4 q0 = s_rec[(g + tt) * 3 + 0];
                const float4 q1 = s_rec[(g + tt) * 3 + 1];
                const float cc2 = reinterpret_cast<const float*>(&s_rec[(g + tt) * 3 + 2])[0];
                const float a0 = q0.x, b0 = q0.y, a1 = q0.w, b1 = q1.x, a2 = q1.z, b2 = q1.w;
                const float c0 = __builtin_fmaf(a0, xf, q0.z);
                const float c1 = __builtin_fmaf(a1, xf, q1.y);
                const float c2 = __builtin_fmaf(a2, xf, cc2);
#pragma unroll
                for (int j = 0; j < kTrLaneY; ++j) {
                    float v0 = __builtin_fmaf(b0, yf[j], c0);
                    float v1 = __builtin_fmaf(b1, yf[j], c1);
                    float v2 = __builtin_fmaf(b2, yf[j], c2);
#pragma unroll
                    for (int i = 0; i < kTrLaneX; ++i) {
                        float m = __builtin_fminf(__builtin_fminf(v0, v1), v2);
                        acc = __builtin_fmaxf(acc, m);
                        v0 += a0; v1 += a1; v2 += a2;
                    }
                }
            }
            const bool pass = (acc >= 0.0f) || no_prefilter;
            if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;

            // Rare path: find the (pixel, triangle) pairs and run the exact test.
            for (int tt = 0; tt < kGroupTris; ++tt) {
                const uint32_t ti = base + g + tt;
                if (ti >= p.n_tri) break;
                const float4 q0 = s_rec[(g + tt) * 3 + 0];
                const float4 q1 = s_rec[(g + tt) * 3 + 1];
                const float cc2 = reinterpret_cast<const float*>(&s_rec[(g + tt) * 3 + 2])[0];
                const float a0 = q0.x, b0 = q0.y, a1 = q0.w, b1 = q1.x, a2 = q1.z, b2 = q1.w;
                const float c0 = __builtin_fmaf(a0, xf, q0.z);
                const float c1 = __builtin_fmaf(a1, xf, q1.y);
                const float c2 = __builtin_fmaf(a2, xf, cc2);
                Mask mask = 0;
#pragma unroll
                for (int j = 0; j < kTrLaneY; ++j) {
                    float v0 = __builtin_fmaf(b0, yf[j], c0);
                    float v1 = __builtin_fmaf(b1, yf[j], c1);
                    float v2 = __builtin_fmaf(b2, yf[j], c2);
#pragma unroll
                    for (int i = 0; i < kTrLaneX; ++i) {
                        float m = __builtin_fminf(__builtin_fminf(v0, v1), v2);
                        if (m >= 0.0f) mask |= (Mask)1 << (j * kTrLaneX + i);
                        v0 += a0; v1 += a1; v2 += a2;
                    }
                }
                if (no_prefilter) mask = kAllPixels;
                if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) continue;
#pragma unroll
                for (int part = 0; part < kListParts; ++part) {
                    const uint32_t bits = (uint32_t)(mask >> (part * 16)) & 0xFFFFu;
                    if (kListParts > 1 && __builtin_amdgcn_ballot_w64(bits != 0) == 0) continue;
                    const uint32_t total = list_pixels<kTrLaneX, kTrTileX>(bits, bx, by + part * kListRows, lane, list, list_count);
                    exact_list<PROJ, true, kTrTileX>(p, ti, total, lane, list, tile_ux, tile_uy, tile_keys, pitch, n_exact, recorded);
                }
            }
        }
        }
    }

    if (wave_active && !single_slice) {
        // tell shade_keys_kernel that this tile has keys worth reading (any wave of any slice may)
        if (p.tile_flags != nullptr && recorded && lane == 0)
            p.tile_flags[(wave_lr0 / kTrTileY) * p.tiles_x + wave_x0 / kTrTileX] = p.frame_tag;
        if (p.exact_counter != nullptr && n_exact) atomicAdd(p.exact_counter, n_exact);
    } else if (wave_active) {
        // The atomics went to the device's L2; make sure the loads below see them.
        __threadfence();
#pragma unroll
        for (int j = 0; j < kTrLaneY; ++j) {
#pragma unroll
            for (int seg = 0; seg < kTrLaneX / 4; ++seg) {
                const size_t off = (size_t)(lr0 + j) * pitch + x0 + 4 * seg;
                const ulonglong2* row = reinterpret_cast<const ulonglong2*>(key_plane + off);
                const ulonglong2 k01 = row[0], k23 = row[1];
                int4 iv;
                float4 tv;
                iv.x = (int32_t)(uint32_t)k01.x; iv.y = (int32_t)(uint32_t)k01.y;
                iv.z = (int32_t)(uint32_t)k23.x; iv.w = (int32_t)(uint32_t)k23.y;
                // intersect.t_ of a miss is -1 (kernel.cl:363).
                tv.x = iv.x < 0 ? -1.0f : __uint_as_float((uint32_t)(k01.x >> 32));
                tv.y = iv.y < 0 ? -1.0f : __uint_as_float((uint32_t)(k01.y >> 32));
                tv.z = iv.z < 0 ? -1.0f : __uint_as_float((uint32_t)(k23.x >> 32));
                tv.w = iv.w < 0 ? -1.0f : __uint_as_float((uint32_t)(k23.y >> 32));
                if (t_out != nullptr) {         // the (t, id) planes are written only for callers that ask
                    *reinterpret_cast<float4*>(t_out + off) = tv;
                    *reinterpret_cast<int4*>(id_out + off) = iv;
                }
                // Shade this row segment here rather than in a second launch: four pixels, one
                // aligned 16-byte RGBA8 store per lane, 128 contiguous bytes per lane row of the wave.
                if (p.fuse_shade) shade_row4<PROJ>(p, x0 + 4 * seg, lr0 + j, tv, iv);
            }
        }
        if (p.exact_counter != nullptr && n_exact) atomicAdd(p.exact_counter, n_exact);
    }
#ifdef MRT_TIMELINE
    if (lane == 0) {
        const uint32_t slot = MRT_TL_SLOT;
        if (slot == 0) g_timeline_count = gridDim.x * gridDim.y * gridDim.z * (kBlockThreads / 64);
        if (slot < kTimelineMax) {
            TimelineEntry& e = g_timeline[slot];
            e.t_end = wall_clock64();
            e.hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID, all 32 bits
            e.xcc_id = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // HW_REG_XCC_ID
            e.block_x = blk_x; e.block_y = blk_y; e.block_z = blk_slice; e.wave = (uint32_t)wave;
            e.cycles = (unsigned long long)clock64() - e.cycles;
        }
    }
#endif
}

namespace {

// What a wave that owns a whole 32x32 pixel tile keeps for it (tile-cull and grid kernels): the
// closest hit of the tile's 1024 pixels in LDS as 64-bit (t, id) keys, a pixel list for the exact
// step, and the uv components of its columns and rows.  All of it belongs to this wave alone.
struct TileState {
    unsigned long long* keys;      // [32 * 32]
    uint16_t* list;                // [32 * 32]
    uint32_t* count;               // fill of the list
    float* tile_ux;                // [32]
    float* tile_uy;                // [32]
    uint32_t bx, by;               // this lane's 4x4 block inside the tile
    float xf, yf0;                 // ... and its first pixel in image coordinates
    int lane;
};

// One sphere that may touch the tile: its ellipse record f per pixel of the lane's 4x4 block,
// then the reference arithmetic over the pixels that pass.
template <int PROJ>
__device__ __forceinline__ void survivor_sphere(const RenderParams& p, uint32_t si, float4 f, const TileState& st,
                                                unsigned long long& n_exact, bool& recorded) {
    // wave-uniform loads land in scalar registers; VALU operands read from those issue at half rate
    const float fx = in_vgpr(f.x), fz = in_vgpr(f.z);
    uint32_t mask = 0;
    float dup = __builtin_fmaf(st.xf, fx, f.y);
#pragma unroll
    for (int i = 0; i < kLanePx; ++i) {
#pragma unroll
        for (int j = 0; j < kLanePx; ++j) {
            const float dvp = __builtin_fmaf(st.yf0 + (float)j, fz, f.w);
            if (__builtin_fmaf(dup, dup, __builtin_fmaf(dvp, dvp, -1.0f)) <= 0.0f) mask |= 1u << (j * kLanePx + i);
        }
        dup += fx;
    }
    if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) return;
    const uint32_t total = list_pixels(mask, st.bx, st.by, st.lane, st.list, st.count);
    exact_list<PROJ, false>(p, si, total, st.lane, st.list, st.tile_ux, st.tile_uy, st.keys, kWaveTile, n_exact, recorded);
}

// One triangle that may touch the tile, record (q0, q1, q2): 4x4 block reject at the best corner
// of the lane's block (via C'_k, see store_record in mrt_records.hip), the three edge functions
// per pixel, then the reference arithmetic over the pixels that pass.
template <int PROJ>
__device__ __forceinline__ void survivor_triangle(const RenderParams& p, uint32_t ti, float4 q0, float4 q1, float4 q2,
                                                  const TileState& st, unsigned long long& n_exact, bool& recorded) {
    const float a0 = in_vgpr(q0.x), b0 = in_vgpr(q0.y), a1 = in_vgpr(q0.w), b1 = in_vgpr(q1.x),
                a2 = in_vgpr(q1.z), b2 = in_vgpr(q1.w);
    const float m0 = __builtin_fmaf(a0, st.xf, __builtin_fmaf(b0, st.yf0, q2.y));
    const float m1 = __builtin_fmaf(a1, st.xf, __builtin_fmaf(b1, st.yf0, q2.z));
    const float m2 = __builtin_fmaf(a2, st.xf, __builtin_fmaf(b2, st.yf0, q2.w));
    const bool blk = __builtin_fminf(__builtin_fminf(m0, m1), m2) >= 0.0f;
    if (__builtin_amdgcn_ballot_w64(blk) == 0) return;
    const float c0 = __builtin_fmaf(a0, st.xf, q0.z);
    const float c1 = __builtin_fmaf(a1, st.xf, q1.y);
    const float c2 = __builtin_fmaf(a2, st.xf, q2.x);
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < kLanePx; ++j) {
        const float y = st.yf0 + (float)j;
        float v0 = __builtin_fmaf(b0, y, c0);
        float v1 = __builtin_fmaf(b1, y, c1);
        float v2 = __builtin_fmaf(b2, y, c2);
#pragma unroll
        for (int i = 0; i < kLanePx; ++i) {
            if (__builtin_fminf(__builtin_fminf(v0, v1), v2) >= 0.0f) mask |= 1u << (j * kLanePx + i);
            v0 += a0; v1 += a1; v2 += a2;
        }
    }
    if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) return;
    const uint32_t total = list_pixels(mask, st.bx, st.by, st.lane, st.list, st.count);
    exact_list<PROJ, true>(p, ti, total, st.lane, st.list, st.tile_ux, st.tile_uy, st.keys, kWaveTile, n_exact, recorded);
}

// Fills the tile's uv tables (gfxraytrace.hpp:850-852: the pixel grid is (x + 1, y + 1); rows
// past the end of this rank's rows repeat the last one, as everywhere), sets every key to
// "no hit" and empties the list.  LDS operations of one wave complete in order and no other wave
// touches these arrays, so a wave-level fence is all the synchronisation they need.
__device__ __forceinline__ void tile_begin(const RenderParams& p, const TileState& st, uint32_t wave_x0, uint32_t wave_lr0) {
    const uint32_t last_row = p.local_rows ? p.local_rows - 1 : 0;
    const uint32_t k = st.lane & 31;
    if (st.lane < 32) {
        st.tile_ux[k] = uv_component(p.left, p.right, (float)(wave_x0 + k) + 1.0f, p.xres_f);
    } else {
        const uint32_t lr = wave_lr0 + k;
        st.tile_uy[k] = uv_component(p.bottom, p.top, (float)image_row(p, lr < last_row ? lr : last_row) + 1.0f, p.yres_f);
    }
#pragma unroll
    for (int k2 = 0; k2 < kWaveTile * kWaveTile / 64; ++k2) st.keys[k2 * 64 + st.lane] = kNoHitKey;
    if (st.lane == 0) *st.count = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The lane's 4x4 block of closest hits out of the tile's keys: (t, id) planes when asked for
// (t_out != nullptr), shaded RGBA8 when SHADE (and p.fuse_shade) is set.
template <int PROJ, bool SHADE>
__device__ __forceinline__ void tile_end(const RenderParams& p, const TileState& st, uint32_t x0, uint32_t lr0,
                                         float* __restrict__ t_out, int32_t* __restrict__ id_out, uint32_t pitch) {
    // The atomics are complete before this wave reads the keys back.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < kLanePx; ++j) {
        unsigned long long k4[kLanePx];
#pragma unroll
        for (int i = 0; i < kLanePx; ++i) k4[i] = st.keys[(st.by + j) * kWaveTile + st.bx + i];
        int4 iv;
        float4 tv;
        iv.x = (int32_t)(uint32_t)k4[0]; iv.y = (int32_t)(uint32_t)k4[1];
        iv.z = (int32_t)(uint32_t)k4[2]; iv.w = (int32_t)(uint32_t)k4[3];
        tv.x = iv.x < 0 ? -1.0f : __uint_as_float((uint32_t)(k4[0] >> 32));      // intersect.t_ of a miss, kernel.cl:363
        tv.y = iv.y < 0 ? -1.0f : __uint_as_float((uint32_t)(k4[1] >> 32));
        tv.z = iv.z < 0 ? -1.0f : __uint_as_float((uint32_t)(k4[2] >> 32));
        tv.w = iv.w < 0 ? -1.0f : __uint_as_float((uint32_t)(k4[3] >> 32));
        if (t_out != nullptr) {
            const size_t off = (size_t)(lr0 + j) * pitch + x0;
            *reinterpret_cast<float4*>(t_out + off) = tv;
            *reinterpret_cast<int4*>(id_out + off) = iv;
        }
        if (SHADE && p.fuse_shade) shade_row4<PROJ>(p, x0, lr0 + j, tv, iv);
    }
}

// Four RGBA8 pixels of local row lr starting at column x0 (a multiple of 4): one aligned 16-byte
// store when p.fuse_shade == 2, single pixels inside the image otherwise.
__device__ __forceinline__ void store_row4(const RenderParams& p, uint32_t x0, uint32_t lr, const uint32_t px[4]) {
    if (!(lr < p.local_rows && x0 < p.xres)) return;
    uint32_t* dst = p.out + (size_t)lr * p.xres + x0;
    if (p.fuse_shade == 2) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(px[0], px[1], px[2], px[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (x0 + i < p.xres) dst[i] = px[i];
    }
}

// A tile nothing hit: background bytes for the lane's 4x4 block, "miss" in the planes if asked.
__device__ __forceinline__ void tile_background(const RenderParams& p, uint32_t bg, uint32_t x0, uint32_t lr0,
                                                float* __restrict__ t_out, int32_t* __restrict__ id_out, uint32_t pitch) {
    const uint32_t px[4] = {bg, bg, bg, bg};
#pragma unroll
    for (int j = 0; j < kLanePx; ++j) {
        if (t_out != nullptr) {
            const size_t off = (size_t)(lr0 + j) * pitch + x0;
            *reinterpret_cast<float4*>(t_out + off) = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);     // kernel.cl:363
            *reinterpret_cast<int4*>(id_out + off) = make_int4(-1, -1, -1, -1);
        }
        store_row4(p, x0, lr0 + j, px);
    }
}

// The end of a tile whose wave shades it: the pixels that were hit are listed (list_pixels) and
// shaded 64 at a time by whichever lane is free -- Blinn-Phong is several hundred dependent
// instructions, and a lane that walks its own 4x4 block would drag its 63 neighbours through them
// for every hit pixel it owns -- with the colour put where the key was (upper word 0: no hit has
// t == 0).  Then every lane stores its block: the colour where there is one, background elsewhere.
template <int PROJ>
__device__ __forceinline__ void tile_end_shaded(const RenderParams& p, const TileState& st, uint32_t bg,
                                                uint32_t wave_x0, uint32_t wave_lr0, uint32_t x0, uint32_t lr0,
                                                float* __restrict__ t_out, int32_t* __restrict__ id_out, uint32_t pitch) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < kLanePx; ++j) {
        int4 iv;
        float4 tv;
        int32_t ids[kLanePx];
        float ts[kLanePx];
#pragma unroll
        for (int i = 0; i < kLanePx; ++i) {
            const unsigned long long k = st.keys[(st.by + j) * kWaveTile + st.bx + i];
            ids[i] = (int32_t)(uint32_t)k;
            ts[i] = ids[i] < 0 ? -1.0f : __uint_as_float((uint32_t)(k >> 32));       // intersect.t_ of a miss, kernel.cl:363
            if (ids[i] >= 0 && x0 + i < p.xres && lr0 + j < p.local_rows) mask |= 1u << (j * kLanePx + i);
        }
        if (t_out != nullptr) {
            iv.x = ids[0]; iv.y = ids[1]; iv.z = ids[2]; iv.w = ids[3];
            tv.x = ts[0]; tv.y = ts[1]; tv.z = ts[2]; tv.w = ts[3];
            const size_t off = (size_t)(lr0 + j) * pitch + x0;
            *reinterpret_cast<float4*>(t_out + off) = tv;
            *reinterpret_cast<int4*>(id_out + off) = iv;
        }
    }
    const uint32_t total = list_pixels(mask, st.bx, st.by, st.lane, st.list, st.count);
    for (uint32_t b = 0; b < total; b += 64) {
        const uint32_t idx = b + (uint32_t)st.lane;
        if (idx < total) {
            const uint32_t px = st.list[idx];
            const unsigned long long k = st.keys[px];
            const uint32_t x = wave_x0 + (px & 31u), lr = wave_lr0 + (px >> 5);
            const uint32_t c = shade_pixel<PROJ>(p, x, image_row(p, lr), __uint_as_float((uint32_t)(k >> 32)),
                                                 (int32_t)(uint32_t)k);
            st.keys[px] = (unsigned long long)c;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < kLanePx; ++j) {
        uint32_t px[kLanePx];
#pragma unroll
        for (int i = 0; i < kLanePx; ++i) {
            const unsigned long long k = st.keys[(st.by + j) * kWaveTile + st.bx + i];
            px[i] = (uint32_t)(k >> 32) == 0u ? (uint32_t)k : bg;
        }
        store_row4(p, x0, lr0 + j, px);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Screen-space tile culling (mrt_accel_*, method MRT_ACCEL_TILE_CULL); never the
// brute-force path.
//
// Same tiles and records as trace_kernel, but the first rejection is done per
// wave tile with the lanes spread over PRIMITIVES: lane l tests primitive
// base + l against the wave's whole 32x32 tile (nearest point of the tile to the
// bounding circle / ellipse, and each edge function at its best tile corner) and
// a ballot gives the survivors.  A survivor is then tested per 4x4 lane block
// and per pixel with the same records; the pixels that pass are written, as a
// compact list, to LDS and the reference arithmetic runs over that list 64
// pixels at a time, whichever lane they belong to -- so a triangle that covers
// 40 pixels costs one pass with 40 lanes busy instead of up to 16 passes with
// two or three.  The closest hit of the tile's 1024 pixels lives in LDS as a
// 64-bit key (t bits << 32 | object id) updated with an atomic minimum: hits
// have t > 0, so the keys order like (t, id), and the smallest key is the hit
// the reference's loop keeps -- the smallest t, the lowest id among equals
// (strict "<" in object order, kernel.cl:326,344) -- in whatever order the
// pairs are evaluated.  Needs a tile's 32 rows to be consecutive image rows
// (band height a multiple of 32); the host falls back to trace_kernel otherwise.
// ---------------------------------------------------------------------------
template <int PROJ>
__global__ __launch_bounds__(kBlockThreads) void cull_trace_kernel(const RenderParams p,
                                                                  float* __restrict__ t_out,
                                                                  int32_t* __restrict__ id_out,
                                                                  uint32_t pitch) {
    constexpr int kWaves = kBlockThreads / 64, kTilePx = kWaveTile * kWaveTile;
    __shared__ unsigned long long s_key[kWaves][kTilePx];     // closest hit per pixel of the wave's tile
    __shared__ uint16_t s_list[kWaves][kTilePx];              // pixels one primitive still has to be tested on
    __shared__ float s_ux[kWaves][kWaveTile], s_uy[kWaves][kWaveTile];
    __shared__ uint32_t s_count[kWaves];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t wave_x0 = blockIdx.x * kBlockTileX + (wave & 1) * kWaveTile;
    const uint32_t wave_lr0 = blockIdx.y * kBlockTileY + (wave >> 1) * kWaveTile;
    if (!((wave_x0 < p.xres) && (wave_lr0 < p.local_rows))) return;   // no block-level barriers in this kernel
    TileState st;
    st.keys = s_key[wave]; st.list = s_list[wave]; st.count = &s_count[wave];
    st.tile_ux = s_ux[wave]; st.tile_uy = s_uy[wave];
    st.bx = (lane & 7) * kLanePx; st.by = (lane >> 3) * kLanePx;     // this lane's block inside the tile
    st.lane = lane;
    const uint32_t x0 = wave_x0 + st.bx;
    const uint32_t lr0 = wave_lr0 + st.by;
    // Tile and lane-block origins in image pixel coordinates.
    const float txf = (float)wave_x0;
    const float tyf = (float)image_row(p, wave_lr0);
    st.xf = (float)x0;
    st.yf0 = tyf + (float)st.by;
    tile_begin(p, st, wave_x0, wave_lr0);
    unsigned long long n_exact = 0;
    bool recorded = false;

    // ---- spheres ----
    for (uint32_t base = 0; base < p.n_sph; base += 64) {
        const float4 e = p.sph_filter[base + lane];          // padded to a multiple of 768
        const bool pass = tile_may_hit_sphere(e, txf, tyf) && (base + lane < p.n_sph);
        unsigned long long live = __builtin_amdgcn_ballot_w64(pass);
        while (live) {
            const uint32_t si = base + (uint32_t)__builtin_ctzll(live);
            live &= live - 1;
            survivor_sphere<PROJ>(p, si, p.sph_filter[si], st, n_exact, recorded);
        }
    }

    // ---- triangles ----
    for (uint32_t base = 0; base < p.n_tri; base += 64) {
        const uint32_t mine = base + lane;                    // records are padded to a multiple of 256
        const float4 c = p.tri_circle[mine];
        const bool pass = tile_may_hit_triangle(c, p.tri_filter + (size_t)mine * 3, txf, tyf);
        unsigned long long live = __builtin_amdgcn_ballot_w64(pass && mine < p.n_tri);
        while (live) {
            const uint32_t ti = base + (uint32_t)__builtin_ctzll(live);
            live &= live - 1;
            survivor_triangle<PROJ>(p, ti, p.tri_filter[(size_t)ti * 3 + 0], p.tri_filter[(size_t)ti * 3 + 1],
                                    p.tri_filter[(size_t)ti * 3 + 2], st, n_exact, recorded);
        }
    }

    tile_end<PROJ, false>(p, st, x0, lr0, t_out, id_out, pitch);       // shade_kernel takes the planes from here
    if (p.exact_counter != nullptr && n_exact) atomicAdd(p.exact_counter, n_exact);
}

// ---------------------------------------------------------------------------
// Screen-space uniform grid (mrt_accel_*, method MRT_ACCEL_GRID); never the
// brute-force path.
//
// The cells of the grid are the 32x32 pixel tiles of the image; per view,
// mrt_accel.hip bins every primitive into the tiles it may touch (the tile
// tests of mrt_tile_tests.hpp, run once) and leaves, per tile, a list of object
// ids: refs[cell_start[c] .. cell_start[c + 1]).  A wave owns a tile exactly as
// in cull_trace_kernel -- closest hits as (t, id) keys in LDS, survivors tested
// per 4x4 block and per pixel, the reference arithmetic over the pixels that
// pass -- but walks only its tile's list.  The prefilter records of a batch of
// 32 listed objects are fetched by all lanes at once and staged in LDS, so the
// per-object loop reads them as LDS broadcasts instead of waiting for one
// scalar load after another; the batches of a long list are dealt to the four
// waves of the block.  The order of a list is whatever the binning
// atomics gave: the keys make the winner independent of it.  The kernel shades
// its own pixels (no (t, id) planes unless asked for): a cell with an empty list,
// or one in which nothing was hit, is filled with the background colour; in the
// others the hit pixels are listed and shaded 64 at a time.  Needs a tile's 32
// rows to be consecutive image rows starting at a multiple of 32.
// ---------------------------------------------------------------------------
constexpr int kGridBatch = 32;

template <int PROJ>
__global__ __launch_bounds__(kBlockThreads) void grid_trace_kernel(const RenderParams p,
                                                                  const uint32_t* __restrict__ cell_start,
                                                                  const uint32_t* __restrict__ refs,
                                                                  uint32_t grid_tiles_x,
                                                                  float* __restrict__ t_out,
                                                                  int32_t* __restrict__ id_out,
                                                                  uint32_t pitch) {
    constexpr int kWaves = kBlockThreads / 64, kTilePx = kWaveTile * kWaveTile;
    __shared__ unsigned long long s_key[kWaves][kTilePx];
    __shared__ uint16_t s_list[kWaves][kTilePx];
    __shared__ float4 s_batch[kWaves][kGridBatch * 3];        // records of the objects being walked
    __shared__ float s_ux[kWaves][kWaveTile], s_uy[kWaves][kWaveTile];
    __shared__ uint32_t s_count[kWaves];

    __shared__ uint32_t s_begin[kWaves], s_end[kWaves], s_hit[kWaves];
    static_assert(kWaves == 4, "the batches of a list are dealt to the four waves of a block");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // Tile t of the block's 2x2, the cell it is, and the wave that owns it (wave t): sets it up,
    // shades and stores it.  The lists are walked by all four waves (see below).
    const uint32_t wave_x0 = blockIdx.x * kBlockTileX + (wave & 1) * kWaveTile;
    const uint32_t wave_lr0 = blockIdx.y * kBlockTileY + (wave >> 1) * kWaveTile;
    const bool own_active = (wave_x0 < p.xres) && (wave_lr0 < p.local_rows);
    TileState own;
    own.keys = s_key[wave]; own.list = s_list[wave]; own.count = &s_count[wave];
    own.tile_ux = s_ux[wave]; own.tile_uy = s_uy[wave];
    own.bx = (lane & 7) * kLanePx; own.by = (lane >> 3) * kLanePx;
    own.lane = lane;
    own.xf = 0.0f; own.yf0 = 0.0f;
    const uint32_t x0 = wave_x0 + own.bx;
    const uint32_t lr0 = wave_lr0 + own.by;
    uint32_t begin = 0, end = 0;
    if (own_active) {
        const uint32_t cell = (image_row(p, wave_lr0) / kWaveTile) * grid_tiles_x + wave_x0 / kWaveTile;
        begin = cell_start[cell];
        end = cell_start[cell + 1];
        if (begin != end) tile_begin(p, own, wave_x0, wave_lr0);
    }
    if (lane == 0) {
        s_begin[wave] = begin;
        s_end[wave] = end;
        s_hit[wave] = 0u;
        s_count[wave] = 0u;
    }
    __syncthreads();

    // A list is cut into batches of 32 objects and batch b of tile t goes to wave (t + b) % 4: a
    // short list stays with the tile's own wave, a long one -- the pole of a mesh where a hundred
    // thin triangles meet in one cell, whatever the resolution -- is shared by the block instead
    // of holding the frame up.  The tile's keys take hits from any wave (LDS atomics).
    unsigned long long n_exact = 0;
    float4* batch = s_batch[wave];
    for (int q = 0; q < kWaves; ++q) {
        const int t = (wave + q) & 3;
        const uint32_t t_begin = s_begin[t], t_end = s_end[t];
        const uint32_t first = t_begin + (uint32_t)((wave - t) & 3) * kGridBatch;
        if (first >= t_end) continue;
        const uint32_t t_x0 = blockIdx.x * kBlockTileX + (t & 1) * kWaveTile;
        const uint32_t t_lr0 = blockIdx.y * kBlockTileY + (t >> 1) * kWaveTile;
        TileState st = own;
        st.keys = s_key[t]; st.tile_ux = s_ux[t]; st.tile_uy = s_uy[t];
        st.xf = (float)(t_x0 + own.bx);
        st.yf0 = (float)image_row(p, t_lr0) + (float)own.by;
        bool recorded = false;
        for (uint32_t base = first; base < t_end; base += kWaves * kGridBatch) {
            const uint32_t n = (t_end - base) < (uint32_t)kGridBatch ? (t_end - base) : (uint32_t)kGridBatch;
            const uint32_t ref = (uint32_t)lane < n ? refs[base + lane] : 0u;
            // 96 record slots, two per lane: slot s is part s % 3 of listed object s / 3.  A sphere
            // has one part (its ellipse), a triangle three (its edge record).
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = lane + 64 * h;
                const int k = s / 3, part = s - 3 * k;
                const uint32_t obj = (uint32_t)__builtin_amdgcn_ds_bpermute(k << 2, (int)ref);
                if (s < kGridBatch * 3 && (uint32_t)k < n) {
                    if (obj < p.n_sph) {
                        if (part == 0) batch[s] = p.sph_filter[obj];
                    } else {
                        batch[s] = p.tri_filter[(size_t)(obj - p.n_sph) * 3 + part];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t obj = (uint32_t)__builtin_amdgcn_readlane((int)ref, (int)k);
                if (obj < p.n_sph)
                    survivor_sphere<PROJ>(p, obj, batch[k * 3], st, n_exact, recorded);
                else
                    survivor_triangle<PROJ>(p, obj - p.n_sph, batch[k * 3 + 0], batch[k * 3 + 1], batch[k * 3 + 2], st,
                                            n_exact, recorded);
            }
            // the next batch overwrites the records only after the reads above
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (recorded && lane == 0) s_hit[t] = 1u;
    }
    if (p.exact_counter != nullptr && n_exact) atomicAdd(p.exact_counter, n_exact);
    __syncthreads();

    if (!own_active) return;
    const uint32_t bg = shade_pixel<PROJ>(p, 0u, 0u, -1.0f, -1);       // the colour of a miss
    if (begin == end || s_hit[wave] == 0u)      // nothing listed for this cell, or nothing hit
        tile_background(p, bg, x0, lr0, t_out, id_out, pitch);
    else if (p.fuse_shade)
        tile_end_shaded<PROJ>(p, own, bg, wave_x0, wave_lr0, x0, lr0, t_out, id_out, pitch);
    else
        tile_end<PROJ, false>(p, own, x0, lr0, t_out, id_out, pitch);
}

// ---------------------------------------------------------------------------
// MRT_FLAG_FULL_TEST: the determinants of the 3x3 solve for EVERY (pixel,
// primitive) pair -- no screen-space rejection of any kind.  (Not a general-ray
// kernel: prep_full_records folds the origin all rays of a pinhole frame
// share, or the direction all rays of an orthographic one share, into the
// per-view records.)
//
// tri_intersect (kernel.cl:197-235) solves by Cramer's rule: four 3x3
// determinants det, Nb, Ng, Nt, then beta = Nb/det, gamma = Ng/det, t = Nt/det
// and the range tests.  With c = s2 x d and k = sol x d (s1 = a - b, s2 = a - c,
// sol = a - origin, d the ray direction):
//     det = s1 . c      Nb = sol . c      Ng = s1 . k      Nt = s1 . (s2 x sol)
// This kernel evaluates, for every pair, both cross products and the three dot
// products Nb, A = det - Nb = (s1 - sol) . c and Ng with fused multiply-adds
// (mul + fma per cross component, three fma per dot product: 21 operations),
// then W = A - Ng = det - Nb - Ng.  A hit needs Nb/det >= 0, Ng/det >= 0,
// (Nb + Ng)/det <= 1 and t = Nt/det > 0; the sign s that det has at a hit is a
// property of the triangle (the sign of Nt for the pinhole camera, where Nt does
// not depend on the pixel; of det itself for the orthographic one), so a pair
// whose s Nb, s Ng or s W is negative cannot be accepted.
//
// The fused values differ from the reference's unfused ones by rounding.  Both
// are within 5 ulp-units of the sum of the absolute values of their six
// products (P below) of the real-number determinant, so they differ by at
// most 10 u P, u = 2^-24.  prep_full_records bounds P per triangle over all
// pixels, takes 32 u P as the margin, adds the band that keeps a quotient
// which would round to -0 (the reference's "< 0" lets it through) and an
// absolute term for products in the denormal range, and the margin enters the
// kernel as the starting value of each fma chain -- at no cost per pair.  So
// the verdict "min(Nb, Ng, W) < 0" (for s = +1; "max > 0" for s = -1) is only
// ever true for pairs the reference arithmetic rejects; every other pair (about
// 0.05 % on the benchmark frames) runs tri_exact from the original vertices,
// which decides, in registers, against the closest hit so far.  Triangles
// whose margins cannot be bounded (overflowing or non-finite products) get a
// record every pixel passes.  Spheres: the discriminant of circle_intersect
// (kernel.cl:248-254) for every pair; sqrt and division only where it is not
// negative.
//
// A lane owns 4x2 pixels, a wave 32x16, a block 64x32; the closest hit stays
// in registers.  Records (4 x float4 per triangle, made by prep_full_records)
// are staged through LDS in chunks of 256 triangles (16 KiB).
// ---------------------------------------------------------------------------
constexpr int kFullChunkFloat4 = kChunkTris * kFullRecFloat4;

namespace {

struct M3 { float m0, m1, m2; };
// The three 2x2 minors determinant3x3 (kernel.cl:149-155) forms from its second
// and third argument, and its expansion along the first: det3(a, b, c) ==
// expand(a, minors(b, c)) operation for operation.
__device__ __forceinline__ M3 minors(F3 b, F3 c) {
    M3 r;
    r.m0 = b.y * c.z - b.z * c.y;
    r.m1 = b.x * c.z - b.z * c.x;
    r.m2 = b.x * c.y - b.y * c.x;
    return r;
}
__device__ __forceinline__ float expand(F3 a, M3 m) {
    float det = a.x * m.m0;
    det -= a.y * m.m1;
    det += a.z * m.m2;
    return det;
}

// a x b with one rounded product and one fused multiply-add per component.
__device__ __forceinline__ F3 cross_fused(F3 a, F3 b) {
    return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)),
               __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
               __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
// a . b + start as a chain of three fused multiply-adds.
__device__ __forceinline__ float dot_fused(F3 a, F3 b, float start) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, __builtin_fmaf(a.x, b.x, start)));
}

// Sum of the absolute values of the six products of det3(a, b, c), for vectors of magnitudes.
__device__ __forceinline__ double permabs(double ax, double ay, double az, double bx, double by, double bz,
                                          double cx, double cy, double cz) {
    return ax * (by * cz + bz * cy) + ay * (bx * cz + bz * cx) + az * (bx * cy + by * cx);
}
// x as a float that is not below it.
__device__ __forceinline__ float float_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + 1u);     // x > 0 here: next float up
    return f;
}

}  // namespace

// Per-view triangle records of full_trace_kernel, four float4 each.
//   pinhole:       (s2 | Mb) (sol | Ma) (s1 - sol | Mg) (s1 | -)     sol = a - eye
//   orthographic:  (s2 x d | Mb) (a | D + Ma) (s1 | Mg) (s1 x s2 | Mt)   d = -w, D = det
// with s1 = a - b, s2 = a - c (kernel.cl:204-206) and M* the signed margins
// described above (sign s of the group).  The kernel forms, per pixel,
//   pinhole:       Nb = sol . (s2 x d) + Mb, A = (s1 - sol) . (s2 x d) + Ma, Ng = s1 . (sol x d) + Mg, W = A - Ng
//   orthographic:  sol = a - origin, Nb = sol . (s2 x d) + Mb, Ng = s1 . (sol x d) + Mg,
//                  W = (D + Ma) - Nb - Ng, Nt = sol . (s1 x s2) + Mt
//
// A hit needs t = Nt/det > 0, so the sign det has at a hit is a property of the
// triangle: the sign of Nt for the pinhole camera (Nt does not depend on the
// pixel), of det for the orthographic one (det does not).  The records are
// written in two groups -- sign +1 from the front of the buffer, sign -1 from
// its end (at n_tri - 1 downwards), counts[0] and counts[1] of them -- so that
// the kernel runs each group through straight-line code made for that sign.
// Triangles with a zero or NaN sign can never be accepted (t would be 0, NaN or
// infinite) and get no record.  The order inside a group is whatever the
// atomics give; the kernel breaks ties in t by object id, which is what
// processing in object order with a strict "<" amounts to (kernel.cl:326).
template <int PROJ>
__global__ void prep_full_records(const RenderParams p, float4* __restrict__ rec, uint32_t* __restrict__ ids,
                                  uint32_t* __restrict__ counts) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.n_tri) return;
    const float4 va = p.tri_exact[(size_t)k * 3 + 0];
    const float4 vb = p.tri_exact[(size_t)k * 3 + 1];
    const float4 vc = p.tri_exact[(size_t)k * 3 + 2];
    const F3 a = mk3(va.x, va.y, va.z);
    const F3 s1 = sub3(a, mk3(vb.x, vb.y, vb.z));
    const F3 s2 = sub3(a, mk3(vc.x, vc.y, vc.z));
    const double u32 = 32.0 / 16777216.0;          // 32 u: the rounding budget per determinant
    const double dx = p.ray_abs[0], dy = p.ray_abs[1], dz = p.ray_abs[2];
    const double a1x = fabs((double)s1.x), a1y = fabs((double)s1.y), a1z = fabs((double)s1.z);
    const double a2x = fabs((double)s2.x), a2y = fabs((double)s2.y), a2z = fabs((double)s2.z);
    float4 r0, r1, r2, r3;
    float sign_src;
    double p_all, mb, ma, mg, mt = 0.0;
    if (PROJ == MRT_PROJ_PERSP) {
        const F3 sol = sub3(a, mk3(p.eye[0], p.eye[1], p.eye[2]));
        const F3 e = sub3(s1, sol);
        sign_src = det3(s1, s2, sol);              // Nt, the reference's own value
        const double sx = fabs((double)sol.x), sy = fabs((double)sol.y), sz = fabs((double)sol.z);
        const double p_det = permabs(a1x, a1y, a1z, a2x, a2y, a2z, dx, dy, dz);
        const double p_b = permabs(sx, sy, sz, a2x, a2y, a2z, dx, dy, dz);
        const double p_g = permabs(a1x, a1y, a1z, sx, sy, sz, dx, dy, dz);
        const double big = fmax(fmax(fmax(a1x, a1y), fmax(a1z, a2x)), fmax(fmax(a2y, a2z), fmax(fmax(sx, sy), fmax(sz, fmax(dx, fmax(dy, dz))))));
        // |quotient| < 2^-149 rounds to -0, which passes "< 0"; products in the denormal range are
        // off by 2^-150 absolutely, scaled by the factor they meet next
        const double band = p_det * 0x1p-100 + 64.0 * 0x1p-149 * (1.0 + big);
        p_all = p_det + p_b + p_g;
        mb = u32 * p_b + band;
        mg = u32 * p_g + band;
        ma = mg + u32 * p_all + band;              // W = A - Ng carries Ng's margin with the other sign
        r0 = make_float4(s2.x, s2.y, s2.z, 0.f);
        r1 = make_float4(sol.x, sol.y, sol.z, 0.f);
        r2 = make_float4(e.x, e.y, e.z, 0.f);
        r3 = make_float4(s1.x, s1.y, s1.z, 0.f);
    } else {
        const F3 d = mk3(-p.w[0], -p.w[1], -p.w[2]);
        const F3 c = cross_fused(s2, d);
        const F3 n = cross_fused(s1, s2);
        const float det = expand(s1, minors(s2, d));   // det3(s1, s2, d), the reference's own value
        sign_src = det;
        const double sx = fabs((double)a.x) + p.org_abs[0], sy = fabs((double)a.y) + p.org_abs[1],
                     sz = fabs((double)a.z) + p.org_abs[2];
        const double p_det = permabs(a1x, a1y, a1z, a2x, a2y, a2z, dx, dy, dz);
        const double p_b = permabs(sx, sy, sz, a2x, a2y, a2z, dx, dy, dz);
        const double p_g = permabs(a1x, a1y, a1z, sx, sy, sz, dx, dy, dz);
        const double p_t = permabs(a1x, a1y, a1z, a2x, a2y, a2z, sx, sy, sz);
        const double big = fmax(fmax(fmax(a1x, a1y), fmax(a1z, a2x)), fmax(fmax(a2y, a2z), fmax(fmax(sx, sy), fmax(sz, fmax(dx, fmax(dy, dz))))));
        const double band = p_det * 0x1p-100 + 64.0 * 0x1p-149 * (1.0 + big);
        p_all = p_det + p_b + p_g + p_t;
        mb = u32 * p_b + band;
        mg = u32 * p_g + band;
        mt = u32 * p_t + band;
        ma = mb + mg + u32 * p_all + band;         // W = D - Nb - Ng carries both margins with the other sign
        r0 = make_float4(c.x, c.y, c.z, 0.f);
        r1 = make_float4(a.x, a.y, a.z, det);
        r2 = make_float4(s1.x, s1.y, s1.z, 0.f);
        r3 = make_float4(n.x, n.y, n.z, 0.f);
    }
    float sg;
    uint32_t slot;
    if (sign_src > 0.0f) { sg = 1.0f; slot = atomicAdd(&counts[0], 1u); }
    else if (sign_src < 0.0f) { sg = -1.0f; slot = p.n_tri - 1u - atomicAdd(&counts[1], 1u); }
    else return;
    if (p_all < 1e36 && ma < 1e36) {               // false for NaN too
        // Records of the s = -1 group carry s2 and s1 (pinhole) or s2 x d, s1 and s1 x s2
        // (orthographic) negated: every quantity the kernel forms is then s times the
        // determinant it stands for, and one verdict -- "none of them negative" -- serves both
        // groups.  Negation is exact and rounding is symmetric, so the margins are unaffected.
        if (sg < 0.0f) {
            if (PROJ == MRT_PROJ_PERSP) {
                r0.x = -r0.x; r0.y = -r0.y; r0.z = -r0.z;      // s2
            } else {
                r0.x = -r0.x; r0.y = -r0.y; r0.z = -r0.z;      // s2 x d
                r2.x = -r2.x; r2.y = -r2.y; r2.z = -r2.z;      // s1
            }
            r3.x = -r3.x; r3.y = -r3.y; r3.z = -r3.z;          // s1 (pinhole), s1 x s2 (orthographic)
        }
        r0.w = float_up(mb);
        r2.w = float_up(mg);
        if (PROJ == MRT_PROJ_PERSP) {
            r1.w = float_up(ma);
        } else {
            r1.w = float_up(fabs((double)r1.w) + ma);          // s D + Ma, rounded up
            r3.w = float_up(mt);
        }
    } else {
        // The margins cannot be bounded: a record every pixel passes (all products zero, so
        // Nb = Ng = 1, W = 3 or 2, Nt = 1); the exact arithmetic decides every pair.
        r0 = make_float4(0.f, 0.f, 0.f, 1.0f);
        r1 = make_float4(0.f, 0.f, 0.f, 4.0f);
        r2 = make_float4(0.f, 0.f, 0.f, 1.0f);
        r3 = make_float4(0.f, 0.f, 0.f, 1.0f);
    }
    rec[(size_t)slot * kFullRecFloat4 + 0] = r0;
    rec[(size_t)slot * kFullRecFloat4 + 1] = r1;
    rec[(size_t)slot * kFullRecFloat4 + 2] = r2;
    rec[(size_t)slot * kFullRecFloat4 + 3] = r3;
    ids[slot] = k;
}

template <int PROJ>
__global__ __launch_bounds__(kBlockThreads, MRT_FULL_WAVES_PER_SIMD) void full_trace_kernel(const RenderParams p,
                                                                  const float4* __restrict__ full_rec,
                                                                  const uint32_t* __restrict__ full_ids,
                                                                  const uint32_t* __restrict__ class_counts,
                                                                  unsigned long long* __restrict__ key_plane,
                                                                  float* __restrict__ t_out,
                                                                  int32_t* __restrict__ id_out, uint32_t pitch) {
    __shared__ float4 s_rec[kFullChunkFloat4];
    __shared__ uint32_t s_ids[kChunkTris];
    static_assert(kChunkTris == kBlockThreads, "one triangle id per thread when a chunk is staged");
    static_assert(kChunkFloat4 <= kFullChunkFloat4, "the sphere chunks reuse the record buffer");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t wave_x0 = blockIdx.x * kBlockTileX + (wave & 1) * kWaveTile;
    const uint32_t wave_lr0 = blockIdx.y * kFullBlockH + (wave >> 1) * kFullWaveH;
    const bool wave_active = (wave_x0 < p.xres) && (wave_lr0 < p.local_rows);
    const uint32_t x0 = wave_x0 + (lane & 7) * kFullPxX;
    const uint32_t lr0 = wave_lr0 + (lane >> 3) * kFullPxY;

    // Per pixel: the part of the primary ray that depends on it (direction for the
    // pinhole camera, origin for the orthographic one) and the closest hit so far.
    F3 ray[kFullSlots];
    float best_t[kFullSlots];
    int32_t best_id[kFullSlots];
    F3 shared_part = mk3(0.f, 0.f, 0.f);   // the other half of the ray, the same for every pixel
    {
        const uint32_t last_row = p.local_rows ? p.local_rows - 1 : 0;
#pragma unroll
        for (int j = 0; j < kFullPxY; ++j) {
            const uint32_t lr = lr0 + j;
            const uint32_t y = image_row(p, lr < last_row ? lr : last_row);
            const float uy = uv_component(p.bottom, p.top, (float)y + 1.0f, p.yres_f);
#pragma unroll
            for (int i = 0; i < kFullPxX; ++i) {
                const float ux = uv_component(p.left, p.right, (float)(x0 + i) + 1.0f, p.xres_f);
                F3 o, d;
                pixel_ray<PROJ>(p, ux, uy, o, d);
                ray[j * kFullPxX + i] = (PROJ == MRT_PROJ_PERSP) ? d : o;
                shared_part = (PROJ == MRT_PROJ_PERSP) ? o : d;
                best_t[j * kFullPxX + i] = FLT_MAX;               // gfxraytrace.hpp:813
                best_id[j * kFullPxX + i] = -1;
            }
        }
    }

    shared_part = mk3(in_vgpr(shared_part.x), in_vgpr(shared_part.y), in_vgpr(shared_part.z));

    // A frame of few waves (a 4K frame is four rounds of them) is cut into p.prim_slices grid
    // layers: layer z takes every prim_slices-th chunk of records, starting at chunk z, and the
    // layers' closest hits meet in the key plane under atomicMin (see kNoHitKey), which
    // shade_keys_kernel then decodes and shades.  One layer keeps everything in registers.
    const uint32_t n_layers = p.prim_slices > 1 ? p.prim_slices : 1u;
    const uint32_t layer = n_layers > 1 ? blockIdx.z : 0u;

    // ---- spheres: object ids 0 .. n_sph-1 ----
    for (uint32_t base = layer * kChunkFloat4; base < p.n_sph; base += n_layers * kChunkFloat4) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kChunkFloat4 / kBlockThreads; ++k) {
            const uint32_t idx = base + tid + k * kBlockThreads;
            s_rec[tid + k * kBlockThreads] = idx < p.n_sph ? p.sph[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (!wave_active) continue;
        const uint32_t count = (p.n_sph - base) < (uint32_t)kChunkFloat4 ? (p.n_sph - base) : (uint32_t)kChunkFloat4;
        for (uint32_t g = 0; g < count; ++g) {
            const float4 sp = s_rec[g];
            const F3 center = mk3(sp.x, sp.y, sp.z);
            // circle_intersect gives up only when the discriminant compares below zero
            // (kernel.cl:256); a NaN discriminant (NaN radius or centre) goes on, so the test
            // is "not less than" and not a maximum, which would lose the NaN.
            bool cand = false;
#pragma unroll
            for (int s = 0; s < kFullSlots; ++s) {
                const F3 o = (PROJ == MRT_PROJ_PERSP) ? shared_part : ray[s];
                const F3 d = (PROJ == MRT_PROJ_PERSP) ? ray[s] : shared_part;
                float a, b;
                cand |= !(sphere_disc(center, sp.w, o, d, a, b) < 0.0f);
            }
            if (__builtin_amdgcn_ballot_w64(cand) == 0) continue;
            if (cand) {
#pragma unroll
                for (int s = 0; s < kFullSlots; ++s) {
                    const F3 o = (PROJ == MRT_PROJ_PERSP) ? shared_part : ray[s];
                    const F3 d = (PROJ == MRT_PROJ_PERSP) ? ray[s] : shared_part;
                    float t;
                    bool two;
                    if (sphere_exact(center, sp.w, o, d, best_t[s], 0.0f, t, two) && t > 0.0f && t < best_t[s]) {
                        best_t[s] = t;
                        best_id[s] = (int32_t)(base + g);
                    }
                }
            }
        }
    }

    // ---- triangles: object ids n_sph .. n_sph+n_tri-1, one sign group after the other ----
    auto sign_group = [&](uint32_t first, uint32_t n) {
        for (uint32_t base = layer * kChunkTris; base < n; base += n_layers * kChunkTris) {
            __syncthreads();
            // the buffer has a chunk of slack beyond the last record
#pragma unroll
            for (int k = 0; k < kFullChunkFloat4 / kBlockThreads; ++k)
                s_rec[tid + k * kBlockThreads] = full_rec[(size_t)(first + base) * kFullRecFloat4 + tid + k * kBlockThreads];
            s_ids[tid] = full_ids[first + base + tid];
            __syncthreads();
            if (!wave_active) continue;
            const uint32_t count = (n - base) < (uint32_t)kChunkTris ? (n - base) : (uint32_t)kChunkTris;
            for (uint32_t g = 0; g < count; ++g) {
                const float4 q0 = s_rec[g * kFullRecFloat4 + 0];
                const float4 q1 = s_rec[g * kFullRecFloat4 + 1];
                const float4 q2 = s_rec[g * kFullRecFloat4 + 2];
                const float4 q3 = s_rec[g * kFullRecFloat4 + 3];
                // The verdict works on sign bits (v_or3_b32 per pixel and an and-tree over the lane's
                // pixels; v_min3_f32 + v_max_f32 per pixel took the same time, with two code paths
                // by sign group): a pair stays a candidate when none of its quantities
                // has the sign bit set, and `neg` keeps the sign bit only while every pixel of the
                // lane has been ruled out.  A quantity that could belong to an accepted pair is
                // at least half its (positive) margin, so -0 never stands for "not negative".
                uint32_t neg = 0xFFFFFFFFu;              // only the sign bit is ever looked at
                if (PROJ == MRT_PROJ_PERSP) {
                    const F3 s2 = mk3(q0.x, q0.y, q0.z), sol = mk3(q1.x, q1.y, q1.z);
                    const F3 e = mk3(q2.x, q2.y, q2.z), s1 = mk3(q3.x, q3.y, q3.z);
#pragma unroll
                    for (int s = 0; s < kFullSlots; ++s) {
                        const F3 c = cross_fused(s2, ray[s]);
                        const F3 kk = cross_fused(sol, ray[s]);
                        const float nb = dot_fused(sol, c, q0.w);
                        const float aa = dot_fused(e, c, q1.w);
                        const float ng = dot_fused(s1, kk, q2.w);
                        const float w = aa - ng;
                        neg &= __float_as_uint(nb) | __float_as_uint(ng) | __float_as_uint(w);
                    }
                } else {
                    const F3 c = mk3(q0.x, q0.y, q0.z), a = mk3(q1.x, q1.y, q1.z);
                    const F3 s1 = mk3(q2.x, q2.y, q2.z), nrm = mk3(q3.x, q3.y, q3.z);
#pragma unroll
                    for (int s = 0; s < kFullSlots; ++s) {
                        const F3 sol = sub3(a, ray[s]);                   // as tri_intersect forms it
                        const F3 kk = cross_fused(sol, shared_part);
                        const float nb = dot_fused(sol, c, q0.w);
                        const float ng = dot_fused(s1, kk, q2.w);
                        const float nt = dot_fused(sol, nrm, q3.w);
                        const float w = (q1.w - nb) - ng;
                        neg &= __float_as_uint(nb) | __float_as_uint(ng) | __float_as_uint(w) | __float_as_uint(nt);
                    }
                }
                // a quantity that is NaN (non-finite rays) may go either way, no such pair is a hit
                // (the compare's lane mask itself is the vote: no round trip through a VGPR)
                const uint64_t cand_mask = __builtin_amdgcn_sicmp((int32_t)neg, -1, 38 /* ICMP_SGT */);
                if (cand_mask == 0) continue;
                const bool cand = (int32_t)neg > -1;
                if (cand) {
                    const uint32_t ti = __builtin_amdgcn_readfirstlane(s_ids[g]);
                    const float4 va = p.tri_exact[(size_t)ti * 3 + 0];
                    const float4 vb = p.tri_exact[(size_t)ti * 3 + 1];
                    const float4 vc = p.tri_exact[(size_t)ti * 3 + 2];
                    const F3 A = mk3(va.x, va.y, va.z), B = mk3(vb.x, vb.y, vb.z), C = mk3(vc.x, vc.y, vc.z);
                    const int32_t obj_id = (int32_t)(p.n_sph + ti);
#pragma unroll
                    for (int s = 0; s < kFullSlots; ++s) {
                        const F3 o = (PROJ == MRT_PROJ_PERSP) ? shared_part : ray[s];
                        const F3 d = (PROJ == MRT_PROJ_PERSP) ? ray[s] : shared_part;
                        float t;
                        // The groups are not in object order: equal t goes to the lower id, which
                        // is what the reference's strict "<" in object order gives (kernel.cl:326).
                        if (tri_exact(A, B, C, o, d, best_t[s], 0.0f, t) && t > 0.0f &&
                            (t < best_t[s] || (t == best_t[s] && obj_id < best_id[s]))) {
                            best_t[s] = t;
                            best_id[s] = obj_id;
                        }
                    }
                }
            }
        }
    };
    const uint32_t n_pos = class_counts[0], n_neg = class_counts[1];
    sign_group(0u, n_pos);
    sign_group(p.n_tri - n_neg, n_neg);

    if (wave_active && n_layers > 1) {
        // this layer's closest hits into the key plane (padded to whole tiles: no bounds checks)
        bool rec = false;
#pragma unroll
        for (int s = 0; s < kFullSlots; ++s) {
            if (best_id[s] >= 0) {
                atomicMin(&key_plane[(size_t)(lr0 + s / kFullPxX) * pitch + x0 + s % kFullPxX],
                          ((unsigned long long)__float_as_uint(best_t[s]) << 32) | (uint32_t)best_id[s]);
                rec = true;
            }
        }
        if (p.tile_flags != nullptr && __builtin_amdgcn_ballot_w64(rec) != 0 && lane == 0)
            p.tile_flags[(wave_lr0 >> p.tile_y_log2) * p.tiles_x + (wave_x0 >> p.tile_x_log2)] = p.frame_tag;
        if (p.exact_counter != nullptr && layer == 0)
            atomicAdd(p.exact_counter, (unsigned long long)kFullSlots * ((unsigned long long)p.n_sph + p.n_tri));
    } else if (wave_active) {
#pragma unroll
        for (int j = 0; j < kFullPxY; ++j) {
            const size_t off = (size_t)(lr0 + j) * pitch + x0;
            float4 tv;
            int4 iv;
            iv.x = best_id[j * 4 + 0]; iv.y = best_id[j * 4 + 1]; iv.z = best_id[j * 4 + 2]; iv.w = best_id[j * 4 + 3];
            tv.x = iv.x < 0 ? -1.0f : best_t[j * 4 + 0];          // intersect.t_ of a miss, kernel.cl:363
            tv.y = iv.y < 0 ? -1.0f : best_t[j * 4 + 1];
            tv.z = iv.z < 0 ? -1.0f : best_t[j * 4 + 2];
            tv.w = iv.w < 0 ? -1.0f : best_t[j * 4 + 3];
            if (t_out != nullptr) {
                *reinterpret_cast<float4*>(t_out + off) = tv;
                *reinterpret_cast<int4*>(id_out + off) = iv;
            }
            if (p.fuse_shade) shade_row4<PROJ>(p, x0, lr0 + j, tv, iv);
        }
        if (p.exact_counter != nullptr)
            atomicAdd(p.exact_counter, (unsigned long long)kFullSlots * ((unsigned long long)p.n_sph + p.n_tri));
    }
}

// The epilogue of trace_kernel as a launch of its own, for frames traced in several slices: one
// thread per pixel of the padded plane decodes the key, writes the (t, id) planes if asked to, and
// shades.  A key nobody touched is all ones: id -1, a miss.  With tile flags (see RenderParams)
// the keys of a tile no wave recorded a hit in are not read at all, and every key that was
// written -- padding pixels included, whose rows repeat the last image row -- is put back to all
// ones, which leaves the plane ready for the next frame without a memset.
template <int PROJ>
__global__ __launch_bounds__(256) void shade_keys_kernel(const RenderParams p,
                                                        unsigned long long* __restrict__ key_plane,
                                                        float* __restrict__ t_out, int32_t* __restrict__ id_out,
                                                        uint32_t pitch) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t lr = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool inside = x < p.xres && lr < p.local_rows;
    if (p.tile_flags == nullptr && !inside) return;
    const size_t src = (size_t)lr * pitch + x;
    unsigned long long key = ~0ull;
    if (p.tile_flags == nullptr) {
        key = key_plane[src];
    } else if (p.tile_flags[(lr >> p.tile_y_log2) * p.tiles_x + (x >> p.tile_x_log2)] == p.frame_tag) {
        key = key_plane[src];
        if (key != ~0ull) key_plane[src] = ~0ull;
    }
    if (!inside) return;
    const int32_t id = (int32_t)(uint32_t)key;
    const float t = id < 0 ? -1.0f : __uint_as_float((uint32_t)(key >> 32));     // kernel.cl:363
    if (t_out != nullptr) {
        t_out[src] = t;
        id_out[src] = id;
    }
    p.out[(size_t)lr * p.xres + x] = shade_pixel<PROJ>(p, x, image_row(p, lr), t, id);
}

// Shading as a launch of its own, one thread per pixel (a wave writes 256 contiguous bytes of
// RGBA8), for cull_trace_kernel: at its three waves per SIMD the long dependent chains of
// phong_light (normalise, pow, divide) are better hidden by a kernel that needs few registers.
template <int PROJ>
__global__ __launch_bounds__(256) void shade_kernel(const RenderParams p,
                                                   const float* __restrict__ t_in,
                                                   const int32_t* __restrict__ id_in,
                                                   uint32_t pitch) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);
    const uint32_t lr = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.xres || lr >= p.local_rows) return;
    const size_t src = (size_t)lr * pitch + x;
    p.out[(size_t)lr * p.xres + x] = shade_pixel<PROJ>(p, x, image_row(p, lr), t_in[src], id_in[src]);
}

template __global__ void trace_kernel<MRT_PROJ_ORTHO, false, 4, 4, false>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_ORTHO, false, 4, 4, true>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP, false, 4, 4, false>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP, false, 4, 4, true>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_ORTHO, false, TraceWide::lane_x, TraceWide::lane_y, false>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_ORTHO, false, TraceWide::lane_x, TraceWide::lane_y, true>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP, false, TraceWide::lane_x, TraceWide::lane_y, false>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP, false, TraceWide::lane_x, TraceWide::lane_y, true>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_ORTHO, true, 4, 4, false>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_ORTHO, true, 4, 4, true>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP, true, 4, 4, false>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void trace_kernel<MRT_PROJ_PERSP, true, 4, 4, true>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void full_trace_kernel<MRT_PROJ_ORTHO>(const RenderParams, const float4*, const uint32_t*, const uint32_t*, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void full_trace_kernel<MRT_PROJ_PERSP>(const RenderParams, const float4*, const uint32_t*, const uint32_t*, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void prep_full_records<MRT_PROJ_ORTHO>(const RenderParams, float4*, uint32_t*, uint32_t*);
template __global__ void prep_full_records<MRT_PROJ_PERSP>(const RenderParams, float4*, uint32_t*, uint32_t*);
template __global__ void cull_trace_kernel<MRT_PROJ_ORTHO>(const RenderParams, float*, int32_t*, uint32_t);
template __global__ void cull_trace_kernel<MRT_PROJ_PERSP>(const RenderParams, float*, int32_t*, uint32_t);
template __global__ void grid_trace_kernel<MRT_PROJ_ORTHO>(const RenderParams, const uint32_t*, const uint32_t*, uint32_t, float*, int32_t*, uint32_t);
template __global__ void grid_trace_kernel<MRT_PROJ_PERSP>(const RenderParams, const uint32_t*, const uint32_t*, uint32_t, float*, int32_t*, uint32_t);

template __global__ void shade_keys_kernel<MRT_PROJ_ORTHO>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void shade_keys_kernel<MRT_PROJ_PERSP>(const RenderParams, unsigned long long*, float*, int32_t*, uint32_t);
template __global__ void shade_kernel<MRT_PROJ_ORTHO>(const RenderParams, const float*, const int32_t*, uint32_t);
template __global__ void shade_kernel<MRT_PROJ_PERSP>(const RenderParams, const float*, const int32_t*, uint32_t);

}  // namespace mrt

#ifdef MRT_TIMELINE
// Diagnostic build only: copies the wave records of everything trace_kernel ran on the current
// device in its last launch (at most max_entries of 96 bytes each) and clears the count.
extern "C" int mrt_debug_timeline_(void* out, uint32_t max_entries, uint32_t* n_out) {
    uint32_t n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return MRT_ERR_HIP;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(mrt::g_timeline_count), sizeof(n)) != hipSuccess) return MRT_ERR_HIP;
    if (n > mrt::kTimelineMax) n = mrt::kTimelineMax;
    if (n > max_entries) n = max_entries;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(mrt::g_timeline), (size_t)n * sizeof(mrt::TimelineEntry)) != hipSuccess) return MRT_ERR_HIP;
    const uint32_t zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mrt::g_timeline_count), &zero, sizeof(zero)) != hipSuccess) return MRT_ERR_HIP;
    *n_out = n;
    return MRT_OK;
}
#endif
