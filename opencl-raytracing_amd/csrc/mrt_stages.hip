// mrt_stages.hip -- stage-level entry points with the reference's own
// array-of-structs types, one per helper of CLHelpers.hpp:76-322.
//
// These exist so a caller that keeps the reference's four-stage
// gfx::scene::render() (gfxraytrace.hpp:830-878) can swap CLHelpers.hpp alone.
// Each call copies its host arrays in, runs one kernel over n work-items (the
// reference's fixed global size of 160000, CLHelpers.hpp:9, is not kept) and
// copies the result out.  The fused path (mrt_render_*) is the fast one; these
// trade speed for the reference's exact call shape.  Arithmetic is the shared
// exact code of mrt_device.hpp.
#include "mrt.h"
#include "mrt_guard.hpp"
#include "mrt_device.hpp"

#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <type_traits>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#pragma clang fp contract(off)

extern "C" int mrt_stage_set_error_(mrt_ctx* ctx, int code, const char* msg);
extern "C" int mrt_stage_device_(const mrt_ctx* ctx);
extern "C" void* mrt_stage_host_buffer_(mrt_ctx* ctx, size_t bytes);
extern "C" void mrt_stage_set_t_switches_(mrt_ctx* ctx, uint32_t n);
extern "C" void* mrt_stage_buffer_(mrt_ctx* ctx, int slot, size_t bytes);
extern "C" void* mrt_ctx_stream_(const mrt_ctx* ctx);
extern "C" void* mrt_stage_event_(mrt_ctx* ctx, int which);
extern "C" void mrt_stage_set_kernel_ms_(mrt_ctx* ctx, double ms);
extern "C" double mrt_stage_get_kernel_ms_(const mrt_ctx* ctx);

namespace mrt {

__global__ void stage_uv_kernel(const float4* __restrict__ pos, float left, float right, float top,
                                float bottom, float xres, float yres, int n, float2* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pos[i];
    out[i] = make_float2(uv_component(left, right, p.x, xres), uv_component(bottom, top, p.y, yres));
}

struct CamK { float eye[3], u[3], v[3], w[3]; };

template <int PROJ>
__global__ void stage_rays_kernel(CamK c, const float2* __restrict__ uv, int n, float focal,
                                  float4* __restrict__ rays /* 2 per ray */) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 q = uv[i];
    F3 o, d;
    if (PROJ == MRT_PROJ_PERSP)
        persp_ray(mk3(c.eye[0], c.eye[1], c.eye[2]), mk3(c.u[0], c.u[1], c.u[2]), mk3(c.v[0], c.v[1], c.v[2]),
                  mk3(c.w[0], c.w[1], c.w[2]), focal, q.x, q.y, o, d);
    else
        ortho_ray(mk3(c.u[0], c.u[1], c.u[2]), mk3(c.v[0], c.v[1], c.v[2]), mk3(c.w[0], c.w[1], c.w[2]), q.x, q.y, o, d);
    // The fourth float of a float3 is never written by the reference kernels.
    rays[2 * (size_t)i] = make_float4(o.x, o.y, o.z, 0.f);
    rays[2 * (size_t)i + 1] = make_float4(d.x, d.y, d.z, 0.f);
}

// ---------------------------------------------------------------------------
// kernel `intersections` (kernel.cl:305-379) for ARBITRARY rays: every ray
// brings its own origin and direction, the caller brings the t range.
//
// Closest hit = the smallest accepted t, the lowest object index among equal t
// (the reference walks the objects in order and keeps a hit only on a strict
// "<", kernel.cl:326,344), over the pairs with t_lower < t < t_upper.  That is
// independent of the order in which pairs are evaluated, so the object list is
// cut into slices (blockIdx.z), every slice of every ray block is a wave of its
// own, and the slices meet in one 64-bit key per ray, (t, object index), under
// atomicMin.
//
// Object record (host side, 3 x float4 per object, in the caller's order):
//   sphere   geo[0] = (cx, cy, cz, 0) geo[1] = (r, -, -, -)
//   triangle geo[0] = (ax, ay, az, 1) geo[1] = (bx, by, bz, -) geo[2] = (cx, cy, cz, -)
//
// Triangles.  tri_intersect (kernel.cl:197-235) solves by Cramer's rule: det,
// Nb, Ng, Nt, then beta = Nb/det, gamma = Ng/det, t = Nt/det and the range
// tests, the determinants being triple products of s1 = a - b, s2 = a - c,
// sol = a - o and d.  Expanding sol and using the moment m = d x o of the ray
// (per ray) and n = s1 x s2, A1 = a x s1, A2 = a x s2, k = a . n (per triangle):
//     det = d . n     Nt = k - o . n     Nb = d . A2 - s2 . m     Ng = s1 . m - d . A1
// -- the four determinants of a general ray as six dot products of the
// ray's (d, m, o) with per-triangle vectors, 18 fused multiply-adds per pair
// and neither a subtraction nor a cross product (15 without Nt, which a wave
// leaves out until it pays: see the kernel).  It is a conservative filter,
// never the decision: a hit needs Nb, Ng and W = det - Nb - Ng (and Nt, when t_lower >= 0)
// to have the sign of det, and since det = Nb + Ng + W that is "all of one
// sign".  The fused values differ from the reference's unfused ones by less than
// 16 u P (u = 2^-24, P the sum of the absolute values of the six products of
// the determinant with |a| + |o| standing for |sol|: expanding a - o gives up
// the cancellation the reference's subtraction enjoys, so a scene far from the
// origin passes more pairs on to the reference arithmetic, never fewer than it
// must); prep_general_records bounds P per triangle over all rays
// of the call (general_ray_bounds), takes M = 32 u (sum of the four P) plus the
// band that keeps a quotient which would round to -0, and the kernel starts
// every fma chain at +M: X = value + M.  A pair goes on to the reference
// arithmetic (tri_exact, from the original vertices) unless some X is negative
// AND some X exceeds 2M; on the shipped mesh about one pair in two thousand
// does.  Rays with a non-finite or huge component ("wild") run the reference
// arithmetic for every object, as do triangles whose products may overflow.
//
// Spheres: the discriminant of circle_intersect (kernel.cl:248-254) in the
// reference's own operation order for every pair; sqrt and the division only
// where it is not negative.
// ---------------------------------------------------------------------------
struct HitOut { float4 loc_t; float4 n_id; };

#ifndef MRT_GEN_RAYS
#define MRT_GEN_RAYS 8
#endif
#ifndef MRT_GEN_WAVES
#define MRT_GEN_WAVES 4
#endif
constexpr int kGenRays = MRT_GEN_RAYS;   // rays per lane
constexpr int kGenThreads = 256;
constexpr int kGenBlockRays = kGenThreads * kGenRays;
#ifndef MRT_GEN_CHUNK
#define MRT_GEN_CHUNK 256
#endif
constexpr int kGenChunk = MRT_GEN_CHUNK; // triangle records per LDS chunk (5 float4 each: 20 KiB)
constexpr int kGenRecFloat4 = 5;
constexpr int kGenSphFloat4 = 2;         // (cx, cy, cz, r) (object index, -, -, -)
constexpr int kGenQueue = 64 + kGenRays * 64;   // (ray, object) pairs awaiting the reference arithmetic, per wave
constexpr float kWildLimit = 0x1p60f;    // rays with a component at or beyond this (or NaN) are "wild"

// Order-preserving map of a float to an unsigned integer (negative t is legal when t_lower < 0).
__device__ __forceinline__ uint32_t ordered_bits(float t) {
    const uint32_t b = __float_as_uint(t + 0.0f);             // -0 and +0 compare equal: one key for both
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ bool is_wild(float4 ro, float4 rd) {
    const float m = fmaxf(fmaxf(fmaxf(fabsf(ro.x), fabsf(ro.y)), fmaxf(fabsf(ro.z), fabsf(rd.x))),
                          fmaxf(fabsf(rd.y), fabsf(rd.z)));
    // fmaxf drops NaNs: test every component on its own
    const bool nan = ro.x != ro.x || ro.y != ro.y || ro.z != ro.z || rd.x != rd.x || rd.y != rd.y || rd.z != rd.z;
    return nan || !(m < kWildLimit);
}

// Per component, the largest |origin| and |direction| over the rays that are not wild, as the
// bit patterns of non-negative floats under atomicMax (bounds[0..2] origin, [3..5] direction):
// four rays per thread, one set of atomics per block.  Also sets every ray's key to "no hit".
__global__ __launch_bounds__(256) void general_ray_bounds(const float4* __restrict__ rays, int n_rays,
                                                          uint32_t* __restrict__ bounds,
                                                          unsigned long long* __restrict__ keys,
                                                          uint32_t* __restrict__ t_switches) {
    __shared__ float s_m[4][6];
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_switches = 0u;      // counted by general_intersect_kernel
    float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += gridDim.x * blockDim.x) {
        const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
        keys[i] = ~0ull;
        if (is_wild(ro, rd)) continue;
        m[0] = fmaxf(m[0], fabsf(ro.x)); m[1] = fmaxf(m[1], fabsf(ro.y)); m[2] = fmaxf(m[2], fabsf(ro.z));
        m[3] = fmaxf(m[3], fabsf(rd.x)); m[4] = fmaxf(m[4], fabsf(rd.y)); m[5] = fmaxf(m[5], fabsf(rd.z));
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float v = m[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const float v = fmaxf(fmaxf(s_m[0][threadIdx.x], s_m[1][threadIdx.x]), fmaxf(s_m[2][threadIdx.x], s_m[3][threadIdx.x]));
        if (v > 0.0f) atomicMax(&bounds[threadIdx.x], __float_as_uint(v));
    }
}

namespace {
// a x b with one rounded product and one fused multiply-add per component.
__device__ __forceinline__ F3 cross_fma(F3 a, F3 b) {
    return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)),
               __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
               __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
// a . b + start as a chain of three fused multiply-adds.
__device__ __forceinline__ float dot_fma(F3 a, F3 b, float start) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, __builtin_fmaf(a.x, b.x, start)));
}
// Sum of the absolute values of the six products of det3(a, b, c), for vectors of magnitudes.
__device__ __forceinline__ double perm_abs(double ax, double ay, double az, double bx, double by, double bz,
                                           double cx, double cy, double cz) {
    return ax * (by * cz + bz * cy) + ay * (bx * cz + bz * cx) + az * (bx * cy + by * cx);
}
// x (positive) as a float that is not below it.
__device__ __forceinline__ float round_up_to_float(double x) {
    float f = (float)x;
    if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + 1u);
    return f;
}
}  // namespace

// Filter records of general_intersect_kernel, five float4 per triangle, in list order:
//   (A2 | M) (-A1 | 3M) (-s2 | 2M) (s1 | object index) (n | k + M)
__global__ void prep_general_records(const float4* __restrict__ geo, const uint32_t* __restrict__ tri_objs,
                                     int n_tri, const uint32_t* __restrict__ bounds, float4* __restrict__ rec) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tri) return;
    const uint32_t obj = tri_objs[k];
    const float4 va = geo[3 * (size_t)obj], vb = geo[3 * (size_t)obj + 1], vc = geo[3 * (size_t)obj + 2];
    const F3 a = mk3(va.x, va.y, va.z);
    const F3 s1 = sub3(a, mk3(vb.x, vb.y, vb.z));            // kernel.cl:204-205, as the reference rounds them
    const F3 s2 = sub3(a, mk3(vc.x, vc.y, vc.z));
    const F3 n = cross_fma(s1, s2);
    const F3 a1 = cross_fma(a, s1);
    const F3 a2 = cross_fma(a, s2);
    const double kt = (double)a.x * (double)n.x + (double)a.y * (double)n.y + (double)a.z * (double)n.z;
    // magnitudes, with room for the roundings of d x o and of the sums
    const double ox = 1.01 * (double)__uint_as_float(bounds[0]), oy = 1.01 * (double)__uint_as_float(bounds[1]),
                 oz = 1.01 * (double)__uint_as_float(bounds[2]);
    const double dx = 1.01 * (double)__uint_as_float(bounds[3]), dy = 1.01 * (double)__uint_as_float(bounds[4]),
                 dz = 1.01 * (double)__uint_as_float(bounds[5]);
    const double a1x = fabs((double)s1.x), a1y = fabs((double)s1.y), a1z = fabs((double)s1.z);
    const double a2x = fabs((double)s2.x), a2y = fabs((double)s2.y), a2z = fabs((double)s2.z);
    const double sx = 1.01 * fabs((double)a.x) + ox, sy = 1.01 * fabs((double)a.y) + oy, sz = 1.01 * fabs((double)a.z) + oz;
    const double p_det = perm_abs(a1x, a1y, a1z, a2x, a2y, a2z, dx, dy, dz);
    const double p_b = perm_abs(sx, sy, sz, a2x, a2y, a2z, dx, dy, dz);
    const double p_g = perm_abs(a1x, a1y, a1z, sx, sy, sz, dx, dy, dz);
    const double p_t = perm_abs(a1x, a1y, a1z, a2x, a2y, a2z, sx, sy, sz);
    const double big = fmax(fmax(fmax(a1x, a1y), fmax(a1z, a2x)), fmax(fmax(a2y, a2z), fmax(fmax(sx, sy), fmax(sz, fmax(dx, fmax(dy, dz))))));
    // |quotient| < 2^-149 rounds to -0, which passes the reference's "< 0"; products in the
    // denormal range are off by 2^-150 absolutely, scaled by the factor they meet next
    const double band = p_det * 0x1p-100 + 64.0 * 0x1p-149 * (1.0 + big * (1.0 + big));
    const double p_all = p_det + p_b + p_g + p_t;
    const double margin = (32.0 / 16777216.0) * p_all + band;
    float4 r0, r1, r2, r3, r4;
    if (p_all < 1e36 && margin < 1e36 && fabs(kt) < 1e36) {   // false for NaN too
        const float m = round_up_to_float(margin);
        r0 = make_float4(a2.x, a2.y, a2.z, m);
        r1 = make_float4(-a1.x, -a1.y, -a1.z, round_up_to_float(3.0 * (double)m));
        r2 = make_float4(-s2.x, -s2.y, -s2.z, 2.0f * m);
        r3 = make_float4(s1.x, s1.y, s1.z, 0.f);
        r4 = make_float4(n.x, n.y, n.z, (float)(kt + (double)m));
    } else {
        // The margin cannot be bounded: a record every ray passes (all products zero: X = M = 1).
        r0 = make_float4(0.f, 0.f, 0.f, 1.0f);
        r1 = make_float4(0.f, 0.f, 0.f, 3.0f);
        r2 = make_float4(0.f, 0.f, 0.f, 2.0f);
        r3 = make_float4(0.f, 0.f, 0.f, 0.f);
        r4 = make_float4(0.f, 0.f, 0.f, 1.0f);
    }
    r3.w = __uint_as_float(obj);
    rec[(size_t)k * kGenRecFloat4 + 0] = r0;
    rec[(size_t)k * kGenRecFloat4 + 1] = r1;
    rec[(size_t)k * kGenRecFloat4 + 2] = r2;
    rec[(size_t)k * kGenRecFloat4 + 3] = r3;
    rec[(size_t)k * kGenRecFloat4 + 4] = r4;
}

// The pairs the filter cannot rule out are rare and scattered over lanes and rays, so they are
// not tested where they are found: the wave appends (ray, object) to a queue of its own in LDS
// and, whenever 64 are waiting (and once more at the end), every lane takes one entry, reads its
// ray and its object from memory and runs the reference arithmetic -- all 64 lanes busy, nothing
// held in registers while the filter loop runs.
//
// The fourth determinant.  When t_lower >= 0 an accepted t is positive, so Nt must have the sign
// of det as well.  That test costs three fused multiply-adds and two verdict instructions per
// pair and rules out only pairs whose LINE meets the triangle behind the ray's origin -- as rare
// as the hits themselves in most scenes, and then cheaper to leave to the reference arithmetic:
// 160 000 x 4034 without it 0.28-0.29 ms against 0.33-0.35 with it, for the camera's primary rays
// and for rays that start inside the mesh alike.  A wave therefore starts without it and takes it
// up (t_filter_allowed: t_lower >= 0) only when the pairs it has passed on so far outnumber
// kSwitchRate per triangle of its 512 rays -- large overlapping triangles behind the rays, where
// the cheaper filter would pass every pair on.  Either filter only ever rejects what the
// reference arithmetic rejects, so the records do not depend on the choice.
constexpr uint32_t kSwitchRate = 10, kSwitchGrace = 8;

// Measurement aid: -DMRT_GEN_SERIAL_CHAINS keeps the instruction scheduler from interleaving a
// ray's dot-product chains (every instruction then depends on the one before it).
#ifdef MRT_GEN_SERIAL_CHAINS
#define MRT_GEN_CHAIN_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define MRT_GEN_CHAIN_FENCE do { } while (0)
#endif

__global__ __launch_bounds__(kGenThreads, MRT_GEN_WAVES) void general_intersect_kernel(
    const float4* __restrict__ rays, int n_rays, const float4* __restrict__ tri_rec, int n_tri, int tris_per_slice,
    const float4* __restrict__ sph_rec, int n_sph, int sph_per_slice, const float4* __restrict__ geo,
    float t_upper, float t_lower, int t_filter_allowed, unsigned long long* __restrict__ keys,
    uint32_t* __restrict__ t_switches) {
    __shared__ float4 s_rec[kGenChunk * kGenRecFloat4];
    __shared__ uint2 s_queue[kGenThreads / 64][kGenQueue];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint2* queue = s_queue[tid >> 6];
    const uint32_t ray0 = blockIdx.x * (uint32_t)kGenBlockRays + (uint32_t)tid;
    const uint32_t last_ray = (uint32_t)n_rays - 1u;

    // The lane's rays (a ray beyond the end repeats the last one: same key, same value).  Wild rays
    // take part in the filter arithmetic as zeros and are candidates for every object.
    F3 rno[kGenRays], rd[kGenRays], rm[kGenRays];           // -origin, direction, moment d x o
    uint32_t wild = 0;
#pragma unroll
    for (int s = 0; s < kGenRays; ++s) {
        uint32_t i = ray0 + (uint32_t)s * kGenThreads;
        if (i > last_ray) i = last_ray;
        const float4 o4 = rays[2 * (size_t)i], d4 = rays[2 * (size_t)i + 1];
        const bool w = is_wild(o4, d4);
        wild |= w ? (1u << s) : 0u;
        const F3 o = w ? mk3(0.f, 0.f, 0.f) : mk3(o4.x, o4.y, o4.z);
        rd[s] = w ? mk3(0.f, 0.f, 0.f) : mk3(d4.x, d4.y, d4.z);
        rm[s] = cross_fma(rd[s], o);
        rno[s] = mk3(-o.x, -o.y, -o.z);
    }
    const uint32_t wild_sign = wild ? 0x80000000u : 0u;
    uint32_t queued = 0;                                  // entries waiting in the wave's queue (wave-uniform)
    uint32_t passed_on = 0;                               // pairs this wave has handed to the reference arithmetic (wave-uniform)
    bool use_t = false;                                   // the fourth determinant takes part in the filter (wave-uniform)

    // The reference arithmetic for `count` queue entries from `from` on, one per lane.
    auto run_queue = [&](uint32_t from, uint32_t count) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if ((uint32_t)lane < count) {
            const uint2 e = queue[from + lane];
            const uint32_t i = e.x, obj = e.y;
            const float4 g0 = geo[3 * (size_t)obj], g1 = geo[3 * (size_t)obj + 1], g2 = geo[3 * (size_t)obj + 2];
            const float4 o4 = rays[2 * (size_t)i], d4 = rays[2 * (size_t)i + 1];
            const F3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
            float t;
            bool hit;
            if (g0.w != 0.0f) {
                hit = tri_exact(mk3(g0.x, g0.y, g0.z), mk3(g1.x, g1.y, g1.z), mk3(g2.x, g2.y, g2.z), o, d, t_upper, t_lower, t);
            } else {
                bool two;
                hit = sphere_exact(mk3(g0.x, g0.y, g0.z), g1.x, o, d, t_upper, t_lower, t, two);
            }
            if (hit && t > t_lower && t < t_upper)                  // kernel.cl:326,344
                atomicMin(&keys[i], ((unsigned long long)ordered_bits(t) << 32) | obj);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // Appends (ray of slot s, obj) for the lanes of `mask`; at most 64 entries per call.
    auto push = [&](uint64_t mask, bool mine, int s, uint32_t obj) {
        if (mine) {
            uint32_t i = ray0 + (uint32_t)s * kGenThreads;
            if (i > last_ray) i = last_ray;
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            queue[queued + before] = make_uint2(i, obj);
        }
        queued += (uint32_t)__builtin_popcountll(mask);
        passed_on += (uint32_t)__builtin_popcountll(mask);
    };
    auto drain = [&](uint32_t keep) {
        while (queued > keep) {
            const uint32_t n = queued < 64u ? queued : 64u;
            queued -= n;
            run_queue(queued, n);
        }
    };

    // ---- triangles of this slice ----
    const int t_begin = (int)blockIdx.z * tris_per_slice;
    const int t_end = t_begin + tris_per_slice < n_tri ? t_begin + tris_per_slice : n_tri;
    for (int cbase = t_begin; cbase < t_end; cbase += kGenChunk) {
        const int count = t_end - cbase < kGenChunk ? t_end - cbase : kGenChunk;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kGenRecFloat4; ++k) {
            const int i = tid + k * kGenThreads;
            if (i < count * kGenRecFloat4) s_rec[i] = tri_rec[(size_t)cbase * kGenRecFloat4 + i];
        }
        __syncthreads();
        // Two loops over the chunk's triangles, one per filter, each compiled on its own (one loop
        // with a branch per triangle around the two bodies spills 52 registers into its survivor
        // path and is slower than either): the wave runs the cheaper one until the rate of pairs
        // passed on says otherwise, then the other from that triangle on, in this chunk and the next.
        auto scan = [&](auto with_t_tag, int g_begin) -> int {
            constexpr bool WITH_T = decltype(with_t_tag)::value;
            for (int g = g_begin; g < count; ++g) {
                const float4 q0 = s_rec[g * kGenRecFloat4 + 0];
                const float4 q1 = s_rec[g * kGenRecFloat4 + 1];
                const float4 q2 = s_rec[g * kGenRecFloat4 + 2];
                const float4 q3 = s_rec[g * kGenRecFloat4 + 3];
                const float4 q4 = s_rec[g * kGenRecFloat4 + 4];
                const F3 a2 = mk3(q0.x, q0.y, q0.z), na1 = mk3(q1.x, q1.y, q1.z), ns2 = mk3(q2.x, q2.y, q2.z);
                const F3 s1 = mk3(q3.x, q3.y, q3.z), n = mk3(q4.x, q4.y, q4.z);
                // The sign bit of c[s] is set when ray s may hit: all X >= 0 (a hit with det > 0) or
                // all X <= 2M (det < 0).
                uint32_t c[kGenRays];
                uint32_t any = wild_sign;
#pragma unroll
                for (int s = 0; s < kGenRays; ++s) {
                    const float xb = dot_fma(ns2, rm[s], dot_fma(a2, rd[s], q0.w));
                    MRT_GEN_CHAIN_FENCE;
                    const float xg = dot_fma(s1, rm[s], dot_fma(na1, rd[s], q0.w));
                    MRT_GEN_CHAIN_FENCE;
                    const float xd = dot_fma(n, rd[s], q1.w);
                    MRT_GEN_CHAIN_FENCE;
                    const float xw = (xd - xb) - xg;
                    float nlo = __builtin_fmaxf(__builtin_fmaxf(-xb, -xg), -xw);
                    float hi = __builtin_fmaxf(__builtin_fmaxf(xb, xg), xw);
                    if (WITH_T) {
                        const float xt = dot_fma(n, rno[s], q4.w);
                        nlo = __builtin_fmaxf(nlo, -xt);
                        hi = __builtin_fmaxf(hi, xt);
                    }
                    c[s] = __float_as_uint(nlo) | __float_as_uint(hi - q2.w);
                    any |= c[s];
                }
                // (the compare's lane mask itself is the vote: no round trip through a VGPR)
                if (__builtin_amdgcn_sicmp((int32_t)any, 0, 40 /* ICMP_SLT */) == 0) continue;
#ifdef MRT_GEN_NO_SURVIVORS
                // timing-only build (wrong records): what the survivors' path costs
                if (any != 0x12345678u) continue;
#endif
                const uint32_t obj = __builtin_amdgcn_readfirstlane(__float_as_uint(q3.w));
#pragma unroll
                for (int s = 0; s < kGenRays; ++s) {
                    const bool mine = (int32_t)(c[s] | ((wild >> s) << 31)) < 0;
                    const uint64_t mask = __builtin_amdgcn_ballot_w64(mine);
                    if (mask) push(mask, mine, s, obj);
                }
                drain(63u);
                // too many pairs passed on for the triangles seen so far: the fourth determinant from here on
                if (!WITH_T && t_filter_allowed && passed_on > kSwitchRate * ((uint32_t)(cbase - t_begin + g) + kSwitchGrace)) {
                    use_t = true;
                    if (lane == 0) atomicAdd(t_switches, 1u);           // (a diagnostic: once per wave at most)
                    return g + 1;
                }
            }
            return count;
        };
        int g_next = 0;
        if (!use_t) g_next = scan(std::false_type{}, 0);
        if (g_next < count) (void)scan(std::true_type{}, g_next);
    }

    // ---- spheres of this slice ----
    const int s_begin = (int)blockIdx.z * sph_per_slice;
    const int s_end = s_begin + sph_per_slice < n_sph ? s_begin + sph_per_slice : n_sph;
    constexpr int kSphChunk = kGenChunk * kGenRecFloat4 / kGenSphFloat4;
    for (int cbase = s_begin; cbase < s_end; cbase += kSphChunk) {
        const int count = s_end - cbase < kSphChunk ? s_end - cbase : kSphChunk;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kGenRecFloat4; ++k) {
            const int i = tid + k * kGenThreads;
            if (i < count * kGenSphFloat4) s_rec[i] = sph_rec[(size_t)cbase * kGenSphFloat4 + i];
        }
        __syncthreads();
        for (int g = 0; g < count; ++g) {
            const float4 sp = s_rec[g * kGenSphFloat4];
            const F3 center = mk3(sp.x, sp.y, sp.z);
            // circle_intersect gives up only when the discriminant compares below zero
            // (kernel.cl:256): a NaN discriminant goes on.
            uint32_t mine_bits = wild;
#pragma unroll
            for (int s = 0; s < kGenRays; ++s) {
                float qa, qb;
                const F3 o = mk3(-rno[s].x, -rno[s].y, -rno[s].z);         // exact: negation twice
                mine_bits |= !(sphere_disc(center, sp.w, o, rd[s], qa, qb) < 0.0f) ? (1u << s) : 0u;
            }
            if (__builtin_amdgcn_ballot_w64(mine_bits != 0) == 0) continue;
            const uint32_t obj = __builtin_amdgcn_readfirstlane(__float_as_uint(s_rec[g * kGenSphFloat4 + 1].x));
#pragma unroll
            for (int s = 0; s < kGenRays; ++s) {
                const bool mine = (mine_bits >> s) & 1u;
                const uint64_t mask = __builtin_amdgcn_ballot_w64(mine);
                if (mask) push(mask, mine, s, obj);
            }
            drain(63u);
        }
    }
    drain(0u);
}

// Decodes the keys: the winner's hit record, with t, location and normal from the reference
// arithmetic (tri_intersect / circle_intersect, kernel.cl:197-303).
__global__ void general_finish_kernel(const float4* __restrict__ geo, const float4* __restrict__ rays, int n_rays,
                                      const unsigned long long* __restrict__ keys, float t_upper, float t_lower,
                                      HitOut* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    const unsigned long long key = keys[i];
    const int best = (key == ~0ull) ? -1 : (int)(uint32_t)key;
    HitOut h;
    h.loc_t = make_float4(0.f, 0.f, 0.f, -1.0f);
    h.n_id = make_float4(0.f, 0.f, 0.f, __int_as_float(best));
    if (best >= 0) {
        const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
        const F3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
        const float4 g0 = geo[3 * (size_t)best], g1 = geo[3 * (size_t)best + 1], g2 = geo[3 * (size_t)best + 2];
        float t = 0.f;
        F3 n, loc;
        if (g0.w != 0.0f) {
            const F3 A = mk3(g0.x, g0.y, g0.z), B = mk3(g1.x, g1.y, g1.z), C = mk3(g2.x, g2.y, g2.z);
            (void)tri_exact(A, B, C, o, d, t_upper, t_lower, t);
            loc = add3(o, scale3(d, t));
            n = normalize3(cross3(sub3(A, B), sub3(A, C)));
        } else {
            const F3 c = mk3(g0.x, g0.y, g0.z);
            bool two = false;
            (void)sphere_exact(c, g1.x, o, d, t_upper, t_lower, t, two);
            loc = add3(o, scale3(d, t));
            if (two) n = normalize3(div3(sub3(loc, c), g1.x));
            else n = normalize3(sub3(loc, c));
        }
        h.loc_t = make_float4(loc.x, loc.y, loc.z, t);
        h.n_id = make_float4(n.x, n.y, n.z, __int_as_float(best));
    }
    out[i] = h;
}

struct PhongK { float ka, kd, ks, amb[3], eye[3], bg[3]; int n_lights; int flat; };

// in: 5 float4 per pixel = the 80-byte intersect record viewed as 20 dwords.
__global__ void stage_shade_kernel(const uint32_t* __restrict__ xs, int n, PhongK k,
                                   const float4* __restrict__ lights, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* r = xs + (size_t)i * 20;
    // intersect layout (CLStructs.hpp:115-128): location @0, normal @16,
    // intersects @32 (one byte), shininess @60, color @64.
    const bool hit = (r[8] & 0xFFu) != 0;
    float4 res;
    if (!hit) {
        res = make_float4(k.bg[0], k.bg[1], k.bg[2], 0.0f);   // w = 0: a miss
    } else {
        const F3 color = mk3(__uint_as_float(r[16]), __uint_as_float(r[17]), __uint_as_float(r[18]));
        float cr, cg, cb;
        if (k.flat) {
            cr = color.x; cg = color.y; cb = color.z;
        } else {
            const F3 loc = mk3(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]));
            const F3 nrm = mk3(__uint_as_float(r[4]), __uint_as_float(r[5]), __uint_as_float(r[6]));
            const float shin = __uint_as_float(r[15]);
            cr = k.ka * k.amb[0]; cg = k.ka * k.amb[1]; cb = k.ka * k.amb[2];
            for (int l = 0; l < k.n_lights; ++l) {
                const float4 lp = lights[2 * l], lc = lights[2 * l + 1];
                phong_light(loc, nrm, color, shin, mk3(k.eye[0], k.eye[1], k.eye[2]), mk3(lp.x, lp.y, lp.z),
                            mk3(lc.x, lc.y, lc.z), k.kd, k.ks, cr, cg, cb);
            }
            cr = clamp01(cr); cg = clamp01(cg); cb = clamp01(cb);
        }
        res = make_float4(cr, cg, cb, 1.0f);
    }
    out[i] = res;
}

}  // namespace mrt

namespace {

// A staging buffer of the ctx (slot 0..2): the ctx owns it, it grows on demand and is reused by
// every later stage call, so a steady stream of calls allocates nothing.
struct DevBuf {
    void* p = nullptr;
    bool take(mrt_ctx* ctx, int slot, size_t bytes) { p = mrt_stage_buffer_(ctx, slot, bytes); return p != nullptr; }
};
#define STAGE_BUF(buf, slot, bytes)                                  \
    do {                                                             \
        if (!(buf).take(ctx, (slot), (bytes))) return MRT_ERR_HIP;    \
    } while (0)
// Copies and kernels of a stage call go to the ctx's own stream; the call returns when the
// results are in the caller's arrays (the reference's helpers block too: CLHelpers.hpp:106-109).
#define STAGE_COPY(dst, src, bytes, kind)                                                    \
    do {                                                                                     \
        STAGE_TRY(hipMemcpyAsync((dst), (src), (bytes), (kind), stage_stream));               \
        if ((kind) == hipMemcpyDeviceToHost) STAGE_TRY(hipStreamSynchronize(stage_stream));   \
    } while (0)

#define STAGE_TRY(expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return mrt_stage_set_error_(ctx, MRT_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

// Whether the caller's result array looks freshly allocated: three of its pages asked after with
// mincore(2), two of them not resident.  (The reference's helpers return a new[] array per call,
// CLHelpers.hpp:91,131,167,208,254,293: memory the allocator has just mapped, every page of which
// faults on its first store.)
bool looks_untouched(const void* p, size_t bytes) {
    const long ps = sysconf(_SC_PAGESIZE);
    if (ps <= 0 || bytes < 16 * (size_t)ps) return false;
    const uintptr_t first = ((uintptr_t)p + (uintptr_t)ps - 1) & ~((uintptr_t)ps - 1);
    const size_t pages = ((uintptr_t)p + bytes - first) / (size_t)ps;
    int cold = 0;
    for (size_t k : {pages / 8, pages / 2, pages - 2}) {
        unsigned char v = 1;
        if (mincore(reinterpret_cast<void*>(first + k * (size_t)ps), 1, &v) == 0 && !(v & 1)) ++cold;
    }
    return cold >= 2;
}

// The host's share of a stage call -- results scattered from the ctx's page-locked buffer into the
// caller's records, `bytes` at `dst` -- on a few threads where that pays: fn(begin, end) stores
// elements [begin, end) and touches nothing else.  Starting a thread costs 60 microseconds or so, a
// tenth of what one thread needs for 160 000 records that are resident, so threads are used for a
// million elements and more, and from 32768 on when the array looks untouched (its page faults are
// then taken side by side: 1.5 -> 1.05 ms for the reference's 160 000 intersect records).  One
// thread per 32768 elements, eight at most; a thread that cannot be started leaves its share to
// the caller's.
template <typename Fn>
void scatter_rows(int n, const void* dst, size_t bytes, Fn fn) {
    constexpr int kPerThread = 32768, kMaxThreads = 8, kAlways = 1 << 20;
    int threads = n / kPerThread;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (unsigned)threads > hw) threads = (int)hw;
    if (threads > kMaxThreads) threads = kMaxThreads;
    if (threads > 1 && n < kAlways && !looks_untouched(dst, bytes)) threads = 1;
    if (threads <= 1) { fn(0, n); return; }
    struct Joiner {
        std::vector<std::thread> t;
        ~Joiner() { for (std::thread& x : t) if (x.joinable()) x.join(); }
    } workers;
    const int share = (n + threads - 1) / threads;
    int begin = share;                                 // [0, share) is the calling thread's
    try {
        workers.t.reserve((size_t)threads - 1);
        for (; begin < n; begin += share) {
            const int end = begin + share < n ? begin + share : n;
            workers.t.emplace_back(fn, begin, end);
        }
    } catch (...) {
        fn(begin, n);                                  // (std::system_error: no more threads to be had)
    }
    fn(0, share < n ? share : n);
}

// Puts the caller's current HIP device back when a stage call returns (begin() below switches
// to the ctx's GPU).
struct RestoreDevice {
    int prev = -1;
    RestoreDevice() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~RestoreDevice() {
        int now = -1;
        if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    }
};

int begin(mrt_ctx* ctx, int n, const void* a, const void* b) {
    if (!ctx) return MRT_ERR_INVALID;
    if (n < 0) return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "negative element count");
    if (n > 0 && (!a || !b)) return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "NULL array");
    hipError_t e = hipSetDevice(mrt_stage_device_(ctx));
    if (e != hipSuccess) return mrt_stage_set_error_(ctx, MRT_ERR_HIP, hipGetErrorString(e));
    return MRT_OK;
}

inline dim3 grid_for(int n) { return dim3((unsigned)((n + 255) / 256)); }

mrt::CamK cam_of(const mrt_cl_cam* c) {
    mrt::CamK k;
    for (int i = 0; i < 3; ++i) { k.eye[i] = c->eye.s[i]; k.u[i] = c->u.s[i]; k.v[i] = c->v.s[i]; k.w[i] = c->w.s[i]; }
    return k;
}

int rays_common(mrt_ctx* ctx, const mrt_cl_cam* camera, const mrt_cl_float2* uv, int n, float focal,
                mrt_cl_ray* rays_out, bool persp) {
    RestoreDevice restore_device;
    int rc = begin(ctx, n, uv, rays_out);
    if (rc != MRT_OK) return rc;
    if (!camera) return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "camera is NULL");
    if (n == 0) return MRT_OK;
    hipStream_t stage_stream = (hipStream_t)mrt_ctx_stream_(ctx);
    DevBuf d_uv, d_rays;
    STAGE_BUF(d_uv, 0, (size_t)n * 8);
    STAGE_BUF(d_rays, 1, (size_t)n * 32);
    STAGE_COPY(d_uv.p, uv, (size_t)n * 8, hipMemcpyHostToDevice);
    // The results come back into the ctx's page-locked buffer and only the three components the
    // kernel computes are stored into the caller's records: the padding floats keep what the
    // caller put there, as they do in the reference (viewRay holds two cl_float3).
    const mrt_cl_ray* tmp = static_cast<const mrt_cl_ray*>(mrt_stage_host_buffer_(ctx, (size_t)n * 32));
    if (!tmp) return MRT_ERR_HIP;
    if (persp)
        hipLaunchKernelGGL(mrt::stage_rays_kernel<MRT_PROJ_PERSP>, grid_for(n), dim3(256), 0, stage_stream, cam_of(camera),
                           (const float2*)d_uv.p, n, focal, (float4*)d_rays.p);
    else
        hipLaunchKernelGGL(mrt::stage_rays_kernel<MRT_PROJ_ORTHO>, grid_for(n), dim3(256), 0, stage_stream, cam_of(camera),
                           (const float2*)d_uv.p, n, focal, (float4*)d_rays.p);
    STAGE_TRY(hipGetLastError());
    STAGE_COPY(const_cast<mrt_cl_ray*>(tmp), d_rays.p, (size_t)n * 32, hipMemcpyDeviceToHost);
    scatter_rows(n, rays_out, (size_t)n * sizeof(mrt_cl_ray), [=](int begin, int end) {
        for (int i = begin; i < end; ++i)
            for (int k = 0; k < 3; ++k) {
                rays_out[i].origin.s[k] = tmp[(size_t)i].origin.s[k];
                rays_out[i].direction.s[k] = tmp[(size_t)i].direction.s[k];
            }
    });
    return MRT_OK;
}

int shade_common(mrt_ctx* ctx, const mrt_cl_intersect* xsects, int n, mrt_cl_rgb background,
                 const mrt_cl_phong* phong, const mrt_cl_cam* camera, const mrt_cl_light* lights,
                 int n_lights, mrt_cl_rgb* out, bool flat) {
    RestoreDevice restore_device;
    int rc = begin(ctx, n, xsects, out);
    if (rc != MRT_OK) return rc;
    if (!flat && (!phong || !camera || n_lights < 0 || (n_lights > 0 && !lights)))
        return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "phong shader arguments missing");
    if (n == 0) return MRT_OK;
    static_assert(sizeof(mrt_cl_intersect) == 80, "intersect must be 80 bytes");
    mrt::PhongK k;
    std::memset(&k, 0, sizeof k);
    k.flat = flat ? 1 : 0;
    k.bg[0] = background.r; k.bg[1] = background.g; k.bg[2] = background.b;
    std::vector<float4> hl;
    if (!flat) {
        k.ka = phong->ambient_coef; k.kd = phong->diffuse_coef; k.ks = phong->specular_coef;
        for (int i = 0; i < 3; ++i) { k.amb[i] = phong->ambient_color.s[i]; k.eye[i] = camera->eye.s[i]; }
        k.n_lights = n_lights;
        for (int l = 0; l < n_lights; ++l) {
            hl.push_back(make_float4(lights[l].location.s[0], lights[l].location.s[1], lights[l].location.s[2], 0.f));
            hl.push_back(make_float4(lights[l].color.s[0], lights[l].color.s[1], lights[l].color.s[2], 0.f));
        }
    }
    hipStream_t stage_stream = (hipStream_t)mrt_ctx_stream_(ctx);
    DevBuf d_x, d_l, d_o;
    STAGE_BUF(d_x, 0, (size_t)n * 80);
    STAGE_BUF(d_l, 1, hl.size() * sizeof(float4));
    STAGE_BUF(d_o, 2, (size_t)n * 16);
    STAGE_COPY(d_x.p, xsects, (size_t)n * 80, hipMemcpyHostToDevice);
    if (!hl.empty()) STAGE_COPY(d_l.p, hl.data(), hl.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mrt::stage_shade_kernel, grid_for(n), dim3(256), 0, stage_stream, (const uint32_t*)d_x.p, n, k,
                       (const float4*)d_l.p, (float4*)d_o.p);
    STAGE_TRY(hipGetLastError());
    float4* res = static_cast<float4*>(mrt_stage_host_buffer_(ctx, (size_t)n * 16));
    if (!res) return MRT_ERR_HIP;
    STAGE_COPY(res, d_o.p, (size_t)n * 16, hipMemcpyDeviceToHost);
    scatter_rows(n, out, (size_t)n * sizeof(mrt_cl_rgb), [=](int begin, int end) {
        for (int i = begin; i < end; ++i) {
            out[i].r = res[(size_t)i].x; out[i].g = res[(size_t)i].y; out[i].b = res[(size_t)i].z;
            // flat_shader never writes .a (kernel.cl:381-396); phong_shader copies
            // the whole background struct on a miss (kernel.cl:443) and leaves .a
            // alone on a hit.
            if (!flat && res[(size_t)i].w == 0.0f) out[i].a = background.a;
        }
    });
    return MRT_OK;
}

}  // namespace

extern "C" {

int mrt_stage_uv(mrt_ctx* ctx, const mrt_cl_float3* positions, const mrt_cl_vp* viewport, int n,
                 mrt_cl_float2* uv_out) try {
    RestoreDevice restore_device;
    int rc = begin(ctx, n, positions, uv_out);
    if (rc != MRT_OK) return rc;
    if (!viewport) return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "viewport is NULL");
    if (n == 0) return MRT_OK;
    hipStream_t stage_stream = (hipStream_t)mrt_ctx_stream_(ctx);
    DevBuf d_pos, d_uv;
    STAGE_BUF(d_pos, 0, (size_t)n * 16);
    STAGE_BUF(d_uv, 1, (size_t)n * 8);
    STAGE_COPY(d_pos.p, positions, (size_t)n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mrt::stage_uv_kernel, grid_for(n), dim3(256), 0, stage_stream, (const float4*)d_pos.p, viewport->left,
                       viewport->right, viewport->top, viewport->bottom, viewport->x_resolution,
                       viewport->y_resolution, n, (float2*)d_uv.p);
    STAGE_TRY(hipGetLastError());
    STAGE_COPY(uv_out, d_uv.p, (size_t)n * 8, hipMemcpyDeviceToHost);
    return MRT_OK;
} MRT_GUARD((void)mrt_stage_set_error_(ctx, code_, what_))

int mrt_stage_ortho_viewrays(mrt_ctx* ctx, const mrt_cl_cam* camera, const mrt_cl_float2* uv, int n,
                             mrt_cl_ray* rays_out) try {
    return rays_common(ctx, camera, uv, n, 0.0f, rays_out, false);
} MRT_GUARD((void)mrt_stage_set_error_(ctx, code_, what_))

int mrt_stage_persp_viewrays(mrt_ctx* ctx, const mrt_cl_cam* camera, const mrt_cl_float2* uv, int n,
                             float focal_length, mrt_cl_ray* rays_out) try {
    return rays_common(ctx, camera, uv, n, focal_length, rays_out, true);
} MRT_GUARD((void)mrt_stage_set_error_(ctx, code_, what_))

int mrt_stage_intersect(mrt_ctx* ctx, const mrt_cl_object* objects, int n_objects, const mrt_cl_ray* rays,
                        int n_rays, float t_upper_bound, float t_lower_bound, mrt_cl_intersect* out) try {
    RestoreDevice restore_device;
    int rc = begin(ctx, n_rays, rays, out);
    if (rc != MRT_OK) return rc;
    if (n_objects < 0 || (n_objects > 0 && !objects))
        return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "bad object array");
    if (n_rays == 0) return MRT_OK;
    static_assert(sizeof(mrt_cl_object) == 144, "object must be 144 bytes");
#ifdef MRT_STAGE_TIMING
    // Measurement aid: where the host's time of one call goes, to stderr.
    using StageClock = std::chrono::steady_clock;
    StageClock::time_point stage_t[6];
    int stage_ti = 0;
#define MRT_STAGE_MARK() (stage_t[stage_ti++] = StageClock::now())
#else
#define MRT_STAGE_MARK() do { } while (0)
#endif
    MRT_STAGE_MARK();
    // One upload for everything the kernels read about the objects: the exact records in the
    // caller's order, the object indices of the triangles, the sphere records, and six zeroed
    // words for the ray bounds.  Sections start on 256-byte boundaries.
    auto align256 = [](size_t v) { return (v + 255) / 256 * 256; };
    int n_tri = 0;
    for (int j = 0; j < n_objects; ++j) n_tri += objects[j].is_triangle ? 1 : 0;
    const int n_sph = n_objects - n_tri;
    const size_t off_geo = 0;
    const size_t off_tri = align256(off_geo + (size_t)n_objects * 3 * sizeof(float4));
    const size_t off_sph = align256(off_tri + (size_t)n_tri * sizeof(uint32_t));
    const size_t off_bounds = align256(off_sph + (size_t)n_sph * mrt::kGenSphFloat4 * sizeof(float4));
    const size_t pack_bytes = off_bounds + 256;
    std::vector<unsigned char> pack(pack_bytes, 0);
    float4* geo = reinterpret_cast<float4*>(pack.data() + off_geo);
    uint32_t* tri_objs = reinterpret_cast<uint32_t*>(pack.data() + off_tri);
    float4* sph_rec = reinterpret_cast<float4*>(pack.data() + off_sph);
    for (int j = 0, it = 0, is = 0; j < n_objects; ++j) {
        const mrt_cl_object& o = objects[j];
        // kernel.cl:322: is_triangle selects the triangle test, anything else is a sphere.
        if (o.is_triangle) {
            geo[3 * (size_t)j] = make_float4(o.triObj.a.s[0], o.triObj.a.s[1], o.triObj.a.s[2], 1.0f);
            geo[3 * (size_t)j + 1] = make_float4(o.triObj.b.s[0], o.triObj.b.s[1], o.triObj.b.s[2], 0.f);
            geo[3 * (size_t)j + 2] = make_float4(o.triObj.c.s[0], o.triObj.c.s[1], o.triObj.c.s[2], 0.f);
            tri_objs[it++] = (uint32_t)j;
        } else {
            geo[3 * (size_t)j] = make_float4(o.circleObj.center.s[0], o.circleObj.center.s[1], o.circleObj.center.s[2], 0.0f);
            geo[3 * (size_t)j + 1] = make_float4(o.circleObj.radius, 0.f, 0.f, 0.f);
            geo[3 * (size_t)j + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            float idx;
            const uint32_t ju = (uint32_t)j;
            std::memcpy(&idx, &ju, 4);
            sph_rec[(size_t)is * mrt::kGenSphFloat4] = geo[3 * (size_t)j];
            sph_rec[(size_t)is * mrt::kGenSphFloat4].w = o.circleObj.radius;
            sph_rec[(size_t)is * mrt::kGenSphFloat4 + 1] = make_float4(idx, 0.f, 0.f, 0.f);
            ++is;
        }
    }
    MRT_STAGE_MARK();                                   // the object pack is built
    hipStream_t stage_stream = (hipStream_t)mrt_ctx_stream_(ctx);
    DevBuf d_pack, d_rays, d_out, d_rec, d_keys;
    STAGE_BUF(d_pack, 0, pack_bytes);
    STAGE_BUF(d_rays, 1, (size_t)n_rays * 32);
    // (the hit records, and behind them the word that counts the waves which took up the fourth determinant)
    const size_t out_bytes = (size_t)n_rays * sizeof(mrt::HitOut);
    STAGE_BUF(d_out, 2, out_bytes + sizeof(mrt::HitOut));
    uint32_t* d_switches = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(d_out.p) + out_bytes);
    STAGE_BUF(d_rec, 3, (size_t)(n_tri ? n_tri : 1) * mrt::kGenRecFloat4 * sizeof(float4));
    STAGE_BUF(d_keys, 4, (size_t)n_rays * sizeof(unsigned long long));
    unsigned char* dp = static_cast<unsigned char*>(d_pack.p);
    const float4* d_geo = reinterpret_cast<const float4*>(dp + off_geo);
    const uint32_t* d_tri = reinterpret_cast<const uint32_t*>(dp + off_tri);
    const float4* d_sph = reinterpret_cast<const float4*>(dp + off_sph);
    uint32_t* d_bounds = reinterpret_cast<uint32_t*>(dp + off_bounds);
    STAGE_COPY(d_pack.p, pack.data(), pack_bytes, hipMemcpyHostToDevice);
    STAGE_COPY(d_rays.p, rays, (size_t)n_rays * 32, hipMemcpyHostToDevice);
    MRT_STAGE_MARK();                                   // pack and rays handed to the runtime

    // The object list in slices (grid layers), so that a call with few rays still fills the GPU.
    // The GPU holds kResidentBlocks blocks at a time (four per compute unit: four waves per SIMD);
    // blocks of one launch take about the same time, so the launch is sized to a whole number of
    // rounds -- about six -- with the last one nearly full, a slice at least 32 triangles.
    const int ray_blocks = (n_rays + mrt::kGenBlockRays - 1) / mrt::kGenBlockRays;
    constexpr int kResidentBlocks = 256 * 4;
    const int max_slices = n_tri / 32 > 1 ? n_tri / 32 : 1;
    int slices = 1;
    if (ray_blocks < 6 * kResidentBlocks) {
        slices = (6 * kResidentBlocks) / ray_blocks;          // rounds down: the last round is not overfull
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
    }
    static const int forced_slices = [] {                      // tuning aid, read once per process
        const char* e = std::getenv("MRT_STAGE_SLICES");
        const long v = e ? std::strtol(e, nullptr, 10) : 0;
        return v > 0 && v < 65536 ? (int)v : 0;
    }();
    if (forced_slices) slices = forced_slices < max_slices ? forced_slices : max_slices;
    const int tris_per_slice = (n_tri + slices - 1) / slices;
    const int sph_per_slice = (n_sph + slices - 1) / slices;

    hipEvent_t ev0 = (hipEvent_t)mrt_stage_event_(ctx, 0), ev1 = (hipEvent_t)mrt_stage_event_(ctx, 1);
    STAGE_TRY(hipEventRecord(ev0, stage_stream));
    {
        const int blocks = (n_rays + 1023) / 1024;
        hipLaunchKernelGGL(mrt::general_ray_bounds, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, stage_stream,
                           (const float4*)d_rays.p, n_rays, d_bounds, (unsigned long long*)d_keys.p, d_switches);
        STAGE_TRY(hipGetLastError());
    }
    if (n_tri) {
        hipLaunchKernelGGL(mrt::prep_general_records, dim3((n_tri + 255) / 256), dim3(256), 0, stage_stream, d_geo, d_tri,
                           n_tri, (const uint32_t*)d_bounds, (float4*)d_rec.p);
        STAGE_TRY(hipGetLastError());
    }
    if (n_objects) {
        const dim3 grid((unsigned)ray_blocks, 1, (unsigned)slices);
        // (tuning aid, read once per process: MRT_STAGE_T_FILTER=0 never lets a wave take up the fourth determinant)
        static const int t_filter = [] {
            const char* e = std::getenv("MRT_STAGE_T_FILTER");
            return e ? (int)std::strtol(e, nullptr, 10) : 1;
        }();
        hipLaunchKernelGGL(mrt::general_intersect_kernel, grid, dim3(mrt::kGenThreads), 0, stage_stream,
                           (const float4*)d_rays.p, n_rays, (const float4*)d_rec.p, n_tri, tris_per_slice, d_sph, n_sph,
                           sph_per_slice, d_geo, t_upper_bound, t_lower_bound, (t_lower_bound >= 0.0f && t_filter) ? 1 : 0,
                           (unsigned long long*)d_keys.p, d_switches);
        STAGE_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(mrt::general_finish_kernel, grid_for(n_rays), dim3(256), 0, stage_stream, d_geo,
                       (const float4*)d_rays.p, n_rays, (const unsigned long long*)d_keys.p, t_upper_bound, t_lower_bound,
                       (mrt::HitOut*)d_out.p);
    STAGE_TRY(hipGetLastError());
    STAGE_TRY(hipEventRecord(ev1, stage_stream));
    MRT_STAGE_MARK();                                   // kernels enqueued
    const mrt::HitOut* res = static_cast<const mrt::HitOut*>(mrt_stage_host_buffer_(ctx, out_bytes + sizeof(mrt::HitOut)));
    if (!res) return MRT_ERR_HIP;
    STAGE_COPY(const_cast<mrt::HitOut*>(res), d_out.p, out_bytes + sizeof(mrt::HitOut), hipMemcpyDeviceToHost);
    MRT_STAGE_MARK();                                   // results in page-locked host memory
    {
        uint32_t switched = 0;
        std::memcpy(&switched, res + n_rays, sizeof switched);
        mrt_stage_set_t_switches_(ctx, switched);
    }
    float kernel_ms = 0.f;
    STAGE_TRY(hipEventElapsedTime(&kernel_ms, ev0, ev1));
    mrt_stage_set_kernel_ms_(ctx, (double)kernel_ms);
    scatter_rows(n_rays, out, (size_t)n_rays * sizeof(mrt_cl_intersect), [=](int begin, int end) {
        for (int i = begin; i < end; ++i) {
            const mrt::HitOut& h = res[(size_t)i];
            int id;
            std::memcpy(&id, &h.n_id.w, 4);
            mrt_cl_intersect& x = out[i];
            if (id < 0) {
                // kernel.cl:361-366: a miss writes these three fields only.
                x.t_ = -1.0f; x.intersects = 0; x.shininess = 0.0f;
            } else {
                const mrt_cl_object& o = objects[id];
                const bool tri = o.is_triangle != 0;
                x.location.s[0] = h.loc_t.x; x.location.s[1] = h.loc_t.y; x.location.s[2] = h.loc_t.z;
                x.normal.s[0] = h.n_id.x; x.normal.s[1] = h.n_id.y; x.normal.s[2] = h.n_id.z;
                x.intersects = 1;
                x.t_ = h.loc_t.w;
                x.shininess = tri ? o.triObj.shininess : o.circleObj.shininess;
                const mrt_cl_float3& col = tri ? o.triObj.color : o.circleObj.color;
                x.color.s[0] = col.s[0]; x.color.s[1] = col.s[1]; x.color.s[2] = col.s[2];
                x.is_triangle = tri ? 1 : 0;
                x.is_circle = tri ? 0 : 1;
                x.dummy2 = x.shininess;   // kernel.cl:377
            }
        }
    });
    MRT_STAGE_MARK();                                   // ... and in the caller's records
#ifdef MRT_STAGE_TIMING
    {
        auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(stage_t[b] - stage_t[a]).count(); };
        std::fprintf(stderr, "stage_intersect host: pack %.0f, uploads %.0f, launches %.0f, wait + download %.0f (kernels %.0f), scatter %.0f us\n",
                     us(0, 1), us(1, 2), us(2, 3), us(3, 4), kernel_ms * 1e3, us(4, 5));
    }
#endif
#undef MRT_STAGE_MARK
    return MRT_OK;
} MRT_GUARD((void)mrt_stage_set_error_(ctx, code_, what_))

double mrt_stage_last_kernel_ms(const mrt_ctx* ctx) { return ctx ? mrt_stage_get_kernel_ms_(ctx) : 0.0; }

int mrt_stage_flat_shader(mrt_ctx* ctx, const mrt_cl_intersect* xsects, int n, mrt_cl_rgb background,
                          mrt_cl_rgb* out) try {
    return shade_common(ctx, xsects, n, background, nullptr, nullptr, nullptr, 0, out, true);
} MRT_GUARD((void)mrt_stage_set_error_(ctx, code_, what_))

int mrt_stage_phong_shader(mrt_ctx* ctx, const mrt_cl_intersect* xsects, const mrt_cl_phong* phong, int n,
                           mrt_cl_rgb background, const mrt_cl_cam* camera, const mrt_cl_light* lights,
                           int n_lights, mrt_cl_rgb* out) try {
    return shade_common(ctx, xsects, n, background, phong, camera, lights, n_lights, out, false);
} MRT_GUARD((void)mrt_stage_set_error_(ctx, code_, what_))

}  // extern "C"
