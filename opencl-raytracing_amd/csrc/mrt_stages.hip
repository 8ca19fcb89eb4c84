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
#include "mrt_device.hpp"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#pragma clang fp contract(off)

extern "C" int mrt_stage_set_error_(mrt_ctx* ctx, int code, const char* msg);
extern "C" int mrt_stage_device_(const mrt_ctx* ctx);
extern "C" void* mrt_stage_buffer_(mrt_ctx* ctx, int slot, size_t bytes);
extern "C" void* mrt_ctx_stream_(const mrt_ctx* ctx);

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

// Compact object record: kind in geo[0].w (0 sphere, 1 triangle).
//   sphere   geo[0] = (cx, cy, cz, kind) geo[1] = (r, -, -, -)
//   triangle geo[0] = (ax, ay, az, kind) geo[1] = (bx, by, bz, -) geo[2] = (cx, cy, cz, -)
struct HitOut { float4 loc_t; float4 n_id; };

__global__ void stage_intersect_kernel(const float4* __restrict__ geo, int n_objects,
                                       const float4* __restrict__ rays, int n_rays, float t_upper,
                                       float t_lower, HitOut* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    const float4 ro = rays[2 * (size_t)i], rd = rays[2 * (size_t)i + 1];
    const F3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
    float temp_max = t_upper;   // kernel.cl:307
    int best = -1;
    for (int j = 0; j < n_objects; ++j) {
        const float4 g0 = geo[3 * (size_t)j], g1 = geo[3 * (size_t)j + 1], g2 = geo[3 * (size_t)j + 2];
        float t;
        bool hit;
        if (g0.w != 0.0f) {
            hit = tri_exact(mk3(g0.x, g0.y, g0.z), mk3(g1.x, g1.y, g1.z), mk3(g2.x, g2.y, g2.z), o, d,
                            temp_max, t_lower, t);
        } else {
            bool two;
            hit = sphere_exact(mk3(g0.x, g0.y, g0.z), g1.x, o, d, temp_max, t_lower, t, two);
        }
        if (hit && t > t_lower && t < temp_max) { temp_max = t; best = j; }   // kernel.cl:326,344
    }
    HitOut h;
    h.loc_t = make_float4(0.f, 0.f, 0.f, -1.0f);
    h.n_id = make_float4(0.f, 0.f, 0.f, __int_as_float(best));
    if (best >= 0) {
        const float4 g0 = geo[3 * (size_t)best], g1 = geo[3 * (size_t)best + 1], g2 = geo[3 * (size_t)best + 2];
        const float t = temp_max;
        const F3 loc = add3(o, scale3(d, t));
        F3 n;
        if (g0.w != 0.0f) {
            const F3 A = mk3(g0.x, g0.y, g0.z);
            n = normalize3(cross3(sub3(A, mk3(g1.x, g1.y, g1.z)), sub3(A, mk3(g2.x, g2.y, g2.z))));
        } else {
            const F3 c = mk3(g0.x, g0.y, g0.z);
            float a, b;
            const float disc = sphere_disc(c, g1.x, o, d, a, b);
            if (disc > 0) n = normalize3(div3(sub3(loc, c), g1.x));
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
    // Keep the caller's padding floats (the kernels never write .s[3]).
    STAGE_COPY(d_rays.p, rays_out, (size_t)n * 32, hipMemcpyHostToDevice);
    std::vector<mrt_cl_ray> tmp((size_t)n);
    if (persp)
        hipLaunchKernelGGL(mrt::stage_rays_kernel<MRT_PROJ_PERSP>, grid_for(n), dim3(256), 0, stage_stream, cam_of(camera),
                           (const float2*)d_uv.p, n, focal, (float4*)d_rays.p);
    else
        hipLaunchKernelGGL(mrt::stage_rays_kernel<MRT_PROJ_ORTHO>, grid_for(n), dim3(256), 0, stage_stream, cam_of(camera),
                           (const float2*)d_uv.p, n, focal, (float4*)d_rays.p);
    STAGE_TRY(hipGetLastError());
    STAGE_COPY(tmp.data(), d_rays.p, (size_t)n * 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            rays_out[i].origin.s[k] = tmp[(size_t)i].origin.s[k];
            rays_out[i].direction.s[k] = tmp[(size_t)i].direction.s[k];
        }
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
    std::vector<float4> res((size_t)n);
    STAGE_COPY(res.data(), d_o.p, (size_t)n * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        out[i].r = res[(size_t)i].x; out[i].g = res[(size_t)i].y; out[i].b = res[(size_t)i].z;
        // flat_shader never writes .a (kernel.cl:381-396); phong_shader copies
        // the whole background struct on a miss (kernel.cl:443) and leaves .a
        // alone on a hit.
        if (!flat && res[(size_t)i].w == 0.0f) out[i].a = background.a;
    }
    return MRT_OK;
}

}  // namespace

extern "C" {

int mrt_stage_uv(mrt_ctx* ctx, const mrt_cl_float3* positions, const mrt_cl_vp* viewport, int n,
                 mrt_cl_float2* uv_out) {
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
}

int mrt_stage_ortho_viewrays(mrt_ctx* ctx, const mrt_cl_cam* camera, const mrt_cl_float2* uv, int n,
                             mrt_cl_ray* rays_out) {
    return rays_common(ctx, camera, uv, n, 0.0f, rays_out, false);
}

int mrt_stage_persp_viewrays(mrt_ctx* ctx, const mrt_cl_cam* camera, const mrt_cl_float2* uv, int n,
                             float focal_length, mrt_cl_ray* rays_out) {
    return rays_common(ctx, camera, uv, n, focal_length, rays_out, true);
}

int mrt_stage_intersect(mrt_ctx* ctx, const mrt_cl_object* objects, int n_objects, const mrt_cl_ray* rays,
                        int n_rays, float t_upper_bound, float t_lower_bound, mrt_cl_intersect* out) {
    RestoreDevice restore_device;
    int rc = begin(ctx, n_rays, rays, out);
    if (rc != MRT_OK) return rc;
    if (n_objects < 0 || (n_objects > 0 && !objects))
        return mrt_stage_set_error_(ctx, MRT_ERR_INVALID, "bad object array");
    if (n_rays == 0) return MRT_OK;
    static_assert(sizeof(mrt_cl_object) == 144, "object must be 144 bytes");
    std::vector<float4> geo((size_t)n_objects * 3);
    for (int j = 0; j < n_objects; ++j) {
        const mrt_cl_object& o = objects[j];
        // kernel.cl:322: is_triangle selects the triangle test, anything else is a sphere.
        if (o.is_triangle) {
            geo[3 * (size_t)j] = make_float4(o.triObj.a.s[0], o.triObj.a.s[1], o.triObj.a.s[2], 1.0f);
            geo[3 * (size_t)j + 1] = make_float4(o.triObj.b.s[0], o.triObj.b.s[1], o.triObj.b.s[2], 0.f);
            geo[3 * (size_t)j + 2] = make_float4(o.triObj.c.s[0], o.triObj.c.s[1], o.triObj.c.s[2], 0.f);
        } else {
            geo[3 * (size_t)j] = make_float4(o.circleObj.center.s[0], o.circleObj.center.s[1], o.circleObj.center.s[2], 0.0f);
            geo[3 * (size_t)j + 1] = make_float4(o.circleObj.radius, 0.f, 0.f, 0.f);
            geo[3 * (size_t)j + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    hipStream_t stage_stream = (hipStream_t)mrt_ctx_stream_(ctx);
    DevBuf d_geo, d_rays, d_out;
    STAGE_BUF(d_geo, 0, geo.size() * sizeof(float4));
    STAGE_BUF(d_rays, 1, (size_t)n_rays * 32);
    STAGE_BUF(d_out, 2, (size_t)n_rays * sizeof(mrt::HitOut));
    if (!geo.empty()) STAGE_COPY(d_geo.p, geo.data(), geo.size() * sizeof(float4), hipMemcpyHostToDevice);
    STAGE_COPY(d_rays.p, rays, (size_t)n_rays * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mrt::stage_intersect_kernel, grid_for(n_rays), dim3(256), 0, stage_stream, (const float4*)d_geo.p,
                       n_objects, (const float4*)d_rays.p, n_rays, t_upper_bound, t_lower_bound,
                       (mrt::HitOut*)d_out.p);
    STAGE_TRY(hipGetLastError());
    std::vector<mrt::HitOut> res((size_t)n_rays);
    STAGE_COPY(res.data(), d_out.p, (size_t)n_rays * sizeof(mrt::HitOut), hipMemcpyDeviceToHost);
    for (int i = 0; i < n_rays; ++i) {
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
    return MRT_OK;
}

int mrt_stage_flat_shader(mrt_ctx* ctx, const mrt_cl_intersect* xsects, int n, mrt_cl_rgb background,
                          mrt_cl_rgb* out) {
    return shade_common(ctx, xsects, n, background, nullptr, nullptr, nullptr, 0, out, true);
}

int mrt_stage_phong_shader(mrt_ctx* ctx, const mrt_cl_intersect* xsects, const mrt_cl_phong* phong, int n,
                           mrt_cl_rgb background, const mrt_cl_cam* camera, const mrt_cl_light* lights,
                           int n_lights, mrt_cl_rgb* out) {
    return shade_common(ctx, xsects, n, background, phong, camera, lights, n_lights, out, false);
}

}  // extern "C"
