// mrt_device.hpp -- device-side arithmetic shared by the render kernel and the
// stage kernels.
//
// Two kinds of code live here and must not be mixed up:
//
//  * "exact" functions restate the reference kernels' FP32 arithmetic
//    (kernel.cl:123-445) operator for operator.  The translation unit is built
//    with -ffp-contract=off and the pragma below, so a*b+c is a rounded
//    multiply followed by a rounded add, as on an OpenCL device that does not
//    contract; division and sqrtf are the correctly rounded forms hipcc emits
//    by default.  Every hit/miss decision, t, location, normal and colour that
//    reaches a pixel comes from these functions.
//
//  * the conservative triangle prefilter (in mrt_kernels.hip) uses explicit
//    __builtin_fmaf and only ever decides "this pixel cannot hit this
//    triangle"; it never produces a value that is written out.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mrt {

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 mk3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 add3(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 scale3(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 div3(F3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }

// determinant3x3, kernel.cl:149-155.
__device__ __forceinline__ float det3(F3 s1, F3 s2, F3 s3) {
    float det = s1.x * (s2.y * s3.z - s2.z * s3.y);
    det -= s1.y * (s2.x * s3.z - s2.z * s3.x);
    det += s1.z * (s2.x * s3.y - s2.y * s3.x);
    return det;
}

// cl_cross, kernel.cl:157-164.
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// cl_dot, kernel.cl:166-172 (the built-in dot at :423 sums in the same order).
__device__ __forceinline__ float dot3(F3 a, F3 b) {
    float p = a.x * b.x;
    p += a.y * b.y;
    p += a.z * b.z;
    return p;
}

// cl_magnitude / cl_normalize, kernel.cl:174-185; also stands for the built-in
// normalize at :227.
__device__ __forceinline__ F3 normalize3(F3 a) {
    float m = a.x * a.x;
    m += a.y * a.y;
    m += a.z * a.z;
    return div3(a, sqrtf(m));
}

// One component of kernel `uv`, kernel.cl:123-128: the 0.5 literal makes the
// right-hand side double; one rounding to float on the store.
__device__ __forceinline__ float uv_component(float lo, float hi, float pos, float res) {
    return (float)((double)lo + ((double)(hi - lo) * ((double)pos + 0.5)) / (double)res);
}

// kernel.cl:130-138: the eye is overwritten; direction = -w.
__device__ __forceinline__ void ortho_ray(F3 u, F3 v, F3 w, float ux, float uy, F3& o, F3& d) {
    o.x = (u.x * ux) + (v.x * uy);
    o.y = (u.y * ux) + (v.y * uy);
    o.z = (u.z * ux) + (v.z * uy);
    d = mk3(-w.x, -w.y, -w.z);
}

// kernel.cl:140-147: direction = -w*f, += u*uv.x, += v*uv.y; not normalised.
__device__ __forceinline__ void persp_ray(F3 eye, F3 u, F3 v, F3 w, float focal, float ux, float uy,
                                          F3& o, F3& d) {
    o = eye;
    F3 dir = mk3(-w.x * focal, -w.y * focal, -w.z * focal);
    dir = add3(dir, scale3(u, ux));
    dir = add3(dir, scale3(v, uy));
    d = dir;
}

// tri_intersect, kernel.cl:197-235, reduced to its decision and t: Cramer's
// rule (solve, :187-195) and the inclusive range tests.
__device__ __forceinline__ bool tri_exact(F3 a, F3 b, F3 c, F3 o, F3 d, float t_upper, float t_lower,
                                          float& t_out) {
    F3 s1 = sub3(a, b);
    F3 s2 = sub3(a, c);
    F3 sol = sub3(a, o);
    float det = det3(s1, s2, d);
    float bx = det3(sol, s2, d) / det;
    float gy = det3(s1, sol, d) / det;
    float tz = det3(s1, s2, sol) / det;
    if (tz < t_lower || tz > t_upper) return false;
    if (gy < 0 || gy > 1) return false;
    if (bx < 0 || bx > 1 - gy) return false;
    t_out = tz;
    return true;
}

// Discriminant of circle_intersect, kernel.cl:248-254.
__device__ __forceinline__ float sphere_disc(F3 center, float radius, F3 o, F3 d, float& a_out,
                                             float& b_out) {
    F3 omc = sub3(o, center);
    float a = dot3(d, d);
    float b = dot3(d, omc);
    float radsqr = radius * radius;
    float c = dot3(omc, omc);
    c = c - radsqr;
    a_out = a;
    b_out = b;
    return b * b - a * c;
}

// circle_intersect, kernel.cl:237-303, reduced to its decision and t (near
// root only, strict range test).  two_roots tells which normal formula applies.
__device__ __forceinline__ bool sphere_exact(F3 center, float radius, F3 o, F3 d, float t_upper,
                                             float t_lower, float& t_out, bool& two_roots) {
    float a, b;
    float disc = sphere_disc(center, radius, o, d, a, b);
    if (disc < 0) return false;
    float t = -b;
    two_roots = disc > 0;
    if (two_roots) t -= sqrtf(disc);
    t /= a;
    if (t < t_upper && t > t_lower) { t_out = t; return true; }
    return false;
}

// Per-light body of kernel `phong_shader`, kernel.cl:408-431.
__device__ __forceinline__ void phong_light(F3 loc, F3 n, F3 color, float shininess, F3 eye, F3 lpos,
                                            F3 lcol, float kd, float ks, float& r, float& g, float& b) {
    F3 L = normalize3(sub3(lpos, loc));
    F3 V = normalize3(sub3(eye, loc));
    F3 H = normalize3(add3(V, L));
    float ndl = dot3(n, L);
    if (ndl < 0) ndl = 0;
    r += color.x * kd * ndl;
    g += color.y * kd * ndl;
    b += color.z * kd * ndl;
    float ndh = dot3(H, n);
    if (ndh < 0) ndh = 0;
    ndh = powf(ndh, shininess);   // powr(), kernel.cl:426
    r += ks * ndh * lcol.x;
    g += ks * ndh * lcol.y;
    b += ks * ndh * lcol.z;
}

__device__ __forceinline__ float clamp01(float x) {   // kernel.cl:434-439
    x = (x < 0) ? 0 : x;
    x = (x > 1) ? 1 : x;
    return x;
}

// hdr_to_byte, gfximage.hpp:46-51: unsigned(x * 255.0f); NaN gives 0.
__device__ __forceinline__ uint32_t to_byte(float x) {
    float s = x * 255.0f;
    return (s == s) ? ((uint32_t)s & 0xFFu) : 0u;
}

}  // namespace mrt
