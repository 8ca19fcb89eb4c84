// mrt_records.hip -- per-view prefilter records of trace_kernel / cull_trace_kernel, built on the GPU.
//
// The reference re-packs every object on the host for every frame
// (scene::intersect, gfxraytrace.hpp:810-828).  Here the geometry stays on the
// GPU and a change of view (mrt_set_camera) only re-derives, in one kernel per
// primitive kind, the records the render kernels filter with:
//
//   triangles   three edge functions of the pixel coordinate and a bounding
//               circle of the region they pass;
//   spheres     an ellipse around the pixel-space bounding box of the outline.
//
// Everything here is double precision, one thread per primitive: the work is
// N-sized (microseconds), and the margins derived here are what makes the FP32
// filters in the render kernels conservative, so they are not the place to save
// arithmetic.
#include "mrt.h"
#include "mrt_kernels.hpp"

#include <hip/hip_runtime.h>

#include <math.h>

namespace mrt {

namespace {

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 add(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 mul(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ D3 absv(D3 a) { return d3(fabs(a.x), fabs(a.y), fabs(a.z)); }
// Sum of the absolute values of the six products of a 3x3 determinant.
__device__ __forceinline__ double permabs(D3 a, D3 b, D3 c) {
    return a.x * (b.y * c.z + b.z * c.y) + a.y * (b.x * c.z + b.z * c.x) + a.z * (b.x * c.y + b.y * c.x);
}
__device__ __forceinline__ double maxabs(D3 v) { return fmax(fabs(v.x), fmax(fabs(v.y), fabs(v.z))); }

constexpr float kAlways = 1e30f;

// The relative rounding budget K assumes normal numbers.  A product that lands in the denormal
// range is off by up to 2^-150 absolutely instead, and the factor it is multiplied by next
// scales that; 64 such errors times the largest factor in play is far more than a determinant
// of these four vectors can collect.
__device__ __forceinline__ double tiny_abs(D3 a, D3 b, D3 c, D3 d) {
    return 64.0 * 0x1p-149 * (1.0 + fmax(fmax(maxabs(a), maxabs(b)), fmax(maxabs(c), maxabs(d))));
}

// One edge function A*x + B*y + C of a triangle record.
struct Edge { float A, B, C; };

__device__ Edge make_edge(double a, double b, double c, double margin, double xres, double yres) {
    const double g = hypot(a, b);
    double A, B, C;
    if (g > 0.0 && isfinite(g)) {
        // Unit gradient: the value is a signed distance in pixels.
        A = a / g; B = b / g; C = (c + margin) / g + 0.0625;
    } else if (isfinite(c)) {
        A = 0; B = 0; C = (c + margin >= 0.0) ? 1.0 : -1.0;
    } else {
        A = 0; B = 0; C = 1.0;
    }
    // Rounding of the three coefficients to float and of the two fused
    // multiply-adds (plus up to seven additions along a lane's row) in the kernel:
    // (3 + 2 + 7) half-ulps of the largest magnitude in play, 6 of the 16 budgeted.
    C += 16.0 * (1.0 / 16777216.0) * (fabs(A) * xres + fabs(B) * yres + fabs(C));
    if (!(fabs(C) < 1e30)) C = (C > 0 || C != C) ? 1e30 : -1e30;
    Edge e;
    e.A = (float)A; e.B = (float)B; e.C = (float)C;
    return e;
}

__device__ Edge const_edge(float c) { Edge e; e.A = 0.f; e.B = 0.f; e.C = c; return e; }

// Triangle record as the kernels read it: (A0,B0,C0,A1) (B1,C1,A2,B2) (C2,C0',C1',C2').
// The last three floats are the block-origin constants of the opt-in 4x4 block
// reject: C'_k = C_k + 3*max(A_k,0) + 3*max(B_k,0), rounded up.
__device__ void store_record(float4* __restrict__ rec, const Edge e[3]) {
    float cb[3];
    for (int k = 0; k < 3; ++k) {
        const double span = 3.0 * fmax((double)e[k].A, 0.0) + 3.0 * fmax((double)e[k].B, 0.0);
        double c = (double)e[k].C + span;
        c += 4.0 / 16777216.0 * (fabs(c) + span) + 1e-30;
        cb[k] = (float)c;
        if ((double)cb[k] < c) cb[k] = nextafterf(cb[k], INFINITY);
        if (!(fabsf(cb[k]) < 1e30f)) cb[k] = cb[k] > 0 ? 1e30f : -1e30f;
    }
    rec[0] = make_float4(e[0].A, e[0].B, e[0].C, e[1].A);
    rec[1] = make_float4(e[1].B, e[1].C, e[2].A, e[2].B);
    rec[2] = make_float4(e[2].C, cb[0], cb[1], cb[2]);
}

// ---------------------------------------------------------------------------
// Edge functions of one triangle.
//
// For a pinhole camera the ray direction is affine in the pixel (px, py) and
// the origin is fixed; for the orthographic camera it is the other way round
// (kernel.cl:130-147 with the pixel grid of gfxraytrace.hpp:850-852).  The
// three numerators Cramer's rule divides by the common determinant
// (kernel.cl:187-195, :208-213) are therefore affine functions of (px, py).
// With s the sign that makes a valid hit (t > 0) have a positive determinant,
// a pixel can only hit if  E0 = s*U >= 0, E1 = s*V >= 0, E2 = s*(Det-U-V) >= 0
// -- up to the rounding error of the FP32 evaluation the exact path performs.
// That error is bounded here term by term and added to each edge's constant,
// so the filter passes every pair the exact arithmetic can accept.  Triangles
// whose sign cannot be decided (eye in the triangle's plane, ray parallel to
// it) get a record that always passes.
// ---------------------------------------------------------------------------
__device__ void triangle_edges(const ViewDerived& w, D3 a, D3 b, D3 c, Edge e[3]) {
    const double K = 32.0 / 16777216.0;   // 32 ulp(1)/2: generous per-term rounding budget
    const D3 eye = d3(w.eye[0], w.eye[1], w.eye[2]), cu = d3(w.u[0], w.u[1], w.u[2]),
             cv = d3(w.v[0], w.v[1], w.v[2]), cw = d3(w.w[0], w.w[1], w.w[2]);
    const bool persp = w.persp != 0;
    // Affine pixel dependence: persp direction / ortho origin = V0 + Vx*px + Vy*py.
    const D3 Vx = mul(cu, w.ux1), Vy = mul(cv, w.uy1);
    const D3 span0 = add(mul(cu, w.ux0), mul(cv, w.uy0));
    const D3 V0 = persp ? add(mul(cw, -w.focal), span0) : span0;
    const D3 spanabs = add(mul(absv(cu), w.mux), mul(absv(cv), w.muy));
    const D3 vabs = persp ? add(mul(absv(cw), w.focal), spanabs) : spanabs;
    const D3 s1 = sub(a, b), s2 = sub(a, c);
    const D3 n = cross(s1, s2);
    e[0] = e[1] = e[2] = const_edge(kAlways);
    if (persp) {
        const D3 sol = sub(a, eye);
        const double T = dot(sol, n);
        const double dT = K * permabs(absv(s1), absv(s2), absv(sol));
        if (!(fabs(T) > dT)) return;
        const double sg = T > 0 ? 1.0 : -1.0;
        const D3 P0 = cross(sol, s2), P1 = cross(s1, sol);
        const D3 P2 = sub(sub(n, P0), P1);
        const double dD = K * permabs(absv(s1), absv(s2), vabs);
        const double maxD = dot(absv(n), vabs);
        // A quotient below half the smallest denormal rounds to -0, which the reference's
        // "beta < 0" lets through: a numerator within 2^-148 |det| of zero must pass too.
        const double zero_band = maxD * 0x1p-148 + tiny_abs(s1, s2, sol, vabs);
        const double dU = K * permabs(absv(sol), absv(s2), vabs) + zero_band;
        const double dV = K * permabs(absv(s1), absv(sol), vabs) + zero_band;
        const double m2 = dU + dV + 2 * dD + K * maxD + zero_band;
        e[0] = make_edge(sg * dot(P0, Vx), sg * dot(P0, Vy), sg * dot(P0, V0), dU, w.xres, w.yres);
        e[1] = make_edge(sg * dot(P1, Vx), sg * dot(P1, Vy), sg * dot(P1, V0), dV, w.xres, w.yres);
        e[2] = make_edge(sg * dot(P2, Vx), sg * dot(P2, Vy), sg * dot(P2, V0), m2, w.xres, w.yres);
    } else {
        const D3 d = mul(cw, -1.0);
        const double Det = dot(d, n);
        const double dD = K * permabs(absv(s1), absv(s2), absv(d));
        if (!(fabs(Det) > dD)) return;
        const double sg = Det > 0 ? 1.0 : -1.0;
        const D3 solabs = add(absv(a), vabs);
        const D3 Q0 = cross(s2, d), Q1 = cross(d, s1);
        const D3 a0 = sub(a, V0);   // sol = a0 - Vx*px - Vy*py
        // as above: quotients that round to -0, products in the denormal range
        const double zero_band = fabs(Det) * 0x1p-148 + tiny_abs(s1, s2, solabs, absv(d));
        const double dU = K * permabs(solabs, absv(s2), absv(d)) + zero_band;
        const double dV = K * permabs(absv(s1), solabs, absv(d)) + zero_band;
        const double m2 = dU + dV + 2 * dD + K * fabs(Det) + zero_band;
        const double u0 = dot(a0, Q0), ux = -dot(Vx, Q0), uy = -dot(Vy, Q0);
        const double v0 = dot(a0, Q1), vx = -dot(Vx, Q1), vy = -dot(Vy, Q1);
        e[0] = make_edge(sg * ux, sg * uy, sg * u0, dU, w.xres, w.yres);
        e[1] = make_edge(sg * vx, sg * vy, sg * v0, dV, w.xres, w.yres);
        e[2] = make_edge(sg * (-ux - vx), sg * (-uy - vy), sg * (Det - u0 - v0), m2, w.xres, w.yres);
    }
}

// ---------------------------------------------------------------------------
// Bounding circle of a triangle's prefilter region, (xc, yc, R^2, -).
//
// The region the three edge functions pass is the triangle whose sides are the
// lines E_k = 0 of the record exactly as stored (floats, margins included).
// Its corners are the pairwise intersections of those lines; the smallest
// circle around them (diameter on the longest side when the third corner lies
// inside, else the circumcircle), grown by a margin for the FP32 evaluation in
// the kernel, therefore contains every pixel the edge test can pass.  Records
// whose region is unbounded, degenerate or cannot be bounded reliably get a
// circle every pixel passes.
// ---------------------------------------------------------------------------
__device__ float4 bounding_circle(const Edge e[3], double xres, double yres) {
    const float4 always = make_float4(0.f, 0.f, 1e30f, 0.f);
    const double A[3] = {e[0].A, e[1].A, e[2].A}, B[3] = {e[0].B, e[1].B, e[2].B}, C[3] = {e[0].C, e[1].C, e[2].C};
    double px[3], py[3];
    for (int k = 0; k < 3; ++k) {
        const int i = k, j = (k + 1) % 3, o = (k + 2) % 3;
        const double det = A[i] * B[j] - A[j] * B[i];
        if (!(fabs(det) > 1e-9)) return always;
        px[k] = (B[i] * C[j] - B[j] * C[i]) / det;
        py[k] = (A[j] * C[i] - A[i] * C[j]) / det;
        // the corner must satisfy the third inequality, or the region is not a bounded triangle
        const double third = A[o] * px[k] + B[o] * py[k] + C[o];
        if (!(third >= -1e-6 * (fabs(A[o] * px[k]) + fabs(B[o] * py[k]) + fabs(C[o]) + 1.0))) return always;
        if (!isfinite(px[k]) || !isfinite(py[k])) return always;
    }
    // smallest enclosing circle of three points
    double best_d2 = -1; int bi = 0, bj = 1;
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3;
        const double d2 = (px[i] - px[j]) * (px[i] - px[j]) + (py[i] - py[j]) * (py[i] - py[j]);
        if (d2 > best_d2) { best_d2 = d2; bi = i; bj = j; }
    }
    const int bk = 3 - bi - bj;
    double cx = 0.5 * (px[bi] + px[bj]), cy = 0.5 * (py[bi] + py[bj]);
    double r = 0.5 * sqrt(best_d2);
    const double dk = hypot(px[bk] - cx, py[bk] - cy);
    if (dk > r) {
        // circumcircle
        const double ax = px[0], ay = py[0], bx = px[1], by = py[1], cx2 = px[2], cy2 = py[2];
        const double d = 2.0 * (ax * (by - cy2) + bx * (cy2 - ay) + cx2 * (ay - by));
        if (fabs(d) > 1e-300) {
            const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx2 * cx2 + cy2 * cy2;
            cx = (a2 * (by - cy2) + b2 * (cy2 - ay) + c2 * (ay - by)) / d;
            cy = (a2 * (cx2 - bx) + b2 * (ax - cx2) + c2 * (bx - ax)) / d;
        } else {
            cx = (px[0] + px[1] + px[2]) / 3.0; cy = (py[0] + py[1] + py[2]) / 3.0;
        }
        r = 0.0;
        for (int i = 0; i < 3; ++i) r = fmax(r, hypot(px[i] - cx, py[i] - cy));
    }
    if (!isfinite(cx) || !isfinite(cy) || !isfinite(r)) return always;
    // FP32 evaluation: x - xc and y - yc are rounded at the magnitude of the larger
    // operand, then squared; cover that and the rounding of the record itself.
    const double reach = fabs(cx) + fabs(cy) + xres + yres + r;
    r += 0.5 + 32.0 / 16777216.0 * reach;
    const double r2 = r * r * (1.0 + 1e-5);
    if (!(r2 < 1e29) || fabs(cx) > 1e15 || fabs(cy) > 1e15) return always;
    return make_float4((float)cx, (float)cy, (float)r2, 0.f);
}

}  // namespace

// One thread per triangle slot of the padded record arrays: edge record + bounding circle for
// the real triangles, records no pixel passes for the padding.
__global__ __launch_bounds__(128) void build_triangle_records(const ViewDerived w, const float4* __restrict__ tri_exact,
                                                              uint32_t n_tri, uint32_t n_tri_padded,
                                                              float4* __restrict__ tri_filter,
                                                              float4* __restrict__ tri_circle) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tri_padded) return;
    Edge e[3];
    if (k >= n_tri) {
        e[0] = e[1] = e[2] = const_edge(-kAlways);
        store_record(tri_filter + (size_t)k * 3, e);
        tri_circle[k] = make_float4(0.f, 0.f, -1.0f, 0.f);
        return;
    }
    const float4 va = tri_exact[(size_t)k * 3 + 0], vb = tri_exact[(size_t)k * 3 + 1], vc = tri_exact[(size_t)k * 3 + 2];
    triangle_edges(w, d3(va.x, va.y, va.z), d3(vb.x, vb.y, vb.z), d3(vc.x, vc.y, vc.z), e);
    store_record(tri_filter + (size_t)k * 3, e);
    tri_circle[k] = bounding_circle(e, w.xres, w.yres);
}

// ---------------------------------------------------------------------------
// Sphere records (sx, ox, sy, oy): a pixel (x, y) can only hit the sphere if
// (x*sx + ox)^2 + (y*sy + oy)^2 <= 1.  The ellipse is the one circumscribing
// the pixel-space bounding box of the sphere's outline (tangent rays from the
// eye for the pinhole camera, centre +- r for the orthographic one), grown by a
// margin that covers the cancellation error of the FP32 discriminant b*b - a*c
// the exact path evaluates (kernel.cl:248-254).  A sphere whose outline is
// unbounded or cannot be bounded reliably (it reaches the eye plane, or the
// camera basis is not orthonormal) gets a record every pixel passes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void build_sphere_records(const ViewDerived w, const float4* __restrict__ sph,
                                                            uint32_t n_sph, uint32_t n_sph_padded,
                                                            float4* __restrict__ sph_filter) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sph_padded) return;
    const float4 always = make_float4(0.f, 0.f, 0.f, 0.f);   // 0 + 0 <= 1
    const float4 never = make_float4(0.f, 2.f, 0.f, 0.f);    // 4 > 1
    if (k >= n_sph) { sph_filter[k] = never; return; }
    const float4 sp = sph[k];
    const D3 c = d3(sp.x, sp.y, sp.z);
    const double r = fabs((double)sp.w);
    const D3 eye = d3(w.eye[0], w.eye[1], w.eye[2]), cu = d3(w.u[0], w.u[1], w.u[2]),
             cv = d3(w.v[0], w.v[1], w.v[2]), cw = d3(w.w[0], w.w[1], w.w[2]);
    const double K = 64.0 / 16777216.0;
    const double f = w.focal;
    float4 out = always;
    if (w.orthonormal && r > 0 && isfinite(r) && w.ux1 > 0 && w.uy1 > 0) {
        double lo_u = 0, hi_u = 0, lo_v = 0, hi_v = 0, margin = 0;
        bool ok = true;
        if (w.persp) {
            const D3 oc = sub(c, eye);
            const double X = dot(oc, cu), Y = dot(oc, cv), Z = -dot(oc, cw);
            const double rr = r * 1.001;
            ok = f > 0 && Z > rr * 1.001 + 1e-30;
            if (ok) {
                const double den = Z * Z - rr * rr;
                const double sx = rr * sqrt(X * X + den), sy = rr * sqrt(Y * Y + den);
                lo_u = f * (X * Z - sx) / den; hi_u = f * (X * Z + sx) / den;
                lo_v = f * (Y * Z - sy) / den; hi_v = f * (Y * Z + sy) / den;
                const double dist = sqrt(dot(oc, oc));
                const double dmax2 = f * f + w.mux * w.mux + w.muy * w.muy;
                // b*b and a*c (kernel.cl:250-254) must stay clear of the denormal range, where the
                // relative error model of the margin does not hold
                if (f * f * r * r < 1e-30) ok = false;
                // error of b*b - a*c over its gradient at the outline, in pixels
                margin = 1.0 + K * dist * dmax2 / (r * fmin(w.ux1, w.uy1) * f);
            }
        } else {
            const double U = dot(c, cu), V = dot(c, cv);
            lo_u = U - r * 1.001; hi_u = U + r * 1.001;
            lo_v = V - r * 1.001; hi_v = V + r * 1.001;
            const double reach = sqrt(dot(c, c)) + w.mux + w.muy + r;
            margin = 1.0 + K * reach * reach / (r * fmin(w.ux1, w.uy1));
            if (r * r < 1e-30) ok = false;      // as for the pinhole camera
        }
        if (ok && isfinite(margin) && margin < 1e6) {
            const double px0 = (lo_u - w.ux0) / w.ux1, px1 = (hi_u - w.ux0) / w.ux1;
            const double py0 = (lo_v - w.uy0) / w.uy1, py1 = (hi_v - w.uy0) / w.uy1;
            const double xc = 0.5 * (px0 + px1), yc = 0.5 * (py0 + py1);
            const double hx = 0.5 * fabs(px1 - px0), hy = 0.5 * fabs(py1 - py0);
            const double ax = 1.41421357 * hx + margin, ay = 1.41421357 * hy + margin;
            if (isfinite(xc) && isfinite(yc) && isfinite(ax) && isfinite(ay)) {
                // grow once more for the FP32 evaluation of the two squares in the kernel
                const double grow = 1.0 + 1e-5 * (1.0 + (fabs(xc) + w.xres) / ax + (fabs(yc) + w.yres) / ay);
                const double sxx = 1.0 / (ax * grow), syy = 1.0 / (ay * grow);
                out = make_float4((float)sxx, (float)(-xc * sxx), (float)syy, (float)(-yc * syy));
                if (!isfinite(out.x) || !isfinite(out.y) || !isfinite(out.z) || !isfinite(out.w)) out = always;
            }
        }
    }
    sph_filter[k] = out;
}

}  // namespace mrt
