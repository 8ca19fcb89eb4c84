/*
 * mrt_oracle.h -- CPU restatement of the reference's render arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against and the CPU baseline bench.py times; nothing in the product path
 * (libmrt_hip, the mrraytracer CLI) links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or fixtures
 * for this path, and its code may not be executed in this build, so this
 * restatement is pinned only by (a) analytic known-answer values derived from
 * the shipped scene files and the formulas, and (b) agreement with an
 * independently written float64 numpy model (oracle/specmodel.py).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout, directory "Ray Tracing/").  Arithmetic is FP32 in the
 * operation order of the OpenCL C source, compiled with -ffp-contract=off, so
 * each float operation rounds once exactly where the source has an operator.
 */
#ifndef MRT_ORACLE_H
#define MRT_ORACLE_H

#include "mrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* camera::camera(eye, view, up), gfxraytrace.hpp:881-889 (double precision):
 * w = -(view/|view|), u = (up x w)/|up x w|, v = w x u. */
void mrt_oracle_camera_basis(const double view[3], const double up[3], double u[3], double v[3],
                             double w[3]);

/* Stage restatements on the reference's AoS structs (one call = one kernel
 * launch over n work-items). */
void mrt_oracle_uv(const mrt_cl_float3 *pos, const mrt_cl_vp *vp, int n, mrt_cl_float2 *out);
void mrt_oracle_ortho_viewrays(const mrt_cl_cam *c, const mrt_cl_float2 *uv, int n, mrt_cl_ray *rays);
void mrt_oracle_persp_viewrays(const mrt_cl_cam *c, const mrt_cl_float2 *uv, int n, float focal,
                               mrt_cl_ray *rays);
void mrt_oracle_intersect(const mrt_cl_object *objects, int n_objects, const mrt_cl_ray *rays,
                          int n_rays, float t_upper, float t_lower, mrt_cl_intersect *out,
                          int32_t *ids_out /* may be NULL */);
void mrt_oracle_flat_shader(const mrt_cl_intersect *x, int n, mrt_cl_rgb background, mrt_cl_rgb *out);
void mrt_oracle_phong_shader(const mrt_cl_intersect *x, const mrt_cl_phong *ph, int n,
                             mrt_cl_rgb background, const mrt_cl_cam *cam, const mrt_cl_light *lights,
                             int n_lights, mrt_cl_rgb *out);

/* hdr_to_byte, gfximage.hpp:46-51: unsigned(x * 255.0f).  NaN maps to 0
 * (the reference asserts instead). */
uint8_t mrt_oracle_to_byte(float x);

/*
 * Whole frame = scene::render() (gfxraytrace.hpp:830-878) with the kernels'
 * global size equal to the pixel count.  Rows [row_begin, row_end) only;
 * outputs are indexed by absolute pixel (y*xres + x) and any may be NULL:
 *   rgba  4 bytes/pixel (A = 255)      ids   winning object id, -1 = miss
 *   tval  intersect.t_ (-1 for miss)   rgbf  3 floats/pixel, shaded colour
 * n_threads <= 1 runs on the calling thread.
 */
int mrt_oracle_render(const mrt_scene *s, uint32_t row_begin, uint32_t row_end, int n_threads,
                      uint8_t *rgba, int32_t *ids, float *tval, float *rgbf);

#ifdef MRT_ORACLE_ULP_LATITUDE
/* libmrt_oracle_ulp.so only (never the oracle): how the results of the reference's built-ins are
 * moved inside the limits OpenCL 1.2 gives a device -- 0 pseudo-randomly, 1 all up, 2 all down.
 * Set before a render, not during one. */
void mrt_oracle_set_ulp_mode(int mode);
#endif

#ifdef __cplusplus
}
#endif
#endif
