/*
 * mrt_oracle.c -- CPU restatement of the reference's render arithmetic.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see mrt_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * x86-64 SSE2 evaluates every float operator in IEEE binary32, so the value
 * of each expression below is the value of the corresponding OpenCL C
 * expression on a device that does not contract multiplies and adds.
 *
 * Reference files restated (directory "Ray Tracing/"): kernel.cl:123-445 for
 * the arithmetic, gfxraytrace.hpp:810-889 for the host glue, gfximage.hpp:46-51
 * for the byte conversion.
 */
#include "mrt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/*
 * The built-ins of the reference.  OpenCL C's `/`, sqrt(), normalize(), dot() and powr() are not
 * correctly rounded by rule: the OpenCL 1.2 full profile (section 7.4) lets a device be off by
 * 2.5 ulp for x / y, 3 ulp for sqrt, 2 + n ulp for normalize of an n-vector and 16 ulp for powr, and
 * bounds dot() by an absolute error that a few ulp of the result stay within.  The oracle takes
 * the correctly rounded division and sqrt of IEEE arithmetic and glibc's powf.  A second build of
 * this file (-DMRT_ORACLE_ULP_LATITUDE, libmrt_oracle_ulp.so, oracle/Makefile) moves the result of
 * every such operation by a whole number of ulps inside those limits -- pseudo-randomly from the
 * result's own bits, or all up, or all down (mrt_oracle_set_ulp_mode) -- to size how far that
 * freedom of a conforming device moves the reference's pixels (oracle/builtin_latitude.py).  That
 * build is never the oracle.  In the oracle itself the macros below are the plain operations.
 */
#ifdef MRT_ORACLE_ULP_LATITUDE
static int g_ulp_mode = 0;            /* 0 pseudo-random in [-k, k], 1 all +k, 2 all -k */
void mrt_oracle_set_ulp_mode(int mode) { g_ulp_mode = mode; }
static float ulp_move(float r, int k_max, uint32_t salt)
{
    uint32_t b;
    memcpy(&b, &r, 4);
    const uint32_t mag = b & 0x7FFFFFFFu;
    if (mag == 0 || mag >= 0x7F800000u) return r;        /* zero, infinity, NaN: as they are */
    int k;
    if (g_ulp_mode == 1) k = k_max;
    else if (g_ulp_mode == 2) k = -k_max;
    else {
        uint32_t h = (b ^ salt) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        k = (int)(h % (uint32_t)(2 * k_max + 1)) - k_max;
    }
    int64_t m = (int64_t)mag + k;                        /* k ulps in magnitude; stays finite and non-zero */
    if (m < 1) m = 1;
    if (m > 0x7F7FFFFF) m = 0x7F7FFFFF;
    b = (b & 0x80000000u) | (uint32_t)m;
    memcpy(&r, &b, 4);
    return r;
}
#define CL_DIV(a, b) ulp_move((a) / (b), 2, 0x9E3779B9u)
#define CL_SQRT(x) ulp_move(sqrtf(x), 3, 0x85EBCA6Bu)
#define CL_POWR(x, y) ulp_move(powf((x), (y)), 16, 0xC2B2AE35u)
#define CL_BUILTIN(r, k) ulp_move((r), (k), 0x27D4EB2Fu)     /* normalize(): 2 + 3 ulp; dot(): 5 ulp */
#else
#define CL_DIV(a, b) ((a) / (b))
#define CL_SQRT(x) sqrtf(x)
#define CL_POWR(x, y) powf((x), (y))
#define CL_BUILTIN(r, k) (r)
#endif

typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
static inline f3 f3_div(f3 a, float s) { return f3_make(CL_DIV(a.x, s), CL_DIV(a.y, s), CL_DIV(a.z, s)); }

/* determinant3x3, kernel.cl:149-155 (cofactor expansion along the first vector). */
static float det3(f3 s1, f3 s2, f3 s3)
{
    float det = s1.x * (s2.y * s3.z - s2.z * s3.y);
    det -= s1.y * (s2.x * s3.z - s2.z * s3.x);
    det += s1.z * (s2.x * s3.y - s2.y * s3.x);
    return det;
}

/* cl_cross, kernel.cl:157-164. */
static f3 cl_cross(f3 a, f3 b)
{
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* cl_dot, kernel.cl:166-172; the OpenCL built-in dot() used at kernel.cl:423
 * is given the same left-to-right summation. */
static float cl_dot(f3 a, f3 b)
{
    float p = a.x * b.x;
    p += a.y * b.y;
    p += a.z * b.z;
    return p;
}

/* cl_magnitude / cl_normalize, kernel.cl:174-185.  The built-in normalize()
 * used at kernel.cl:227 is given the same definition (vector / length):
 * cl_builtin_normalize below. */
static float cl_magnitude(f3 a)
{
    float m = a.x * a.x;
    m += a.y * a.y;
    m += a.z * a.z;
    return CL_SQRT(m);
}
static f3 cl_normalize(f3 a) { return f3_div(a, cl_magnitude(a)); }
static f3 cl_builtin_normalize(f3 a)
{
    f3 r = cl_normalize(a);
    return f3_make(CL_BUILTIN(r.x, 5), CL_BUILTIN(r.y, 5), CL_BUILTIN(r.z, 5));
}

typedef struct {
    int is_tri;
    f3 a, b, c;      /* triangle vertices                 */
    f3 center;       /* sphere                            */
    float radius;
    f3 color;
    float shininess;
} obj_t;

typedef struct {
    int hit;
    float t;
    f3 loc, n, color;
    float shininess;
    int is_tri;
} hit_t;

/* tri_intersect, kernel.cl:197-235: Cramer's rule on [a-b | a-c | d] x = a-o
 * (solve(), kernel.cl:187-195), inclusive range tests. */
static hit_t tri_intersect(const obj_t *tr, f3 o, f3 d, float t_upper, float t_lower)
{
    hit_t h;
    memset(&h, 0, sizeof h);
    h.t = -1.0f;
    f3 s1 = f3_sub(tr->a, tr->b);
    f3 s2 = f3_sub(tr->a, tr->c);
    f3 s3 = d;
    f3 sol = f3_sub(tr->a, o);
    float det = det3(s1, s2, s3);
    float bx = CL_DIV(det3(sol, s2, s3), det);
    float gy = CL_DIV(det3(s1, sol, s3), det);
    float tz = CL_DIV(det3(s1, s2, sol), det);
    if (tz < t_lower || tz > t_upper) return h;
    if (gy < 0 || gy > 1) return h;
    if (bx < 0 || bx > 1 - gy) return h;
    h.loc = f3_add(o, f3_scale(d, tz));
    h.n = cl_builtin_normalize(cl_cross(s1, s2));   /* normalize(), kernel.cl:227 */
    h.hit = 1;
    h.t = tz;
    h.shininess = tr->shininess;
    h.color = tr->color;
    h.is_tri = 1;
    return h;
}

/* circle_intersect, kernel.cl:237-303: near root only, strict range test. */
static hit_t circle_intersect(const obj_t *sp, f3 o, f3 d, float t_upper, float t_lower)
{
    hit_t h;
    memset(&h, 0, sizeof h);
    h.t = -1.0f;
    f3 omc = f3_sub(o, sp->center);
    float a = cl_dot(d, d);
    float b = cl_dot(d, omc);
    float radsqr = sp->radius * sp->radius;
    float c = cl_dot(omc, omc);
    c = c - radsqr;
    float disc = b * b - a * c;
    if (disc < 0) return h;
    float t = -b;
    if (disc > 0) {
        t -= CL_SQRT(disc);
        t = CL_DIV(t, a);
        if (t < t_upper && t > t_lower) {
            h.loc = f3_add(o, f3_scale(d, t));
            h.n = cl_normalize(f3_div(f3_sub(h.loc, sp->center), sp->radius));
        } else {
            return h;
        }
    } else {
        t = CL_DIV(t, a);
        if (t < t_upper && t > t_lower) {
            h.loc = f3_add(o, f3_scale(d, t));
            h.n = cl_normalize(f3_sub(h.loc, sp->center));
        } else {
            return h;
        }
    }
    h.hit = 1;
    h.t = t;
    h.shininess = sp->shininess;
    h.color = sp->color;
    h.is_tri = 0;
    return h;
}

/* Body of kernel `intersections`, kernel.cl:305-379: objects in array order,
 * accept iff hit && t > lower && t < current best (strict), shrink the bound. */
static int closest_hit(const obj_t *objs, int n, f3 o, f3 d, float t_upper, float t_lower, hit_t *out)
{
    float temp_max = t_upper;
    int best = -1;
    hit_t closest;
    memset(&closest, 0, sizeof closest);
    closest.t = -1.0f;
    for (int j = 0; j < n; ++j) {
        hit_t tmp = objs[j].is_tri ? tri_intersect(&objs[j], o, d, temp_max, t_lower)
                                   : circle_intersect(&objs[j], o, d, temp_max, t_lower);
        if (tmp.hit && tmp.t > t_lower && tmp.t < temp_max) {
            closest = tmp;
            temp_max = tmp.t;
            best = j;
        }
    }
    *out = closest;
    return best;
}

/* Per-pixel body of kernel `phong_shader`, kernel.cl:398-445 (hit case). */
typedef struct { f3 loc, color; } light_t;
static f3 phong_pixel(const hit_t *h, float ka, float kd, float ks, f3 amb, f3 eye,
                      const light_t *lights, int n_lights)
{
    float r = ka * amb.x, g = ka * amb.y, b = ka * amb.z;
    for (int j = 0; j < n_lights; ++j) {
        f3 L = cl_normalize(f3_sub(lights[j].loc, h->loc));
        f3 V = cl_normalize(f3_sub(eye, h->loc));
        f3 H = cl_normalize(f3_add(V, L));
        float ndl = cl_dot(h->n, L);
        if (ndl < 0) ndl = 0;
        r += h->color.x * kd * ndl;
        g += h->color.y * kd * ndl;
        b += h->color.z * kd * ndl;
        float ndh = CL_BUILTIN(cl_dot(H, h->n), 5);   /* dot(), kernel.cl:423 */
        if (ndh < 0) ndh = 0;
        ndh = CL_POWR(ndh, h->shininess);   /* powr(), kernel.cl:426 */
        r += ks * ndh * lights[j].color.x;
        g += ks * ndh * lights[j].color.y;
        b += ks * ndh * lights[j].color.z;
    }
    r = (r < 0) ? 0 : r;  r = (r > 1) ? 1 : r;
    g = (g < 0) ? 0 : g;  g = (g > 1) ? 1 : g;
    b = (b < 0) ? 0 : b;  b = (b > 1) ? 1 : b;
    return f3_make(r, g, b);
}

/* One component of kernel `uv`, kernel.cl:123-128.  The literal 0.5 is a
 * double, so everything to the right of `lo +` is evaluated in double and the
 * sum is rounded to float once, on the store. */
static float uv_component(float lo, float hi, float pos, float res)
{
    return (float)((double)lo + ((double)(hi - lo) * ((double)pos + 0.5)) / (double)res);
}

uint8_t mrt_oracle_to_byte(float x)
{
    if (!(x == x)) return 0;
    return (uint8_t)(unsigned)(x * 255.0f);
}

void mrt_oracle_camera_basis(const double view[3], const double up[3], double u[3], double v[3],
                             double w[3])
{
    double vm = sqrt(view[0] * view[0] + view[1] * view[1] + view[2] * view[2]);
    for (int i = 0; i < 3; ++i) w[i] = -(view[i] / vm);
    double c[3] = {up[1] * w[2] - up[2] * w[1], up[2] * w[0] - up[0] * w[2], up[0] * w[1] - up[1] * w[0]};
    double cm = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    for (int i = 0; i < 3; ++i) u[i] = c[i] / cm;
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = w[2] * u[0] - w[0] * u[2];
    v[2] = w[0] * u[1] - w[1] * u[0];
}

/* ---------------- stage restatements on the AoS structs ---------------- */

static inline f3 from_cl(const mrt_cl_float3 *p) { return f3_make(p->s[0], p->s[1], p->s[2]); }
static inline void to_cl(mrt_cl_float3 *p, f3 v) { p->s[0] = v.x; p->s[1] = v.y; p->s[2] = v.z; }

void mrt_oracle_uv(const mrt_cl_float3 *pos, const mrt_cl_vp *vp, int n, mrt_cl_float2 *out)
{
    for (int i = 0; i < n; ++i) {
        out[i].s[0] = uv_component(vp->left, vp->right, pos[i].s[0], vp->x_resolution);
        out[i].s[1] = uv_component(vp->bottom, vp->top, pos[i].s[1], vp->y_resolution);
    }
}

/* kernel.cl:130-138: origin = u*uv.x + v*uv.y (the eye is overwritten), direction = -w. */
static void ortho_ray(f3 u, f3 v, f3 w, float ux, float uy, f3 *o, f3 *d)
{
    o->x = (u.x * ux) + (v.x * uy);
    o->y = (u.y * ux) + (v.y * uy);
    o->z = (u.z * ux) + (v.z * uy);
    *d = f3_make(-w.x, -w.y, -w.z);
}

/* kernel.cl:140-147: origin = eye, direction = -w*f, += u*uv.x, += v*uv.y (not normalised). */
static void persp_ray(f3 eye, f3 u, f3 v, f3 w, float focal, float ux, float uy, f3 *o, f3 *d)
{
    *o = eye;
    f3 dir = f3_make(-w.x * focal, -w.y * focal, -w.z * focal);
    dir = f3_add(dir, f3_scale(u, ux));
    dir = f3_add(dir, f3_scale(v, uy));
    *d = dir;
}

void mrt_oracle_ortho_viewrays(const mrt_cl_cam *c, const mrt_cl_float2 *uv, int n, mrt_cl_ray *rays)
{
    for (int i = 0; i < n; ++i) {
        f3 o, d;
        ortho_ray(from_cl(&c->u), from_cl(&c->v), from_cl(&c->w), uv[i].s[0], uv[i].s[1], &o, &d);
        to_cl(&rays[i].origin, o);
        to_cl(&rays[i].direction, d);
    }
}

void mrt_oracle_persp_viewrays(const mrt_cl_cam *c, const mrt_cl_float2 *uv, int n, float focal,
                               mrt_cl_ray *rays)
{
    for (int i = 0; i < n; ++i) {
        f3 o, d;
        persp_ray(from_cl(&c->eye), from_cl(&c->u), from_cl(&c->v), from_cl(&c->w), focal,
                  uv[i].s[0], uv[i].s[1], &o, &d);
        to_cl(&rays[i].origin, o);
        to_cl(&rays[i].direction, d);
    }
}

void mrt_oracle_intersect(const mrt_cl_object *objects, int n_objects, const mrt_cl_ray *rays,
                          int n_rays, float t_upper, float t_lower, mrt_cl_intersect *out,
                          int32_t *ids_out)
{
    obj_t *objs = (obj_t *)calloc((size_t)(n_objects > 0 ? n_objects : 1), sizeof(obj_t));
    for (int j = 0; j < n_objects; ++j) {
        const mrt_cl_object *s = &objects[j];
        obj_t *o = &objs[j];
        o->is_tri = s->is_triangle != 0;   /* kernel.cl:322: anything not a triangle is a sphere */
        if (o->is_tri) {
            o->a = from_cl(&s->triObj.a); o->b = from_cl(&s->triObj.b); o->c = from_cl(&s->triObj.c);
            o->color = from_cl(&s->triObj.color); o->shininess = s->triObj.shininess;
        } else {
            o->center = from_cl(&s->circleObj.center); o->radius = s->circleObj.radius;
            o->color = from_cl(&s->circleObj.color); o->shininess = s->circleObj.shininess;
        }
    }
    for (int i = 0; i < n_rays; ++i) {
        hit_t h;
        int id = closest_hit(objs, n_objects, from_cl(&rays[i].origin), from_cl(&rays[i].direction),
                             t_upper, t_lower, &h);
        if (ids_out) ids_out[i] = id;
        /* kernel.cl:361-378: a miss writes only t_, intersects and shininess;
         * the other bytes of out[i] keep what the caller put there. */
        if (id < 0) {
            out[i].t_ = -1.0f;
            out[i].intersects = 0;
            out[i].shininess = 0.0f;
        } else {
            to_cl(&out[i].location, h.loc);
            to_cl(&out[i].normal, h.n);
            out[i].intersects = 1;
            out[i].t_ = h.t;
            out[i].shininess = h.shininess;
            to_cl(&out[i].color, h.color);
            out[i].is_triangle = (uint8_t)h.is_tri;
            out[i].is_circle = (uint8_t)!h.is_tri;
            out[i].dummy2 = h.shininess;
        }
    }
    free(objs);
}

void mrt_oracle_flat_shader(const mrt_cl_intersect *x, int n, mrt_cl_rgb background, mrt_cl_rgb *out)
{
    /* kernel.cl:381-396: .a is never written. */
    for (int i = 0; i < n; ++i) {
        if (!x[i].intersects) {
            out[i].r = background.r; out[i].g = background.g; out[i].b = background.b;
        } else {
            out[i].r = x[i].color.s[0]; out[i].g = x[i].color.s[1]; out[i].b = x[i].color.s[2];
        }
    }
}

void mrt_oracle_phong_shader(const mrt_cl_intersect *x, const mrt_cl_phong *ph, int n,
                             mrt_cl_rgb background, const mrt_cl_cam *cam, const mrt_cl_light *lights,
                             int n_lights, mrt_cl_rgb *out)
{
    light_t *ls = (light_t *)calloc((size_t)(n_lights > 0 ? n_lights : 1), sizeof(light_t));
    for (int j = 0; j < n_lights; ++j) {
        ls[j].loc = from_cl(&lights[j].location);
        ls[j].color = from_cl(&lights[j].color);
    }
    for (int i = 0; i < n; ++i) {
        if (x[i].intersects) {
            hit_t h;
            memset(&h, 0, sizeof h);
            h.loc = from_cl(&x[i].location); h.n = from_cl(&x[i].normal);
            h.color = from_cl(&x[i].color); h.shininess = x[i].shininess;
            f3 c = phong_pixel(&h, ph->ambient_coef, ph->diffuse_coef, ph->specular_coef,
                               from_cl(&ph->ambient_color), from_cl(&cam->eye), ls, n_lights);
            out[i].r = c.x; out[i].g = c.y; out[i].b = c.z;   /* .a untouched, kernel.cl:404-439 */
        } else {
            out[i] = background;                               /* kernel.cl:443 */
        }
    }
    free(ls);
}

/* ---------------- whole frame ---------------- */

typedef struct {
    const mrt_scene *s;
    const obj_t *objs;
    int n_objs;
    const light_t *lights;
    uint32_t row_begin, row_end;
    uint8_t *rgba;
    int32_t *ids;
    float *tval;
    float *rgbf;
} job_t;

static void render_rows(const job_t *jb)
{
    const mrt_scene *s = jb->s;
    const uint32_t w = s->xres;
    const f3 eye = f3_make(s->eye[0], s->eye[1], s->eye[2]);
    const f3 cu = f3_make(s->u[0], s->u[1], s->u[2]);
    const f3 cv = f3_make(s->v[0], s->v[1], s->v[2]);
    const f3 cw = f3_make(s->w[0], s->w[1], s->w[2]);
    const f3 amb = f3_make(s->ambient[0], s->ambient[1], s->ambient[2]);
    const f3 bg = f3_make(s->background[0], s->background[1], s->background[2]);
    /* viewport::to_struct, gfxraytrace.hpp:736-740: resolutions become floats. */
    const float xres_f = (float)s->xres, yres_f = (float)s->yres;
    /* scene::intersect, gfxraytrace.hpp:812-813. */
    const float tmin = 0.0f, tmax = FLT_MAX;

    for (uint32_t y = jb->row_begin; y < jb->row_end; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            /* gfxraytrace.hpp:850-852: pixel grid (i%w + 1, i/w + 1). */
            const float px = (float)x + 1.0f, py = (float)y + 1.0f;
            const float ux = uv_component(s->left, s->right, px, xres_f);
            const float uy = uv_component(s->bottom, s->top, py, yres_f);
            f3 o, d;
            if (s->projection == MRT_PROJ_ORTHO) ortho_ray(cu, cv, cw, ux, uy, &o, &d);
            else persp_ray(eye, cu, cv, cw, s->focal, ux, uy, &o, &d);
            hit_t h;
            int id = closest_hit(jb->objs, jb->n_objs, o, d, tmax, tmin, &h);
            f3 col;
            if (id < 0) col = bg;
            else if (s->shader == MRT_SHADER_FLAT) col = h.color;
            else col = phong_pixel(&h, s->ka, s->kd, s->ks, amb, eye, jb->lights, (int)s->n_lights);
            if (jb->rgba) {
                jb->rgba[4 * i + 0] = mrt_oracle_to_byte(col.x);
                jb->rgba[4 * i + 1] = mrt_oracle_to_byte(col.y);
                jb->rgba[4 * i + 2] = mrt_oracle_to_byte(col.z);
                jb->rgba[4 * i + 3] = 255;
            }
            if (jb->ids) jb->ids[i] = id;
            if (jb->tval) jb->tval[i] = (id < 0) ? -1.0f : h.t;
            if (jb->rgbf) { jb->rgbf[3 * i] = col.x; jb->rgbf[3 * i + 1] = col.y; jb->rgbf[3 * i + 2] = col.z; }
        }
    }
}

static void *render_thread(void *p) { render_rows((const job_t *)p); return NULL; }

int mrt_oracle_render(const mrt_scene *s, uint32_t row_begin, uint32_t row_end, int n_threads,
                      uint8_t *rgba, int32_t *ids, float *tval, float *rgbf)
{
    if (!s || s->xres == 0 || s->yres == 0 || row_end > s->yres || row_begin > row_end) return -1;
    for (uint32_t k = 0; k < s->n_spheres; ++k) if (s->sph_mat[k] >= s->n_materials) return -1;
    for (uint32_t k = 0; k < s->n_tris; ++k) if (s->tri_mat[k] >= s->n_materials) return -1;

    /* Object array in the order of gfx::scene::read_json (gfxraytrace.hpp:701-714):
     * spheres first, then triangles. */
    const int n = (int)(s->n_spheres + s->n_tris);
    obj_t *objs = (obj_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(obj_t));
    light_t *lights = (light_t *)calloc((size_t)(s->n_lights > 0 ? s->n_lights : 1), sizeof(light_t));
    if (!objs || !lights) { free(objs); free(lights); return -1; }
    for (uint32_t k = 0; k < s->n_spheres; ++k) {
        obj_t *o = &objs[k];
        uint32_t m = s->sph_mat[k];
        o->is_tri = 0;
        o->center = f3_make(s->sph[0][k], s->sph[1][k], s->sph[2][k]);
        o->radius = s->sph[3][k];
        o->color = f3_make(s->mat[0][m], s->mat[1][m], s->mat[2][m]);
        o->shininess = s->mat[3][m];
    }
    for (uint32_t k = 0; k < s->n_tris; ++k) {
        obj_t *o = &objs[s->n_spheres + k];
        uint32_t m = s->tri_mat[k];
        o->is_tri = 1;
        o->a = f3_make(s->tri[0][k], s->tri[1][k], s->tri[2][k]);
        o->b = f3_make(s->tri[3][k], s->tri[4][k], s->tri[5][k]);
        o->c = f3_make(s->tri[6][k], s->tri[7][k], s->tri[8][k]);
        o->color = f3_make(s->mat[0][m], s->mat[1][m], s->mat[2][m]);
        o->shininess = s->mat[3][m];
    }
    for (uint32_t k = 0; k < s->n_lights; ++k) {
        lights[k].loc = f3_make(s->light[0][k], s->light[1][k], s->light[2][k]);
        lights[k].color = f3_make(s->light[3][k], s->light[4][k], s->light[5][k]);
    }

    uint32_t rows = row_end - row_begin;
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > rows) n_threads = rows ? (int)rows : 1;
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int k = 0; k < n_threads; ++k) {
        jobs[k].s = s; jobs[k].objs = objs; jobs[k].n_objs = n; jobs[k].lights = lights;
        jobs[k].row_begin = row_begin + (uint32_t)(((uint64_t)rows * k) / n_threads);
        jobs[k].row_end = row_begin + (uint32_t)(((uint64_t)rows * (k + 1)) / n_threads);
        jobs[k].rgba = rgba; jobs[k].ids = ids; jobs[k].tval = tval; jobs[k].rgbf = rgbf;
    }
    if (n_threads == 1) {
        render_rows(&jobs[0]);
    } else {
        for (int k = 0; k < n_threads; ++k) pthread_create(&th[k], NULL, render_thread, &jobs[k]);
        for (int k = 0; k < n_threads; ++k) pthread_join(th[k], NULL);
    }
    free(jobs); free(th); free(objs); free(lights);
    return 0;
}
