/*
 * mrt.h -- C ABI of libmrt_hip, the MI355X (gfx950) render hot path.
 *
 * One fused device pass replaces the six blocking OpenCL stage helpers of the
 * reference (uv -> view rays -> closest-hit intersections -> shader) and the
 * host glue around them.  Everything here is plain C: pointers, sizes, PODs.
 * No C++ types, no torch types, no exceptions cross this boundary and the
 * library never calls exit().
 *
 * Reference interfaces replaced (paths relative to the reference checkout,
 * directory "Ray Tracing/"):
 *   CLHelpers.hpp:17   initCL()                    -> mrt_create
 *   CLHelpers.hpp:76   cl_uv()                     -> mrt_render_* (fused), mrt_stage_uv
 *   CLHelpers.hpp:116  cl_ortho_viewrays()         -> mrt_render_* (fused), mrt_stage_ortho_viewrays
 *   CLHelpers.hpp:152  cl_persp_viewrays()         -> mrt_render_* (fused), mrt_stage_persp_viewrays
 *   CLHelpers.hpp:191  cl_intersect()              -> mrt_render_* (fused), mrt_stage_intersect
 *   CLHelpers.hpp:239  cl_flat_shader()            -> mrt_render_* (fused), mrt_stage_flat_shader
 *   CLHelpers.hpp:276  cl_phong_shader()           -> mrt_render_* (fused), mrt_stage_phong_shader
 *   gfxraytrace.hpp:830 gfx::scene::render()       -> mrt_set_scene + mrt_render_rgba8
 *   CLStructs.hpp:8-128 packed AoS structs         -> mrt_scene (SoA planes) / mrt_cl_* (AoS mirrors)
 *
 * Threading: one mrt_ctx may be used from one thread at a time, and frames of
 * one ctx must be enqueued on one stream at a time (a ctx owns a single set of
 * working buffers; the library orders a frame behind the previous frame of the
 * same ctx when the stream changes, at the cost of a wait).  Distinct contexts
 * are independent.
 * Every entry point makes its ctx's GPU the current HIP device for its own
 * duration and restores the caller's current device before it returns.  All
 * entry points return MRT_OK (0) or a negative mrt_status; mrt_last_error()
 * gives the text of the last failure on a ctx.
 */
#ifndef MRT_H
#define MRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRT_ABI_VERSION 8

typedef enum mrt_status {
    MRT_OK = 0,
    MRT_ERR_INVALID = -1,      /* bad argument / inconsistent scene            */
    MRT_ERR_NO_DEVICE = -2,    /* no usable HIP device                         */
    MRT_ERR_HIP = -3,          /* a HIP runtime call failed                    */
    MRT_ERR_NO_SCENE = -4,     /* render called before mrt_set_scene           */
    MRT_ERR_TOO_LARGE = -5,    /* size beyond a documented limit               */
    MRT_ERR_IO = -6,           /* file could not be read / written             */
    MRT_ERR_PARSE = -7,        /* scene file rejected (message has the reason) */
    MRT_ERR_NO_MEMORY = -8     /* host memory could not be allocated (the C++ runtime's
                                  std::bad_alloc, caught at the boundary)      */
} mrt_status;

enum { MRT_PROJ_ORTHO = 0, MRT_PROJ_PERSP = 1 };
enum { MRT_SHADER_FLAT = 0, MRT_SHADER_PHONG = 1 };

/*
 * Scene as the device sees it: camera/viewport/shader constants in FP32 (the
 * double->float casts of gfxraytrace.hpp:727-808 already applied by the
 * caller) and primitive data as structure-of-arrays planes.  Replaces the
 * 144-byte AoS `object`, `light`, `phong`, `cam`, `vp` of CLStructs.hpp.
 *
 * Object order is "all spheres, then all triangles" (gfxraytrace.hpp:701-714);
 * object id j < n_spheres is sphere j, otherwise triangle j - n_spheres.
 * All pointers are borrowed for the duration of the call that receives the
 * struct and may be freed afterwards.
 */
typedef struct mrt_scene {
    float eye[3], u[3], v[3], w[3];     /* camera basis (cam, CLStructs.hpp:40)  */
    uint32_t xres, yres;                /* image size in pixels                  */
    float left, right, top, bottom;     /* viewport (vp, CLStructs.hpp:48)       */
    int32_t projection;                 /* MRT_PROJ_*                            */
    float focal;                        /* persp focal length (ignored if ortho) */
    int32_t shader;                     /* MRT_SHADER_*                          */
    float ka, kd, ks;                   /* Blinn-Phong coefficients              */
    float ambient[3];                   /* ambient colour                        */
    float background[3];                /* miss colour                           */

    uint32_t n_spheres;
    const float *sph[4];                /* planes cx, cy, cz, r      [n_spheres] */
    const uint32_t *sph_mat;            /* material index            [n_spheres] */

    uint32_t n_tris;
    const float *tri[9];                /* planes ax,ay,az,bx,by,bz,cx,cy,cz [n_tris] */
    const uint32_t *tri_mat;            /* material index            [n_tris]    */

    uint32_t n_materials;
    const float *mat[4];                /* planes r, g, b, shininess [n_materials] */

    uint32_t n_lights;
    const float *light[6];              /* planes x, y, z, r, g, b   [n_lights]  */
} mrt_scene;

typedef struct mrt_config {
    int32_t device;       /* HIP device ordinal; -1 = current device            */
    uint32_t flags;       /* MRT_FLAG_*                                         */
} mrt_config;

enum {
    MRT_FLAG_NONE = 0,
    /* Test every (pixel, primitive) pair with the exact arithmetic only,
     * without the conservative edge-function prefilter.  Same pixels out;
     * exists so the prefilter can be validated against it. */
    MRT_FLAG_NO_PREFILTER = 1u << 0,
    /* Count the (pixel, primitive) pairs that reach the exact test into
     * mrt_stats.exact_tests (one atomic per wavefront; off by default). */
    MRT_FLAG_COUNT_EXACT = 1u << 1,
    /* (bit 2 is reserved) */
    /* Skip the per-pixel bounding-circle test that precedes the per-pixel
     * edge-function test of a triangle (measurement aid; same pixels out). */
    MRT_FLAG_NO_CIRCLE = 1u << 3,
    /* (bit 4 is reserved; screen-space culling is not a flag of this context but
     * a separate entry-point family, mrt_accel_*, below) */
    /* No rejection in screen space at all: every (pixel, triangle) pair
     * evaluates the determinants of tri_intersect's Cramer solve
     * (kernel.cl:187-213) -- two cross products and three dot products with
     * fused multiply-adds, as a conservative filter whose margins are derived
     * per triangle and view -- and every (pixel, sphere) pair the discriminant
     * of circle_intersect (kernel.cl:248-254); the pairs the filter cannot rule
     * out (about 0.05 % on the benchmark frames) run the reference's own
     * unfused arithmetic, which decides.  The rays of a frame share their
     * origin (pinhole) or direction (orthographic) and the per-view records
     * fold that half in: 37 floating-point operations per pair, not the 45 of
     * a test that takes origin and direction per ray (that kernel is behind
     * mrt_stage_intersect).  Same pixels out.  Takes precedence over the
     * other flags. */
    MRT_FLAG_FULL_TEST = 1u << 5,
    /* The default kernel exists in two shapes: a lane owns 4x4 pixels (32x32
     * pixel wave tiles) or 8x4 (64x32 tiles, 10 % fewer instructions per
     * (pixel, primitive) pair, four times fewer tiles).  The library takes 8x4
     * for frames -- a rank's rows of one -- of at least 3 Mpixel; these two
     * flags force one shape whatever the size (measurement and test aid; same
     * pixels out).  Both set: MRT_ERR_INVALID. */
    MRT_FLAG_TRACE_NARROW = 1u << 6,
    MRT_FLAG_TRACE_WIDE = 1u << 7,
    /* mrt_create normally brings the fixed array of
     * mrt_render_bands_device_noalloc onto the GPU (64 MiB: 23 ms of the first
     * context's start-up, and 5 ms of the first frame's), so that no call of
     * that entry point ever sets anything up.  A caller that will not use it --
     * the mrraytracer CLI -- sets this flag and mrt_create leaves the array
     * alone: a process that never calls the entry point never pays for it, and
     * one that does after all pays on its first such call on that GPU (which
     * then loads the array; later calls allocate nothing, as documented).
     * mrt_multi_*, mrt_sequence_* and mrt_accel_* have no such entry point
     * and always leave the array alone.  (ABI version 7.) */
    MRT_FLAG_DEFER_NOALLOC_SETUP = 1u << 8
    /* (bit 30 is reserved: the library marks the slots of an mrt_sequence with it) */
};

typedef struct mrt_stats {
    double upload_ms;       /* scene H2D + per-frame setup (set_scene)          */
    double kernel_ms;       /* device time of both render kernels (HIP events)  */
    double trace_ms;        /* ... of the closest-hit kernel, which also shades  */
    double shade_ms;        /* ... of the separate shading launch (tile cull only), else 0 */
    double download_ms;     /* D2H of the pixels (0 for device-resident output) */
    uint64_t rays;          /* primary rays = pixels rendered                   */
    uint64_t prim_tests;    /* rays x (n_spheres + n_tris), brute force         */
    uint64_t exact_tests;   /* (pixel, primitive) pairs that reached the exact test (0 unless counted) */
    uint32_t kernel;        /* MRT_KERNEL_*: the closest-hit kernel this frame ran                     */
    uint32_t prim_slices;   /* grid layers the triangle list was cut into (trace kernel), else 1       */
} mrt_stats;

/* Which closest-hit kernel a frame ran (mrt_stats.kernel).  A brute-force
 * context runs TRACE or FULL; the others belong to mrt_accel_*, where the
 * method asks and the band geometry decides (see mrt_accel_config). */
enum {
    MRT_KERNEL_TRACE = 0,         /* per-pixel circle -> edge -> exact funnel (default)   */
    MRT_KERNEL_TRACE_BLOCK = 1,   /* ... preceded by the 4x4 block reject                 */
    MRT_KERNEL_CULL = 2,          /* 32x32 tile cull                                      */
    MRT_KERNEL_FULL = 3,          /* a 3x3 solve for every pair (MRT_FLAG_FULL_TEST)      */
    MRT_KERNEL_GRID = 4,          /* per-tile primitive lists of the screen-space grid    */
    MRT_KERNEL_TRACE_WIDE = 5     /* the default funnel, 8x4 pixels per lane (large frames) */
};

typedef struct mrt_ctx mrt_ctx;

/* Create / destroy a context bound to one GPU.  Replaces initCL()
 * (CLHelpers.hpp:17-74): no program build at run time, kernels are
 * precompiled for gfx950 into the library.
 * Environment, read once by mrt_create: MRT_FORCE_SLICES=<1..65536> fixes the
 * number of slices / layers the primitive list of this context's frames is cut
 * into instead of the automatic choice (a tuning aid: same pixels, other
 * timing; anything else in the variable is ignored). */
int mrt_create(const mrt_config *cfg, mrt_ctx **out);
void mrt_destroy(mrt_ctx *ctx);

/* Text of the last error on ctx (or of the last failed mrt_create when ctx is
 * NULL).  Never NULL. */
const char *mrt_last_error(const mrt_ctx *ctx);

/* ABI version of the loaded library (== MRT_ABI_VERSION it was built with). */
int mrt_abi_version(void);

/* Copy the scene to the GPU (SoA planes) and derive the per-frame primitive
 * records the kernel consumes.  Replaces the to_struct() packers and VLAs of
 * gfxraytrace.hpp:727-828. */
int mrt_set_scene(mrt_ctx *ctx, const mrt_scene *scene);

/* Camera, viewport and projection of a frame: the fields of mrt_scene that
 * change when only the view changes. */
typedef struct mrt_camera {
    float eye[3], u[3], v[3], w[3];
    uint32_t xres, yres;
    float left, right, top, bottom;
    int32_t projection;                 /* MRT_PROJ_*                            */
    float focal;
} mrt_camera;

/* Replace the view of the scene last given to mrt_set_scene: geometry,
 * materials, lights and shader stay on the GPU, only the per-view records
 * are rebuilt, by kernels on the ctx's own stream: the call does not wait for
 * the GPU, it is ordered behind the frames already enqueued and the next
 * frame, on whatever stream, behind it.  The
 * reference has no equivalent: it re-creates context, program and buffers in
 * every stage call (CLHelpers.hpp:79,119,155,196,242,281). */
int mrt_set_camera(mrt_ctx *ctx, const mrt_camera *camera);

/* Returns when everything this ctx has enqueued so far is over: the record
 * kernels of mrt_set_camera and the frames, on whatever stream they were put.
 * What queue.finish() is to the reference's stage helpers (CLHelpers.hpp:111). */
int mrt_synchronize(mrt_ctx *ctx);

/* Render the whole frame into host memory: xres*yres RGBA8, row-major, pixel
 * (x, y) at 4*(y*xres + x), A = 255.  Bytes are unsigned(x * 255.0f) of the
 * shaded float colour (gfximage.hpp:46-51).  Replaces gfx::scene::render()
 * (gfxraytrace.hpp:830-878) and the per-stage calls it makes. */
int mrt_render_rgba8(mrt_ctx *ctx, uint8_t *host_out, mrt_stats *stats /* may be NULL */);

/* mrt_render_rgba8 for a caller who can use the rows as they arrive (an encoder,
 * a display): the frame is rendered in strips of strip_rows consecutive rows
 * (0 = the library's choice: about eight for a frame of 2^40 ray-primitive pairs
 * and more, one below, where the kernels are over before there is anything to
 * hide behind them), every strip is copied to
 * host_out behind its kernels and beside the strips that follow, and
 * ready(user, row_begin, row_end) is called -- on the calling thread, in row
 * order -- as soon as rows [row_begin, row_end) of host_out are final.  Same
 * bytes as mrt_render_rgba8.  stats->trace_ms covers all strips.  host_out is
 * best page-locked.  Takes the place of the host loops that assemble the image
 * after the last stage (gfxraytrace.hpp:872-875). */
typedef void (*mrt_rows_ready_fn)(void *user, uint32_t row_begin, uint32_t row_end);
int mrt_render_rgba8_rows(mrt_ctx *ctx, uint8_t *host_out, uint32_t strip_rows, mrt_rows_ready_fn ready,
                          void *user, mrt_stats *stats /* may be NULL */);

/* Page-locked host memory for mrt_render_rgba8's host_out: the device-to-host
 * copy then runs at PCIe rate instead of through the runtime's bounce buffers
 * (about 3x at 3840x2160).  Any other host memory works too.  Takes the place
 * of the `new rgbColor[count]` result arrays of the reference's stage helpers
 * (CLHelpers.hpp:254,293), which the caller owns and never frees. */
int mrt_host_alloc(size_t bytes, void **out);
void mrt_host_free(void *p);

/* As mrt_render_rgba8, additionally returning per pixel the winning object id
 * (-1 = miss) and its ray parameter t (-1.0f for a miss), the fields
 * intersect.t_ / intersects of CLStructs.hpp:115-128.  ids/t may be NULL. */
int mrt_render_debug(mrt_ctx *ctx, uint8_t *host_out, int32_t *ids, float *t,
                     mrt_stats *stats);

/*
 * Interleaved row-band partition used for multi-GPU rendering: the image is
 * cut into bands of band_rows rows (the last may be short); band b belongs to
 * rank b % world.  A rank's bands are stored back-to-back, in increasing b.
 * mrt_band_rows() returns how many image rows rank owns.
 */
uint32_t mrt_band_rows(uint32_t yres, uint32_t band_rows, uint32_t rank, uint32_t world);
/* The band height, 64 rows or 32 (whole wave tiles), that leaves the busiest of
 * `world` ranks the fewer rows of a yres-row image (64 when it makes no difference):
 * 32 for 4320 rows over 8 ranks, where
 * bands of 64 would give one rank 576 rows against a mean of 540.  The frame is as
 * slow as its busiest rank.  (Takes the place of the reference's single-device
 * choice, CLHelpers.hpp:28-45, like the rest of the band partition.) */
uint32_t mrt_balanced_band_rows(uint32_t yres, uint32_t world);

/* Render only this rank's bands into DEVICE memory (dev_out, at least
 * mrt_band_rows()*xres*4 bytes), asynchronously on hip_stream (a hipStream_t,
 * NULL = the null stream).  Nothing is copied to the host; the caller gathers
 * (RCCL) and synchronises.  kernel_ms in stats is filled only when
 * stats != NULL, which makes the call wait for the kernel. */
int mrt_render_bands_device(mrt_ctx *ctx, uint32_t band_rows, uint32_t rank, uint32_t world,
                            void *dev_out, size_t dev_out_bytes, void *hip_stream,
                            mrt_stats *stats);

/* mrt_render_bands_device for callers that cannot allocate device memory
 * (tight serving loops, harnesses that forbid hipMalloc): same arguments without stats, same pixels, but the closest-hit
 * key plane the trace kernel works in lives in a fixed array inside the
 * library instead of a buffer the ctx grows on demand, and the call makes no
 * hipMalloc / hipFree (nor any other allocating call).  The array holds
 * MRT_NOALLOC_MAX_KEYS 64-bit keys (64 MiB on every GPU the library has been
 * used on, until the process ends): the rows this rank renders, padded to
 * whole block tiles (64x64 pixels, 128x64 for frames of 3 Mpixel and more),
 * must fit -- a 3840x2160 frame does -- or the call fails with
 * MRT_ERR_TOO_LARGE (a MRT_FLAG_FULL_TEST context, whose kernel only merges
 * its layers in the key plane, then runs as one layer instead).  Every call on
 * one GPU, from whatever ctx, shares that GPU's array: two such frames in
 * flight at once on one GPU race; enqueue them on one stream or synchronise
 * between them.
 * Not allocating is all it promises.  HIP graph capture is NOT supported: on a
 * stream that is capturing (hipStreamBeginCapture) this call -- like every other
 * entry point that takes a stream -- fails with MRT_ERR_INVALID before it
 * enqueues anything, and the capture stays valid.  What is known about replay
 * (diagnostic -DMRT_ALLOW_CAPTURE builds only): a recorded frame -- a memset
 * node on the fixed array, trace kernel, shading kernel, in a chain -- holds
 * the grid, the block and, byte for byte, the arguments of the eager call, and
 * replayed from a plain C++ program it writes the eager frame's bytes, on the
 * HIP 7.2 runtime and on the 7.0 runtime the PyTorch wheel brings alike
 * (tools/graph_frame_probe.cpp --replay; tools/graph_probe.hip for the parts on
 * their own).  Replayed inside a Python process in which PyTorch has started
 * that same runtime, the first hipGraphLaunch ends in hipErrorIllegalAddress
 * and leaves the GPU faulted (four times out of four, torch.cuda.graph or
 * hipStreamBeginCapture alike).  Why was not taken apart further, every
 * attempt costs a GPU; hence the refusal, everywhere.
 * No counterpart in the reference, which allocates in every stage call
 * (CLHelpers.hpp:93-104). */
#define MRT_NOALLOC_MAX_KEYS (8u * 1024u * 1024u)
int mrt_render_bands_device_noalloc(mrt_ctx *ctx, uint32_t band_rows, uint32_t rank, uint32_t world,
                                    void *dev_out, size_t dev_out_bytes, void *hip_stream);

/* Host-side inverse of the band partition: scatter the gathered per-rank
 * buffers (rank r's bytes start at gathered + r*stride_bytes) into a
 * scanline-ordered xres*yres RGBA8 image. */
int mrt_bands_to_image(const uint8_t *gathered, size_t stride_bytes, uint32_t xres, uint32_t yres,
                       uint32_t band_rows, uint32_t world, uint8_t *image_out);

/* ------------------------------------------------------------------------
 * One frame over several GPUs of a node, from one process.  The image is cut
 * into the interleaved row bands of mrt_band_rows, every GPU holds the whole
 * scene and renders its own bands on its own stream, all GPUs at once, and the
 * bands end up in one scanline-ordered host image.  Takes the place of the
 * reference's single-device choice (CLHelpers.hpp:28-45) and of its host-side
 * image assembly (gfxraytrace.hpp:872-875).
 * ---------------------------------------------------------------------- */
enum {
    MRT_GATHER_AUTO = 0,   /* RCCL when there are several distinct GPUs and it loads, else PEER   */
    MRT_GATHER_RCCL = 1,   /* ncclGather to rank 0 over xGMI, then rank 0 copies to the host      */
    MRT_GATHER_PEER = 2,   /* hipMemcpyPeerAsync to rank 0, then rank 0 copies to the host        */
    MRT_GATHER_HOST = 3    /* no gather: every GPU copies its own bands to the host image itself  */
};

typedef struct mrt_multi_config {
    uint32_t n_gpus;          /* ranks; 0 = one per visible GPU                                   */
    const int32_t *devices;   /* n_gpus HIP ordinals, NULL = 0 .. n_gpus-1.  An ordinal may repeat:
                                 several ranks then share a GPU (a rehearsal of N ranks on fewer
                                 GPUs; RCCL cannot be used then)                                  */
    uint32_t flags;           /* MRT_FLAG_* for every rank                                        */
    uint32_t band_rows;       /* rows per band, 0 = mrt_balanced_band_rows of every frame's height   */
    uint32_t gather;          /* MRT_GATHER_*                                                     */
} mrt_multi_config;

typedef struct mrt_multi_stats {
    double render_ms_max;   /* device time of the slowest rank's render kernels (HIP events)  */
    double render_ms_sum;   /* ... summed over the ranks                                      */
    double total_ms;        /* host wall time of the call: render, gather, copies to the host */
    uint64_t rays, prim_tests;
    uint32_t n_gpus;        /* ranks                                                          */
    uint32_t gather;        /* MRT_GATHER_* actually used                                     */
} mrt_multi_stats;

typedef struct mrt_multi mrt_multi;

int mrt_multi_create(const mrt_multi_config *cfg, mrt_multi **out);
void mrt_multi_destroy(mrt_multi *m);
/* Text of the last error on m (of the last failed mrt_multi_create when m is NULL). */
const char *mrt_multi_last_error(const mrt_multi *m);
/* Ranks, the gather transport in use, and a remark when MRT_GATHER_AUTO had to
 * settle for less than RCCL (empty otherwise). */
uint32_t mrt_multi_world(const mrt_multi *m);
uint32_t mrt_multi_gather(const mrt_multi *m);
const char *mrt_multi_note(const mrt_multi *m);
/* mrt_set_scene / mrt_set_camera on every rank. */
int mrt_multi_set_scene(mrt_multi *m, const mrt_scene *scene);
int mrt_multi_set_camera(mrt_multi *m, const mrt_camera *camera);
/* mrt_render_rgba8 over all ranks: same bytes as one GPU renders.  host_out is
 * best page-locked (mrt_host_alloc): the band copies then run asynchronously. */
int mrt_multi_render_rgba8(mrt_multi *m, uint8_t *host_out, mrt_multi_stats *stats /* may be NULL */);
/* ... telling the caller of every band as it lands in host_out, in row order
 * (see mrt_render_rgba8_rows): an encoder fed from `ready` works under the
 * other bands' copies and the other GPUs' rendering. */
int mrt_multi_render_rgba8_rows(mrt_multi *m, uint8_t *host_out, mrt_rows_ready_fn ready, void *user,
                                mrt_multi_stats *stats /* may be NULL */);

/* ------------------------------------------------------------------------
 * Many frames of one scene on one GPU, several of them in flight: a turntable,
 * an animation, a stream of views to serve.  A frame alone leaves part of the
 * GPU idle while its last waves drain, its shading launch runs and the next
 * frame's first waves start (about 9 % of a 3840x2160 frame); a sequence keeps
 * in_flight frames enqueued side by side, each in a slot with working buffers,
 * per-view records and a stream of its own, so that the next frame fills those
 * gaps.  Frame k goes to slot k mod in_flight and runs behind frame
 * k - in_flight.  One thread at a time per sequence.  The multi-frame caller
 * the reference cannot serve: it sets up
 * context, program and buffers anew in every stage call
 * (CLHelpers.hpp:79,119,155,196,242,281).
 * ---------------------------------------------------------------------- */
typedef struct mrt_sequence_config {
    int32_t device;       /* HIP ordinal, -1 = the current device             */
    uint32_t flags;       /* MRT_FLAG_* of the slots' brute-force contexts     */
    uint32_t in_flight;   /* slots, 1 .. 4; 0 = 2                              */
} mrt_sequence_config;

typedef struct mrt_sequence mrt_sequence;

int mrt_sequence_create(const mrt_sequence_config *cfg, mrt_sequence **out);
/* Waits for the frames in flight, then destroys the slots and their streams:
 * whatever the caller chained on those streams must be over, and a framework
 * that remembers streams (an allocator that records an event on the stream a
 * buffer was last used on when the buffer is freed) must have let go of them. */
void mrt_sequence_destroy(mrt_sequence *q);
/* Text of the last error on q (of the last failed mrt_sequence_create when q is NULL). */
const char *mrt_sequence_last_error(const mrt_sequence *q);
uint32_t mrt_sequence_in_flight(const mrt_sequence *q);
/* mrt_set_scene on every slot; waits for the frames in flight. */
int mrt_sequence_set_scene(mrt_sequence *q, const mrt_scene *scene);
/* Enqueue the next frame and return at once: `view` (NULL = the view of the
 * frame before, or of the scene as given) and this rank's interleaved bands
 * into dev_out, exactly as mrt_render_bands_device writes them.  The frame runs
 * on its slot's stream; *frame_stream (may be NULL) receives that hipStream_t,
 * so that the caller can chain what consumes dev_out (a gather, a copy) behind
 * the frame, or make a stream of its own wait for it.  dev_out belongs to the
 * frame until that stream has passed it (or mrt_sequence_wait has returned);
 * work the caller chained on the slot's stream is what frame k + in_flight
 * runs behind. */
int mrt_sequence_submit(mrt_sequence *q, const mrt_camera *view, uint32_t band_rows, uint32_t rank,
                        uint32_t world, void *dev_out, size_t dev_out_bytes, void **frame_stream);
/* The same for a caller who wants the frames in host memory: the whole frame
 * (xres*yres RGBA8 of the view, as mrt_render_rgba8 writes it) is rendered
 * into a device buffer of the sequence (2 x in_flight of them, in rotation)
 * and copied to host_out by a copy stream of the sequence's own, behind the
 * frame and beside the frames that follow: in a stream of frames the copies
 * cost nothing (1.20 ms per 3840x2160 frame in page-locked memory, against 1.81
 * for mrt_render_rgba8 frame by frame).  Returns at once; host_out -- best
 * page-locked (mrt_host_alloc) -- is complete when mrt_sequence_wait returns. */
int mrt_sequence_submit_host(mrt_sequence *q, const mrt_camera *view, uint8_t *host_out);
/* Returns when every frame submitted so far (and whatever the caller chained on
 * the slots' streams) has completed. */
int mrt_sequence_wait(mrt_sequence *q);

/* ------------------------------------------------------------------------
 * Screen-space acceleration: the same pixels without testing every pair.
 *
 * The reference's closest-hit loop is O(pixels x objects) (kernel.cl:320-360)
 * and so is an mrt_ctx; the benchmark's tests/s are defined on that loop.  An
 * mrt_accel is the algorithmic successor behind entry points of its own: it
 * renders byte-identical frames (same winners, same t) but skips the pairs
 * that cannot meet.  No counterpart in the reference.
 * ---------------------------------------------------------------------- */
enum {
    /* Uniform grid over the image with cells of 32x32 pixels, built on the GPU
     * for every view (mrt_accel_set_scene / _set_camera: bin count, exclusive
     * scan, fill): each cell lists the objects whose conservative screen-space
     * region touches it, and a frame walks one list per cell. */
    MRT_ACCEL_GRID = 0,
    /* No structure: every frame tests every object against every 32x32 tile
     * (64 objects per wave instruction) and only the survivors per pixel. */
    MRT_ACCEL_TILE_CULL = 1,
    /* The brute-force per-pixel funnel preceded by one test per 4x4 pixel block. */
    MRT_ACCEL_BLOCK_REJECT = 2
};

typedef struct mrt_accel_config {
    int32_t device;       /* HIP device ordinal; -1 = current device            */
    uint32_t method;      /* MRT_ACCEL_*                                        */
    uint32_t flags;       /* MRT_FLAG_COUNT_EXACT or 0                          */
} mrt_accel_config;

typedef struct mrt_accel_stats {
    mrt_stats frame;      /* as for mrt_ctx frames; frame.kernel says what ran  */
    double build_ms;      /* GRID: device time of the last build (per view)     */
    uint64_t cells;       /* GRID: cells of the grid                            */
    uint64_t refs;        /* GRID: (cell, object) references in all lists       */
} mrt_accel_stats;

typedef struct mrt_accel mrt_accel;

int mrt_accel_create(const mrt_accel_config *cfg, mrt_accel **out);
void mrt_accel_destroy(mrt_accel *a);
/* Text of the last error on a (of the last failed mrt_accel_create when a is NULL). */
const char *mrt_accel_last_error(const mrt_accel *a);
uint32_t mrt_accel_method(const mrt_accel *a);
/* As mrt_set_scene / mrt_set_camera; GRID also (re)builds the grid for the view,
 * on the GPU.  mrt_accel_set_camera does not wait for it: beyond 2^28 references
 * mrt_accel_set_scene fails with MRT_ERR_TOO_LARGE at once, after
 * mrt_accel_set_camera the first frame of that view does. */
int mrt_accel_set_scene(mrt_accel *a, const mrt_scene *scene);
int mrt_accel_set_camera(mrt_accel *a, const mrt_camera *camera);
/* As mrt_render_rgba8 / mrt_render_debug / mrt_render_bands_device.  Interleaved
 * bands (world > 1) must be a multiple of 32 rows for GRID; TILE_CULL and
 * BLOCK_REJECT fall back to the plain per-pixel kernel for band heights they
 * cannot tile (32 and 4 rows) and say so in stats->frame.kernel. */
int mrt_accel_render_rgba8(mrt_accel *a, uint8_t *host_out, mrt_accel_stats *stats /* may be NULL */);
int mrt_accel_render_debug(mrt_accel *a, uint8_t *host_out, int32_t *ids, float *t, mrt_accel_stats *stats);
int mrt_accel_render_bands_device(mrt_accel *a, uint32_t band_rows, uint32_t rank, uint32_t world,
                                  void *dev_out, size_t dev_out_bytes, void *hip_stream,
                                  mrt_accel_stats *stats);

/* ------------------------------------------------------------------------
 * Scene files.  Same schema, validation order and error strings as the
 * reference loader (rayson.hpp:380-638) and the conversion that follows it
 * (gfxraytrace.hpp:645-721, camera basis in double :881-889).
 * ---------------------------------------------------------------------- */
typedef struct mrt_loaded_scene mrt_loaded_scene;

/* Parse a scene file.  On failure returns MRT_ERR_IO / MRT_ERR_PARSE and
 * writes the reference's message (without the `error loading scene "...": `
 * prefix) into err (NUL-terminated, truncated to err_cap). */
int mrt_scene_load_file(const char *path, mrt_loaded_scene **out, char *err, size_t err_cap);
int mrt_scene_load_text(const char *json_text, size_t len, mrt_loaded_scene **out, char *err,
                        size_t err_cap);
/* Borrowed view valid until mrt_scene_free(). */
const mrt_scene *mrt_scene_view(const mrt_loaded_scene *s);
/* Values the CLI prints that are not part of mrt_scene (doubles before the
 * float casts; light intensity is parsed but never used by the shaders). */
typedef struct mrt_scene_info {
    double eye[3], u[3], v[3], w[3];
    double left, top, right, bottom;
    double focal, ka, kd, ks;
    uint32_t n_lights;
    const double *light_pos;        /* [n_lights][3] */
    const double *light_intensity;  /* [n_lights]    */
} mrt_scene_info;
const mrt_scene_info *mrt_scene_info_view(const mrt_loaded_scene *s);
void mrt_scene_free(mrt_loaded_scene *s);

/* 8-bit RGB PNG (no alpha) from an RGBA8 buffer, the format of
 * gfxpng.hpp:63-86. */
int mrt_write_png_rgb(const char *path, const uint8_t *rgba, uint32_t xres, uint32_t yres);

/* The same file, encoded while the frame is still arriving: rgba is the whole
 * xres*yres RGBA8 image buffer, filled by someone else; mrt_png_stream_rows
 * reports rows [row_begin, row_end) of it final (any order, every row exactly
 * once), host threads deflate each stripe of rows as soon as it is whole, and
 * mrt_png_stream_end waits for them, writes the file and frees the stream
 * (MRT_ERR_INVALID, and no file, if rows were never reported).  Byte for byte
 * the file mrt_write_png_rgb writes.  gfxpng.hpp:63-86 encodes after the last
 * pixel is in place; at 16384 x 16384 that is the longest step of the run. */
typedef struct mrt_png_stream mrt_png_stream;
int mrt_png_stream_begin(const char *path, const uint8_t *rgba, uint32_t xres, uint32_t yres,
                         mrt_png_stream **out);
int mrt_png_stream_rows(mrt_png_stream *s, uint32_t row_begin, uint32_t row_end);
int mrt_png_stream_end(mrt_png_stream *s);

/* ------------------------------------------------------------------------
 * Stage-level entry points with the reference's own array-of-structs types,
 * for callers that keep gfxraytrace.hpp's four-stage render() and only swap
 * CLHelpers.hpp.  Layouts mirror CLStructs.hpp:8-128 (packed; float3 = 16 B).
 * Inputs and outputs are HOST arrays owned by the caller; n may be any size
 * (the reference's fixed 160000 global size, CLHelpers.hpp:9, is not kept).
 * Every call blocks until its results are in the caller's arrays, like the
 * reference's helpers.  The device staging buffers belong to the ctx: they
 * grow to the largest call seen and are freed by mrt_destroy, so repeated
 * calls of one size allocate nothing (the reference creates and releases its
 * cl::Buffers in every call, CLHelpers.hpp:93-104); copies and kernels run on
 * the ctx's own stream, not on the null stream.  Results cross the link into
 * one page-locked host buffer of the ctx, as large as the largest result seen
 * (32 bytes per ray for mrt_stage_intersect and the view rays, 16 per pixel
 * for the shaders), from which they are scattered into the caller's records;
 * it too is held until mrt_destroy.  The scatter runs on the calling thread,
 * except into an array of a million elements and more or one whose pages
 * have not been touched yet (a fresh allocation, as the reference's helpers
 * return): then on up to eight host threads that live for that call.
 * ---------------------------------------------------------------------- */
#pragma pack(push, 1)
typedef struct mrt_cl_float3 { float s[4]; } mrt_cl_float3;              /* cl_float3  16 B */
typedef struct mrt_cl_float2 { float s[2]; } mrt_cl_float2;              /* cl_float2   8 B */
typedef struct mrt_cl_rgb { float r, g, b, a; } mrt_cl_rgb;              /* rgbColor   16 B */
typedef struct mrt_cl_light { mrt_cl_float3 location, color; float intensity, pad[3]; } mrt_cl_light; /* 48 B */
typedef struct mrt_cl_cam { mrt_cl_float3 eye, u, v, w; } mrt_cl_cam;    /* cam        64 B */
typedef struct mrt_cl_vp { float x_resolution, y_resolution, left, right, top, bottom, pad[2]; } mrt_cl_vp; /* 32 B */
typedef struct mrt_cl_ray { mrt_cl_float3 origin, direction; } mrt_cl_ray; /* viewRay  32 B */
typedef struct mrt_cl_phong { float ambient_coef, diffuse_coef, specular_coef; mrt_cl_float3 ambient_color; float pad; } mrt_cl_phong; /* 32 B */
typedef struct mrt_cl_triangle { mrt_cl_float3 color; float shininess; mrt_cl_float3 a, b, c; float pad[3]; } mrt_cl_triangle; /* 80 B */
typedef struct mrt_cl_circle { mrt_cl_float3 color; float shininess; mrt_cl_float3 center; float radius; float pad[2]; } mrt_cl_circle; /* 48 B */
typedef struct mrt_cl_object { mrt_cl_triangle triObj; mrt_cl_circle circleObj; uint8_t is_triangle, is_circle, pad0, pad1; float pad[3]; } mrt_cl_object; /* 144 B */
typedef struct mrt_cl_intersect {
    mrt_cl_float3 location, normal;   /*  0, 16 */
    uint8_t intersects;               /* 32     */
    float t_;                         /* 33     */
    float dummy2;                     /* 37     */
    mrt_cl_float3 dummy3;             /* 41     */
    uint8_t is_triangle, is_circle, dummy1; /* 57, 58, 59 */
    float shininess;                  /* 60     */
    mrt_cl_float3 color;              /* 64     */
} mrt_cl_intersect;                   /* 80 B   */
#pragma pack(pop)

/* kernel `uv` (kernel.cl:123-128) / cl_uv (CLHelpers.hpp:76). */
int mrt_stage_uv(mrt_ctx *ctx, const mrt_cl_float3 *positions, const mrt_cl_vp *viewport, int n,
                 mrt_cl_float2 *uv_out);
/* kernel `ortho_viewrays` (kernel.cl:130-138) / cl_ortho_viewrays (CLHelpers.hpp:116). */
int mrt_stage_ortho_viewrays(mrt_ctx *ctx, const mrt_cl_cam *camera, const mrt_cl_float2 *uv, int n,
                             mrt_cl_ray *rays_out);
/* kernel `persp_viewrays` (kernel.cl:140-147) / cl_persp_viewrays (CLHelpers.hpp:152). */
int mrt_stage_persp_viewrays(mrt_ctx *ctx, const mrt_cl_cam *camera, const mrt_cl_float2 *uv, int n,
                             float focal_length, mrt_cl_ray *rays_out);
/* kernel `intersections` (kernel.cl:305-379) / cl_intersect (CLHelpers.hpp:191):
 * arbitrary rays (an origin and a direction each) against arbitrary objects
 * (spheres and triangles in any order), closest hit in (t_lower, t_upper), the
 * lower object index among equal t.  Every (ray, object) pair is tested: rays
 * go through registers eight to a lane, objects through LDS in chunks, and the
 * object list is cut into slices (grid layers) that meet in one 64-bit (t, index)
 * key per ray, so that the reference's own 160000 rays fill the GPU.  Per
 * (ray, triangle) pair the kernel forms the determinants of tri_intersect's
 * Cramer solve (kernel.cl:187-213) as dot products of the ray's direction and
 * moment (direction x origin) with per-triangle vectors, in fused
 * multiply-adds -- det, N_beta, N_gamma always; N_t, which carries the sign of
 * t, only in waves whose rays have much geometry behind them -- as a
 * conservative filter; the pairs it cannot rule out (about one in two thousand
 * on the shipped mesh) and every pair of a ray with non-finite or huge
 * components run the reference's own unfused arithmetic, which decides.
 * Bit-identical records to the per-ray loop of the reference. */
int mrt_stage_intersect(mrt_ctx *ctx, const mrt_cl_object *objects, int n_objects,
                        const mrt_cl_ray *rays, int n_rays, float t_upper_bound,
                        float t_lower_bound, mrt_cl_intersect *out);
/* Device time (HIP events, milliseconds) of the kernels of the last
 * mrt_stage_intersect call on ctx, without the copies either side: what the
 * reference never measures (CLHelpers.hpp:230-236 only waits for the queue). */
double mrt_stage_last_kernel_ms(const mrt_ctx *ctx);
/* kernel `flat_shader` (kernel.cl:381-396) / cl_flat_shader (CLHelpers.hpp:239). */
int mrt_stage_flat_shader(mrt_ctx *ctx, const mrt_cl_intersect *xsects, int n, mrt_cl_rgb background,
                          mrt_cl_rgb *out);
/* kernel `phong_shader` (kernel.cl:398-445) / cl_phong_shader (CLHelpers.hpp:276). */
int mrt_stage_phong_shader(mrt_ctx *ctx, const mrt_cl_intersect *xsects, const mrt_cl_phong *phong,
                           int n, mrt_cl_rgb background, const mrt_cl_cam *camera,
                           const mrt_cl_light *lights, int n_lights, mrt_cl_rgb *out);

#ifdef __cplusplus
}
#endif
#endif /* MRT_H */
