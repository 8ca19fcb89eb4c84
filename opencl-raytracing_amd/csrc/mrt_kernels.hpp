// mrt_kernels.hpp -- parameter block shared by host code and the render kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrt {

// Geometry of the pixel->lane mapping of every kernel but trace_kernel (which has two shapes,
// TraceShape below).  One 64-lane wavefront owns a 32x32 pixel tile: lanes form an 8x8 grid and
// each lane owns a 4x4 block, so every record a wave fetches from LDS is reused for 16 pixels
// per lane and each lane's row segment is one aligned 16-byte RGBA8 store.
constexpr int kLanePx = 4;                 // pixels per lane, per axis
constexpr int kWaveTile = 8 * kLanePx;     // 32
#ifndef MRT_BLOCK_WAVES_Y
#define MRT_BLOCK_WAVES_Y 2
#endif
constexpr int kBlockWavesX = 2, kBlockWavesY = MRT_BLOCK_WAVES_Y;
constexpr int kBlockThreads = 64 * kBlockWavesX * kBlockWavesY;   // 256
constexpr int kBlockTileX = kWaveTile * kBlockWavesX;             // 64
constexpr int kBlockTileY = kWaveTile * kBlockWavesY;             // 64

// trace_kernel's lane block is a template parameter: a lane owns LX x LY pixels, a wave 8 x 8
// lanes (its tile), a block 2 x 2 waves.  4 x 4 is the shape of every other kernel; 8 x 4 spends
// 10 % fewer instructions per (pixel, primitive) pair (the per-lane set-up of a record is shared
// by 32 pixels instead of 16) and is what large frames use; small frames keep 4 x 4, whose
// four times smaller tiles fill the GPU.  Block tiles are multiples of 64 pixels per axis, so
// planes padded for either shape fit the other kernels.
template <int LX, int LY>
struct TraceShape {
    static_assert(LX % 4 == 0 && LX <= 8 && LY % 4 == 0 && LY <= 8, "lane rows are stored 16 bytes at a time; block tiles are multiples of 64");
    static constexpr int lane_x = LX, lane_y = LY;
    static constexpr int tile_x = 8 * LX, tile_y = 8 * LY;
    static constexpr int block_x = tile_x * kBlockWavesX, block_y = tile_y * kBlockWavesY;
};
using TraceNarrow = TraceShape<4, 4>;      // 32 x 32 tiles, 64 x 64 blocks
#ifndef MRT_WIDE_LANE_Y
#define MRT_WIDE_LANE_Y 4                  // (8 builds the 8x8 shape in its place, for measurements)
#endif
using TraceWide = TraceShape<8, MRT_WIDE_LANE_Y>;        // 64 x 32 tiles, 128 x 64 blocks

// Prefilter records are staged through LDS in chunks of kChunkFloat4 float4
// (12 KiB): 256 triangle records of three float4, or 768 sphere records of
// one.  kGroup* primitives are tested between two "does any lane pass?" votes.
constexpr int kChunkTris = 256;
constexpr int kChunkFloat4 = kChunkTris * 3;
constexpr int kFullRecFloat4 = 4;          // float4 per triangle record of full_trace_kernel
// full_trace_kernel: a lane owns kFullPxX x kFullPxY pixels, a wave 8 x 8 lanes, a block 2 x 2 waves.
#ifndef MRT_FULL_PX_Y
#define MRT_FULL_PX_Y 2
#endif
#ifndef MRT_FULL_WAVES_PER_SIMD
#define MRT_FULL_WAVES_PER_SIMD 4
#endif
constexpr int kFullPxX = 4, kFullPxY = MRT_FULL_PX_Y, kFullSlots = kFullPxX * kFullPxY;
constexpr int kFullWaveH = 8 * kFullPxY;          // 16 rows per wave
constexpr int kFullBlockH = 2 * kFullWaveH;       // 32 rows per block
static_assert(kBlockTileY % kFullBlockH == 0, "the padded planes are whole full_trace_kernel blocks");
#ifndef MRT_GROUP_TRIS
#define MRT_GROUP_TRIS 4
#endif
constexpr int kGroupTris = MRT_GROUP_TRIS;
constexpr int kGroupSph = 4;
#ifndef MRT_GROUP_CIRCLE
#define MRT_GROUP_CIRCLE 8
#endif
constexpr int kGroupCircle = MRT_GROUP_CIRCLE;   // triangles per vote of the bounding-circle test (a multiple of kGroupTris)

// Register budget of trace_kernel in waves per SIMD.  6 (<= 80 VGPRs) is what
// the kernel needs without spilling in its hot loop; 7 measured the same and 8
// (64 VGPRs, spills) 7 % slower on teatime 4K.  The 8x4 shape would take 98
// VGPRs; held to 80 it spills 19 outside the hot loop and is 2-3 % faster at
// six waves per SIMD than at four (five: the same as six).
#ifndef MRT_TRACE_WAVES_PER_SIMD
#define MRT_TRACE_WAVES_PER_SIMD 6
#endif

// trace_kernel's launch order when the triangle list is cut into slices (RenderParams::tile_major).
// Tile-major (the slices of one block tile are consecutive blocks, tiles follow row by row) makes
// the last blocks of a launch those of the image's last rows, for every slice; slice-major makes
// them the last slice over the whole image.  The frame ends when its last wave does, and the waves
// that take several times the median are those of tiles where many small triangles meet: in
// tile-major order they are in flight together, early, wherever that region is not the image's
// last rows, and the launch can then do with fewer, longer slices.  Measured on teatime (MI355X,
// a frame alone, each order at its best slice count): 4K 1.097 against 1.180 ms, 1080p 0.335
// against 0.380 ms, an eighth of the 8K frame 0.590 against 0.596 ms.  With fewer than eight
// slices the tail is a small share either way and slice-major order is 0.7-1 % faster (the 8K
// frame, 4 slices: 4.105 against 4.161 ms): a tile's slices meet in the same keys at the same
// time in tile-major order.  The host asks for tile-major from kTileMajorSlices slices up.
// MRT_TILE_MAJOR=0 / 1 builds one order only, for measurements.
#ifndef MRT_TILE_MAJOR
#define MRT_TILE_MAJOR -1
#endif
constexpr uint32_t kTileMajorSlices = 8;

struct RenderParams {
    float eye[3], u[3], v[3], w[3];
    float left, right, top, bottom;
    float xres_f, yres_f;
    float focal;
    float ray_abs[3];               // per component: upper bound of |direction| over all primary rays
    float org_abs[3];               // per component: upper bound of |origin| over all primary rays
    float ka, kd, ks;
    float amb[3];
    float bg[3];
    int32_t shader;
    uint32_t flags;
    uint32_t fuse_shade;            // RGBA8 stores of the epilogue: 0 none (planes only), 1 single pixels, 2 aligned 16 bytes
    uint32_t xres, yres;            // whole image
    uint32_t band_rows, rank, world;  // interleaved row-band partition
    uint32_t local_rows;            // rows this launch renders (packed)
    uint32_t n_sph, n_tri, n_tri_padded, n_sph_padded, n_lights;
    uint32_t prim_slices, slice_tris;   // trace_kernel: triangle list cut into slices, one grid layer each (1 = whole list)
    uint32_t tile_major;                // ... and the slice is blockIdx.x, the tile (blockIdx.y, blockIdx.z); 0: the tile is (x, y), the slice z
    const float4* sph_filter;       // 1 x float4 per sphere, padded to kChunkFloat4
    const float4* tri_filter;       // 3 x float4 per triangle, padded to kChunkTris
    const float4* tri_circle;       // (xc, yc, R^2, -) per triangle, padded to kChunkTris
    const float4* tri_exact;        // 3 x float4 per triangle: (a, mat) (b, -) (c, -)
    const float4* sph;              // (cx, cy, cz, r)
    const uint32_t* sph_mat;
    const float4* mats;             // (r, g, b, shininess)
    const float4* lights;           // 2 x float4 per light: (pos, -) (colour, -)
    uint32_t* out;                  // RGBA8, local_rows x xres
    int32_t* ids;                   // optional, local_rows x xres
    float* tvals;                   // optional
    unsigned long long* exact_counter;  // optional
    // Sliced trace_kernel frames with ctx-owned planes: one word per wave tile, set to this
    // frame's tag by any wave that recorded a hit there, so that shade_keys_kernel reads (and puts
    // back to "no hit") only the keys of tiles something hit.  A word that holds another frame's
    // tag means "nothing hit", so the flags are never cleared between frames.  nullptr: every key
    // is read and none is put back.
    uint32_t* tile_flags;
    uint32_t frame_tag;             // never 0 (what a fresh flag buffer holds)
    uint32_t tiles_x;               // trace_kernel wave tiles per row of the padded plane
    uint32_t tile_x_log2, tile_y_log2;  // ... and their size (32 x 32 or 64 x 32)
};

// What the record-building kernels (mrt_records.hip) need to know about a view, in double:
// the camera, and the affine map from the 0-based pixel index to uv (kernel.cl:126-127 with the
// pixel grid (x + 1, y + 1) of gfxraytrace.hpp:850-852): uv(px) = ux0 + ux1 * px.
struct ViewDerived {
    double eye[3], u[3], v[3], w[3];
    double focal;
    double ux0, ux1, uy0, uy1;
    double mux, muy;              // largest |uv.x|, |uv.y| over the image
    double xres, yres;
    int32_t persp;
    int32_t orthonormal;          // the basis is orthonormal to 1e-4 (the sphere outlines assume it)
};

}  // namespace mrt
