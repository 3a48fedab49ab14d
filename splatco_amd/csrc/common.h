// splatco_amd/csrc/common.h -- shared declarations of the gfx950 rasterizer kernels.
// Written for CDNA4 (wave64, 256 CUs / 8 XCDs) only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/splatco_raster.h"

namespace scr {

constexpr int TILE = SCR_TILE;          // 16x16 pixels
constexpr int TILE_PIX = TILE * TILE;   // 256
constexpr int WAVE = 64;                // CDNA wavefront
constexpr int NUM_XCD = 8;
constexpr int PRE_BLOCK = 256;          // Gaussians per preprocess / scatter / reduce workgroup
constexpr int REC_F = 12;               // floats per splat record (48 B, three float4)
constexpr int GRAD_F = 12;              // floats per per-instance gradient record (9 used)

inline __host__ __device__ size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Grid {
    int W, H, gx, gy, tiles;
    __host__ __device__ Grid(int H_, int W_)
        : W(W_), H(H_), gx((W_ + TILE - 1) / TILE), gy((H_ + TILE - 1) / TILE), tiles(gx * gy) {}
};

// ---- geom buffer: per-Gaussian state (written by the plan phase) + per-tile counters ----
struct GeomView {
    float4* rec;            // [P][3] float4 : (mx,my,A,B) (C,o,r,g) (b,depth,rect_lo,rect_hi)
    uint32_t* tiles_touched;  // [P]
    uint32_t* point_offsets;  // [P] inclusive scan of tiles_touched (written by the scatter kernel)
    uint8_t* clamped;         // [P] bit c set when SH colour channel c was clamped at 0
    uint32_t* block_sums;     // [ceil(P/PRE_BLOCK)] -> exclusive prefix after scan
    uint32_t* tile_count;     // [tiles]
    uint32_t* ranges;         // [tiles][2] (start, end)
    uint32_t* cursor;         // [tiles]
    unsigned long long* total;  // [1] number of instances (64-bit)
    size_t bytes;
};

inline __host__ GeomView geom_view(void* base, int64_t P, int H, int W) {
    Grid g(H, W);
    char* p = (char*)base;
    size_t off = 0;
    GeomView v;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align_up(n); return q; };
    size_t nblk = (size_t)((P + PRE_BLOCK - 1) / PRE_BLOCK);
    v.rec = (float4*)take((size_t)P * REC_F * 4);
    v.tiles_touched = (uint32_t*)take((size_t)P * 4);
    v.point_offsets = (uint32_t*)take((size_t)P * 4);
    v.clamped = (uint8_t*)take((size_t)P);
    v.block_sums = (uint32_t*)take((nblk + 1) * 4);
    v.tile_count = (uint32_t*)take((size_t)g.tiles * 4);
    v.ranges = (uint32_t*)take((size_t)g.tiles * 8);
    v.cursor = (uint32_t*)take((size_t)g.tiles * 4);
    v.total = (unsigned long long*)take(8);
    v.bytes = off;
    return v;
}

// ---- binning buffer: per (Gaussian, tile) instance lists ----
struct BinView {
    unsigned long long* keys;  // [I] (depth_bits << 32 | gaussian id), grouped by tile, unsorted
    uint32_t* inst_slot;       // [I] indexed by (point_offset_exclusive + k): slot in the tile-grouped arrays
    uint32_t* point_list;      // [I] Gaussian ids, per tile sorted by (depth, id)
    uint32_t* orig_slot;       // [I] sorted position -> slot the instance occupied before the sort
    size_t bytes;
};

inline __host__ BinView bin_view(void* base, int64_t I) {
    char* p = (char*)base;
    size_t off = 0;
    BinView v;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align_up(n); return q; };
    size_t n = (size_t)(I > 0 ? I : 1);
    v.keys = (unsigned long long*)take(n * 8);
    v.inst_slot = (uint32_t*)take(n * 4);
    v.point_list = (uint32_t*)take(n * 4);
    v.orig_slot = (uint32_t*)take(n * 4);
    v.bytes = off;
    return v;
}

// ---- image buffer ----
struct ImgView {
    float* final_T;       // [H*W]
    uint32_t* n_contrib;  // [H*W]
    size_t bytes;
};

inline __host__ ImgView img_view(void* base, int H, int W) {
    char* p = (char*)base;
    size_t off = 0;
    ImgView v;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align_up(n); return q; };
    v.final_T = (float*)take((size_t)H * W * 4);
    v.n_contrib = (uint32_t*)take((size_t)H * W * 4);
    v.bytes = off;
    return v;
}

// Kernel-side copy of the settings (scalars by value, matrices by device pointer).
struct KSettings {
    int H, W;
    float tanfovx, tanfovy, fx, fy;
    const float* bg;
    float scale_modifier;
    const float* view;
    const float* proj;
    int sh_degree;
    const float* campos;
};

inline __host__ KSettings ksettings(const scr_settings* s) {
    KSettings k;
    k.H = s->image_height;
    k.W = s->image_width;
    k.tanfovx = s->tanfovx;
    k.tanfovy = s->tanfovy;
    k.fx = (float)s->image_width / (2.0f * s->tanfovx);   // normative: binary32, this order
    k.fy = (float)s->image_height / (2.0f * s->tanfovy);
    k.bg = s->bg;
    k.scale_modifier = s->scale_modifier;
    k.view = s->viewmatrix;
    k.proj = s->projmatrix;
    k.sh_degree = s->sh_degree;
    k.campos = s->campos;
    return k;
}

// XCD-aware block -> tile map: consecutive tiles (which share splats) run on one XCD so their
// record gathers hit that XCD's L2.  Blocks are observed to be dealt round-robin to the 8 XCDs;
// the map is a speed choice only.  Returns -1 for the padding blocks.
__device__ inline int xcd_tile(int block, int tiles) {
    int chunk = (tiles + NUM_XCD - 1) / NUM_XCD;
    int t = (block % NUM_XCD) * chunk + block / NUM_XCD;
    return t < tiles ? t : -1;
}
inline __host__ int xcd_grid(int tiles) { return ((tiles + NUM_XCD - 1) / NUM_XCD) * NUM_XCD; }

// launchers (defined in the .hip files)
void launch_filter(int64_t P, const float* means3D, const float* scales, const float* rotations,
                   const float* cov3D, const KSettings& ks, int32_t* radii, hipStream_t st);
void launch_mark_visible(int64_t P, const float* means3D, const float* view, uint8_t* out, hipStream_t st);
void launch_preprocess(int64_t P, int M, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D, const float* opacities, const float* shs, const float* colors,
                       const KSettings& ks, const GeomView& gv, int32_t* radii, hipStream_t st);
void launch_plan_scans(int64_t P, const KSettings& ks, const GeomView& gv, hipStream_t st);
void launch_scatter(int64_t P, const KSettings& ks, const GeomView& gv, const BinView& bv, hipStream_t st);
void launch_tile_sort(const KSettings& ks, const GeomView& gv, const BinView& bv, hipStream_t st);
void launch_blend_forward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                          float* out_color, hipStream_t st);
void launch_blend_backward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                           const float* dL_dcolor, float4* grad_rec, hipStream_t st);
void launch_preprocess_backward(int64_t P, int M, const float* means3D, const float* scales,
                                const float* rotations, const float* cov3D, const float* shs,
                                const KSettings& ks, const int32_t* radii, const GeomView& gv,
                                const BinView& bv, const float4* grad_rec, float* dL_dmeans3D,
                                float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh, float* dL_dopacity,
                                float* dL_dscales, float* dL_drotations, float* dL_dcov3D, hipStream_t st);

}  // namespace scr
