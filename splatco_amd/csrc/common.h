// splatco_amd/csrc/common.h -- shared declarations of the gfx950 rasterizer kernels.
// Written for CDNA4 (wave64, 256 CUs / 8 XCDs) only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/splatco_raster.h"

namespace scr {

// min(a + b, 2^32 - 1)
__host__ __device__ __forceinline__ uint32_t sat_add_u32(uint32_t a, uint32_t b) {
    const uint32_t s = a + b;
    return s < a ? 0xFFFFFFFFu : s;
}


constexpr int TILE = SCR_TILE;          // 16x16 pixels
constexpr int TILE_PIX = TILE * TILE;   // 256
constexpr int WAVE = 64;                // CDNA wavefront
constexpr int NUM_XCD = 8;
constexpr int PRE_BLOCK = 256;          // Gaussians per preprocess-backward workgroup
constexpr int BIN_THREADS = 1024;       // threads of a preprocess / scatter workgroup
constexpr int BIN_ROUNDS = 4;           // Gaussians per thread
constexpr int BIN_GPW = BIN_THREADS * BIN_ROUNDS;  // Gaussians per preprocess / scatter workgroup
constexpr int LDS_HIST_MAX_TILES = 40000;  // per-tile LDS histogram (4 B / tile) in the CU's 160 KB of LDS (4K images: 32400 tiles)
constexpr int ID_BITS = 28;             // sort key = depth:32 | id:28 | quadrant mask:4  ->  P < 2^28
constexpr int WG_SORT_MAX = 8192;       // keys an on-chip workgroup sort takes (binning.hip tile_sort_wg_kernel): one wave per 1024
constexpr int REC_F = 12;               // floats per splat record (48 B, three float4)
constexpr int GRAD_F = 9;               // floats per per-instance gradient record (GradRec)

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
    uint2* gm_base;           // [P] (b, rw): Gaussian-major index of the instance in tile (tx, ty) = b + ty*rw + tx
                              //     (b = first index - y0*rw - x0 mod 2^32, rw = rect width; written by the scatter kernel)
    uint32_t* live_bits;      // [P] bit k: the k-th tile of the Gaussian's rect walk (row-major) can be reached by one of its
                              //     quadrants (quadrant mask != 0), k < 32; written by the scatter kernel.  An instance whose bit
                              //     is clear gets no gradient record (blend backward) and none is read (preprocess backward)
    uint8_t* has_rec;         // [P] deep lists only: set by the blend backward for every Gaussian that got at least one gradient
                              //     record, so that preprocess_backward skips the record loop of the others (at 20 M anchors most
                              //     Gaussians sit behind every pixel's last contributor in all of their tiles)
    uint32_t* block_sums;     // [ceil(P/BIN_GPW)] -> exclusive prefix after scan
    uint32_t* tile_count;     // [tiles] instances per tile (preprocess -> plan scan); afterwards the blend launches' tile order
    uint32_t* ranges;         // [tiles][2] (start, end)
    uint32_t* cursor;         // [tiles]
    int64_t P;                  // Gaussians (for the per-Gaussian arrays' sizes)
    unsigned long long* total;  // [16] number of instances, largest per-tile instance count, [2] 1 = tile_count holds the blend tile
                                //      order, [3] plan flags (SCR_PLAN_*, raised by preprocess_kernel), [4..12] first entry of
                                //      every XCD's list in the tile order
    size_t bytes;
};

inline __host__ GeomView geom_view(void* base, int64_t P, int H, int W) {
    Grid g(H, W);
    char* p = (char*)base;
    size_t off = 0;
    GeomView v;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align_up(n); return q; };
    size_t nblk = (size_t)((P + BIN_GPW - 1) / BIN_GPW);
    v.rec = (float4*)take((size_t)P * REC_F * 4);
    v.tiles_touched = (uint32_t*)take((size_t)P * 4);
    v.point_offsets = (uint32_t*)take((size_t)P * 4);
    v.clamped = (uint8_t*)take((size_t)P);
    v.gm_base = (uint2*)take((size_t)P * 8);
    v.live_bits = (uint32_t*)take((size_t)P * 4);
    v.has_rec = (uint8_t*)take((size_t)P);
    v.block_sums = (uint32_t*)take((nblk + 1) * 4);
    v.tile_count = (uint32_t*)take((size_t)g.tiles * 4);
    v.ranges = (uint32_t*)take((size_t)g.tiles * 8);
    v.cursor = (uint32_t*)take((size_t)g.tiles * 4);
    v.total = (unsigned long long*)take(128);
    v.bytes = off;
    v.P = P;
    return v;
}

// Deep tile lists (MatrixCity scale: tens of thousands of entries per tile, a few hundred of them in front of the pixels'
// last contributors): the tile sort does not materialise gm_index -- one 8-byte gm_base gather per instance, most of them
// for entries the backward pass never visits -- and the blend backward derives the index of the entries it does visit
// from gm_base itself.  With every entry live (the benchmark density) the sort's coalesced gm_index is the cheaper way
// round (measured both ways: tile sort 5.1 -> 4.0 ms at 20 M anchors; blend backward +4 % at cfg1 / cfg2).
// Both passes derive the choice from the same two numbers.
extern int g_force_deep_lists;      // capi.hip; scr_debug_force_deep_lists: -1 = by the rule below, 0 / 1 = forced (parity tests)
inline __host__ bool deep_lists(int64_t I, int tiles) {
    return g_force_deep_lists >= 0 ? g_force_deep_lists != 0 : I > (int64_t)8192 * tiles;
}

// per-Gaussian record flags (has_rec): from which mean list length on the blend backward marks the Gaussians it wrote a
// record for and preprocess_backward treats the others like invisible ones.  Measured (round 3): with 2048 entries per
// tile on average and more the flags pay (preprocess_backward 0.88 -> 0.79 ms at cfg2, 0.97 -> 0.86 at cfg3, 4.6 -> 2.2 at
// cfg4; the byte stores are lost in the blend backward); at the benchmark density (540 per tile, nearly every Gaussian has
// a record) they cost the blend backward 1 - 2 %.  Forced together with the deep-list variants by the test hook.
constexpr int FLAGS_MIN_MEAN = 2048;
inline __host__ bool record_flags(int64_t I, int tiles) {
    return g_force_deep_lists >= 0 ? g_force_deep_lists != 0 : I > (int64_t)FLAGS_MIN_MEAN * tiles;
}

// ---- binning buffer: per (Gaussian, tile) instance lists ----
struct BinView {
    unsigned long long* keys;  // [I] grouped by tile, unsorted: depth bits << 32 | gaussian id << 4 | quadrant mask
    uint32_t* point_list;      // [I] Gaussian ids, per tile sorted by (depth, id)
    uint32_t* gm_index;        // [I] sorted position -> Gaussian-major instance index (where its gradient record goes)
    uint8_t* qmask;            // [I] sorted position -> 4-bit mask of the tile's 8x8 quadrants the splat can touch
    unsigned long long* keys2; // [I] second buffer of the merge passes (only when a tile exceeds one sort chunk)
    size_t bytes;
};

inline __host__ BinView bin_view(void* base, int64_t I, int64_t max_tile_instances) {
    char* p = (char*)base;
    size_t off = 0;
    BinView v;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align_up(n); return q; };
    size_t n = (size_t)(I > 0 ? I : 1);
    v.keys = (unsigned long long*)take(n * 8);
    v.point_list = (uint32_t*)take(n * 4);
    v.gm_index = (uint32_t*)take(n * 4);
    v.qmask = (uint8_t*)take(n);
    const bool merge = max_tile_instances > WG_SORT_MAX;  // tiles above one workgroup sort chunk (binning.hip) take global merge passes
    v.keys2 = merge ? (unsigned long long*)take(n * 8) : nullptr;
    v.bytes = off;
    return v;
}

// ---- backward scratch: one gradient record per (Gaussian, tile) instance at its Gaussian-major index: nine floats
// (Y = opacity G dL/dalpha: sum Y dx, sum Y dy, sum Y dx^2, sum Y dx dy | sum Y dy^2, sum Y, c0, c1 | c2), 36 bytes, back to back.  The two
// four-float parts are moved with dword-aligned 16-byte accesses (gfx950 global memory needs dword alignment only);
// separate arrays per part were measured and rejected: the lone 4-byte stores cost a whole 32-byte sector each.
struct __attribute__((packed, aligned(4))) GradQuad { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) GradRec { GradQuad a, b; float c; };
static_assert(sizeof(GradRec) == 36, "36-byte gradient records");

// ---- image buffer ----
struct ImgView {
    float* final_T;       // [H*W]
    uint32_t* n_contrib;  // [H*W]
    unsigned long long* cut_key;  // [tiles] written by the blend backward: sort key (depth bits << 32 | id) of the tile's first
                                  //         list entry that got no gradient record (~0: all have one)
    size_t bytes;
};

inline __host__ ImgView img_view(void* base, int H, int W) {
    char* p = (char*)base;
    size_t off = 0;
    ImgView v;
    auto take = [&](size_t n) { char* q = p ? p + off : nullptr; off += align_up(n); return q; };
    v.final_T = (float*)take((size_t)H * W * 4);
    v.n_contrib = (uint32_t*)take((size_t)H * W * 4);
    v.cut_key = (unsigned long long*)take((size_t)Grid(H, W).tiles * 8 + 8);   // [tiles] is the stamp word, see blend.hip
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

// Per-lane select through an SGPR-pair lane mask.  hipcc usually emits the VOP2 form
// `v_cndmask_b32_e32 ..., vcc`, which gfx950 issues at ~23 cycles per wave instruction against
// ~4.5 for the VOP3 form with the mask in an SGPR pair (tools/exp/valu_probe2.hip), so the hot
// loops spell the VOP3 form out.  mask bit set -> t, clear -> f.
__device__ __forceinline__ float sel(unsigned long long mask, float t, float f) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t sel(unsigned long long mask, uint32_t t, uint32_t f) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask));
    return r;
}
// IEEE minNum without the canonicalising v_max the compiler puts in front of fminf's v_min
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned long long lanes(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// any of three values NaN or +-Inf (v_cmp_class_f32: signalling NaN, quiet NaN, -Inf, +Inf)
__device__ __forceinline__ bool nonfinite3(float a, float b, float c) {
    constexpr int NAN_OR_INF = 0x1 | 0x2 | 0x4 | 0x200;
    return __builtin_amdgcn_classf(a, NAN_OR_INF) || __builtin_amdgcn_classf(b, NAN_OR_INF) || __builtin_amdgcn_classf(c, NAN_OR_INF);
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
// XCD of a tile when the scene is not uniform (tile_order_kernel): 4x4-tile blocks dealt round-robin, so that every XCD
// gets the same mix of dense and sparse image regions (the contiguous bands of xcd_tile leave the XCDs of the image
// centre with most of the work, and blocks are dealt to the XCDs in turn: the launch runs at the pace of the slowest)
__host__ __device__ inline int xcd_of_tile(int t, int gx) {
    const int tx = t % gx, ty = t / gx;
    return ((tx >> 2) + (ty >> 2) * ((gx + 3) >> 2)) % NUM_XCD;
}

// Conservative sub-tile culling.  A 16x16 tile is blended by four waves, one per 8x8-pixel
// quadrant (bit q = (y half << 1) | x half).  Bit q is CLEARED only when no pixel of the quadrant
// can pass the blend kernels' `alpha >= 1/255` test: the minimum of the quadratic form
// q(d) = -power(d) over the quadrant's (continuous) pixel box is bounded below, and the splat is
// dropped when o * exp(-qmin) is below 1/255 by a safety margin that dwarfs the rounding
// difference between this bound and the per-pixel evaluation.  Dropping such a splat cannot
// change any result (it would be skipped by every pixel anyway), so lists, n_contrib and images
// stay bit-identical to the un-culled semantics.  NaNs compare false -> kept.
// Conservative sub-tile culling.  A splat contributes to a pixel only where its quadratic form
// q(d) = a dx^2 + b dx dy + c dy^2 stays below thr = ln(255 * opacity) (alpha >= 1/255); a pixel box can be
// skipped when the exact minimum of q over the box exceeds thr by more than the rounding of the terms.
struct SplatForm {
    float mx, my, a, b, c, thr;
    float vy, vx;  // minimiser slopes along an edge: dy* = vy * dx on a vertical edge, dx* = vx * dy on a horizontal one
    bool ok;       // finite and positive definite; anything else is simply kept everywhere
};
__device__ inline SplatForm splat_form(float4 r0, float4 r1) {
    SplatForm f;
    f.mx = r0.x; f.my = r0.y;
    f.a = -r0.z; f.b = -r0.w; f.c = -r1.x;  // q = a dx^2 + b dx dy + c dy^2, a,c > 0
    f.thr = __logf(255.0f * r1.y) + 0.01f;  // q above this -> alpha < 1/255 (with margin)
    f.ok = (f.a > 0.0f && f.c > 0.0f && 4.0f * f.a * f.c - f.b * f.b > 0.0f) && (f.mx - f.mx == 0.0f) &&
           (f.my - f.my == 0.0f) && (f.thr - f.thr == 0.0f);
    // one division per direction and Gaussian instead of one per edge: q is stationary at the minimiser, so
    // the last-bit difference to (-b * fix) / (2 k) moves the edge minimum by a second-order amount, far
    // inside the margins below
    f.vy = -f.b / (2.0f * f.c);
    f.vx = -f.b / (2.0f * f.a);
    return f;
}
// The four quadrants of a tile at once.  Same minimum as box_reachable, found with two edge minima per quadrant
// instead of four: q is convex with its minimiser at d = 0, so over a box that does not contain 0 the minimum lies on a
// face that is VISIBLE from 0 (from a point of any other face a step towards 0 stays inside the box and lowers q) --
// at most the vertical edge dx = cx and the horizontal edge dy = cy, where (cx, cy) = 0 clamped into the box
// (v_med3_f32).  A coordinate of 0 that already lies inside the box's range gives cx = 0: the "edge" dx = 0 is then just
// another line through the box (a valid upper bound of the minimum, equal to it when 0 is inside the box: q = 0), so no
// case distinction is left.  Everything that depends on one axis only is shared by the two quadrants of a row / column.
// The splat is kept when either candidate stays within the threshold plus the rounding margin of its own terms.
// S = edge of the four boxes in pixels: 8 = the quadrants of a tile (origin = the tile's), 4 = the 4x4-pixel sub-blocks of a
// quadrant (origin = the quadrant's; blend.hip packs four of them into a wave).  Bit (hy << 1) | hx.
template <int S>
__device__ inline uint32_t box_mask4(float4 r0, float4 r1, int box_x0, int box_y0) {
    const SplatForm f = splat_form(r0, r1);
    if (!f.ok) return 0xfu;
    const float x0 = (float)box_x0, y0 = (float)box_y0;
    float dxl[2], dxh[2], dyl[2], dyh[2];     // d = mean - pixel over the half's columns / rows (low, high end)
    float cx[2], cy[2], qx[2], qy[2], bcx[2], bcy[2], sy[2], sx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        dxh[h] = f.mx - (x0 + (float)(S * h));
        dxl[h] = f.mx - (x0 + (float)(S * h + S - 1));
        dyh[h] = f.my - (y0 + (float)(S * h));
        dyl[h] = f.my - (y0 + (float)(S * h + S - 1));
        cx[h] = __builtin_amdgcn_fmed3f(0.0f, dxl[h], dxh[h]);
        cy[h] = __builtin_amdgcn_fmed3f(0.0f, dyl[h], dyh[h]);
        qx[h] = f.a * cx[h] * cx[h];          // the fixed coordinate's own term
        qy[h] = f.c * cy[h] * cy[h];
        bcx[h] = f.b * cx[h];
        bcy[h] = f.b * cy[h];
        sy[h] = f.vy * cx[h];                 // unclamped minimiser along the edge
        sx[h] = f.vx * cy[h];
    }
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int hx = q & 1, hy = q >> 1;
        const float v = __builtin_amdgcn_fmed3f(sy[hx], dyl[hy], dyh[hy]);
        const float t1 = bcx[hx] * v, t2 = f.c * v * v;
        const float qv = qx[hx] + t1 + t2, magv = qx[hx] + fabsf(t1) + t2;
        const float u = __builtin_amdgcn_fmed3f(sx[hy], dxl[hx], dxh[hx]);
        const float s1 = bcy[hy] * u, s2 = f.a * u * u;
        const float qh = qy[hy] + s1 + s2, magh = qy[hy] + fabsf(s1) + s2;
        const bool far = qv > __builtin_fmaf(1.0e-5f, magv, f.thr) && qh > __builtin_fmaf(1.0e-5f, magh, f.thr);
        if (!far) mask |= 1u << q;
    }
    return mask;
}
__device__ inline uint32_t quadrant_mask(float4 r0, float4 r1, int tile_x0, int tile_y0) { return box_mask4<8>(r0, r1, tile_x0, tile_y0); }


// Several buffers cleared by ONE kernel launch (preprocess.hip).  hipMemsetAsync costs a 5 us fill kernel AND 8 - 11 us of
// idle stream time in front of it per call (profiles/r03x_step_gaps.txt): one per forward pass at cfg1, eighteen per
// tri-plane backward.  Pointers 4-byte aligned, sizes in bytes (multiples of 4).
struct ZeroList {
    static constexpr int MAX = 20;
    void* p[MAX];
    unsigned long long n[MAX];
    int count = 0;
    bool full() const { return count >= MAX; }
    // queue one buffer; a full list is launched first (on `st`) and restarted, so that no buffer is ever dropped
    void add(void* ptr, size_t bytes, hipStream_t st);
};
void launch_zero(const ZeroList& z, hipStream_t st);      // launches nothing for an empty list
void launch_zero_far_records(int64_t P, const GeomView& gv, GradRec* grad_rec, hipStream_t st);
inline void ZeroList::add(void* ptr, size_t bytes, hipStream_t st) {
    if (!ptr || !bytes) return;
    if (full()) { launch_zero(*this, st); count = 0; }
    p[count] = ptr;
    n[count] = bytes;
    ++count;
}

// launchers (defined in the .hip files)
void launch_box_coords(int64_t V, const float* xyz, const float* lo, const float* hi, float* out, hipStream_t st);
int launch_nl_fold(int L, int d, const int* dd, const int* col, const unsigned char* col_at, const float* const* W,
                   const float* const* b, const float* const* gamma, const float* const* beta, float* G, float* c,
                   hipStream_t st);
int launch_nl_fold_backward(int L, int d, const int* dd, const int* col, const unsigned char* col_at, const float* const* W,
                            const float* const* gamma, const float* const* beta, const float* dG, const float* dc,
                            float* const* dW, float* const* db, float* const* dgamma, float* const* dbeta, hipStream_t st);
int launch_nl_running_stats(int L, int d, const int* dd, const int* col, const unsigned char* col_at, const float* momentum,
                            float* const* run_mean, float* const* run_var, long long* const* batches, const float* mean,
                            const float* var, int64_t n, hipStream_t st);
void launch_filter(int64_t P, const float* means3D, const float* scales, const float* rotations,
                   const float* cov3D, const KSettings& ks, int32_t* radii, hipStream_t st);
void launch_mark_visible(int64_t P, const float* means3D, const float* view, uint8_t* out, hipStream_t st);
void launch_preprocess(int64_t P, int M, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D, const float* opacities, const float* shs, const float* colors,
                       const KSettings& ks, const GeomView& gv, int32_t* radii, hipStream_t st);
void launch_plan_scans(int64_t P, const KSettings& ks, const GeomView& gv, unsigned long long* mailbox,
                       unsigned long long seq, hipStream_t st);
void launch_scatter(int64_t P, const KSettings& ks, const GeomView& gv, const BinView& bv, unsigned long long cap_instances,
                    hipStream_t st);
void launch_tile_sort(const KSettings& ks, const GeomView& gv, const BinView& bv, int64_t max_tile_instances,
                      bool with_gm_index, hipStream_t st);
void launch_blend_forward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                          float* out_color, bool longest_first, bool safe, hipStream_t st);
void launch_blend_backward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                           const float* dL_dcolor, GradRec* grad_rec, unsigned long long stamp, bool deep, bool flags,
                           bool safe, hipStream_t st);
void launch_preprocess_backward(int64_t P, int M, const float* means3D, const float* scales,
                                const float* rotations, const float* cov3D, const float* shs,
                                const KSettings& ks, const int32_t* radii, const GeomView& gv,
                                const BinView& bv, const GradRec* grad_rec, const unsigned long long* cut_key,
                                unsigned long long stamp, bool deep, float* dL_dmeans3D,
                                float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh, float* dL_dopacity,
                                float* dL_dscales, float* dL_drotations, float* dL_dcov3D, hipStream_t st);

void launch_expand_count(int64_t n, const float* neural_opacity, uint32_t* wg_count, unsigned long long* total,
                         unsigned long long* mailbox, unsigned long long seq, hipStream_t st);
void launch_mask_count(int64_t n, const uint8_t* mask, uint32_t* wg_count, unsigned long long* total,
                       unsigned long long* mailbox, unsigned long long seq, hipStream_t st);
void launch_mask_index(int64_t n, const uint8_t* mask, const uint32_t* wg_offset, int64_t* index, int64_t* inverse,
                       hipStream_t st);
void launch_expand_run(int64_t n, int k, const float* neural_opacity, const float* color, const float* scale_rot,
                       const float* offsets, int ldo, const float* grid_scaling, const float* anchor,
                       const uint32_t* wg_offset, int32_t* out_index, uint8_t* mask_out, float* xyz,
                       float* color_out, float* opacity, float* scaling, float* rot, hipStream_t st);
void launch_expand_backward(int64_t V, int k, const float* scale_rot, const float* offsets, int ldo,
                            const float* grid_scaling, const int32_t* out_index, const float* g_xyz,
                            const float* g_color, const float* g_opacity, const float* g_scaling,
                            const float* g_rot, float* d_neural_opacity, float* d_color, float* d_scale_rot,
                            float* d_offsets, float* d_grid_scaling, float* d_anchor, const float* g_reg, int64_t P,
                            hipStream_t st);

void launch_statis_compute(int64_t V, int k, const float* neural_opacity, const int32_t* out_index,
                           const uint8_t* update_filter, const float* grad, int gstride, float* inc_opacity,
                           float* inc_grad, hipStream_t st);
void launch_statis_apply(int64_t V, int k, const int64_t* visible_index, const float* inc_opacity, const float* inc_grad,
                         float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                         hipStream_t st);

constexpr int NL_DP = 80;               // padded width of the BatchNorm-Linear's coefficient rows AND of a row of column statistics
int anchor_gather_stat_rows(int64_t V);  // rows of column statistics the gather hands to the BatchNorm-Linear
int64_t anchor_gather_stat_buffer_rows(int64_t V);      // rows of the buffer it needs for them (per-tile rows behind the result)
void launch_anchor_gather(int64_t V, const int64_t* idx, const float* p_feat, const float* p_anchor, const float* p_offset,
                          const float* p_scaling, float* feat, float* anchor, float* offsets, float* grid_scaling,
                          float* g_fea, int ldg, float* stats, hipStream_t st);
void launch_anchor_gather_backward(int64_t N, int64_t V, const int64_t* inv, const float* grid_scaling, const float* d_feat,
                                   const float* d_anchor, const float* d_offsets, const float* d_grid_scaling,
                                   const float* d_g_fea, int ldg, float* g_feat, float* g_anchor, float* g_offset,
                                   float* g_scaling, int accumulate, const float* nl_coef, const float* nl_dy, int nl_lddy,
                                   const float* nl_x, int nl_ldx, hipStream_t st);
void launch_knn(int64_t N, int k, const float* grid9, const float* sorted_pts, const int64_t* sorted_id,
                const int32_t* cell_start, int64_t* out_idx, hipStream_t st);
void launch_knn_curvature(int64_t N, int k, const float* pts, const int64_t* idx, float* curvature, hipStream_t st);
size_t norm_linear_scratch_bytes(int64_t V);
int launch_norm_linear_forward(int64_t V, int d, const float* x, int ldx, const float* G, const float* c, float eps, float* y,
                               float* mean, float* var, float* inv, void* scratch, const float* stats, int stat_rows,
                               hipStream_t st);
int launch_norm_linear_backward(int64_t V, int d, const float* x, int ldx, const float* dy, int lddy, const float* G,
                                const float* mean, const float* inv, float* dx, int lddx, float* dG, float* dc, void* scratch,
                                float* coef_out, hipStream_t st);
size_t mlp_heads_hidden_bytes(int64_t V);
size_t mlp_heads_partial_bytes(int64_t V);
void launch_mlp_heads_forward(int64_t V, const float* feat, int ldf, const float* anchor, const float* campos, const float* geo_a, const float* geo_b,
                              const float* w1, const float* b1, const float* w2o, const float* b2o, const float* w2c,
                              const float* b2c, const float* w2v, const float* b2v, void* hidden_save, float* out_o,
                              float* out_c, float* out_v, hipStream_t st);
void launch_mlp_heads_backward(int64_t V, const float* feat, int ldf, const float* anchor, const float* campos, const float* geo_a, const float* geo_b,
                               const float* w1, const float* w2o, const float* w2c, const float* w2v,
                               const void* hidden_save, const float* out_o, const float* out_c, const float* g_o,
                               const float* g_c, const float* g_v, void* partial, float* d_feat, float* d_anchor,
                               float* d_geo_a, float* d_geo_b, float* d_w1, float* d_b1, float* d_w2o, float* d_b2o, float* d_w2c,
                               float* d_b2c, float* d_w2v, float* d_b2v, hipStream_t st);

// plane attention of the level-0 grid (attention.hip)
size_t tpa_scratch_bytes(int R, int H, int W);
void launch_tpa_stats(int R, int64_t HW, const float* p0, const float* p1, const float* p2, float* avg, float* mx,
                      int* arg, void* scratch, hipStream_t st);
void launch_tpa_forward(int R, int H, int W, const float* p0, const float* p1, const float* p2, const float* ca,
                        const float* w, float* s, uint8_t* am, float* sa, float* o0, float* o1, float* o2, hipStream_t st);
void launch_tpa_backward(int R, int H, int W, const float* p0, const float* p1, const float* p2, const float* ca,
                         const float* w, const float* s, const uint8_t* am, const float* sa, const float* g0,
                         const float* g1, const float* g2, float* d0, float* d1, float* d2, float* dca, float* dw,
                         void* scratch, hipStream_t st);
void launch_tpa_backward_stats(int R, int64_t HW, const float* davg, const float* dmx, const int* arg, float* d0,
                               float* d1, float* d2, hipStream_t st);

size_t l1_ssim_scratch_bytes(int C, int H, int W, int with_grad);
size_t scaling_reg_scratch_bytes(int64_t P);
size_t pair_l1_scratch_bytes(int64_t n);
void launch_pair_l1_forward(int64_t n, const float* g1, const float* g2, const float* r1, const float* r2, void* scratch, float* out,
                            hipStream_t st);
void launch_pair_l1_backward(int64_t n, const float* g1, const float* g2, const float* r1, const float* r2, const float* g, float* d1,
                             float* d2, hipStream_t st);
void launch_scaling_reg_forward(int64_t P, const float* s, void* scratch, float* out, hipStream_t st);
void launch_scaling_reg_backward(int64_t P, const float* s, const float* g, float* ds, hipStream_t st);
void launch_l1_ssim_forward(int C, int H, int W, const float* img1, const float* img2, void* scratch,
                            int with_grad, float* out2, hipStream_t st);
void launch_l1_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const void* scratch,
                             const float* g_l1, const float* g_ssim, float* dimg1, hipStream_t st);
size_t triplane_multi_scratch_bytes(int64_t V, int ngrids, const int* R, const int* X, const int* Y, const int* Z);
int launch_triplane_backward_multi(int64_t V, const float* coords, int cs, int ngrids, const int* R, const int* X, const int* Y,
                                   const int* Z, const int* col, const float* grad, int ld, float* const* grad_planes,
                                   void* scratch, const float* nl_coef, const float* nl_dy, int nl_lddy, const float* nl_x,
                                   int nl_ldx, hipStream_t st);
int launch_norm_linear_dx(int64_t V, int d, const float* x, int ldx, const float* dy, int lddy, const float* coef3, float* dx,
                          int lddx, hipStream_t st);
int launch_plane_row_pairs(int R, int A, int B, const float* plane, float* pairs, hipStream_t st);
size_t triplane_scratch_bytes(int64_t V, int A, int B, int channels);
size_t triplane_backward_scratch_bytes(int64_t V, int X, int Y, int Z, int channels);
int launch_triplane_backward(int64_t V, const float* coords, int cs, int R, int X, int Y, int Z, int planes,
                             const float* grad_out, int ld, const int* cols, float* const* grad_planes, void* scratch,
                             hipStream_t st);
int launch_plane_sample_backward(int64_t V, const float* coords, int cs, int cx, int cy, int R, int A, int B, int planes,
                                 const float* grad_out0, const float* grad_out1, int ld, float* grad_plane0,
                                 float* grad_plane1, void* scratch, hipStream_t st);
int launch_triplane_forward(int64_t V, const float* coords, int cs, const float* xy, const float* xz, const float* yz,
                             int R, int X, int Y, int Z, int channel_last, float* out, int ld, int col_xy, int col_xz,
                             int col_yz, hipStream_t st);

}  // namespace scr
