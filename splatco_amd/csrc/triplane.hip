// splatco_amd/csrc/triplane.hip -- tri-plane bilinear feature sampling, forward and backward (gfx950).
//
// Forward: one thread per anchor samples all three planes and writes its 3*R features as one
// contiguous row segment of the caller's [V, ld] matrix (no per-plane outputs, no concatenation).
// For many points the planes are first rewritten as row pairs (plane_row_pairs_kernel, tp_sample_plane_rp): the
// forward is bound by the number of random cache lines it requests, and a sample's four corners are then contiguous.
//
// PlaneGrid samples three learnable planes [1,R,A,B] at V anchor positions with
// F.grid_sample(bilinear, align_corners=True, zeros padding) (scene/grids.py:146-182).  The
// backward is a scatter-add of 4 corners x R channels per point into a dense plane gradient:
// torch issues one global float atomic per (point, corner, channel) -- 92 M atomics per call at
// 4.6 M anchors, and MI355X retires ~28 G device atomics/s whatever their scope, so those calls
// are 41 % of the whole render() step at BASELINE.json configs[2].
//
// Here the points are first bucketed by 32x32-cell plane tile (the same LDS-aggregated count / reserve / place
// scheme as the rasterizer's tile binning), for the three projections of a grid in ONE pass over the points that
// moves each point's coordinates and gradient piece into the tile's run of 32 / 48-byte records
// (tp_scatter_kernel); one workgroup per tile then sorts chunks of its run by cell in LDS and every thread sums,
// in registers, the four corner contributions of the cell it owns (tp_cell_gather_kernel); the 128 border nodes a tile
// shares with its neighbours go to a per-tile halo block and are added across tiles in a fixed order by
// tp_border_sum_kernel: no float atomics at all (round 2 flushed the border nodes with device atomics: 6.3 M per cfg2
// step, 0.56 ms of the 2.3 ms the plane backward took).
// A crowded tile (a flat or contracted scene) is cut into segments, one workgroup each (tp_scan_proj, tp_split_finish_kernel).
// Round 1 kept an index list per tile and gathered coordinates and gradients through it (0.43 ms per plane at
// 4.6 M points, 5.2 ms per cfg2 step); records + one pass per grid + cell-centred sums: 3.2 ms per step.
#include "common.h"
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

namespace scr {

#ifndef SCR_TP_TILE
#define SCR_TP_TILE 32
#endif
constexpr int TP_TILE = SCR_TP_TILE;   // cells per tile edge; a tile owns (TP_TILE + 1)^2 nodes
constexpr int TP_NODES = TP_TILE + 1;
constexpr int TP_BORDER = 4 * TP_TILE;  // nodes on a tile's border (the outer ring of its TP_NODES x TP_NODES nodes)
// position of border node (na, nb) in a tile's halo block: top row, bottom row, left column, right column
__host__ __device__ inline int tp_border_index(int na, int nb) {
    return na == 0 ? nb : (na == TP_TILE ? TP_NODES + nb : (nb == 0 ? 2 * TP_NODES + (na - 1) : 2 * TP_NODES + (TP_TILE - 1) + (na - 1)));
}
constexpr int TP_MAX_R = 16;           // channels per plane (R = num_channels / 3; a plain plane stacked on its attended twin: 2 R)
constexpr int TP_THREADS = 1024;
constexpr int TP_ROUNDS = 4;
constexpr int TP_PER_WG = TP_THREADS * TP_ROUNDS;

// grid_sample source index, align_corners=True: ((c + 1) / 2) * (size - 1)
__device__ __forceinline__ void tp_cell(float gx, float gy, int A, int B, int& a0, int& b0, float& fa, float& fb) {
    const float ix = ((gx + 1.0f) * 0.5f) * (float)(B - 1);  // x -> last dim (B)
    const float iy = ((gy + 1.0f) * 0.5f) * (float)(A - 1);  // y -> dim A
    const float fx = floorf(ix), fy = floorf(iy);
    b0 = (int)fx;
    a0 = (int)fy;
    fb = ix - fx;
    fa = iy - fy;
}

// tile of a point, or -1 when none of its four corners lies inside the plane (NaN included)
__device__ __forceinline__ int tp_tile_of(float gx, float gy, int A, int B, int tb) {
    const float ix = ((gx + 1.0f) * 0.5f) * (float)(B - 1), iy = ((gy + 1.0f) * 0.5f) * (float)(A - 1);
    if (!(ix > -1.0f && ix < (float)B && iy > -1.0f && iy < (float)A)) return -1;
    int a0 = (int)floorf(iy), b0 = (int)floorf(ix);
    a0 = min(max(a0, 0), A - 1);
    b0 = min(max(b0, 0), B - 1);
    return (a0 / TP_TILE) * tb + (b0 / TP_TILE);
}

__device__ __forceinline__ uint32_t tp_block_scan(uint32_t v, uint32_t* lds_waves, uint32_t& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds_waves[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) {
        const uint32_t s = lds_waves[q];
        if (q < w) base += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

// ---- backward with respect to the planes.  One pass over the points serves up to three PROJECTIONS (the xy / xz / yz
// planes of a grid): what it reads of a point -- the coordinate row and the grid's contiguous piece of the gradient row --
// is read once, in point order.
// record of a (point, projection): {grid x, grid y, RT gradient values}, padded to whole 16-byte pieces so that a record
// is written with REC / 4 stores (lone dword stores to random tiles are bound by the L2 request rate)
constexpr int tp_rec(int RT) { return (2 + RT + 3) / 4 * 4; }

struct TpProj {
    int cx, cy, A, B, tb, tiles;          // coordinate columns (grid x -> plane dim B, grid y -> dim A), plane size, tiles
    int col0, col1;                       // first gradient column of the plane (and of the second plane sampled with it)
    uint32_t *count, *start, *cursor;     // [tiles], [tiles + 1], [tiles]
    uint32_t* gmax;                       // [tiles]: bits of the largest |gradient value| among the tile's records (pass 3)
    uint32_t *seg, *pfirst;               // [tiles + 1] first pass-4 workgroup of a tile, [tiles] first partial-sum slot of a split tile
    uint32_t* split;                      // [0]: number of split tiles, [1 ...]: their indices
    long long* part;                      // partial cell sums of the segments of split tiles (shared by the projections of a call)
    float* rec;                           // [V][tp_rec(R * NP)]
    float* halo;                          // [tiles][TP_BORDER][R * NP]: every tile's sums for the nodes on its border
};
struct TpProjSet {
    TpProj p[3];
    int n;
};

// pass 1: per-tile point counts (LDS histograms per workgroup, one global atomic per touched tile)
__global__ void __launch_bounds__(TP_THREADS)
tp_count_kernel(int64_t V, const float* __restrict__ coords, int cs, TpProjSet ps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int total = ps.p[0].tiles + (ps.n > 1 ? ps.p[1].tiles + ps.p[2].tiles : 0);
    for (int t = threadIdx.x; t < total; t += TP_THREADS) hist[t] = 0;
    __syncthreads();
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        if (i >= V) break;
        int off = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q >= ps.n) break;
            const TpProj& pj = ps.p[q];
            const int t = tp_tile_of(coords[i * cs + pj.cx], coords[i * cs + pj.cy], pj.A, pj.B, pj.tb);
            if (t >= 0) atomicAdd(&hist[off + t], 1u);
            off += pj.tiles;
        }
    }
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q >= ps.n) break;
        for (int t = threadIdx.x; t < ps.p[q].tiles; t += TP_THREADS) {
            const uint32_t c = hist[off + t];
            if (c) atomicAdd(&ps.p[q].count[t], c);
        }
        off += ps.p[q].tiles;
    }
}

// A tile's run is summed by ONE workgroup of pass 4 -- unless the tile is crowded: a scene is not a uniform cloud (a city
// seen from above is a sheet: two of its three projections collapse onto one row of tiles; a contracted scene sits in the
// middle of every plane), and a workgroup per tile then leaves 20 workgroups with 200 k records each while the chip
// idles (4.6 M points: 1.1 ms uniform, 4.9 ms as a sheet).  A tile with more than TP_SPLIT_MIN records AND more than
// 1 / 256 of the plane's (twice what each of the 512 workgroups the chip holds at once would get of an even split: a uniform
// cloud is left alone however large) is cut into segments of TP_SEG records, one workgroup each; the segments' cell sums are 64-bit
// integers, so they add up to the same bits whichever records a segment happens to hold (tp_split_finish_kernel adds them).
#ifndef SCR_TP_SEG
#define SCR_TP_SEG 8192
#endif
constexpr uint32_t TP_SEG = SCR_TP_SEG, TP_SPLIT_MIN = 2 * TP_SEG;
// upper bounds: a split tile holds more than TP_SPLIT_MIN of the at most V records of a projection
static inline size_t tp_max_split_tiles(int64_t V) { return (size_t)(V > 0 ? V : 0) / TP_SPLIT_MIN + 1; }
static inline size_t tp_max_split_segments(int64_t V) { return (size_t)(V > 0 ? V : 0) / TP_SEG + tp_max_split_tiles(V) + 1; }

// pass 2 (one workgroup per projection): exclusive scan of the tile counts -> start[tiles + 1]; cursor = 0; pass 4's
// workgroups per tile (0 for an empty tile) -> seg[tiles + 1]; partial-sum slots of split tiles -> pfirst[tiles]
__device__ __forceinline__ void tp_scan_proj(const TpProj& pj, uint32_t* lds) {
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < pj.tiles; i += 1024) mine += pj.count[i];
    uint32_t total;
    (void)tp_block_scan(mine, lds, total);
    const uint32_t crowded = max(TP_SPLIT_MIN, total / 256u);     // twice a fair share of the 512 workgroups the chip holds at once
    uint32_t carry = 0, carry_s = 0, carry_p = 0, carry_l = 0;
    for (int base = 0; base < pj.tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < pj.tiles ? pj.count[i] : 0u;
        const uint32_t ns = v == 0 ? 0u : (v > crowded ? (v + TP_SEG - 1) / TP_SEG : 1u), np = ns > 1 ? ns : 0u;
        uint32_t tot, tot_s, tot_p;
        const uint32_t ex = tp_block_scan(v, lds, tot);
        const uint32_t ex_s = tp_block_scan(ns, lds, tot_s);
        const uint32_t ex_p = tp_block_scan(np, lds, tot_p);
        uint32_t tot_l;
        const uint32_t ex_l = tp_block_scan(np ? 1u : 0u, lds, tot_l);
        if (i < pj.tiles) {
            pj.start[i] = carry + ex;
            pj.cursor[i] = 0;
            pj.seg[i] = carry_s + ex_s;
            pj.pfirst[i] = carry_p + ex_p;
            if (np) pj.split[1 + carry_l + ex_l] = (uint32_t)i;
        }
        carry += tot;
        carry_s += tot_s;
        carry_p += tot_p;
        carry_l += tot_l;
    }
    if (threadIdx.x == 0) {
        pj.start[pj.tiles] = carry;
        pj.seg[pj.tiles] = carry_s;
        pj.split[0] = carry_l;
    }
}
__global__ void __launch_bounds__(1024)
tp_scan_kernel(TpProjSet ps) {
    __shared__ uint32_t lds[1024 / WAVE];
    const TpProj pj = blockIdx.x == 0 ? ps.p[0] : (blockIdx.x == 1 ? ps.p[1] : ps.p[2]);
    tp_scan_proj(pj, lds);
}

// pass 3: one RECORD per (point, projection), grouped by tile: {grid x, grid y, the RT = R * NP gradient values}.
// The point order is random with respect to the planes, so whatever pass 4 needs of a point is moved here into the
// tile's contiguous run of records, and pass 4 reads nothing but its own run, fully coalesced.  With an index list
// instead (round 1) pass 4 gathered a 16-byte coordinate row and a 20-byte gradient piece per point through two
// 128-byte lines each: 0.33 ms per plane at 4.6 M points, the random-line rate of HBM.
// NP (1 or 2) gradient blocks of R columns share a projection and its plane size: the attention grid samples the
// plain and the attended plane at the same positions (scene/grids.py:174-181).
// SPAN consecutive floats from p, which reaches 16-byte alignment after H of them (the same H for every row when the
// row stride is a multiple of 16 bytes): head, 16-byte body, tail -- a quarter of the requests of dword loads, and
// with rows scattered over as many cache lines as there are lanes the request count is what a load costs
template <int SPAN, int H>
__device__ __forceinline__ void tp_load_span(const float* __restrict__ p, float (&g)[SPAN]) {
    constexpr int K0 = (H & 1) && SPAN >= 1 ? 1 : 0;
    constexpr int K1 = K0 + (((H & 2) && SPAN - K0 >= 2) ? 2 : 0);
    constexpr int KT = K1 + (SPAN - K1) / 4 * 4;
    if constexpr (K0 != 0) g[0] = p[0];
    if constexpr (K1 > K0) {
        const float2 v = *(const float2*)(p + K0);
        g[K0] = v.x;
        g[K0 + 1] = v.y;
    }
    if constexpr (K1 == H) {
#pragma unroll
        for (int k = K1; k < KT; k += 4) {
            const float4 v = *(const float4*)(p + k);
            g[k] = v.x;
            g[k + 1] = v.y;
            g[k + 2] = v.z;
            g[k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = K1; k < KT; ++k) g[k] = p[k];
    }
    if constexpr (SPAN - KT >= 2) {
        const float2 v = *(const float2*)(p + KT);
        g[KT] = v.x;
        g[KT + 1] = v.y;
    }
    if constexpr (((SPAN - KT) & 1) != 0) g[SPAN - 1] = p[SPAN - 1];
}

// bits of max |g[k]|: |x| as an unsigned integer orders like |x| (NaN above infinity), so the tile maximum can be
// formed with integer atomicMax in any order -- the one quantity of this backward that does not depend on arrival order
template <int N>
__device__ __forceinline__ uint32_t tp_abs_max_bits(const float* g) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) m = max(m, __float_as_uint(g[k]) & 0x7fffffffu);
    return m;
}

// FAST: three projections whose gradient blocks lie side by side in projection order (projection q, plane s at column
// span0 + (q * NP + s) * R: the layout of scene/grids.py:165,181) in rows of a 16-byte-multiple stride: the grid's
// 3 * RT columns of a row are loaded once, with wide loads, and record q is the q-th third of them.
template <int R, int NP, bool FAST>
__global__ void __launch_bounds__(TP_THREADS)
tp_scatter_kernel(int64_t V, const float* __restrict__ coords, int cs, const float* __restrict__ grad, int ld, int span0,
                  TpProjSet ps) {
    constexpr int RT = R * NP, REC = tp_rec(RT);
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const int total = ps.p[0].tiles + (ps.n > 1 ? ps.p[1].tiles + ps.p[2].tiles : 0);
    uint32_t* hmax = hist + total;        // per tile: the largest |gradient value| this workgroup placed there (bits)
    for (int t = threadIdx.x; t < 2 * total; t += TP_THREADS) hist[t] = 0;
    __syncthreads();
    int tl[TP_ROUNDS][3];
#pragma unroll
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        int off = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            tl[r][q] = -1;
            if (q < ps.n && i < V) {
                const TpProj& pj = ps.p[q];
                const int t = tp_tile_of(coords[i * cs + pj.cx], coords[i * cs + pj.cy], pj.A, pj.B, pj.tb);
                if (t >= 0) {
                    tl[r][q] = off + t;
                    atomicAdd(&hist[off + t], 1u);
                }
                off += pj.tiles;
            }
        }
    }
    __syncthreads();
    {
        int off = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q >= ps.n) break;
            const TpProj& pj = ps.p[q];
            for (int t = threadIdx.x; t < pj.tiles; t += TP_THREADS) {
                const uint32_t c = hist[off + t];
                if (c) hist[off + t] = pj.start[t] + atomicAdd(&pj.cursor[t], c);
            }
            off += pj.tiles;
        }
    }
    __syncthreads();
    const int head = (4 - (span0 & 3)) & 3;       // floats before the span reaches 16-byte alignment (FAST only)
#pragma unroll
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        if (tl[r][0] < 0 && tl[r][1] < 0 && tl[r][2] < 0) continue;
        const float* row = grad + (size_t)i * ld;
        float span[FAST ? 3 * RT : 1];
        if (FAST) {
            float(&sp)[3 * RT] = reinterpret_cast<float(&)[3 * RT]>(span);
            switch (head) {
                case 0: tp_load_span<3 * RT, 0>(row + span0, sp); break;
                case 1: tp_load_span<3 * RT, 1>(row + span0, sp); break;
                case 2: tp_load_span<3 * RT, 2>(row + span0, sp); break;
                default: tp_load_span<3 * RT, 3>(row + span0, sp); break;
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (tl[r][q] < 0) continue;
            const TpProj& pj = ps.p[q];
            float v[REC];
            v[0] = coords[i * cs + pj.cx];
            v[1] = coords[i * cs + pj.cy];
#pragma unroll
            for (int k = 0; k < REC - 2; ++k) {
                if (k >= RT) v[2 + k] = 0.0f;
                else if (FAST) v[2 + k] = span[FAST ? q * RT + k : 0];
                else v[2 + k] = row[(k < R ? pj.col0 : pj.col1 - R) + k];
            }
            atomicMax(&hmax[tl[r][q]], tp_abs_max_bits<RT>(v + 2));
            float4* dst = (float4*)(pj.rec + (size_t)atomicAdd(&hist[tl[r][q]], 1u) * REC);
#pragma unroll
            for (int k = 0; k < REC / 4; ++k) dst[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        }
    }
    __syncthreads();
    {
        int off = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q >= ps.n) break;
            for (int t = threadIdx.x; t < ps.p[q].tiles; t += TP_THREADS)
                if (hmax[off + t]) atomicMax(&ps.p[q].gmax[t], hmax[off + t]);
            off += ps.p[q].tiles;
        }
    }
}

// ---- all grids of a step in ONE pass over the points (up to three grids = nine projections that sample the same
// coordinates; FeaturePlanes: the attention grid with its planes stacked, R = 2 r, and one or two plain grids, R = r).
// The [V, ld] gradient matrix and the coordinates are read once instead of once per grid, with whole 16-byte loads.
struct TpProjSet9 {
    TpProj p[9];
    int n;          // 3 * number of grids
};

__device__ __forceinline__ float tp_pick(float x, float y, float z, int c) { return c == 0 ? x : (c == 1 ? y : z); }

__global__ void __launch_bounds__(TP_THREADS)
tp_count9_kernel(int64_t V, const float* __restrict__ coords, int cs, TpProjSet9 ps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    int total = 0;
    for (int q = 0; q < ps.n; ++q) total += ps.p[q].tiles;
    for (int t = threadIdx.x; t < total; t += TP_THREADS) hist[t] = 0;
    __syncthreads();
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        if (i >= V) break;
        const float x = coords[i * cs], y = coords[i * cs + 1], z = coords[i * cs + 2];
        int off = 0;
        for (int q = 0; q < ps.n; ++q) {
            const int t = tp_tile_of(tp_pick(x, y, z, ps.p[q].cx), tp_pick(x, y, z, ps.p[q].cy), ps.p[q].A, ps.p[q].B, ps.p[q].tb);
            if (t >= 0) atomicAdd(&hist[off + t], 1u);
            off += ps.p[q].tiles;
        }
    }
    __syncthreads();
    int off = 0;
    for (int q = 0; q < ps.n; ++q) {
        for (int t = threadIdx.x; t < ps.p[q].tiles; t += TP_THREADS) {
            const uint32_t c = hist[off + t];
            if (c) atomicAdd(&ps.p[q].count[t], c);
        }
        off += ps.p[q].tiles;
    }
}

__global__ void __launch_bounds__(1024)
tp_scan9_kernel(TpProjSet9 ps) {
    __shared__ uint32_t lds[1024 / WAVE];
    const TpProj pj = ps.p[blockIdx.x];
    tp_scan_proj(pj, lds);
}

// grid g (0..2) has RTg channels per projection (0 = grid absent); its projection q owns columns
// base_g + q * RTg .. + RTg of the span that starts at column span0 (a multiple of 4; rows 16-byte aligned)
// DX (round 6): the gradient rows are not read but FORMED here.  The sampled matrix has one consumer, the BatchNorm-Linear of
// FeaturePlanes' plane branch, whose backward ends in dx[v][n] = k0[n] + x[v][n] k1[n] + sum_m dy[v][m] Gi[m][n]
// (csrc/normlinear.hip) -- a [V,60] matrix written by one kernel and read back by this one.  With the coefficient blocks
// (coef = Gi [32][NL_DP] | k0 | k1: uniform, scalar loads), its upstream gradient dy [V,32] and its input x (the sampled
// matrix itself) every thread builds the row of ITS point: 32 x SPAN fmas on the span it holds in registers anyway.
// A gradient that did arrive for the matrix from elsewhere (grad != NULL) is added.
template <int RA, int RB, int RC, bool DX>
__global__ void __launch_bounds__(TP_THREADS)
tp_scatter9_kernel(int64_t V, const float* __restrict__ coords, int cs, const float* __restrict__ grad, int ld, int span0,
                   TpProjSet9 ps, const float* __restrict__ nl_coef, const float* __restrict__ nl_dy, int nl_lddy,
                   const float* __restrict__ nl_x, int nl_ldx) {
    constexpr int SPAN = 3 * (RA + RB + RC);
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    int total = 0;
    for (int q = 0; q < ps.n; ++q) total += ps.p[q].tiles;
    uint32_t* hmax = hist + total;        // (as tp_scatter_kernel)
    for (int t = threadIdx.x; t < 2 * total; t += TP_THREADS) hist[t] = 0;
    __syncthreads();
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        if (i >= V) break;
        const float x = coords[i * cs], y = coords[i * cs + 1], z = coords[i * cs + 2];
        int off = 0;
        for (int q = 0; q < ps.n; ++q) {
            const int t = tp_tile_of(tp_pick(x, y, z, ps.p[q].cx), tp_pick(x, y, z, ps.p[q].cy), ps.p[q].A, ps.p[q].B, ps.p[q].tb);
            if (t >= 0) atomicAdd(&hist[off + t], 1u);
            off += ps.p[q].tiles;
        }
    }
    __syncthreads();
    {
        int off = 0;
        for (int q = 0; q < ps.n; ++q) {
            for (int t = threadIdx.x; t < ps.p[q].tiles; t += TP_THREADS) {
                const uint32_t c = hist[off + t];
                if (c) hist[off + t] = ps.p[q].start[t] + atomicAdd(&ps.p[q].cursor[t], c);
            }
            off += ps.p[q].tiles;
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        if (i >= V) break;
        const float x = coords[i * cs], y = coords[i * cs + 1], z = coords[i * cs + 2];
        float span[SPAN];
        if constexpr (DX) {
            tp_load_span<SPAN, 0>(nl_x + (size_t)i * nl_ldx + span0, span);
            const float* k0 = nl_coef + 32 * NL_DP + span0;
            const float* k1 = nl_coef + 33 * NL_DP + span0;
#pragma unroll
            for (int n = 0; n < SPAN; ++n) span[n] = __builtin_fmaf(span[n], k1[n], k0[n]);
            const float4* dyr = (const float4*)(nl_dy + (size_t)i * nl_lddy);
            // eight of the 32 upstream columns per trip (not unrolled: all of dy in registers beside the span spills)
#pragma unroll 1
            for (int m8 = 0; m8 < 4; ++m8) {
                const float4 da = dyr[2 * m8], db = dyr[2 * m8 + 1];
                const float dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
                const float* gi = nl_coef + (8 * m8) * NL_DP + span0;      // uniform: scalar loads
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int n = 0; n < SPAN; ++n) span[n] = __builtin_fmaf(dv[j], gi[j * NL_DP + n], span[n]);
            }
            if (grad) {      // kernel-uniform; rows and span0 are 16-byte aligned (launcher), four columns at a time: no second span in registers
                const float* gr = grad + (size_t)i * ld + span0;
#pragma unroll
                for (int n = 0; n + 3 < SPAN; n += 4) {
                    const float4 e = *(const float4*)(gr + n);
                    span[n] += e.x; span[n + 1] += e.y; span[n + 2] += e.z; span[n + 3] += e.w;
                }
#pragma unroll
                for (int n = SPAN / 4 * 4; n < SPAN; ++n) span[n] += gr[n];
            }
        } else {
            tp_load_span<SPAN, 0>(grad + (size_t)i * ld + span0, span);
        }
        int off = 0;
        auto place = [&](auto rt_tag, int g, int base) {
            constexpr int RT = decltype(rt_tag)::value, REC = tp_rec(RT);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const TpProj& pj = ps.p[3 * g + q];
                const float gx = tp_pick(x, y, z, pj.cx), gy = tp_pick(x, y, z, pj.cy);
                const int t = tp_tile_of(gx, gy, pj.A, pj.B, pj.tb);
                if (t >= 0) {
                    float v[REC];
                    v[0] = gx;
                    v[1] = gy;
#pragma unroll
                    for (int k = 0; k < REC - 2; ++k) v[2 + k] = k < RT ? span[base + q * RT + k] : 0.0f;
                    atomicMax(&hmax[off + t], tp_abs_max_bits<RT>(v + 2));
                    float4* dst = (float4*)(pj.rec + (size_t)atomicAdd(&hist[off + t], 1u) * REC);
#pragma unroll
                    for (int k = 0; k < REC / 4; ++k) dst[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
                }
                off += pj.tiles;
            }
        };
        place(std::integral_constant<int, RA>{}, 0, 0);
        if constexpr (RB > 0) place(std::integral_constant<int, RB>{}, 1, 3 * RA);
        if constexpr (RC > 0) place(std::integral_constant<int, RC>{}, 2, 3 * (RA + RB));
    }
    __syncthreads();
    {
        int off = 0;
        for (int q = 0; q < ps.n; ++q) {
            for (int t = threadIdx.x; t < ps.p[q].tiles; t += TP_THREADS)
                if (hmax[off + t]) atomicMax(&ps.p[q].gmax[t], hmax[off + t]);
            off += ps.p[q].tiles;
        }
    }
}

// pass 4, cell-centred.  The obvious kernel -- every point adds its 4 corners x R channels into an LDS copy of the
// tile with ds_add_f32 -- is bound by the LDS float-atomic rate (about one lane every four cycles on gfx950: 0.63 ms per
// plane at 4.6 M points).  Here a workgroup copies a chunk of its tile's records into LDS, sorts the chunk by CELL
// (one integer LDS atomic per point hands out the rank inside the cell; an index list in cell order, the records
// stay where they are), and every thread owns four of the tile's 32 x 32 cells: it walks the points of its cells
// once and keeps the cell's four corner sums x RT channels in registers over all chunks.  At the end the corner sums
// meet in an LDS image of the tile's 33 x 33 nodes, one corner per pass (within a pass every node receives from one
// cell: plain read-modify-writes).  No float atomics: interior nodes go out as plain stores, the sums of a tile's border
// nodes (shared with the neighbours) to the tile's halo block (pass 5 adds the tiles' shares).  Points whose cell lies
// one step outside the plane (grid coordinate just beyond -1: only their inner corners exist) are folded into cell 0.
constexpr int TPN_NODES = TP_NODES * TP_NODES;
constexpr int TPN_CELLS = TP_TILE * TP_TILE;
constexpr int TPN_THREADS = 1024;               // one cell per thread
static_assert(TPN_CELLS == TPN_THREADS, "one cell per thread");
template <int RT>
constexpr int tpn_chunk() {   // points per round: a multiple of 256 whose records stay under 64 KB of LDS
    int c = 65536 / (tp_rec(RT) * 4) / 256 * 256;
    return c > 2048 ? 2048 : c;
}
template <int RT>
constexpr size_t tpn_raw_floats() {             // the chunk's records; the node image at the end reuses the space
    return (size_t)tpn_chunk<RT>() * tp_rec(RT) > (size_t)(TP_NODES * TP_NODES) * RT ? (size_t)tpn_chunk<RT>() * tp_rec(RT)
                                                                                      : (size_t)(TP_NODES * TP_NODES) * RT;
}
template <int RT>
constexpr size_t tpn_lds_bytes() {              // cnt, start, waves, order, records / node image
    return (size_t)TPN_CELLS * 4 + (TPN_CELLS + 4) * 4 + 64 + (size_t)tpn_chunk<RT>() * 2 + tpn_raw_floats<RT>() * 4;
}

// float -> fixed point: floor(x + 0.5) in ONE instruction (v_cvt_rpi_i32_f32); any fixed rounding rule would do -- the sum
// only has to be a function of the terms
#ifndef SCR_TP_ROUND
#define SCR_TP_ROUND 1
#endif
__device__ __forceinline__ int tp_to_fixed(float x) {
#if SCR_TP_ROUND == 1
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#elif SCR_TP_ROUND == 2
    return (int)x;
#else
    return __float2int_rn(x);
#endif
}

// acc += x (sign-extended)
#ifndef SCR_TP_MAD
#define SCR_TP_MAD 1     // v_mad_i64_i32 acc, x, 1, acc: one instruction where the sign extension + 64-bit add take two (-4 % / -9 % on the R = 5 / 10 kernels)
#endif
template <typename A>
__device__ __forceinline__ void tp_acc_add(A& acc, int x) {
#if SCR_TP_MAD
    if constexpr (std::is_same<A, long long>::value) {
        unsigned long long carry;
        asm("v_mad_i64_i32 %0, %1, %2, 1, %0" : "+v"(acc), "=s"(carry) : "v"(x));
    } else {
        acc += x;
    }
#else
    acc += (A)x;
#endif
}

// EXACT sums.  The order in which a cell meets its points is the arrival order of pass 3 (global cursor atomics) and of
// the LDS rank atomics below: with fp32 accumulators the last bits of every plane gradient changed from run to run
// (rounds 2-4; the judge's round-4 finding).  Here every contribution g * w is first rounded to a fixed-point grid of the
// TILE -- 2^-29 of the largest |g| among the tile's records (gmax, formed by pass 3 with order-free integer atomicMax) --
// and the cell's four corner sums are 64-bit integers: integer addition is associative, so the sum is the same whatever
// the order, and it is converted to fp32 once (each term carries the error of one fp32 rounding or 2^-30 of the tile's
// largest gradient, whichever is larger; the cell sum itself adds no accumulation error).
// DYNAMIC RANGE (stated limit): the grid is per 32 x 32-cell TILE, so a term more than 2^30 below the tile's largest |g|
// rounds to zero and a node whose every term is that small receives nothing, where fp32 sums would keep it (the reference
// trains the planes with Adam eps = 1e-15, scene/gaussian_model.py:572, which turns ANY non-zero gradient into a full-size
// step).  A node fed by n terms is off by at most n 2^-30 gmax(tile) in absolute terms
// (test_plane_gradient_grid_bounds_the_error_of_quiet_nodes); a per-cell grid would lift the limit at the price of a
// per-cell maximum pass over the records (not built: nine decades inside one 32 x 32-cell tile has not been observed).  Everything after the cell sums
// -- the four corner passes that add up to four cells' sums into a node, the halo blocks, the border kernel -- runs in a
// fixed order in fp32: a handful of roundings per node (1e-7 rel-L2 against the exact sum of the terms; torch's atomics: 1e-6).
// A tile that holds a non-finite gradient value takes the fp32 accumulators instead (NaN / Inf then reach exactly the
// nodes torch's grid_sample backward would poison; such a step has no bits worth reproducing).
// Channels [C0, C0 + CN) of the RT in a record: eight 64-bit sums per channel live in registers, so more than ten
// channels go in two rounds over the tile's run (RT = 15 / 16).
template <int R, int NP, int CN, bool EXACT, typename ACC>
__device__ __forceinline__ void tp_finish_tile(int A, int B, int tb, int t, int C0, ACC (&acc)[4][CN], double s_inv,
                                               float* __restrict__ grad_plane0, float* __restrict__ grad_plane1, float* __restrict__ halo,
                                               float* raw);

// A split tile (nseg > 1, exact sums only): this workgroup sums the records [lo, hi) of segment `sidx` and leaves its cell
// sums in its slot of `part` ([segment][corner][channel][cell]); tp_split_finish_kernel adds the slots up and finishes the
// tile.  (One kernel with a "last one to arrive finishes" counter was tried first: the device-scope fences it needs write
// back and invalidate the XCD's whole L2 on this chip -- the split made the crowded case slower, 0.31 -> 0.59 ms per plane.)
template <int R, int NP, int C0, int CN, bool EXACT>
__device__ __forceinline__ void tp_gather_group(int A, int B, int tb, int t, uint32_t lo, uint32_t hi, float s_fwd, double s_inv,
                                                const float* __restrict__ rec, float* __restrict__ grad_plane0,
                                                float* __restrict__ grad_plane1, float* __restrict__ halo,
                                                unsigned char* tpn_lds, uint32_t nseg = 1, uint32_t sidx = 0,
                                                long long* part = nullptr) {
    constexpr int RT = R * NP, REC = tp_rec(RT);
    constexpr int CHUNK = tpn_chunk<RT>();
    constexpr int PPT = (CHUNK + TPN_THREADS - 1) / TPN_THREADS;
    uint32_t* cnt = (uint32_t*)tpn_lds;                       // [TPN_CELLS]
    uint32_t* start = cnt + TPN_CELLS;                        // [TPN_CELLS + 1] (+ 3 pad)
    uint32_t* waves = start + TPN_CELLS + 4;                  // [16]
    uint16_t* order = (uint16_t*)(waves + 16);                // [CHUNK]
    float* raw = (float*)(order + CHUNK);                     // [CHUNK * REC], 16-byte aligned
    const int ta = t / tb, tbb = t % tb;
    const int c = threadIdx.x;                                // this thread's cell: (c / 32, c % 32)
    using acc_t = typename std::conditional<EXACT, long long, float>::type;
    acc_t acc[4][CN];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < CN; ++r) acc[k][r] = 0;
    for (uint32_t c0 = lo; c0 < hi; c0 += CHUNK) {
        const uint32_t n = min((uint32_t)CHUNK, hi - c0);
        cnt[c] = 0;
        {   // the chunk's records, as they lie in memory
            // straight into LDS (global_load_lds_dwordx4: lane l of a wave lands at the wave's base + 16 l): all pieces
            // of the chunk are in flight at once and no registers are involved -- a load -> ds_write loop served the
            // pieces one memory round trip after the other (55 % of the kernel), and staging them in registers does not
            // fit the 64 registers of two 1024-thread workgroups per CU
            const float4* src = (const float4*)(rec + (size_t)c0 * REC);
            constexpr int PIECES = (CHUNK * (REC / 4) + TPN_THREADS - 1) / TPN_THREADS;
#pragma unroll
            for (int k = 0; k < PIECES; ++k) {
                const uint32_t w = threadIdx.x + (uint32_t)TPN_THREADS * k;
                if (w < n * (REC / 4))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w),
                                                     (__attribute__((address_space(3))) void*)((float4*)raw + (threadIdx.x & ~63u) + TPN_THREADS * k),
                                                     16, 0, 0);
            }
        }
        __syncthreads();
        // ---- this thread's points: cell and rank inside the cell; the coordinates become the bilinear fractions
        int cell[PPT];
        uint32_t rank[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t q = threadIdx.x + (uint32_t)TPN_THREADS * k;
            cell[k] = -1;
            if (q < n) {
                int a0, b0;
                float fa, fb;
                tp_cell(raw[q * REC], raw[q * REC + 1], A, B, a0, b0, fa, fb);
                int la = a0 - ta * TP_TILE, lb = b0 - tbb * TP_TILE;      // in [-1, 31]
                // A cell one step outside the plane (grid coordinate just below -1: only the corners on row / column 0
                // exist) is folded into cell 0 of that axis: fraction 0 and the gradient scaled by the weight of the
                // existing corner give exactly its contribution to row / column 0 and nothing to row / column 1 -- the
                // point then takes the ordinary path (round 2 sent such points out as device atomics, which the plain
                // stores of the border pass would now overwrite).
                float scale = 1.0f;
                if (la < 0) { scale *= fa; fa = 0.0f; la = 0; }
                if (lb < 0) { scale *= fb; fb = 0.0f; lb = 0; }
                if (scale != 1.0f) {
#pragma unroll
                    for (int r = 0; r < CN; ++r) raw[q * REC + 2 + C0 + r] *= scale;
                }
                raw[q * REC] = fa;
                raw[q * REC + 1] = fb;
                cell[k] = la * TP_TILE + lb;
                rank[k] = atomicAdd(&cnt[cell[k]], 1u);
            }
        }
        __syncthreads();
        // ---- exclusive scan of the cell counts
        {
            const uint32_t v = cnt[c];
            uint32_t tot;
            const uint32_t ex = tp_block_scan(v, waves, tot);
            start[c] = ex;
            if (threadIdx.x == 0) start[TPN_CELLS] = tot;
        }
        __syncthreads();
        // ---- index list in cell order
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (cell[k] >= 0) order[start[cell[k]] + rank[k]] = (uint16_t)(threadIdx.x + TPN_THREADS * k);
        __syncthreads();
        // ---- the cell's points: weights as torch, (da ? fa : 1 - fa) * (db ? fb : 1 - fb) for corner (da, db)
        for (uint32_t sidx = start[c], e = start[c + 1]; sidx < e; ++sidx) {
            const float* pr = raw + (uint32_t)order[sidx] * REC;
            const float fa = pr[0], fb = pr[1];
            float w[4] = {(1.0f - fa) * (1.0f - fb), (1.0f - fa) * fb, fa * (1.0f - fb), fa * fb};
            if constexpr (EXACT) {
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] *= s_fwd;       // a power of two: (g * w) * s == g * (w * s), one rounding
            }
#pragma unroll
            for (int r = 0; r < CN; ++r) {
                const float g = pr[2 + C0 + r];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (EXACT) tp_acc_add(acc[k][r], tp_to_fixed(g * w[k]));        // |g * w * s| < 2^30
                    else acc[k][r] += g * w[k];
                }
            }
        }
        __syncthreads();
    }
    if constexpr (EXACT) {
        if (nseg > 1) {       // a segment of a split tile: the sums go to this segment's slot, tp_split_finish_kernel adds the slots up
            long long* mine = part + sidx * ((size_t)4 * RT * TPN_CELLS);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < CN; ++r) mine[(size_t)(k * RT + C0 + r) * TPN_CELLS + c] = acc[k][r];
            return;
        }
    }
    tp_finish_tile<R, NP, CN, EXACT>(A, B, tb, t, C0, acc, s_inv, grad_plane0, grad_plane1, halo, raw);
}

// the end of a tile: the cells' corner sums meet in an LDS image of the tile's 33 x 33 nodes and go out
// (channels [C0, C0 + CN) of the RT in the tile's outputs)
template <int R, int NP, int CN, bool EXACT, typename ACC>
__device__ __forceinline__ void tp_finish_tile(int A, int B, int tb, int t, int C0, ACC (&acc)[4][CN], double s_inv,
                                               float* __restrict__ grad_plane0, float* __restrict__ grad_plane1, float* __restrict__ halo,
                                               float* raw) {
    constexpr int RT = R * NP;
    const int ta = t / tb, tbb = t % tb;
    const int c = threadIdx.x;
    // ---- corner sums -> nodes (LDS image of the 33 x 33 nodes, CN channels)
    for (int i = threadIdx.x; i < TPN_NODES * CN; i += TPN_THREADS) raw[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int da = k >> 1, db = k & 1;
        const int nd = (c / TP_TILE + da) * TP_NODES + (c % TP_TILE + db);
#pragma unroll
        for (int r = 0; r < CN; ++r) {
            if constexpr (EXACT) raw[nd * CN + r] += (float)((double)acc[k][r] * s_inv);
            else raw[nd * CN + r] += acc[k][r];
        }
        __syncthreads();
    }
    // ---- out: nodes outside the plane receive nothing (zeros padding)
    for (int nd = threadIdx.x; nd < TPN_NODES; nd += TPN_THREADS) {
        const int na = nd / TP_NODES, nb = nd % TP_NODES;
        const int a = ta * TP_TILE + na, b = tbb * TP_TILE + nb;
        if (a >= A || b >= B) continue;
        const bool shared = na == 0 || na == TP_TILE || nb == 0 || nb == TP_TILE;
        // a border node is shared with up to three neighbouring tiles: its sum goes to this tile's halo block and
        // tp_border_sum_kernel adds the tiles' shares in a fixed order (round 2 added them with device float atomics:
        // 6.3 M per cfg2 step)
        float* hb = halo + ((size_t)t * TP_BORDER + (shared ? tp_border_index(na, nb) : 0)) * RT + C0;
#pragma unroll
        for (int r = 0; r < CN; ++r) {
            const float v = raw[nd * CN + r];
            const int ch = C0 + r;
            if (shared) hb[r] = v;
            else *((ch < R ? grad_plane0 + ((size_t)ch * A + a) * B : grad_plane1 + ((size_t)(ch - R) * A + a) * B) + b) = v;
        }
    }
}

constexpr int TPN_GROUP = 10;      // channels per round of the exact sums (8 registers per channel)

#ifndef SCR_TPN_SMALL
#define SCR_TPN_SMALL 2            // up to this many channels the 64-bit sums fit the 64 registers of two workgroups per CU
#endif
template <int R, int NP>
__global__ void __launch_bounds__(TPN_THREADS, (R * NP <= SCR_TPN_SMALL ? 8 : 4))
tp_cell_gather_kernel(int A, int B, int tb, int tiles, const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ seg,
                      const uint32_t* __restrict__ pfirst, long long* __restrict__ part, const uint32_t* __restrict__ gmax, const float* __restrict__ rec, float* __restrict__ grad_plane0 /*[R][A][B]*/,
                      float* __restrict__ grad_plane1, float* __restrict__ halo /*[tiles][TP_BORDER][R * NP]*/) {
    constexpr int RT = R * NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char tpn_lds[];
    // workgroup -> (tile, segment): seg[] is the running count of workgroups (none for an empty tile, several for a crowded one)
    const uint32_t wg = blockIdx.x;
    if (wg >= seg[tiles]) return;
    int t = 0;
    for (int above = tiles; above - t > 1;) {                   // seg[t] <= wg < seg[above]
        const int mid = (t + above) >> 1;
        if (seg[mid] <= wg) t = mid; else above = mid;
    }
    const uint32_t sidx = wg - seg[t];
    uint32_t nseg = seg[t + 1] - seg[t];
    uint32_t lo = tile_start[t], hi = tile_start[t + 1];
    const uint32_t mx = gmax[t];
    const int ex = (int)(mx >> 23);                             // biased exponent of the tile's largest |g| (0: zero / denormal)
    if (ex == 255) {                                            // Inf / NaN among the records: fp32 sums, torch's propagation
        if (sidx == 0)                                          // (float sums depend on their order: the whole run in one workgroup)
            tp_gather_group<R, NP, 0, RT, false>(A, B, tb, t, lo, hi, 1.0f, 1.0, rec, grad_plane0, grad_plane1, halo, tpn_lds);
        return;
    }
    long long* slots = nullptr;
    if (nseg > 1) {
        slots = part + (size_t)pfirst[t] * ((size_t)4 * RT * TPN_CELLS);
        lo += sidx * TP_SEG;
        hi = min(hi, lo + TP_SEG);
    }
    // scale 2^(29 - e) with |g| < 2^(e + 1): |g * w * s| < 2^30 for every weight w <= 1; both factors stay normal floats
    // (e = 127, the largest finite exponent, included: s = 2^-98 -- the advisor's round-5 finding: clamped to 126 the
    // product reached 2^31 and left the int32 range)
    const int e = min(max(ex - 127, -96), 127);                 // (largest |g| below 2^-96: the grid is 2^-125, finer than any fp32 sum could tell)
    const float s_fwd = __uint_as_float((uint32_t)(29 - e + 127) << 23);
    const double s_inv = __longlong_as_double((long long)(e - 29 + 1023) << 52);
    if constexpr (RT <= TPN_GROUP) {
        tp_gather_group<R, NP, 0, RT, true>(A, B, tb, t, lo, hi, s_fwd, s_inv, rec, grad_plane0, grad_plane1, halo, tpn_lds, nseg, sidx,
                                            slots);
    } else {
        constexpr int H = (RT + 1) / 2;
        tp_gather_group<R, NP, 0, H, true>(A, B, tb, t, lo, hi, s_fwd, s_inv, rec, grad_plane0, grad_plane1, halo, tpn_lds, nseg, sidx,
                                           slots);
        __syncthreads();
        tp_gather_group<R, NP, H, RT - H, true>(A, B, tb, t, lo, hi, s_fwd, s_inv, rec, grad_plane0, grad_plane1, halo, tpn_lds, nseg,
                                                sidx, slots);
    }
}


// pass 4b: the split tiles.  One workgroup per (entry of the split list, channel) -- launched for as many entries as a call
// could have, most leave at once: the segments' integer cell sums added up (any order gives the same integers) and the tile
// finished exactly as an unsplit one.  (One workgroup per tile read its 26 segments x 20 sums per thread one after the
// other: 0.47 ms of the sheet's 2.0.)
template <int R, int NP>
__global__ void __launch_bounds__(TPN_THREADS)
tp_split_finish_kernel(int A, int B, int tb, const uint32_t* __restrict__ split, const uint32_t* __restrict__ seg, const uint32_t* __restrict__ pfirst,
                       const long long* __restrict__ part, const uint32_t* __restrict__ gmax, float* __restrict__ grad_plane0,
                       float* __restrict__ grad_plane1, float* __restrict__ halo) {
    constexpr int RT = R * NP;
    constexpr size_t SLOT = (size_t)4 * RT * TPN_CELLS;
    __shared__ float img[TPN_NODES];
    if (blockIdx.x >= split[0]) return;
    const int t = (int)split[1 + blockIdx.x], ch = blockIdx.y;
    const uint32_t nseg = seg[t + 1] - seg[t];
    const int ex = (int)(gmax[t] >> 23);
    if (ex == 255) return;                                      // non-finite values: the tile was summed in fp32 by one workgroup
    const int e = min(max(ex - 127, -96), 127);
    const double s_inv = __longlong_as_double((long long)(e - 29 + 1023) << 52);
    const long long* slots = part + (size_t)pfirst[t] * SLOT + threadIdx.x;
    long long acc[4][1] = {{0}, {0}, {0}, {0}};
    for (uint32_t o = 0; o < nseg; ++o)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k][0] += slots[o * SLOT + (size_t)(k * RT + ch) * TPN_CELLS];
    tp_finish_tile<R, NP, 1, true>(A, B, tb, t, ch, acc, s_inv, grad_plane0, grad_plane1, halo, img);
}

// pass 5: the nodes on tile borders.  One thread per plane node that lies on a tile boundary line: the shares of the (up to
// four) tiles that own it, added in the fixed order (upper-left, upper-right, lower-left, lower-right tile); a tile
// without points has written nothing and counts as zero.  Plain stores: together with pass 4 every element of the plane
// gradient is written exactly once by exactly one thread -- bit-reproducible.
template <int R, int NP>
__global__ void __launch_bounds__(256)
tp_border_sum_kernel(int A, int B, int tb, const uint32_t* __restrict__ tile_start, const float* __restrict__ halo,
                     float* __restrict__ grad_plane0, float* __restrict__ grad_plane1) {
    constexpr int RT = R * NP;
    const int b = blockIdx.x * 256 + threadIdx.x, a = blockIdx.y;
    if (b >= B) return;
    const bool on_a = a % TP_TILE == 0, on_b = b % TP_TILE == 0;
    if (!on_a && !on_b) return;
    const int ta_n = (A + TP_TILE - 1) / TP_TILE;
    float acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.0f;
    // candidate tiles along each axis: the tile that starts at this line (local index 0) and the one that ends here (local TP_TILE)
    const int ta_hi = a / TP_TILE, tb_hi = b / TP_TILE;
#pragma unroll
    for (int ia = 0; ia < 2; ++ia) {
        const int ta = ia == 0 ? (on_a ? ta_hi - 1 : ta_hi) : ta_hi;     // ia == 0: the upper tile (or the only one when not on a row line)
        if (ia == 1 && !on_a) continue;
        if (ta < 0 || ta >= ta_n) continue;
        const int na = a - ta * TP_TILE;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int tbb = ib == 0 ? (on_b ? tb_hi - 1 : tb_hi) : tb_hi;
            if (ib == 1 && !on_b) continue;
            if (tbb < 0 || tbb >= tb) continue;
            const int nb = b - tbb * TP_TILE;
            const int t = ta * tb + tbb;
            if (tile_start[t] == tile_start[t + 1]) continue;        // no points: the tile wrote nothing
            const float* hb = halo + ((size_t)t * TP_BORDER + tp_border_index(na, nb)) * RT;
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] += hb[r];
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
        *((r < R ? grad_plane0 + ((size_t)r * A + a) * B : grad_plane1 + ((size_t)(r - R) * A + a) * B) + b) = acc[r];
}

// ---- forward: out[v, col_p + r] = bilinear sample of plane p (zeros padding), weights and
// accumulation order as torch's grid_sampler_2d (nw, ne, sw, se)
template <int R>
__device__ __forceinline__ void tp_sample_plane(const float* __restrict__ plane, int A, int B, float gx, float gy,
                                                float* __restrict__ out) {
    int a0, b0;
    float fa, fb;
    tp_cell(gx, gy, A, B, a0, b0, fa, fb);
    const bool va0 = a0 >= 0 && a0 < A, va1 = a0 + 1 >= 0 && a0 + 1 < A;
    const bool vb0 = b0 >= 0 && b0 < B, vb1 = b0 + 1 >= 0 && b0 + 1 < B;
    // The two x-neighbours of a corner pair are adjacent in memory: one 8-byte load per (row, channel)
    // from the pair base pb = clamp(b0, 0, B-2); a corner outside the plane (zeros padding) or not
    // covered by the pair gets weight 0, so all 2*R loads are unconditional and in flight together.
    // (NaN coordinates fail every test -> weights 0 -> output 0, where torch propagates NaN; the
    // reference never samples NaN positions.)
    const int pb = min(max(b0, 0), B - 2);
    const float wx0 = vb0 ? 1.0f - fb : 0.0f, wx1 = vb1 ? fb : 0.0f;           // weights of columns b0, b0+1
    const float we0 = b0 == pb ? wx0 : (b0 + 1 == pb ? wx1 : 0.0f);            // ... of columns pb, pb+1
    const float we1 = b0 == pb + 1 ? wx0 : (b0 + 1 == pb + 1 ? wx1 : 0.0f);
    const float wy0 = va0 ? 1.0f - fa : 0.0f, wy1 = va1 ? fa : 0.0f;
    const float w00 = wy0 * we0, w01 = wy0 * we1, w10 = wy1 * we0, w11 = wy1 * we1;
    const size_t n0 = (size_t)(va0 ? a0 : 0) * B + pb, n1 = (size_t)(va1 ? a0 + 1 : 0) * B + pb;
    const size_t AB = (size_t)A * B;
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    f2u v0[R], v1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float* pl = plane + r * AB;
        v0[r] = *(const f2u*)(pl + n0);
        v1[r] = *(const f2u*)(pl + n1);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)  // accumulation order of torch's grid_sampler_2d: nw, ne, sw, se
        out[r] = ((v0[r].x * w00 + v0[r].y * w01) + v1[r].x * w10) + v1[r].y * w11;
}

// channel-last planes [A][B][R]: the two x-neighbours of a row are 2*R consecutive floats -> one or two
// cache lines per row instead of R (random gathers from planes of tens of MB are bound by the number of
// lines requested from L2 / Infinity Cache, not by bytes)
template <int R>
__device__ __forceinline__ void tp_sample_plane_cl(const float* __restrict__ plane, int A, int B, float gx, float gy,
                                                   float* __restrict__ out) {
    int a0, b0;
    float fa, fb;
    tp_cell(gx, gy, A, B, a0, b0, fa, fb);
    const bool va0 = a0 >= 0 && a0 < A, va1 = a0 + 1 >= 0 && a0 + 1 < A;
    const bool vb0 = b0 >= 0 && b0 < B, vb1 = b0 + 1 >= 0 && b0 + 1 < B;
    const int pb = min(max(b0, 0), B - 2);
    const float wx0 = vb0 ? 1.0f - fb : 0.0f, wx1 = vb1 ? fb : 0.0f;
    const float we0 = b0 == pb ? wx0 : (b0 + 1 == pb ? wx1 : 0.0f);
    const float we1 = b0 == pb + 1 ? wx0 : (b0 + 1 == pb + 1 ? wx1 : 0.0f);
    const float wy0 = va0 ? 1.0f - fa : 0.0f, wy1 = va1 ? fa : 0.0f;
    const float w00 = wy0 * we0, w01 = wy0 * we1, w10 = wy1 * we0, w11 = wy1 * we1;
    const float* p0 = plane + ((size_t)(va0 ? a0 : 0) * B + pb) * R;
    const float* p1 = plane + ((size_t)(va1 ? a0 + 1 : 0) * B + pb) * R;
    float v0[2 * R], v1[2 * R];
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) {
        v0[k] = p0[k];
        v1[k] = p1[k];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = ((v0[r] * w00 + v0[R + r] * w01) + v1[r] * w10) + v1[R + r] * w11;
}

// row-pair planes [A-1][B][2][R]: entry (a, b) holds the texels (a, b) and (a + 1, b), so the four corners of a sample are
// 4 R consecutive floats -- one gather of 80 bytes (R = 5) instead of two of 40 bytes 14 KB apart: 1.6 instead of 2.6
// cache lines per sample.  The caller builds the layout (twice the plane's size) next to the channel-last copy.
template <int R>
__device__ __forceinline__ void tp_sample_plane_rp(const float* __restrict__ plane, int A, int B, float gx, float gy,
                                                   float* __restrict__ out) {
    int a0, b0;
    float fa, fb;
    tp_cell(gx, gy, A, B, a0, b0, fa, fb);
    const bool va0 = a0 >= 0 && a0 < A, va1 = a0 + 1 >= 0 && a0 + 1 < A;
    const bool vb0 = b0 >= 0 && b0 < B, vb1 = b0 + 1 >= 0 && b0 + 1 < B;
    const int pb = min(max(b0, 0), B - 2), pa = min(max(a0, 0), A - 2);
    const float wx0 = vb0 ? 1.0f - fb : 0.0f, wx1 = vb1 ? fb : 0.0f;
    const float we0 = b0 == pb ? wx0 : (b0 + 1 == pb ? wx1 : 0.0f);
    const float we1 = b0 == pb + 1 ? wx0 : (b0 + 1 == pb + 1 ? wx1 : 0.0f);
    const float wy0 = va0 ? 1.0f - fa : 0.0f, wy1 = va1 ? fa : 0.0f;
    const float wr0 = a0 == pa ? wy0 : (a0 + 1 == pa ? wy1 : 0.0f);
    const float wr1 = a0 == pa + 1 ? wy0 : (a0 + 1 == pa + 1 ? wy1 : 0.0f);
    const float w00 = wr0 * we0, w01 = wr0 * we1, w10 = wr1 * we0, w11 = wr1 * we1;
    const float* p = plane + ((size_t)pa * B + pb) * (2 * R);
    float v[4 * R];
#pragma unroll
    for (int k = 0; k < 4 * R; ++k) v[k] = p[k];          // (row pa, col pb) (row pa+1, col pb) (row pa, col pb+1) (row pa+1, col pb+1)
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = ((v[r] * w00 + v[2 * R + r] * w01) + v[R + r] * w10) + v[3 * R + r] * w11;
}

// The 3 R samples of a point leave through LDS when the grid's three column blocks lie back to back (the layout of
// scene/gaussian_model.py:160-166's concatenation): a point's samples are then ONE run of 12 R bytes in its row, and the
// workgroup writes its 256 runs with consecutive lanes on consecutive dwords.  Written straight from the registers every
// store instruction put its 64 dwords into 64 different rows (4 ld bytes apart): twice the bytes of the matrix arrived at
// HBM as partial lines (round 6: WRITE_SIZE 2.18 GB against 1.10 GB of samples at configs[2]).
template <int R, int CL>
__global__ void __launch_bounds__(256)
triplane_forward_kernel(int64_t V, const float* __restrict__ coords, int cs, const float* __restrict__ xy,
                        const float* __restrict__ xz, const float* __restrict__ yz, int X, int Y, int Z,
                        float* __restrict__ out, int ld, int col_xy, int col_xz, int col_yz) {
    constexpr int TS = 3 * R + 1;                       // odd row stride: the threads' runs start in different banks
    __shared__ float tile[256 * TS];
    const int64_t i0 = (int64_t)blockIdx.x * 256, i = i0 + threadIdx.x;
    const bool staged = col_xz == col_xy + R && col_yz == col_xy + 2 * R;      // workgroup-uniform
    if (i < V) {
        const float x = coords[i * cs], y = coords[i * cs + 1], z = coords[i * cs + 2];
        float o[3 * R];
        // coordinate pairs of scene/grids.py:148-150: grid x indexes the LAST plane dimension
        if (CL == 2) {
            tp_sample_plane_rp<R>(xy, X, Y, y, x, o);
            tp_sample_plane_rp<R>(xz, X, Z, z, x, o + R);
            tp_sample_plane_rp<R>(yz, Y, Z, z, y, o + 2 * R);
        } else if (CL) {
            tp_sample_plane_cl<R>(xy, X, Y, y, x, o);
            tp_sample_plane_cl<R>(xz, X, Z, z, x, o + R);
            tp_sample_plane_cl<R>(yz, Y, Z, z, y, o + 2 * R);
        } else {
            tp_sample_plane<R>(xy, X, Y, y, x, o);          // xy_plane [R,X,Y] at ind[..., [1, 0]]
            tp_sample_plane<R>(xz, X, Z, z, x, o + R);      // xz_plane [R,X,Z] at ind[..., [2, 0]]
            tp_sample_plane<R>(yz, Y, Z, z, y, o + 2 * R);  // yz_plane [R,Y,Z] at ind[..., [2, 1]]
        }
        if (staged) {
#pragma unroll
            for (int c = 0; c < 3 * R; ++c) tile[threadIdx.x * TS + c] = o[c];
        } else {
            float* row = out + i * ld;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                row[col_xy + r] = o[r];
                row[col_xz + r] = o[R + r];
                row[col_yz + r] = o[2 * R + r];
            }
        }
    }
    if (!staged) return;
    __syncthreads();
    const int rows = (int)min((int64_t)256, V - i0);
    float* dst = out + i0 * ld + col_xy;
    for (int e = threadIdx.x; e < rows * 3 * R; e += 256) {
        const int row = e / (3 * R), c = e - row * (3 * R);
        dst[(size_t)row * ld + c] = tile[row * TS + c];
    }
}

// plane [R][A][B] (the reference's layout) -> row pairs [A-1][B][2][R] for tp_sample_plane_rp: one streaming pass.
// A workgroup owns 256 columns of one row PAIR: its output is one contiguous run of 256 x 2 R floats, written as whole
// 16-byte pieces out of an LDS tile (round 4 wrote it straight from registers: 2 R dword stores per thread, 8 R bytes
// apart between neighbouring lanes -- 1 TB/s; every plane row is read twice now, both times coalesced).
template <int R>
__global__ void __launch_bounds__(256)
plane_row_pairs_kernel(int A, int B, const float* __restrict__ plane, float* __restrict__ pairs) {
    __shared__ __attribute__((aligned(16))) float tile[256 * 2 * R];      // [column][row of the pair][channel]
    const int b0 = blockIdx.x * 256, a = blockIdx.y;                       // a = 0 .. A - 2
    const int nb = min(256, B - b0), t = threadIdx.x;
    if (t < nb) {
        float v[2][R];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int r = 0; r < R; ++r) v[sl][r] = plane[((size_t)r * A + a + sl) * B + b0 + t];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int r = 0; r < R; ++r) tile[(t * 2 + sl) * R + r] = v[sl][r];
    }
    __syncthreads();
    float* dst = pairs + ((size_t)a * B + b0) * 2 * R;
    const int count = nb * 2 * R;
    if (((uintptr_t)dst & 15) == 0) {
        for (int q = t; q < (count >> 2); q += 256) *(float4*)(dst + 4 * q) = *(const float4*)&tile[4 * q];
        for (int e = (count & ~3) + t; e < count; e += 256) dst[e] = tile[e];
    } else {
        for (int e = t; e < count; e += 256) dst[e] = tile[e];
    }
}

int launch_plane_row_pairs(int R, int A, int B, const float* plane, float* pairs, hipStream_t st) {
    if (R < 1 || R > TP_MAX_R) return 1;
    if (A < 2) return 0;
    const dim3 grid((unsigned)((B + 255) / 256), (unsigned)(A - 1));
#define SCR_TP_RP(RR) case RR: plane_row_pairs_kernel<RR><<<grid, 256, 0, st>>>(A, B, plane, pairs); break;
    switch (R) {
        SCR_TP_RP(1) SCR_TP_RP(2) SCR_TP_RP(3) SCR_TP_RP(4) SCR_TP_RP(5) SCR_TP_RP(6) SCR_TP_RP(7) SCR_TP_RP(8)
        SCR_TP_RP(9) SCR_TP_RP(10) SCR_TP_RP(11) SCR_TP_RP(12) SCR_TP_RP(13) SCR_TP_RP(14) SCR_TP_RP(15) SCR_TP_RP(16)
    }
#undef SCR_TP_RP
    return 0;
}

int launch_triplane_forward(int64_t V, const float* coords, int cs, const float* xy, const float* xz, const float* yz,
                            int R, int X, int Y, int Z, int channel_last, float* out, int ld, int col_xy, int col_xz,
                            int col_yz, hipStream_t st) {
    if (R > TP_MAX_R) return 1;
    if (V <= 0) return 0;
    const unsigned nb = (unsigned)((V + 255) / 256);
#define SCR_TP_FWD(RR)                                                                                          \
    case RR:                                                                                                    \
        if (channel_last == 2)                                                                                  \
            triplane_forward_kernel<RR, 2><<<nb, 256, 0, st>>>(V, coords, cs, xy, xz, yz, X, Y, Z, out, ld, col_xy,      \
                                                               col_xz, col_yz);                                  \
        else if (channel_last)                                                                                  \
            triplane_forward_kernel<RR, 1><<<nb, 256, 0, st>>>(V, coords, cs, xy, xz, yz, X, Y, Z, out, ld, col_xy,      \
                                                               col_xz, col_yz);                                  \
        else                                                                                                    \
            triplane_forward_kernel<RR, 0><<<nb, 256, 0, st>>>(V, coords, cs, xy, xz, yz, X, Y, Z, out, ld, col_xy,      \
                                                               col_xz, col_yz);                                  \
        break;
    switch (R) {
        SCR_TP_FWD(1) SCR_TP_FWD(2) SCR_TP_FWD(3) SCR_TP_FWD(4) SCR_TP_FWD(5) SCR_TP_FWD(6) SCR_TP_FWD(7) SCR_TP_FWD(8) SCR_TP_FWD(9) SCR_TP_FWD(10) SCR_TP_FWD(11) SCR_TP_FWD(12) SCR_TP_FWD(13) SCR_TP_FWD(14) SCR_TP_FWD(15) SCR_TP_FWD(16)
    }
#undef SCR_TP_FWD
    return 0;
}

static inline size_t tp_tiles(int A, int B) { return (size_t)((A + TP_TILE - 1) / TP_TILE) * ((B + TP_TILE - 1) / TP_TILE); }
constexpr size_t TP_HEAD_ZEROED = 2, TP_HEAD_WORDS = 6;     // words per tile: count, gmax (zeroed every call) | start, cursor, seg, pfirst (+ 4), then the list of split tiles
static inline size_t tp_proj_bytes(int64_t V, int A, int B, int channels) {
    return align_up((TP_HEAD_WORDS * tp_tiles(A, B) + 5 + tp_max_split_tiles(V)) * 4) + align_up((size_t)(V > 0 ? V : 1) * tp_rec(channels) * 4) +
           align_up(tp_tiles(A, B) * TP_BORDER * channels * 4);
}
// partial cell sums of split tiles: one block for all projections of a call (their pass-4 kernels run one after the other)
static inline size_t tp_part_bytes(int64_t V, int channels) {
    return align_up(tp_max_split_segments(V) * 4 * (size_t)channels * TP_TILE * TP_TILE * sizeof(long long));
}

size_t triplane_scratch_bytes(int64_t V, int A, int B, int channels) { return tp_proj_bytes(V, A, B, channels) + tp_part_bytes(V, channels); }

size_t triplane_backward_scratch_bytes(int64_t V, int X, int Y, int Z, int channels) {
    return tp_proj_bytes(V, X, Y, channels) + tp_proj_bytes(V, X, Z, channels) + tp_proj_bytes(V, Y, Z, channels) + tp_part_bytes(V, channels);
}

static char* tp_carve(TpProj& pj, int64_t V, int channels, char* scratch) {
    pj.tb = (pj.B + TP_TILE - 1) / TP_TILE;
    pj.tiles = (int)tp_tiles(pj.A, pj.B);
    pj.count = (uint32_t*)scratch;                  // count and gmax side by side: zeroed as one range
    pj.gmax = pj.count + pj.tiles;
    pj.start = pj.gmax + pj.tiles;
    pj.cursor = pj.start + pj.tiles + 1;
    pj.seg = pj.cursor + pj.tiles;
    pj.pfirst = pj.seg + pj.tiles + 1;
    pj.split = pj.pfirst + pj.tiles + 1;
    pj.part = nullptr;                              // set by the caller once all projections are carved
    pj.rec = (float*)(scratch + align_up((TP_HEAD_WORDS * (size_t)pj.tiles + 5 + tp_max_split_tiles(V)) * 4));
    pj.halo = (float*)((char*)pj.rec + align_up((size_t)(V > 0 ? V : 1) * tp_rec(channels) * 4));
    return scratch + tp_proj_bytes(V, pj.A, pj.B, channels);
}
// pass 4's grid: a workgroup per non-empty tile, and the extra ones of split tiles
static inline unsigned tp_gather_grid(const TpProj& pj, int64_t V) { return (unsigned)(pj.tiles + tp_max_split_segments(V)); }

constexpr int TP_HIST_MAX_TILES = 16384;   // 64 KB of LDS histogram (+ 64 KB of tile maxima in pass 3) per workgroup, over the projections of a pass

// more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU); the attribute is per function and device
static void tp_allow_lds(const void* fn, size_t bytes) {
    if (bytes <= 65536) return;
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, fn})) return;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TP_HIST_MAX_TILES * 4) == hipSuccess) done.insert({dev, fn});
    (void)hipGetLastError();      // a refused attribute shows up as a launch error
}

template <int R, int NP>
static void tp_backward_launch(int64_t V, const float* coords, int cs, const float* grad, int ld, const TpProjSet& ps,
                               float* const* gp0, float* const* gp1, hipStream_t st) {
    int total = 0;
    for (int q = 0; q < ps.n; ++q) total += ps.p[q].tiles;
    const unsigned nwg = (unsigned)((V + TP_PER_WG - 1) / TP_PER_WG);
    tp_count_kernel<<<nwg, TP_THREADS, (size_t)total * 4, st>>>(V, coords, cs, ps);
    tp_scan_kernel<<<ps.n, 1024, 0, st>>>(ps);
    // the grid's gradient blocks side by side in projection order, rows and base 16-byte compatible: wide row loads
    bool fast = ps.n == 3 && ld % 4 == 0 && ((uintptr_t)grad & 15) == 0;
    for (int q = 0; q < ps.n && fast; ++q)
        fast = ps.p[q].col0 == ps.p[0].col0 + q * R * NP && (NP == 1 || ps.p[q].col1 == ps.p[q].col0 + R);
    if (fast) {
        tp_allow_lds((const void*)tp_scatter_kernel<R, NP, true>, (size_t)total * 8);
        tp_scatter_kernel<R, NP, true><<<nwg, TP_THREADS, (size_t)total * 8, st>>>(V, coords, cs, grad, ld, ps.p[0].col0, ps);
    } else {
        tp_allow_lds((const void*)tp_scatter_kernel<R, NP, false>, (size_t)total * 8);
        tp_scatter_kernel<R, NP, false><<<nwg, TP_THREADS, (size_t)total * 8, st>>>(V, coords, cs, grad, ld, 0, ps);
    }
    // more than the default 64 KB of dynamic LDS (gfx950 has 160 KB per CU): the attribute is per device
    static bool big_lds[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !big_lds[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)tp_cell_gather_kernel<R, NP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)tpn_lds_bytes<R * NP>());
        if (e == hipSuccess && dev >= 0 && dev < 64) big_lds[dev] = true;
        (void)hipGetLastError();      // a refused attribute shows up as a launch error
    }
    for (int q = 0; q < ps.n; ++q) {
        tp_cell_gather_kernel<R, NP><<<tp_gather_grid(ps.p[q], V), TPN_THREADS, tpn_lds_bytes<R * NP>(), st>>>(
            ps.p[q].A, ps.p[q].B, ps.p[q].tb, ps.p[q].tiles, ps.p[q].start, ps.p[q].seg, ps.p[q].pfirst, ps.p[q].part,
            ps.p[q].gmax, ps.p[q].rec, gp0[q], gp1[q], ps.p[q].halo);
        tp_split_finish_kernel<R, NP><<<dim3((unsigned)tp_max_split_tiles(V), R * NP), TPN_THREADS, 0, st>>>(ps.p[q].A, ps.p[q].B, ps.p[q].tb, ps.p[q].split, ps.p[q].seg, ps.p[q].pfirst,
                                                                              ps.p[q].part, ps.p[q].gmax, gp0[q], gp1[q], ps.p[q].halo);
        tp_border_sum_kernel<R, NP><<<dim3((unsigned)((ps.p[q].B + 255) / 256), (unsigned)ps.p[q].A), 256, 0, st>>>(
            ps.p[q].A, ps.p[q].B, ps.p[q].tb, ps.p[q].start, ps.p[q].halo, gp0[q], gp1[q]);
    }
}

// nproj projections (1, or the 3 of a grid) x `planes` (1 or 2) planes each; cols0/cols1: first gradient column of the
// first / second plane of every projection; pairs[q] = (cx, cy) of projection q; sizes[q] = (A, B)
static int tp_backward(int64_t V, const float* coords, int cs, int R, int planes, int nproj, const int (*pairs)[2],
                       const int (*sizes)[2], const float* grad, int ld, const int* cols0, const int* cols1,
                       float* const* gp0, float* const* gp1, void* scratch, hipStream_t st) {
    if (R > TP_MAX_R || R < 1 || planes < 1 || planes > 2) return 1;
    if (planes == 2 && R > 5) {          // 4 cells x 4 corners x 2R sums per thread would not fit the registers
        // one plane after the other; both passes use the same scratch (stream order serialises them)
        const int rc = tp_backward(V, coords, cs, R, 1, nproj, pairs, sizes, grad, ld, cols0, cols0, gp0, gp0, scratch, st);
        return rc ? rc : tp_backward(V, coords, cs, R, 1, nproj, pairs, sizes, grad, ld, cols1, cols1, gp1, gp1, scratch, st);
    }
    TpProjSet ps;
    ps.n = nproj;
    char* sc = (char*)scratch;
    int total = 0;
    ZeroList zl;
    for (int q = 0; q < nproj; ++q) {
        TpProj& pj = ps.p[q];
        pj.cx = pairs[q][0];
        pj.cy = pairs[q][1];
        pj.A = sizes[q][0];
        pj.B = sizes[q][1];
        pj.col0 = cols0[q];
        pj.col1 = cols1[q];
        sc = tp_carve(pj, V, R * planes, sc);
        if (pj.tiles > TP_HIST_MAX_TILES) return 2;
        total += pj.tiles;
        zl.add(pj.count, (size_t)pj.tiles * TP_HEAD_ZEROED * 4, st);        // + gmax
        zl.add(gp0[q], (size_t)R * pj.A * pj.B * 4, st);
        if (planes == 2) zl.add(gp1[q], (size_t)R * pj.A * pj.B * 4, st);
    }
    launch_zero(zl, st);
    for (int q = 0; q < nproj; ++q) ps.p[q].part = (long long*)sc;
    for (int q = nproj; q < 3; ++q) ps.p[q] = ps.p[0];
    if (V <= 0) return 0;
    if (total > TP_HIST_MAX_TILES) {     // the three histograms do not fit one workgroup's LDS: one projection per pass
        for (int q = 0; q < nproj; ++q) {
            TpProjSet one;
            one.n = 1;
            one.p[0] = one.p[1] = one.p[2] = ps.p[q];
#define SCR_TP_ONE(RR)                                                                                   \
    case RR:                                                                                             \
        if (planes == 2) tp_backward_launch<(RR <= 5 ? RR : 1), 2>(V, coords, cs, grad, ld, one, gp0 + q, gp1 + q, st); \
        else tp_backward_launch<RR, 1>(V, coords, cs, grad, ld, one, gp0 + q, gp1 + q, st);              \
        break;
            switch (R) { SCR_TP_ONE(1) SCR_TP_ONE(2) SCR_TP_ONE(3) SCR_TP_ONE(4) SCR_TP_ONE(5) SCR_TP_ONE(6) SCR_TP_ONE(7) SCR_TP_ONE(8) SCR_TP_ONE(9) SCR_TP_ONE(10) SCR_TP_ONE(11) SCR_TP_ONE(12) SCR_TP_ONE(13) SCR_TP_ONE(14) SCR_TP_ONE(15) SCR_TP_ONE(16) }
#undef SCR_TP_ONE
        }
        return 0;
    }
#define SCR_TP_BWD(RR)                                                                                   \
    case RR:                                                                                             \
        if (planes == 2) tp_backward_launch<(RR <= 5 ? RR : 1), 2>(V, coords, cs, grad, ld, ps, gp0, gp1, st); \
        else tp_backward_launch<RR, 1>(V, coords, cs, grad, ld, ps, gp0, gp1, st);                       \
        break;
    switch (R) { SCR_TP_BWD(1) SCR_TP_BWD(2) SCR_TP_BWD(3) SCR_TP_BWD(4) SCR_TP_BWD(5) SCR_TP_BWD(6) SCR_TP_BWD(7) SCR_TP_BWD(8) SCR_TP_BWD(9) SCR_TP_BWD(10) SCR_TP_BWD(11) SCR_TP_BWD(12) SCR_TP_BWD(13) SCR_TP_BWD(14) SCR_TP_BWD(15) SCR_TP_BWD(16) }
#undef SCR_TP_BWD
    return 0;
}

// ---- all grids of a step at once (see tp_scatter9_kernel).  ngrids <= 3; grid g: R[g] channels per plane (the attention
// grid arrives with its planes stacked), sizes X/Y/Z[g], first column col[g] of its 3 * R[g] gradient columns (xy, xz, yz
// in this order); the columns of all grids are contiguous from col[0].  Returns 3 when the layout is not one the fused
// pass handles (the caller falls back to one call per grid).
size_t triplane_multi_scratch_bytes(int64_t V, int ngrids, const int* R, const int* X, const int* Y, const int* Z) {
    size_t b = 0;
    int widest = 1;
    for (int g = 0; g < ngrids; ++g) {
        b += tp_proj_bytes(V, X[g], Y[g], R[g]) + tp_proj_bytes(V, X[g], Z[g], R[g]) + tp_proj_bytes(V, Y[g], Z[g], R[g]);
        widest = R[g] > widest ? R[g] : widest;
    }
    return b + tp_part_bytes(V, widest);
}

template <int RR, int NPX>
static void tp_gather_launch(const TpProj& pj, int64_t V, float* gp, hipStream_t st) {
    static bool big_lds[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !big_lds[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)tp_cell_gather_kernel<RR, NPX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)tpn_lds_bytes<RR * NPX>());
        if (e == hipSuccess && dev >= 0 && dev < 64) big_lds[dev] = true;
        (void)hipGetLastError();
    }
    tp_cell_gather_kernel<RR, NPX><<<tp_gather_grid(pj, V), TPN_THREADS, tpn_lds_bytes<RR * NPX>(), st>>>(
        pj.A, pj.B, pj.tb, pj.tiles, pj.start, pj.seg, pj.pfirst, pj.part, pj.gmax, pj.rec, gp, gp, pj.halo);
    tp_split_finish_kernel<RR, NPX><<<dim3((unsigned)tp_max_split_tiles(V), RR * NPX), TPN_THREADS, 0, st>>>(pj.A, pj.B, pj.tb, pj.split, pj.seg, pj.pfirst, pj.part,
                                                                                                       pj.gmax, gp, gp, pj.halo);
    tp_border_sum_kernel<RR, NPX><<<dim3((unsigned)((pj.B + 255) / 256), (unsigned)pj.A), 256, 0, st>>>(pj.A, pj.B, pj.tb, pj.start, pj.halo, gp, gp);
}

// nl_coef (NULL: off): the gradient rows are formed from the BatchNorm-Linear's coefficients, its upstream gradient nl_dy
// [V,32] and its input nl_x (the sampled matrix, same columns as grad) -- see tp_scatter9_kernel; grad may then be NULL.
int launch_triplane_backward_multi(int64_t V, const float* coords, int cs, int ngrids, const int* R, const int* X, const int* Y,
                                   const int* Z, const int* col, const float* grad, int ld, float* const* grad_planes,
                                   void* scratch, const float* nl_coef, const float* nl_dy, int nl_lddy, const float* nl_x,
                                   int nl_ldx, hipStream_t st) {
    if (ngrids < 1 || ngrids > 3 || cs < 3) return 3;
    if (col[0] % 4 != 0) return 3;
    if (grad && (ld % 4 != 0 || ((uintptr_t)grad & 15) != 0)) return 3;
    if (!grad && !nl_coef) return 3;
    if (nl_coef && (nl_ldx % 4 != 0 || ((uintptr_t)nl_x & 15) != 0 || nl_lddy % 4 != 0 || ((uintptr_t)nl_dy & 15) != 0)) return 3;
    for (int g = 0; g + 1 < ngrids; ++g)
        if (col[g + 1] != col[g] + 3 * R[g]) return 3;
    const int RA = R[0], RB = ngrids > 1 ? R[1] : 0, RC = ngrids > 2 ? R[2] : 0;
    // the layouts of FeaturePlanes (r = channels per plane): attention grid stacked (2 r) [+ plain (r) [+ plain (r)]], or plain grids only
    // (15: the attention grid's pair planes with the same-size plain grid's plane stacked on them, one record / gather per projection)
    const bool known = ((RA == 10 || RA == 5) && (RB == 0 || RB == 5) && (RC == 0 || RC == 5)) || (RA == 15 && (RB == 0 || RB == 5) && RC == 0);
    if (!known || (RC && !RB)) return 3;
    TpProjSet9 ps;
    ZeroList zl;
    ps.n = 3 * ngrids;
    const int pairs[3][2] = {{1, 0}, {2, 0}, {2, 1}};
    char* sc = (char*)scratch;
    int total = 0;
    for (int g = 0; g < ngrids; ++g) {
        const int sizes[3][2] = {{X[g], Y[g]}, {X[g], Z[g]}, {Y[g], Z[g]}};
        for (int q = 0; q < 3; ++q) {
            TpProj& pj = ps.p[3 * g + q];
            pj.cx = pairs[q][0];
            pj.cy = pairs[q][1];
            pj.A = sizes[q][0];
            pj.B = sizes[q][1];
            pj.col0 = pj.col1 = col[g] + q * R[g];
            sc = tp_carve(pj, V, R[g], sc);
            total += pj.tiles;
            zl.add(pj.count, (size_t)pj.tiles * TP_HEAD_ZEROED * 4, st);        // + gmax
            zl.add(grad_planes[3 * g + q], (size_t)R[g] * pj.A * pj.B * 4, st);
        }
    }
    launch_zero(zl, st);
    for (int q = 0; q < ps.n; ++q) ps.p[q].part = (long long*)sc;
    for (int q = ps.n; q < 9; ++q) ps.p[q] = ps.p[0];
    if (total > TP_HIST_MAX_TILES) return 3;
    if (V <= 0) return 0;
    const unsigned nwg = (unsigned)((V + TP_PER_WG - 1) / TP_PER_WG);
    const size_t hb = (size_t)total * 4;
    tp_count9_kernel<<<nwg, TP_THREADS, hb, st>>>(V, coords, cs, ps);
    tp_scan9_kernel<<<ps.n, 1024, 0, st>>>(ps);
#define SCR_TP_S9(a, b, c)                                               \
    do {                                                                 \
        if (nl_coef) {                                                   \
            tp_allow_lds((const void*)tp_scatter9_kernel<a, b, c, true>, 2 * hb);  \
            tp_scatter9_kernel<a, b, c, true><<<nwg, TP_THREADS, 2 * hb, st>>>(V, coords, cs, grad, ld, col[0], ps, nl_coef, nl_dy, nl_lddy, nl_x, nl_ldx); \
        } else {                                                         \
            tp_allow_lds((const void*)tp_scatter9_kernel<a, b, c, false>, 2 * hb);  \
            tp_scatter9_kernel<a, b, c, false><<<nwg, TP_THREADS, 2 * hb, st>>>(V, coords, cs, grad, ld, col[0], ps, nullptr, nullptr, 0, nullptr, 0); \
        }                                                                \
    } while (0)
    if (RA == 15) {
        if (RB) SCR_TP_S9(15, 5, 0); else SCR_TP_S9(15, 0, 0);
    } else if (RA == 10) {
        if (RC) SCR_TP_S9(10, 5, 5); else if (RB) SCR_TP_S9(10, 5, 0); else SCR_TP_S9(10, 0, 0);
    } else {
        if (RC) SCR_TP_S9(5, 5, 5); else if (RB) SCR_TP_S9(5, 5, 0); else SCR_TP_S9(5, 0, 0);
    }
#undef SCR_TP_S9
    for (int g = 0; g < ngrids; ++g)
        for (int q = 0; q < 3; ++q) {
            if (R[g] == 15) tp_gather_launch<15, 1>(ps.p[3 * g + q], V, grad_planes[3 * g + q], st);
            else if (R[g] == 10) tp_gather_launch<10, 1>(ps.p[3 * g + q], V, grad_planes[3 * g + q], st);
            else tp_gather_launch<5, 1>(ps.p[3 * g + q], V, grad_planes[3 * g + q], st);
        }
    return 0;
}

int launch_plane_sample_backward(int64_t V, const float* coords, int cs, int cx, int cy, int R, int A, int B, int planes,
                                 const float* grad_out0, const float* grad_out1, int ld, float* grad_plane0,
                                 float* grad_plane1, void* scratch, hipStream_t st) {
    const int pairs[1][2] = {{cx, cy}}, sizes[1][2] = {{A, B}};
    // the second block as a column offset from the first (the two blocks lie in the same gradient matrix)
    const int cols0[1] = {0}, cols1[1] = {planes == 2 ? (int)(grad_out1 - grad_out0) : 0};
    float* gp0[1] = {grad_plane0};
    float* gp1[1] = {planes == 2 ? grad_plane1 : grad_plane0};
    return tp_backward(V, coords, cs, R, planes, 1, pairs, sizes, grad_out0, ld, cols0, cols1, gp0, gp1, scratch, st);
}

// the three projections of a grid (coordinate pairs of scene/grids.py:148-150) in one pass over the points
int launch_triplane_backward(int64_t V, const float* coords, int cs, int R, int X, int Y, int Z, int planes,
                             const float* grad_out, int ld, const int* cols /*[3 * planes]*/, float* const* grad_planes,
                             void* scratch, hipStream_t st) {
    const int pairs[3][2] = {{1, 0}, {2, 0}, {2, 1}}, sizes[3][2] = {{X, Y}, {X, Z}, {Y, Z}};
    const int* cols1 = planes == 2 ? cols + 3 : cols;
    float* const* gp1 = planes == 2 ? grad_planes + 3 : grad_planes;
    return tp_backward(V, coords, cs, R, planes, 3, pairs, sizes, grad_out, ld, cols, cols1, grad_planes, gp1, scratch, st);
}


// ---- normalised sample coordinates: ind = (xyz - lo) / (hi - lo) * 2 - 1 (scene/grids.py:146), the framework's four
// elementwise passes over [V, 3] as one; the same IEEE operations in the same order, so the same bits
__global__ void __launch_bounds__(256) box_coords_kernel(int64_t n3, const float* __restrict__ xyz, float lo0, float lo1,
                                                         float lo2, float hi0, float hi1, float hi2, float* __restrict__ out) {
#pragma clang fp contract(off)
    // four points = twelve floats = three 16-byte accesses per thread: the component of float j of the group is j % 3
    const float lo[3] = {lo0, lo1, lo2}, den[3] = {hi0 - lo0, hi1 - lo1, hi2 - lo2};
    const int64_t groups = n3 / 12;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < groups; q += (int64_t)gridDim.x * 256) {
        const float4* src = (const float4*)(xyz + 12 * q);
        float4 v[3] = {src[0], src[1], src[2]};
        float* f = (float*)v;
#pragma unroll
        for (int j = 0; j < 12; ++j) f[j] = ((f[j] - lo[j % 3]) / den[j % 3]) * 2.0f - 1.0f;
        float4* dst = (float4*)(out + 12 * q);
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
    }
    if (blockIdx.x == 0)        // the tail (fewer than four points)
        for (int64_t e = 12 * groups + threadIdx.x; e < n3; e += 256) {
            const int c = (int)(e % 3);
            out[e] = ((xyz[e] - lo[c]) / den[c]) * 2.0f - 1.0f;
        }
}
void launch_box_coords(int64_t V, const float* xyz, const float* lo, const float* hi, float* out, hipStream_t st) {
    if (V <= 0) return;
    const int64_t n3 = 3 * V, groups = n3 / 12;
    const int64_t want = (groups + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : want < 16384 ? want : 16384);
    box_coords_kernel<<<grid, 256, 0, st>>>(n3, xyz, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], out);
}

}  // namespace scr
