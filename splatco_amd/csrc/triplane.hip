// splatco_amd/csrc/triplane.hip -- tri-plane bilinear feature sampling, forward and backward (gfx950).
//
// Forward: one thread per anchor samples all three planes and writes its 3*R features as one
// contiguous row segment of the caller's [V, ld] matrix (no per-plane outputs, no concatenation).
//
// PlaneGrid samples three learnable planes [1,R,A,B] at V anchor positions with
// F.grid_sample(bilinear, align_corners=True, zeros padding) (scene/grids.py:146-182).  The
// backward is a scatter-add of 4 corners x R channels per point into a dense plane gradient:
// torch issues one global float atomic per (point, corner, channel) -- 92 M atomics per call at
// 4.6 M anchors, and MI355X retires ~28 G device atomics/s whatever their scope, so those calls
// are 41 % of the whole render() step at BASELINE.json configs[2].
//
// Here the points are first bucketed by 32x32-cell plane tile (the same LDS-aggregated
// count / reserve / place scheme as the rasterizer's tile binning); one workgroup per tile then
// sorts its points by cell in LDS and lets every thread sum, in registers, the contributions that reach
// the tile nodes it owns (tp_node_gather_kernel): no float atomics inside a tile, global atomics only
// for the 128 border nodes a tile shares with its neighbours (0.33 ms per plane at 4.6 M points).
#include "common.h"

namespace scr {

#ifndef SCR_TP_TILE
#define SCR_TP_TILE 32
#endif
constexpr int TP_TILE = SCR_TP_TILE;   // cells per tile edge; a tile owns (TP_TILE + 1)^2 nodes
constexpr int TP_NODES = TP_TILE + 1;
constexpr int TP_MAX_R = 8;            // channels per plane supported by the LDS tile (R = num_channels / 3)
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

// pass 1: per-tile point counts (LDS histogram per workgroup, one global atomic per touched tile)
__global__ void __launch_bounds__(TP_THREADS)
tp_count_kernel(int64_t V, const float* __restrict__ coords, int cs, int cx, int cy, int A, int B, int tb, int tiles,
                uint32_t* __restrict__ tile_count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    for (int t = threadIdx.x; t < tiles; t += TP_THREADS) hist[t] = 0;
    __syncthreads();
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        if (i >= V) break;
        const int t = tp_tile_of(coords[i * cs + cx], coords[i * cs + cy], A, B, tb);
        if (t >= 0) atomicAdd(&hist[t], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += TP_THREADS) {
        const uint32_t c = hist[t];
        if (c) atomicAdd(&tile_count[t], c);
    }
}

// pass 2 (one workgroup): exclusive scan of the tile counts -> tile_start[tiles + 1]; cursor = 0
__global__ void __launch_bounds__(1024)
tp_scan_kernel(int tiles, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_start,
               uint32_t* __restrict__ cursor) {
    __shared__ uint32_t lds[1024 / WAVE];
    uint32_t carry = 0;
    for (int base = 0; base < tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < tiles ? tile_count[i] : 0u;
        uint32_t tot;
        const uint32_t ex = tp_block_scan(v, lds, tot);
        if (i < tiles) {
            tile_start[i] = carry + ex;
            cursor[i] = 0;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) tile_start[tiles] = carry;
}

// pass 3: point indices grouped by tile
__global__ void __launch_bounds__(TP_THREADS)
tp_scatter_kernel(int64_t V, const float* __restrict__ coords, int cs, int cx, int cy, int A, int B, int tb, int tiles,
                  const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ cursor,
                  uint32_t* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    for (int t = threadIdx.x; t < tiles; t += TP_THREADS) hist[t] = 0;
    __syncthreads();
    int tl[TP_ROUNDS];
#pragma unroll
    for (int r = 0; r < TP_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        tl[r] = i < V ? tp_tile_of(coords[i * cs + cx], coords[i * cs + cy], A, B, tb) : -1;
        if (tl[r] >= 0) atomicAdd(&hist[tl[r]], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += TP_THREADS) {
        const uint32_t c = hist[t];
        if (c) hist[t] = tile_start[t] + atomicAdd(&cursor[t], c);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TP_ROUNDS; ++r) {
        if (tl[r] < 0) continue;
        const int64_t i = (int64_t)blockIdx.x * TP_PER_WG + r * TP_THREADS + threadIdx.x;
        perm[atomicAdd(&hist[tl[r]], 1u)] = (uint32_t)i;
    }
}

// pass 4, node-centred.  The obvious kernel -- every point adds its 4 corners x R channels into an LDS copy of
// the tile with ds_add_f32 -- is bound by the LDS float-atomic rate (about one lane every four cycles on
// gfx950: 0.63 ms per plane at 4.6 M points, 0.37 ms with plain, wrong, read-modify-writes).  Here a
// workgroup sorts a chunk of its tile's points by
// CELL inside LDS (one integer LDS atomic per point hands out the slot), stages the points' bilinear
// fractions and gradient rows in that order, and then every thread owns a few of the tile's 33 x 33 NODES and
// sums, in registers, the contributions of the points in the node's four adjacent cells.  No float atomics
// inside the tile; the sums of a tile's border nodes (shared with the neighbours) go out as global atomics,
// interior nodes as plain stores.
constexpr int TPN_CHUNK = 1024;                 // points staged per round
constexpr int TPN_CELLS = TP_NODES * TP_NODES;  // local cells (la + 1, lb + 1), la, lb in [-1, 31]
constexpr int TPN_NPT = (TP_NODES * TP_NODES + 255) / 256;  // nodes per thread
template <int R>
__global__ void __launch_bounds__(256)
tp_node_gather_kernel(const float* __restrict__ coords, int cs, int cx, int cy, int A, int B, int tb,
                      const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ perm,
                      const float* __restrict__ grad_out /*[V][ld]*/, int ld, float* __restrict__ grad_plane /*[R][A][B]*/) {
    __shared__ uint32_t cnt[TPN_CELLS + 1], start[TPN_CELLS + 1];
    __shared__ uint32_t waves[4];
    __shared__ float pfa[TPN_CHUNK], pfb[TPN_CHUNK];
    __shared__ float pg[TPN_CHUNK][R];
    const int t = blockIdx.x;
    const uint32_t lo = tile_start[t], hi = tile_start[t + 1];
    if (lo == hi) return;
    const int ta = t / tb, tbb = t % tb;
    float acc[TPN_NPT][R];
#pragma unroll
    for (int j = 0; j < TPN_NPT; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[j][r] = 0.0f;
    for (uint32_t c0 = lo; c0 < hi; c0 += TPN_CHUNK) {
        for (int i = threadIdx.x; i <= TPN_CELLS; i += 256) cnt[i] = 0;
        __syncthreads();
        // ---- this thread's points: cell, rank inside the cell, fractions, gradient row (kept in registers)
        int cell[TPN_CHUNK / 256];
        uint32_t rank[TPN_CHUNK / 256];
        float fa[TPN_CHUNK / 256], fb[TPN_CHUNK / 256], g[TPN_CHUNK / 256][R];
#pragma unroll
        for (int k = 0; k < TPN_CHUNK / 256; ++k) {
            const uint32_t q = c0 + threadIdx.x + 256u * k;
            cell[k] = -1;
            if (q < hi) {
                const uint32_t i = perm[q];
                int a0, b0;
                tp_cell(coords[(size_t)i * cs + cx], coords[(size_t)i * cs + cy], A, B, a0, b0, fa[k], fb[k]);
                cell[k] = (a0 - ta * TP_TILE + 1) * TP_NODES + (b0 - tbb * TP_TILE + 1);  // la, lb in [-1, 31]
                rank[k] = atomicAdd(&cnt[cell[k]], 1u);
#pragma unroll
                for (int r = 0; r < R; ++r) g[k][r] = grad_out[(size_t)i * ld + r];
            }
        }
        __syncthreads();
        // ---- exclusive scan of the cell counts (1089 cells, 5 per thread)
        {
            uint32_t v[TPN_NPT], s = 0;
#pragma unroll
            for (int j = 0; j < TPN_NPT; ++j) {
                const int c = threadIdx.x * TPN_NPT + j;
                v[j] = c < TPN_CELLS ? cnt[c] : 0u;
                s += v[j];
            }
            uint32_t tot;
            uint32_t ex = tp_block_scan(s, waves, tot);
#pragma unroll
            for (int j = 0; j < TPN_NPT; ++j) {
                const int c = threadIdx.x * TPN_NPT + j;
                if (c < TPN_CELLS) start[c] = ex;
                ex += v[j];
            }
            if (threadIdx.x == 0) start[TPN_CELLS] = tot;
        }
        __syncthreads();
        // ---- stage in cell order
#pragma unroll
        for (int k = 0; k < TPN_CHUNK / 256; ++k)
            if (cell[k] >= 0) {
                const uint32_t slot = start[cell[k]] + rank[k];
                pfa[slot] = fa[k];
                pfb[slot] = fb[k];
#pragma unroll
                for (int r = 0; r < R; ++r) pg[slot][r] = g[k][r];
            }
        __syncthreads();
        // ---- nodes: node (na, nb) is corner (1,1) of cell (na-1, nb-1), (1,0) of (na-1, nb), (0,1) of (na, nb-1),
        // (0,0) of (na, nb); weights as torch: (da ? fa : 1 - fa) * (db ? fb : 1 - fb)
#pragma unroll
        for (int j = 0; j < TPN_NPT; ++j) {
            const int nd = threadIdx.x + 256 * j;
            if (nd >= TPN_CELLS) break;
            const int na = nd / TP_NODES, nb = nd % TP_NODES;
#pragma unroll
            for (int corner = 0; corner < 4; ++corner) {
                const int da = corner >> 1, db = corner & 1;          // which corner of the cell this node is
                const int ca = na - da, cb = nb - db;                 // local cell (la, lb) = (ca, cb) in [-1, 31]
                if (ca > TP_TILE - 1 || cb > TP_TILE - 1) continue;   // cell of the next tile
                const int c = (ca + 1) * TP_NODES + (cb + 1);
                for (uint32_t sidx = start[c], e = start[c + 1]; sidx < e; ++sidx) {
                    const float wa = da ? pfa[sidx] : 1.0f - pfa[sidx], wb = db ? pfb[sidx] : 1.0f - pfb[sidx];
                    const float w = wa * wb;
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[j][r] += pg[sidx][r] * w;
                }
            }
        }
        __syncthreads();
    }
    // ---- out: nodes outside the plane receive nothing (zeros padding)
#pragma unroll
    for (int j = 0; j < TPN_NPT; ++j) {
        const int nd = threadIdx.x + 256 * j;
        if (nd >= TPN_CELLS) break;
        const int na = nd / TP_NODES, nb = nd % TP_NODES;
        const int a = ta * TP_TILE + na, b = tbb * TP_TILE + nb;
        if (a >= A || b >= B) continue;
        const bool shared = na == 0 || na == TP_TILE || nb == 0 || nb == TP_TILE;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float* dst = grad_plane + ((size_t)r * A + a) * B + b;
            if (!shared) *dst = acc[j][r];
            else if (acc[j][r] != 0.0f) unsafeAtomicAdd(dst, acc[j][r]);
        }
    }
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

template <int R, bool CL>
__global__ void __launch_bounds__(256)
triplane_forward_kernel(int64_t V, const float* __restrict__ coords, int cs, const float* __restrict__ xy,
                        const float* __restrict__ xz, const float* __restrict__ yz, int X, int Y, int Z,
                        float* __restrict__ out, int ld, int col_xy, int col_xz, int col_yz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const float x = coords[i * cs], y = coords[i * cs + 1], z = coords[i * cs + 2];
    float* o = out + i * ld;
    // coordinate pairs of scene/grids.py:148-150: grid x indexes the LAST plane dimension
    if (CL) {
        tp_sample_plane_cl<R>(xy, X, Y, y, x, o + col_xy);
        tp_sample_plane_cl<R>(xz, X, Z, z, x, o + col_xz);
        tp_sample_plane_cl<R>(yz, Y, Z, z, y, o + col_yz);
    } else {
        tp_sample_plane<R>(xy, X, Y, y, x, o + col_xy);  // xy_plane [R,X,Y] at ind[..., [1, 0]]
        tp_sample_plane<R>(xz, X, Z, z, x, o + col_xz);  // xz_plane [R,X,Z] at ind[..., [2, 0]]
        tp_sample_plane<R>(yz, Y, Z, z, y, o + col_yz);  // yz_plane [R,Y,Z] at ind[..., [2, 1]]
    }
}

int launch_triplane_forward(int64_t V, const float* coords, int cs, const float* xy, const float* xz, const float* yz,
                            int R, int X, int Y, int Z, int channel_last, float* out, int ld, int col_xy, int col_xz,
                            int col_yz, hipStream_t st) {
    if (R > TP_MAX_R) return 1;
    if (V <= 0) return 0;
    const unsigned nb = (unsigned)((V + 255) / 256);
#define SCR_TP_FWD(RR)                                                                                          \
    case RR:                                                                                                    \
        if (channel_last)                                                                                       \
            triplane_forward_kernel<RR, true><<<nb, 256, 0, st>>>(V, coords, cs, xy, xz, yz, X, Y, Z, out, ld, col_xy,   \
                                                                  col_xz, col_yz);                               \
        else                                                                                                    \
            triplane_forward_kernel<RR, false><<<nb, 256, 0, st>>>(V, coords, cs, xy, xz, yz, X, Y, Z, out, ld, col_xy,  \
                                                                   col_xz, col_yz);                              \
        break;
    switch (R) {
        SCR_TP_FWD(1) SCR_TP_FWD(2) SCR_TP_FWD(3) SCR_TP_FWD(4) SCR_TP_FWD(5) SCR_TP_FWD(6) SCR_TP_FWD(7) SCR_TP_FWD(8)
    }
#undef SCR_TP_FWD
    return 0;
}

size_t triplane_scratch_bytes(int64_t V, int A, int B) {
    const size_t tiles = (size_t)((A + TP_TILE - 1) / TP_TILE) * ((B + TP_TILE - 1) / TP_TILE);
    return align_up((3 * tiles + 2) * 4) + align_up((size_t)(V > 0 ? V : 1) * 4);
}

int launch_plane_sample_backward(int64_t V, const float* coords, int cs, int cx, int cy, int R, int A, int B,
                                 const float* grad_out, int ld, float* grad_plane, void* scratch, hipStream_t st) {
    if (R > TP_MAX_R) return 1;
    const int ta = (A + TP_TILE - 1) / TP_TILE, tb = (B + TP_TILE - 1) / TP_TILE, tiles = ta * tb;
    if ((size_t)tiles * 4 > 64 * 1024) return 2;  // LDS histogram of the tile counts
    uint32_t* tile_count = (uint32_t*)scratch;
    uint32_t* tile_start = tile_count + tiles;
    uint32_t* cursor = tile_start + tiles + 1;
    uint32_t* perm = (uint32_t*)((char*)scratch + align_up((3 * (size_t)tiles + 2) * 4));
    (void)hipMemsetAsync(tile_count, 0, (size_t)tiles * 4, st);
    (void)hipMemsetAsync(grad_plane, 0, (size_t)R * A * B * 4, st);
    if (V <= 0) return 0;
    const unsigned nwg = (unsigned)((V + TP_PER_WG - 1) / TP_PER_WG);
    tp_count_kernel<<<nwg, TP_THREADS, (size_t)tiles * 4, st>>>(V, coords, cs, cx, cy, A, B, tb, tiles, tile_count);
    tp_scan_kernel<<<1, 1024, 0, st>>>(tiles, tile_count, tile_start, cursor);
    tp_scatter_kernel<<<nwg, TP_THREADS, (size_t)tiles * 4, st>>>(V, coords, cs, cx, cy, A, B, tb, tiles, tile_start, cursor, perm);
#define SCR_TP_BWD(RR)                                                                                         \
    case RR:                                                                                                   \
        tp_node_gather_kernel<RR><<<tiles, 256, 0, st>>>(coords, cs, cx, cy, A, B, tb, tile_start, perm, grad_out, ld, \
                                                         grad_plane);                                          \
        break;
    switch (R) {
        SCR_TP_BWD(1) SCR_TP_BWD(2) SCR_TP_BWD(3) SCR_TP_BWD(4) SCR_TP_BWD(5) SCR_TP_BWD(6) SCR_TP_BWD(7) SCR_TP_BWD(8)
    }
#undef SCR_TP_BWD
    return 0;
}

}  // namespace scr
