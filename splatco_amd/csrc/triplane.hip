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
// accumulates its points into a 33x33xR LDS copy of the tile with LDS float atomics and flushes
// the tile once (global atomics only on that flush: ~5 k per tile instead of 20 per point).
#include "common.h"

namespace scr {

constexpr int TP_TILE = 32;            // cells per tile edge; a tile owns 33 x 33 nodes
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

// pass 4: one workgroup per tile accumulates its points in LDS, then flushes the tile
__global__ void __launch_bounds__(256)
tp_accumulate_kernel(const float* __restrict__ coords, int cs, int cx, int cy, int A, int B, int tb, int R,
                     const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ perm,
                     const float* __restrict__ grad_out /*[V][ld]*/, int ld, float* __restrict__ grad_plane /*[R][A][B]*/) {
    __shared__ float acc[TP_MAX_R][TP_NODES * TP_NODES];
    const int t = blockIdx.x;
    const uint32_t lo = tile_start[t], hi = tile_start[t + 1];
    if (lo == hi) return;
    const int ta = t / tb, tbb = t % tb;
    for (int i = threadIdx.x; i < R * TP_NODES * TP_NODES; i += 256) (&acc[0][0])[i] = 0.0f;
    __syncthreads();
    for (uint32_t q = lo + threadIdx.x; q < hi; q += 256) {
        const uint32_t i = perm[q];
        int a0, b0;
        float fa, fb;
        tp_cell(coords[(size_t)i * cs + cx], coords[(size_t)i * cs + cy], A, B, a0, b0, fa, fb);
        // weights as torch: nw = (ix_se - ix)(iy_se - iy) ... ; corners outside the plane contribute nothing
        const float w00 = (1.0f - fa) * (1.0f - fb), w01 = (1.0f - fa) * fb, w10 = fa * (1.0f - fb), w11 = fa * fb;
        const bool va0 = a0 >= 0 && a0 < A, va1 = a0 + 1 >= 0 && a0 + 1 < A;
        const bool vb0 = b0 >= 0 && b0 < B, vb1 = b0 + 1 >= 0 && b0 + 1 < B;
        const int la = a0 - ta * TP_TILE, lb = b0 - tbb * TP_TILE;  // in [-1, 31]
        const int n00 = la * TP_NODES + lb;
        for (int r = 0; r < R; ++r) {
            const float g = grad_out[(size_t)i * ld + r];
            float* base = acc[r];
            if (va0 && vb0) unsafeAtomicAdd(base + n00, g * w00);
            if (va0 && vb1) unsafeAtomicAdd(base + n00 + 1, g * w01);
            if (va1 && vb0) unsafeAtomicAdd(base + n00 + TP_NODES, g * w10);
            if (va1 && vb1) unsafeAtomicAdd(base + n00 + TP_NODES + 1, g * w11);
        }
    }
    __syncthreads();
    // flush: nodes on the tile's first / last row or column are shared with the neighbouring tile
    for (int i = threadIdx.x; i < TP_NODES * TP_NODES; i += 256) {
        const int la = i / TP_NODES, lb = i % TP_NODES;
        const int a = ta * TP_TILE + la, b = tbb * TP_TILE + lb;
        if (a >= A || b >= B) continue;
        for (int r = 0; r < R; ++r) {
            const float v = acc[r][i];
            if (v != 0.0f) unsafeAtomicAdd(grad_plane + ((size_t)r * A + a) * B + b, v);
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
    tp_accumulate_kernel<<<tiles, 256, 0, st>>>(coords, cs, cx, cy, A, B, tb, R, tile_start, perm, grad_out, ld, grad_plane);
    return 0;
}

}  // namespace scr
