// splatco_amd/csrc/anchor_gather.hip -- the head of generate_neural_gaussians as one pass per direction (gfx950).
//
// Reference, gaussian_renderer/__init__.py:23-31: four boolean-mask gathers of the per-anchor parameters
//     feat = _anchor_feat[visible], anchor = _anchor[visible], grid_offsets = _offset[visible],
//     grid_scaling = exp(_scaling)[visible]
// and their concatenation cat(feat, anchor, grid_offsets.view(V,-1), grid_scaling) [V,71], the input of the
// attribute branch of FeaturePlanes.  In PyTorch that is four index_select kernels, an exp over all N anchors, a cat,
// and in the backward four index_add_ into zero-filled [N,.] buffers plus the slicing of the cat's gradient.
//
// The rows are 12 .. 284 bytes long and mostly misaligned, so a thread-per-element kernel moves 4 bytes per lane and
// instruction and stalls at 1.3 TB/s (measured).  Here a workgroup owns 64 consecutive rows of the OUTPUT side, whose
// five (forward) / four (backward) destination chunks are contiguous and 16-byte aligned:
//   forward : the 64 visible anchors' parameter rows are gathered through the index into a [64][71] tile in LDS
//             (16 / 8-byte loads where the row alignment allows), exp applied to the scaling columns; then the tile
//             streams out as float4 -- once as the concatenated matrix, once split into feat / anchor / offsets /
//             exp(scaling) for the kernels that read those;
//   backward: the visible anchors among 64 consecutive anchors own CONSECUTIVE rows of the upstream gradients (the
//             index is ascending), so those rows are loaded as contiguous chunks into LDS, summed (d g_fea + d part),
//             d exp applied, and the four parameter-gradient chunks are written whole -- zeros for invisible anchors:
//             every element exactly once, no atomics, no memset.
#include "common.h"

namespace scr {

constexpr int AG_FEAT = 32, AG_OFF = 30, AG_COLS = 71;   // feat | anchor 3 | offsets 30 | scaling 6
constexpr int AG_ROWS = 64;                              // rows per workgroup
constexpr int AG_LD = 72;                                // LDS row stride (floats): 288 B, 16-byte aligned rows
constexpr int AG_THREADS = 256;

// streams `count` floats (a contiguous, 16-byte aligned chunk that starts at dst) out of the tile; element e of the
// chunk is tile[(e / width) * AG_LD + col0 + e % width].  Whole float4s, the tail (count not a multiple of 4, last
// workgroup only) as scalars.
__device__ __forceinline__ void ag_store_chunk(float* __restrict__ dst, const float* tile, int count, int width, int col0) {
    const int n4 = count >> 2;
    for (int q = threadIdx.x; q < n4; q += AG_THREADS) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 4 * q + j, r = e / width, c = e - r * width;
            v[j] = tile[r * AG_LD + col0 + c];
        }
        *(float4*)(dst + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    }
    for (int e = 4 * n4 + threadIdx.x; e < count; e += AG_THREADS) {
        const int r = e / width, c = e - r * width;
        dst[e] = tile[r * AG_LD + col0 + c];
    }
}

__global__ void __launch_bounds__(AG_THREADS)
anchor_gather_kernel(int64_t V, const int64_t* __restrict__ idx, const float* __restrict__ p_feat,
                     const float* __restrict__ p_anchor, const float* __restrict__ p_offset,
                     const float* __restrict__ p_scaling, float* __restrict__ feat, float* __restrict__ anchor,
                     float* __restrict__ offsets, float* __restrict__ grid_scaling, float* __restrict__ g_fea, int ldg) {
    __shared__ __attribute__((aligned(16))) float tile[AG_ROWS * AG_LD];
    __shared__ int64_t rowsrc[AG_ROWS];
    const int64_t v0 = (int64_t)blockIdx.x * AG_ROWS;
    const int rows = (int)min((int64_t)AG_ROWS, V - v0);
    if (threadIdx.x < AG_ROWS) rowsrc[threadIdx.x] = threadIdx.x < rows ? idx[v0 + threadIdx.x] : 0;
    __syncthreads();
    // ---- gather: feat rows are 128 B (float4), offset rows 120 B and scaling rows 24 B (float2), anchor rows 12 B
    for (int e = threadIdx.x; e < rows * 8; e += AG_THREADS) {
        const int r = e >> 3, q = e & 7;
        *(float4*)&tile[r * AG_LD + 4 * q] = *(const float4*)(p_feat + rowsrc[r] * AG_FEAT + 4 * q);
    }
    for (int e = threadIdx.x; e < rows * 15; e += AG_THREADS) {
        const int r = e / 15, q = e - 15 * r;
        const float2 x = *(const float2*)(p_offset + rowsrc[r] * AG_OFF + 2 * q);
        tile[r * AG_LD + 35 + 2 * q] = x.x;
        tile[r * AG_LD + 36 + 2 * q] = x.y;
    }
    for (int e = threadIdx.x; e < rows * 3; e += AG_THREADS) {
        const int r = e / 3, q = e - 3 * r;
        tile[r * AG_LD + 32 + q] = p_anchor[rowsrc[r] * 3 + q];
        const float2 x = *(const float2*)(p_scaling + rowsrc[r] * 6 + 2 * q);
        tile[r * AG_LD + 65 + 2 * q] = expf(x.x);        // get_scaling = exp(_scaling)
        tile[r * AG_LD + 66 + 2 * q] = expf(x.y);
        if (q == 0) tile[r * AG_LD + AG_COLS] = 0.0f;    // the pad column of 16-byte aligned g_fea rows (ldg = 72)
    }
    __syncthreads();
    // ---- the workgroup's chunks of the five outputs are contiguous
    ag_store_chunk(g_fea + v0 * ldg, tile, rows * ldg, ldg, 0);      // ldg = 71 (packed) or 72 = AG_LD (a straight copy)
    ag_store_chunk(feat + v0 * AG_FEAT, tile, rows * AG_FEAT, AG_FEAT, 0);
    ag_store_chunk(anchor + v0 * 3, tile, rows * 3, 3, 32);
    ag_store_chunk(offsets + v0 * AG_OFF, tile, rows * AG_OFF, AG_OFF, 35);
    ag_store_chunk(grid_scaling + v0 * 6, tile, rows * 6, 6, 65);
}

// loads `count` floats of a contiguous chunk starting at src (any 4-byte alignment: the chunk starts at the first
// visible row of the workgroup) and ADDS them into the tile at (e / width, col0 + e % width)
__device__ __forceinline__ void ag_add_chunk(const float* __restrict__ src, float* tile, int count, int width, int col0) {
    if (!src) return;
    auto add = [&](int e, float x) {
        const int r = e / width, c = e - r * width;
        tile[r * AG_LD + col0 + c] += x;
    };
    // scalars up to the first 16-byte boundary, float4s, scalars for the rest
    const int head = min(count, (int)(((16u - (uint32_t)((uintptr_t)src & 15u)) & 15u) >> 2));
    const int n4 = (count - head) >> 2;
    if ((int)threadIdx.x < head) add(threadIdx.x, src[threadIdx.x]);
    for (int q = threadIdx.x; q < n4; q += AG_THREADS) {
        const float4 x = *(const float4*)(src + head + 4 * q);
        add(head + 4 * q, x.x); add(head + 4 * q + 1, x.y); add(head + 4 * q + 2, x.z); add(head + 4 * q + 3, x.w);
    }
    for (int e = head + 4 * n4 + threadIdx.x; e < count; e += AG_THREADS) add(e, src[e]);
}

__global__ void __launch_bounds__(AG_THREADS)
anchor_gather_backward_kernel(int64_t N, const int64_t* __restrict__ inv, const float* __restrict__ grid_scaling,
                              const float* __restrict__ d_feat, const float* __restrict__ d_anchor,
                              const float* __restrict__ d_offsets, const float* __restrict__ d_grid_scaling,
                              const float* __restrict__ d_g_fea, int ldg, float* __restrict__ g_feat,
                              float* __restrict__ g_anchor, float* __restrict__ g_offset,
                              float* __restrict__ g_scaling) {
    __shared__ __attribute__((aligned(16))) float tile[AG_ROWS * AG_LD];   // rows = the VISIBLE anchors of the block, in order
    __shared__ __attribute__((aligned(16))) float outt[AG_ROWS * AG_LD];   // rows = the block's 64 anchors
    __shared__ int rowv[AG_ROWS];                                          // row of `tile` of every anchor, -1 = invisible
    __shared__ int64_t first_v;
    __shared__ int nvis;
    const int64_t n0 = (int64_t)blockIdx.x * AG_ROWS;
    const int rows = (int)min((int64_t)AG_ROWS, N - n0);
    // the visible anchors of 64 consecutive anchors own consecutive upstream rows first_v .. first_v + nvis - 1
    int64_t myv = -1;
    if (threadIdx.x < AG_ROWS && (int)threadIdx.x < rows) myv = inv[n0 + threadIdx.x];
    if (threadIdx.x < 64) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(myv >= 0);
        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        rowv[threadIdx.x] = myv >= 0 ? below : -1;
        if (myv >= 0 && below == 0) first_v = myv;
        if (threadIdx.x == 0) nvis = __builtin_popcountll(bal);
    }
    for (int e = threadIdx.x; e < AG_ROWS * AG_LD; e += AG_THREADS) tile[e] = 0.0f;
    __syncthreads();
    const int nv = nvis;
    if (nv) {
        const int64_t fv = first_v;
        ag_add_chunk(d_g_fea ? d_g_fea + fv * ldg : nullptr, tile, nv * ldg, ldg, 0);   // (the pad column lands in tile column 71: unused)
        __syncthreads();   // the parts below add into the same cells
        ag_add_chunk(d_feat ? d_feat + fv * AG_FEAT : nullptr, tile, nv * AG_FEAT, AG_FEAT, 0);
        ag_add_chunk(d_anchor ? d_anchor + fv * 3 : nullptr, tile, nv * 3, 3, 32);
        ag_add_chunk(d_offsets ? d_offsets + fv * AG_OFF : nullptr, tile, nv * AG_OFF, AG_OFF, 35);
        ag_add_chunk(d_grid_scaling ? d_grid_scaling + fv * 6 : nullptr, tile, nv * 6, 6, 65);
        __syncthreads();
        for (int e = threadIdx.x; e < nv * 6; e += AG_THREADS) {            // d exp(s) = exp(s) ds
            const int r = e / 6, c = e - 6 * r;
            tile[r * AG_LD + 65 + c] *= grid_scaling[(fv + r) * 6 + c];
        }
        __syncthreads();
    }
    // ---- the block's rows in anchor order (zeros for invisible anchors), then the four contiguous chunks
    for (int e = threadIdx.x; e < rows * AG_LD; e += AG_THREADS) {
        const int r = e / AG_LD, c = e - r * AG_LD;
        const int rv = rowv[r];
        outt[e] = (rv >= 0 && c < AG_COLS) ? tile[rv * AG_LD + c] : 0.0f;
    }
    __syncthreads();
    ag_store_chunk(g_feat + n0 * AG_FEAT, outt, rows * AG_FEAT, AG_FEAT, 0);
    ag_store_chunk(g_anchor + n0 * 3, outt, rows * 3, 3, 32);
    ag_store_chunk(g_offset + n0 * AG_OFF, outt, rows * AG_OFF, AG_OFF, 35);
    ag_store_chunk(g_scaling + n0 * 6, outt, rows * 6, 6, 65);
}

void launch_anchor_gather(int64_t V, const int64_t* idx, const float* p_feat, const float* p_anchor, const float* p_offset,
                          const float* p_scaling, float* feat, float* anchor, float* offsets, float* grid_scaling,
                          float* g_fea, int ldg, hipStream_t st) {
    if (V <= 0) return;
    anchor_gather_kernel<<<(unsigned)((V + AG_ROWS - 1) / AG_ROWS), AG_THREADS, 0, st>>>(
        V, idx, p_feat, p_anchor, p_offset, p_scaling, feat, anchor, offsets, grid_scaling, g_fea, ldg);
}

void launch_anchor_gather_backward(int64_t N, const int64_t* inv, const float* grid_scaling, const float* d_feat,
                                   const float* d_anchor, const float* d_offsets, const float* d_grid_scaling,
                                   const float* d_g_fea, int ldg, float* g_feat, float* g_anchor, float* g_offset,
                                   float* g_scaling, hipStream_t st) {
    if (N <= 0) return;
    anchor_gather_backward_kernel<<<(unsigned)((N + AG_ROWS - 1) / AG_ROWS), AG_THREADS, 0, st>>>(
        N, inv, grid_scaling, d_feat, d_anchor, d_offsets, d_grid_scaling, d_g_fea, ldg, g_feat, g_anchor, g_offset, g_scaling);
}

}  // namespace scr
