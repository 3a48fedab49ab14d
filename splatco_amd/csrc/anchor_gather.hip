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
// `add`: the chunk is added to what dst holds (gradients of a further view into the same buffer).
__device__ __forceinline__ void ag_store_chunk(float* __restrict__ dst, const float* tile, int count, int width, int col0,
                                               bool add = false) {
    const int n4 = count >> 2;
    for (int q = threadIdx.x; q < n4; q += AG_THREADS) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 4 * q + j, r = e / width, c = e - r * width;
            v[j] = tile[r * AG_LD + col0 + c];
        }
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (add) {
            const float4 old = *(const float4*)(dst + 4 * q);
            o = make_float4(old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w);
        }
        *(float4*)(dst + 4 * q) = o;
    }
    for (int e = 4 * n4 + threadIdx.x; e < count; e += AG_THREADS) {
        const int r = e / width, c = e - r * width;
        dst[e] = (add ? dst[e] : 0.0f) + tile[r * AG_LD + col0 + c];
    }
}

// Column statistics of g_fea for the BatchNorm that reads it (csrc/normlinear.hip): the workgroup has every 64 rows of the
// matrix in LDS anyway, so it also sums (x - x[0]) and (x - x[0])^2 per column over ITS rows and writes one row [2][NL_DP] of
// partial sums; beyond 2048 tiles a second kernel adds the tiles' rows into 2048 (fixed order), and the single-workgroup
// finish kernel of the BatchNorm-Linear combines those in fp64 exactly as it combines the partials of its own statistics
// pass, which is then skipped.  (A capped grid of workgroups WALKING the tiles, one row of sums per workgroup and no second
// kernel, was measured first: the gather went from 0.57 to 0.68 ms with the statistics switched off -- workgroups that loop
// in step load together and store together; one tile per workgroup with the reduction and the row written per TILE: 0.74 ms.
// Eight consecutive tiles per workgroup: 0.65 ms + 5 us for the second kernel, same box.)  What is skipped: 1.3 GB read and
// 0.26 ms at configs[2].  Thread (q = t % 18, rg = t / 18 < 14) sums the four columns 4 q .. 4 q + 3 over rows rg, rg + 14, ..:
// one ds_read_b128 per row.
#ifndef SCR_AG_MAX_WGS
#define SCR_AG_MAX_WGS 2048
#endif
constexpr int AG_MAX_WGS = SCR_AG_MAX_WGS;
#ifndef SCR_AG_TPW
#define SCR_AG_TPW 8
#endif
constexpr int AG_TPW = SCR_AG_TPW;      // consecutive 64-row tiles per workgroup: one row of statistics (and its reduction) per workgroup
int anchor_gather_stat_rows(int64_t V) {      // rows the consumer reads
    const int64_t tiles = (V + AG_ROWS * AG_TPW - 1) / (AG_ROWS * AG_TPW);
    return (int)(tiles < AG_MAX_WGS ? (tiles > 0 ? tiles : 1) : AG_MAX_WGS);
}
int64_t anchor_gather_stat_buffer_rows(int64_t V) {      // rows of the buffer: the consumer's, then one per workgroup when a second kernel reduces them
    const int64_t tiles = (V + AG_ROWS * AG_TPW - 1) / (AG_ROWS * AG_TPW);
    return tiles <= AG_MAX_WGS ? (tiles > 0 ? tiles : 1) : AG_MAX_WGS + tiles;
}

// rows AG_MAX_WGS.. (one per tile) -> rows 0 .. AG_MAX_WGS - 1: row r = sum of tile rows r, r + AG_MAX_WGS, ... in that order
__global__ void __launch_bounds__(2 * NL_DP) anchor_gather_stat_reduce_kernel(int64_t tiles, float* __restrict__ stats) {
    const float* in = stats + (size_t)AG_MAX_WGS * 2 * NL_DP;
    float a = 0.0f;
    if (threadIdx.x % NL_DP < AG_LD)      // (the tiles write 72 of a row's 80 columns)
        for (int64_t t = blockIdx.x; t < tiles; t += AG_MAX_WGS) a += in[(size_t)t * 2 * NL_DP + threadIdx.x];
    stats[(size_t)blockIdx.x * 2 * NL_DP + threadIdx.x] = a;
}

__global__ void __launch_bounds__(AG_THREADS)
anchor_gather_kernel(int64_t V, const int64_t* __restrict__ idx, const float* __restrict__ p_feat,
                     const float* __restrict__ p_anchor, const float* __restrict__ p_offset,
                     const float* __restrict__ p_scaling, float* __restrict__ feat, float* __restrict__ anchor,
                     float* __restrict__ offsets, float* __restrict__ grid_scaling, float* __restrict__ g_fea, int ldg,
                     float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float tile[AG_ROWS * AG_LD];
    __shared__ int64_t rowsrc[AG_ROWS];
    constexpr int SQ = AG_LD / 4, SRG = AG_THREADS / SQ;               // 18 column quads x 14 row groups = 252 threads
    const int sq = threadIdx.x % SQ, srg = threadIdx.x / SQ;           // statistics: this thread's column quad and row group
    float shift[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ssum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ssq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (stats && srg < SRG) {      // row 0 of the matrix = the first visible anchor's parameters (the same expf as below)
        const int64_t i0 = idx[0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sc = 4 * sq + j;
            shift[j] = sc < 32 ? p_feat[i0 * AG_FEAT + sc] : sc < 35 ? p_anchor[i0 * 3 + sc - 32]
                     : sc < 65 ? p_offset[i0 * AG_OFF + sc - 35] : sc < AG_COLS ? expf(p_scaling[i0 * 6 + sc - 65]) : 0.0f;
        }
    }
    for (int tk = 0; tk < AG_TPW; ++tk) {
    const int64_t v0 = ((int64_t)blockIdx.x * AG_TPW + tk) * AG_ROWS;
    if (v0 >= V) break;
    if (tk) __syncthreads();      // the previous tile's readers are done with the LDS tile
    const int rows = (int)min((int64_t)AG_ROWS, V - v0);
    if (threadIdx.x < AG_ROWS) rowsrc[threadIdx.x] = threadIdx.x < rows ? idx[v0 + threadIdx.x] : 0;
    __syncthreads();
    // ---- gather: feat rows are 128 B (float4), offset rows 120 B and scaling rows 24 B (float2), anchor rows 12 B.
    // All eight loads of a thread are issued before the first LDS write, from clamped (always valid) rows and without a
    // branch around them: as loops of `load -> LDS write` the compiler closed every iteration with s_waitcnt vmcnt(0),
    // seven memory round trips one after the other per workgroup.
    {
        const int last = rows - 1;
        float4 vf[2];
        float2 vo[4], vs;
        float va;
        int rf[2], qf[2], ro[4], qo[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = threadIdx.x + AG_THREADS * k;          // rows * 8 float4 pieces of the feature rows
            rf[k] = e >> 3;
            qf[k] = e & 7;
            vf[k] = *(const float4*)(p_feat + rowsrc[min(rf[k], last)] * AG_FEAT + 4 * qf[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = threadIdx.x + AG_THREADS * k;          // rows * 15 float2 pieces of the offset rows
            ro[k] = e / 15;
            qo[k] = e - 15 * ro[k];
            vo[k] = *(const float2*)(p_offset + rowsrc[min(ro[k], last)] * AG_OFF + 2 * qo[k]);
        }
        const int ra = threadIdx.x / 3, qa = threadIdx.x - 3 * ra;     // rows * 3: one anchor coordinate + two scalings
        va = p_anchor[rowsrc[min(ra, last)] * 3 + qa];
        vs = *(const float2*)(p_scaling + rowsrc[min(ra, last)] * 6 + 2 * qa);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (rf[k] < rows) *(float4*)&tile[rf[k] * AG_LD + 4 * qf[k]] = vf[k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ro[k] < rows) {
                tile[ro[k] * AG_LD + 35 + 2 * qo[k]] = vo[k].x;
                tile[ro[k] * AG_LD + 36 + 2 * qo[k]] = vo[k].y;
            }
        if (ra < rows) {
            tile[ra * AG_LD + 32 + qa] = va;
            tile[ra * AG_LD + 65 + 2 * qa] = expf(vs.x);        // get_scaling = exp(_scaling)
            tile[ra * AG_LD + 66 + 2 * qa] = expf(vs.y);
            if (qa == 0) tile[ra * AG_LD + AG_COLS] = 0.0f;     // the pad column of 16-byte aligned g_fea rows (ldg = 72)
        }
    }
    __syncthreads();
    // ---- the workgroup's chunks of the five outputs are contiguous
    ag_store_chunk(g_fea + v0 * ldg, tile, rows * ldg, ldg, 0);      // ldg = 71 (packed) or 72 = AG_LD (a straight copy)
    // (feat / offsets NULL: their consumers -- the MLP heads, the expansion -- read the columns of g_fea through a row stride)
    if (feat) ag_store_chunk(feat + v0 * AG_FEAT, tile, rows * AG_FEAT, AG_FEAT, 0);
    ag_store_chunk(anchor + v0 * 3, tile, rows * 3, 3, 32);
    if (offsets) ag_store_chunk(offsets + v0 * AG_OFF, tile, rows * AG_OFF, AG_OFF, 35);
    ag_store_chunk(grid_scaling + v0 * 6, tile, rows * 6, 6, 65);
    if (stats && srg < SRG) {
        for (int r = srg; r < rows; r += SRG) {
            const float4 v = *(const float4*)&tile[r * AG_LD + 4 * sq];      // (the pad column holds 0 and has shift 0)
            const float t[4] = {v.x - shift[0], v.y - shift[1], v.z - shift[2], v.w - shift[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ssum[j] += t[j];
                ssq[j] += t[j] * t[j];
            }
        }
    }
    }
    if (stats) {
        __syncthreads();      // every thread is done with the tile
        float* red = tile;      // [2][SRG][AG_LD]: 8 KB of the (now free) tile
        if (srg < SRG) {
            *(float4*)&red[(0 * SRG + srg) * AG_LD + 4 * sq] = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
            *(float4*)&red[(1 * SRG + srg) * AG_LD + 4 * sq] = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
        }
        __syncthreads();
        if (threadIdx.x < 2 * AG_LD) {
            const int which = threadIdx.x / AG_LD, c = threadIdx.x % AG_LD;
            float a = 0.0f;
#pragma unroll
            for (int g = 0; g < SRG; ++g) a += red[(which * SRG + g) * AG_LD + c];      // fixed order
            stats[((size_t)blockIdx.x * 2 + which) * NL_DP + c] = c < AG_COLS ? a : 0.0f;
        }
    }
}

// A contiguous chunk of `count` upstream floats starting at src (any 4-byte alignment: it starts at the first visible row
// of the workgroup), as 16-byte pieces from the aligned address below src: piece q holds chunk elements 4 q - h .. 4 q - h + 3
// (h = floats between that address and src).  ISSUE and USE are separate: all pieces of all five chunks of a workgroup
// are in flight before the first one is added into the LDS tile -- as `load -> LDS add` loops the compiler closed every
// iteration with s_waitcnt vmcnt(0), ten memory round trips one after the other per workgroup.
template <int NIT>
struct AgPieces {
    float4 v[NIT];
    int h, count;
};
// TAIL: the aligned pieces may reach past the end of the tensor (its last rows only): element-wise, guarded
template <int NIT, bool TAIL>
__device__ __forceinline__ void ag_issue(const float* __restrict__ src, int count, const float* __restrict__ end, AgPieces<NIT>& p) {
    p.count = src ? count : 0;
    p.h = src ? (int)(((uintptr_t)src & 15u) >> 2) : 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k) p.v[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!src) return;                                    // kernel argument: uniform
    const float* base = src - p.h;
    const int n4 = (p.h + count + 3) >> 2;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const float* a = base + 4 * min((int)threadIdx.x + AG_THREADS * k, n4 - 1);      // clamped: no branch around the load
        if (!TAIL) p.v[k] = *(const float4*)a;
        else {
            if (a < end) p.v[k].x = a[0];
            if (a + 1 < end) p.v[k].y = a[1];
            if (a + 2 < end) p.v[k].z = a[2];
            if (a + 3 < end) p.v[k].w = a[3];
        }
    }
}
// adds (STORE: writes) the chunk into the tile at (e / width, col0 + e % width)
template <int NIT, bool STORE>
__device__ __forceinline__ void ag_use(const AgPieces<NIT>& p, float* tile, int width, int col0) {
    if (!p.count) return;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int e0 = 4 * ((int)threadIdx.x + AG_THREADS * k) - p.h;
        const float x[4] = {p.v[k].x, p.v[k].y, p.v[k].z, p.v[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e0 + j;
            if (e >= 0 && e < p.count) {
                const int r = e / width, c = e - r * width;
                if (STORE) tile[r * AG_LD + col0 + c] = x[j];
                else tile[r * AG_LD + col0 + c] += x[j];
            }
        }
    }
}

struct AgUpstream {
    AgPieces<5> gf;          // d g_fea: up to 64 rows x 72 floats = 1152 pieces + 1
    AgPieces<3> feat, off;   // 512 + 1, 480 + 1
    AgPieces<1> anc, gs;     // 48 + 1, 96 + 1
};
template <bool TAIL>
__device__ __forceinline__ void ag_issue_all(AgUpstream& u, int64_t fv, int nv, int64_t V, int ldg, const float* __restrict__ d_g_fea,
                                             const float* __restrict__ d_feat, const float* __restrict__ d_anchor,
                                             const float* __restrict__ d_offsets, const float* __restrict__ d_grid_scaling) {
    ag_issue<5, TAIL>(d_g_fea ? d_g_fea + fv * ldg : nullptr, nv * ldg, d_g_fea + V * ldg, u.gf);
    ag_issue<3, TAIL>(d_feat ? d_feat + fv * AG_FEAT : nullptr, nv * AG_FEAT, d_feat + V * AG_FEAT, u.feat);
    ag_issue<3, TAIL>(d_offsets ? d_offsets + fv * AG_OFF : nullptr, nv * AG_OFF, d_offsets + V * AG_OFF, u.off);
    ag_issue<1, TAIL>(d_anchor ? d_anchor + fv * 3 : nullptr, nv * 3, d_anchor + V * 3, u.anc);
    ag_issue<1, TAIL>(d_grid_scaling ? d_grid_scaling + fv * 6 : nullptr, nv * 6, d_grid_scaling + V * 6, u.gs);
}

// DX (round 6): the gradient of g_fea is not read but FORMED here.  g_fea's only consumer is the BatchNorm-Linear of the
// attribute branch, whose backward ends in dx[v][n] = k0[n] + x[v][n] k1[n] + sum_m dy[v][m] Gi[m][n] (csrc/normlinear.hip,
// nl_bwd_dx_kernel) -- a [V,72] matrix written by one kernel (288 B per row) and read back by this one.  With the three
// coefficient blocks of that backward (AgDx::coef = Gi [32][NL_DP] | k0 [NL_DP] | k1 [NL_DP]), its upstream gradient dy [V,32]
// and its input x = g_fea [V,72] this kernel builds the rows of its workgroup itself: wave w takes the sixteen visible rows
// 16 w .. of the block, 5 column tiles x 8 v_mfma_f32_16x16x4_f32 each (the A operands -- Gi, 10 KB -- come from the CU's L1), and stores them into the LDS tile the other upstream parts are added to.  1.15 GB less
// written and read at configs[2], one kernel less, and the tail of the backward pass of a step runs PER ANCHOR RANGE: the
// exchange of a range's gradients goes on the wire while the next range's dx is still being formed (multiview.GradArena).
struct AgDx {
    const float* coef;      // NULL: no fused dx
    const float* dy;        // [V][lddy], 32 columns, 16-byte aligned rows
    const float* x;         // [V][ldx], AG_COLS columns
    int lddy, ldx;
};
typedef float agf4 __attribute__((ext_vector_type(4)));

template <bool DX, int BPW>
__global__ void __launch_bounds__(AG_THREADS, 4)      // 128 registers: the 37 KB of LDS allow four workgroups per CU      // DX: 168 registers (three workgroups per CU); else 128, four (37 KB of LDS each)
anchor_gather_backward_kernel(int64_t N, int64_t V, const int64_t* __restrict__ inv, const float* __restrict__ grid_scaling,
                              const float* __restrict__ d_feat, const float* __restrict__ d_anchor,
                              const float* __restrict__ d_offsets, const float* __restrict__ d_grid_scaling,
                              const float* __restrict__ d_g_fea, int ldg, float* __restrict__ g_feat,
                              float* __restrict__ g_anchor, float* __restrict__ g_offset,
                              float* __restrict__ g_scaling, int accumulate, AgDx nl) {
    __shared__ __attribute__((aligned(16))) float tile[AG_ROWS * AG_LD];   // rows = the VISIBLE anchors of the block, in order
    __shared__ __attribute__((aligned(16))) float outt[AG_ROWS * AG_LD];   // rows = the block's 64 anchors
    __shared__ int rowv[AG_ROWS];                                          // row of `tile` of every anchor, -1 = invisible
    __shared__ int64_t first_v;
    __shared__ int nvis;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
#pragma unroll 1      // (unrolled, the eight blocks' loads are interleaved and the kernel takes 248 registers: two workgroups per CU)
  for (int bi = 0; bi < BPW; ++bi) {
    const int64_t n0 = ((int64_t)blockIdx.x * BPW + bi) * AG_ROWS;
    if (n0 >= N) break;                                   // workgroup-uniform
    if (bi) __syncthreads();                              // the previous block's reads of tile / outt / rowv are done
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
        AgUpstream u;
        // the aligned 16-byte pieces of a chunk may start up to 12 bytes before it (inside the tensor: a chunk that starts
        // the tensor is aligned) and end up to 12 bytes after it: only the workgroup holding the last visible rows can
        // leave the tensor that way, and it takes the guarded element-wise path
        // (DX: a gradient that did arrive for g_fea -- a second consumer, rare -- is added in a phase of its own below; its
        // twenty registers beside the dx phase's would not fit the 128 of four workgroups per CU)
        const float* gf_now = DX ? nullptr : d_g_fea;
        float gsc = 0.0f, gsc2 = 0.0f;
        auto issue_upstream = [&]() {
            if (fv + nv + 1 < V) ag_issue_all<false>(u, fv, nv, V, ldg, gf_now, d_feat, d_anchor, d_offsets, d_grid_scaling);
            else ag_issue_all<true>(u, fv, nv, V, ldg, gf_now, d_feat, d_anchor, d_offsets, d_grid_scaling);
            gsc = (int)threadIdx.x < nv * 6 ? grid_scaling[fv * 6 + threadIdx.x] : 0.0f;      // exp(s) of the rows, for d exp
            gsc2 = (int)threadIdx.x + AG_THREADS < nv * 6 ? grid_scaling[fv * 6 + threadIdx.x + AG_THREADS] : 0.0f;
        };
        // DX: the upstream pieces are issued BEHIND the dx phase.  Issued in front of it (one memory round trip for both) the
        // two register sets need 158 registers, i.e. three workgroups per CU: 1.22 ms at configs[2] against 1.04 ms this way
        if (!DX) issue_upstream();
        if (DX) {
            // the block's rows of dx, sixteen per wave (at most one round: 4 waves x 16 rows); lane (r16, g4) holds row
            // 16 rb + r16, columns 16 nt + 4 g4 .. + 3 of every tile nt
            for (int rb = wave; 16 * rb < nv; rb += AG_THREADS / 64) {
                const int tr = 16 * rb + r16;
                const bool ok = tr < nv;
                const int64_t row = fv + (ok ? tr : 0);
                const float* dr = nl.dy + (size_t)row * nl.lddy;
                const float* xr = nl.x + (size_t)row * nl.ldx;
                agf4 dq[2], xv[5];
#pragma unroll
                for (int q = 0; q < 2; ++q) dq[q] = ok ? *(const agf4*)(dr + 16 * q + 4 * g4) : agf4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    const int k = 16 * nt + 4 * g4;
                    agf4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (ok) {
                        if (k + 3 < AG_COLS && (nl.ldx & 3) == 0) v = *(const agf4*)(xr + k);
                        else {
                            if (k < AG_COLS) v.x = xr[k];
                            if (k + 1 < AG_COLS) v.y = xr[k + 1];
                            if (k + 2 < AG_COLS) v.z = xr[k + 2];
                        }
                    }
                    xv[nt] = v;
                }
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    const int k = 16 * nt + 4 * g4;
                    const agf4 k0 = *(const agf4*)(nl.coef + 32 * NL_DP + k), k1 = *(const agf4*)(nl.coef + 33 * NL_DP + k);
                    agf4 acc = k0 + xv[nt] * k1;
                    // A[m = column 16 nt + r16][k = dy column 16 q + 4 g4 + j], K-step s = 4 q + j: 10 KB that every wave of the
                    // chip reads -- they come from the CU's L1 (held in registers across a workgroup's blocks they cost 40
                    // of the 128 a four-workgroup CU allows, and the kernel spilled)
                    float ga[8];
#pragma unroll
                    for (int s_ = 0; s_ < 8; ++s_) ga[s_] = nl.coef[(16 * (s_ >> 2) + 4 * g4 + (s_ & 3)) * NL_DP + 16 * nt + r16];
#pragma unroll
                    for (int s_ = 0; s_ < 8; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s_], dq[s_ >> 2][s_ & 3], acc, 0, 0, 0);
                    if (ok && k < AG_LD) *(agf4*)&tile[tr * AG_LD + k] = acc;      // (column 71, the pad: k0 = k1 = Gi = 0 there)
                }
            }
        } else {
            ag_use<5, false>(u.gf, tile, ldg, 0);            // (the pad column lands in tile column 71: unused)
        }
        if (DX) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); issue_upstream(); }
        __syncthreads();   // the parts below add into the same cells
        ag_use<3, false>(u.feat, tile, AG_FEAT, 0);
        ag_use<1, false>(u.anc, tile, 3, 32);
        ag_use<3, false>(u.off, tile, AG_OFF, 35);
        ag_use<1, false>(u.gs, tile, 6, 65);
        if (DX && d_g_fea) {       // kernel-uniform
            __syncthreads();
            if (fv + nv + 1 < V) ag_issue<5, false>(d_g_fea + fv * ldg, nv * ldg, d_g_fea + V * ldg, u.gf);
            else ag_issue<5, true>(d_g_fea + fv * ldg, nv * ldg, d_g_fea + V * ldg, u.gf);
            ag_use<5, false>(u.gf, tile, ldg, 0);
        }
        __syncthreads();
        {   // d exp(s) = exp(s) ds
            const int e = threadIdx.x;
            if (e < nv * 6) tile[(e / 6) * AG_LD + 65 + e % 6] *= gsc;
            const int e2 = e + AG_THREADS;
            if (e2 < nv * 6) tile[(e2 / 6) * AG_LD + 65 + e2 % 6] *= gsc2;
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
    const bool add = accumulate != 0;
    ag_store_chunk(g_feat + n0 * AG_FEAT, outt, rows * AG_FEAT, AG_FEAT, 0, add);
    ag_store_chunk(g_anchor + n0 * 3, outt, rows * 3, 3, 32, add);
    ag_store_chunk(g_offset + n0 * AG_OFF, outt, rows * AG_OFF, AG_OFF, 35, add);
    ag_store_chunk(g_scaling + n0 * 6, outt, rows * 6, 6, 65, add);
  }
}

void launch_anchor_gather(int64_t V, const int64_t* idx, const float* p_feat, const float* p_anchor, const float* p_offset,
                          const float* p_scaling, float* feat, float* anchor, float* offsets, float* grid_scaling,
                          float* g_fea, int ldg, float* stats, hipStream_t st) {
    if (V <= 0) return;
    const int64_t tiles = (V + AG_ROWS * AG_TPW - 1) / (AG_ROWS * AG_TPW);      // workgroups
    // per-workgroup statistics rows: the consumer's rows themselves up to AG_MAX_WGS workgroups, behind them otherwise
    float* tile_stats = stats ? stats + (tiles <= AG_MAX_WGS ? 0 : (size_t)AG_MAX_WGS * 2 * NL_DP) : nullptr;
    anchor_gather_kernel<<<(unsigned)tiles, AG_THREADS, 0, st>>>(
        V, idx, p_feat, p_anchor, p_offset, p_scaling, feat, anchor, offsets, grid_scaling, g_fea, ldg, tile_stats);
    if (stats && tiles > AG_MAX_WGS) anchor_gather_stat_reduce_kernel<<<AG_MAX_WGS, 2 * NL_DP, 0, st>>>(tiles, stats);
}

constexpr int AG_DX_BPW = 1;      // blocks of 64 anchors per workgroup of the dx-forming instantiation

void launch_anchor_gather_backward(int64_t N, int64_t V, const int64_t* inv, const float* grid_scaling, const float* d_feat,
                                   const float* d_anchor, const float* d_offsets, const float* d_grid_scaling,
                                   const float* d_g_fea, int ldg, float* g_feat, float* g_anchor, float* g_offset,
                                   float* g_scaling, int accumulate, const float* nl_coef, const float* nl_dy, int nl_lddy,
                                   const float* nl_x, int nl_ldx, hipStream_t st) {
    if (N <= 0) return;
    const int64_t blocks = (N + AG_ROWS - 1) / AG_ROWS;
    const AgDx nl{nl_coef, nl_dy, nl_x, nl_lddy, nl_ldx};
    if (nl_coef)
        anchor_gather_backward_kernel<true, AG_DX_BPW><<<(unsigned)((blocks + AG_DX_BPW - 1) / AG_DX_BPW), AG_THREADS, 0, st>>>(
            N, V, inv, grid_scaling, d_feat, d_anchor, d_offsets, d_grid_scaling, d_g_fea, ldg, g_feat, g_anchor, g_offset, g_scaling, accumulate, nl);
    else
        anchor_gather_backward_kernel<false, 1><<<(unsigned)blocks, AG_THREADS, 0, st>>>(
            N, V, inv, grid_scaling, d_feat, d_anchor, d_offsets, d_grid_scaling, d_g_fea, ldg, g_feat, g_anchor, g_offset, g_scaling, accumulate, nl);
}

}  // namespace scr
