// splatco_amd/csrc/expand.hip -- fused neural-Gaussian expansion + opacity-mask compaction (gfx950).
//
// Replaces the torch op chain of gaussian_renderer/__init__.py:68-111 (mask = neural_opacity > 0;
// repeat / cat / boolean-index / split; sigmoid, normalize, FMA): ~12 full-size temporaries become
// one streaming pass.  Candidate c = anchor v * k + offset slot; kept candidates keep their order
// (stable compaction), exactly like `concatenated_all[mask]`.
//
//   opacity = neural_opacity[c]                         color_out = color[c]
//   scaling = grid_scaling[v,3:6] * sigmoid(scale_rot[c,0:3])
//   rot     = scale_rot[c,3:7] / max(||.||, 1e-12)      (torch.nn.functional.normalize)
//   xyz     = anchor[v] + offsets[c] * grid_scaling[v,0:3]
//
// HBM-bound: 56 B read per candidate (+36 B per anchor), 56 B written per kept Gaussian.
// Compaction = wave ballot + mbcnt prefix inside a workgroup, workgroup offsets from a scan of
// per-workgroup counts (count pass reads 4 B per candidate).  Backward is one thread per candidate;
// the per-anchor sums over its k candidates go through LDS in slot order -> no atomics, deterministic.
#include "common.h"

namespace scr {

constexpr int EXP_THREADS = 256;
constexpr int EXP_ITEMS = 4;                      // candidates per thread
constexpr int EXP_PER_WG = EXP_THREADS * EXP_ITEMS;  // candidates per workgroup

__device__ __forceinline__ uint32_t lanes_below64(unsigned long long ballot) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

// pass 1: kept candidates per workgroup
__global__ void __launch_bounds__(EXP_THREADS)
expand_count_kernel(int64_t n, const float* __restrict__ neural_opacity, uint32_t* __restrict__ wg_count) {
    __shared__ uint32_t wsum[EXP_THREADS / WAVE];
    uint32_t c = 0;
    float no[EXP_ITEMS];      // all loads first, from clamped indices: `i < n && load` put a branch around every load and
#pragma unroll                // the ballot behind it -- four memory round trips one after the other per workgroup
    for (int r = 0; r < EXP_ITEMS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        no[r] = neural_opacity[min(i, n - 1)];
    }
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        const bool keep = i < n && no[r] > 0.0f;
        c += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(keep));
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;  // every lane of a wave holds the wave's count
    __syncthreads();
    if (threadIdx.x == 0) wg_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// pass 2 (one workgroup): exclusive scan of the workgroup counts, total to *total
__global__ void __launch_bounds__(1024) expand_scan_kernel(uint32_t nwg, uint32_t* __restrict__ wg_count,
                                                           unsigned long long* __restrict__ total,
                                                           volatile unsigned long long* mailbox, unsigned long long seq) {
    __shared__ uint32_t lds[1024 / WAVE];
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < nwg; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nwg ? wg_count[i] : 0u;
        uint32_t inc = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            uint32_t o = __shfl_up(inc, d, WAVE);
            if (lane >= d) inc += o;
        }
        if (lane == 63) lds[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 1024 / WAVE; ++q) {
            const uint32_t s = lds[q];
            if (q < w) wbase += s;
            tot += s;
        }
        if (i < nwg) wg_count[i] = (uint32_t)carry + wbase + inc - v;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry;
        if (mailbox) {  // pinned host words the caller polls (capi.hip): value, then the stamp
            mailbox[0] = carry;
            __threadfence_system();
            mailbox[1] = seq;
        }
    }
}

// ---- mask -> index list (the visible-anchor index of gaussian_renderer/__init__.py:23-29, `t[visible_mask]`): the same
// count / scan / write scheme on a byte mask; replaces torch.nonzero (a 60 GB/s int64 reduction + a select)
__global__ void __launch_bounds__(EXP_THREADS)
mask_count_kernel(int64_t n, const uint8_t* __restrict__ mask, uint32_t* __restrict__ wg_count) {
    __shared__ uint32_t wsum[EXP_THREADS / WAVE];
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        const bool keep = i < n && mask[i] != 0;
        c += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(keep));
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) wg_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void __launch_bounds__(EXP_THREADS)
mask_index_kernel(int64_t n, const uint8_t* __restrict__ mask, const uint32_t* __restrict__ wg_offset,
                  int64_t* __restrict__ index, int64_t* __restrict__ inverse) {
    __shared__ uint32_t wcnt[EXP_ITEMS][EXP_THREADS / WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool keep[EXP_ITEMS];
    uint32_t below[EXP_ITEMS];
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        keep[r] = i < n && mask[i] != 0;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(keep[r]);
        below[r] = lanes_below64(b);
        if (lane == 0) wcnt[r][w] = (uint32_t)__builtin_popcountll(b);
    }
    __syncthreads();
    uint32_t base = wg_offset[blockIdx.x];
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        uint32_t before = 0;
#pragma unroll
        for (int q = 0; q < EXP_THREADS / WAVE; ++q) before += q < w ? wcnt[r][q] : 0u;
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        if (keep[r]) index[base + before + below[r]] = i;
        if (inverse && i < n) inverse[i] = keep[r] ? (int64_t)(base + before + below[r]) : -1ll;
#pragma unroll
        for (int q = 0; q < EXP_THREADS / WAVE; ++q) base += wcnt[r][q];
    }
}

// pass 3: expand + compact
__global__ void __launch_bounds__(EXP_THREADS)
expand_run_kernel(int64_t n, int k, const float* __restrict__ neural_opacity, const float* __restrict__ color,
                  const float* __restrict__ scale_rot, const float* __restrict__ offsets, int ldo,
                  const float* __restrict__ grid_scaling, const float* __restrict__ anchor,
                  const uint32_t* __restrict__ wg_offset, int32_t* __restrict__ out_index,
                  uint8_t* __restrict__ mask_out, float* __restrict__ xyz, float* __restrict__ color_out,
                  float* __restrict__ opacity, float* __restrict__ scaling, float* __restrict__ rot) {
    __shared__ uint32_t wcnt[EXP_ITEMS][EXP_THREADS / WAVE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool keep[EXP_ITEMS];
    uint32_t below[EXP_ITEMS];
    float no[EXP_ITEMS];      // (all loads first, as in the count kernel)
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        no[r] = neural_opacity[min(i, n - 1)];
    }
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        keep[r] = i < n && no[r] > 0.0f;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep[r]);
        below[r] = lanes_below64(bal);
        if (lane == 0) wcnt[r][w] = (uint32_t)__builtin_popcountll(bal);
    }
    __syncthreads();
    uint32_t run = wg_offset[blockIdx.x];
#pragma unroll
    for (int r = 0; r < EXP_ITEMS; ++r) {
        uint32_t base = run;
#pragma unroll
        for (int q = 0; q < EXP_THREADS / WAVE; ++q) {
            const uint32_t c = wcnt[r][q];
            if (q < w) base += c;
            run += c;
        }
        const int64_t i = (int64_t)blockIdx.x * EXP_PER_WG + r * EXP_THREADS + threadIdx.x;
        if (i >= n) continue;
        if (mask_out) mask_out[i] = keep[r];
        if (!keep[r]) {
            out_index[i] = -1;
            continue;
        }
        const size_t p = (size_t)base + below[r];
        out_index[i] = (int32_t)p;
        const int64_t v = i / k;
        const float* sr = scale_rot + 7 * i;
        const float* gs = grid_scaling + 6 * v;
        opacity[p] = no[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            color_out[3 * p + c] = color[3 * i + c];
            scaling[3 * p + c] = gs[3 + c] * (1.0f / (1.0f + __expf(-sr[c])));
            xyz[3 * p + c] = anchor[3 * v + c] + offsets[v * ldo + 3 * (i - v * k) + c] * gs[c];     // offsets rows: ldo floats apart (3 k when packed)
        }
        const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
        const float inv = 1.0f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
        rot[4 * p + 0] = q0 * inv;
        rot[4 * p + 1] = q1 * inv;
        rot[4 * p + 2] = q2 * inv;
        rot[4 * p + 3] = q3 * inv;
    }
}

// backward: one thread per CANDIDATE (coalesced per-candidate reads / writes); the per-anchor sums
// (d grid_scaling, d anchor) of the k candidates of an anchor are added in slot order through LDS
// -> no atomics, deterministic.  A workgroup owns `apw` consecutive anchors (32, fewer when k is
// so large that 32 k partial records would not fit 48 KB of LDS).

__global__ void __launch_bounds__(1024)
expand_backward_kernel(int64_t V, int k, int apw, const float* __restrict__ scale_rot,
                       const float* __restrict__ offsets, int ldo, const float* __restrict__ grid_scaling,
                       const int32_t* __restrict__ out_index,
                       const float* __restrict__ g_xyz, const float* __restrict__ g_color,
                       const float* __restrict__ g_opacity, const float* __restrict__ g_scaling,
                       const float* __restrict__ g_rot, float* __restrict__ d_neural_opacity,
                       float* __restrict__ d_color, float* __restrict__ d_scale_rot, float* __restrict__ d_offsets,
                       float* __restrict__ d_grid_scaling, float* __restrict__ d_anchor,
                       const float* __restrict__ g_reg, float inv_P) {
    extern __shared__ __attribute__((aligned(16))) float part[];  // [apw * k][9]
    const int64_t v0 = (int64_t)blockIdx.x * apw;
    const int nloc = (int)min((int64_t)apw, V - v0) * k;  // candidates of this workgroup
    for (int c = threadIdx.x; c < nloc; c += blockDim.x) {
        const int64_t i = v0 * k + c, v = i / k;
        const int32_t p = out_index[i];
        const float* gs = grid_scaling + 6 * v;
        float dsr[7] = {0, 0, 0, 0, 0, 0, 0}, dof[3] = {0, 0, 0}, dcol[3] = {0, 0, 0}, dop = 0.0f;
        float acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // (d grid_scaling[0..5], d anchor[0..2]) of this candidate
        if (p >= 0) {
            const float* sr = scale_rot + 7 * i;
            dop = g_opacity[p];
            float sg[3], gsc[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                sg[ch] = 1.0f / (1.0f + __expf(-sr[ch]));
                gsc[ch] = g_scaling[3 * (size_t)p + ch];
            }
            if (g_reg) {   // + dL/dreg * d mean(prod(scaling)) / d scaling: the regulariser's gradient never exists as a tensor
                const float a = gs[3] * sg[0], b = gs[4] * sg[1], c = gs[5] * sg[2];     // the forward's scaling, recomputed
                const float w = g_reg[0] * inv_P;
                gsc[0] = gsc[0] + w * (b * c);
                gsc[1] = gsc[1] + w * (a * c);
                gsc[2] = gsc[2] + w * (a * b);
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                dcol[ch] = g_color[3 * (size_t)p + ch];
                const float gx = g_xyz[3 * (size_t)p + ch];
                acc9[6 + ch] = gx;
                dof[ch] = gx * gs[ch];
                acc9[ch] = gx * offsets[v * ldo + 3 * (i - v * k) + ch];
                acc9[3 + ch] = gsc[ch] * sg[ch];
                dsr[ch] = gsc[ch] * gs[3 + ch] * sg[ch] * (1.0f - sg[ch]);
            }
            const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
            const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            const float g0 = g_rot[4 * (size_t)p], g1 = g_rot[4 * (size_t)p + 1], g2 = g_rot[4 * (size_t)p + 2],
                        g3 = g_rot[4 * (size_t)p + 3];
            if (nrm > 1e-12f) {  // y = x/|x| : dx = (g - y (y.g)) / |x|
                const float inv = 1.0f / nrm;
                const float y0 = q0 * inv, y1 = q1 * inv, y2 = q2 * inv, y3 = q3 * inv;
                const float dot = y0 * g0 + y1 * g1 + y2 * g2 + y3 * g3;
                dsr[3] = (g0 - y0 * dot) * inv;
                dsr[4] = (g1 - y1 * dot) * inv;
                dsr[5] = (g2 - y2 * dot) * inv;
                dsr[6] = (g3 - y3 * dot) * inv;
            } else {  // clamped denominator: y = x / 1e-12
                dsr[3] = g0 * 1e12f; dsr[4] = g1 * 1e12f; dsr[5] = g2 * 1e12f; dsr[6] = g3 * 1e12f;
            }
        }
        d_neural_opacity[i] = dop;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            d_color[3 * i + ch] = dcol[ch];
            d_offsets[3 * i + ch] = dof[ch];
        }
#pragma unroll
        for (int ch = 0; ch < 7; ++ch) d_scale_rot[7 * i + ch] = dsr[ch];
#pragma unroll
        for (int ch = 0; ch < 9; ++ch) part[c * 9 + ch] = acc9[ch];
    }
    __syncthreads();
    // anchor sums: thread (anchor a, component ch) adds the k candidates in slot order
    const int na = nloc / k;
    for (int j = threadIdx.x; j < na * 9; j += blockDim.x) {
        const int a_ = j / 9, ch = j % 9;
        float sum = 0.0f;
        for (int sidx = 0; sidx < k; ++sidx) sum += part[(a_ * k + sidx) * 9 + ch];
        if (ch < 6) d_grid_scaling[6 * (v0 + a_) + ch] = sum;
        else d_anchor[3 * (v0 + a_) + ch - 6] = sum;
    }
}

void launch_expand_count(int64_t n, const float* neural_opacity, uint32_t* wg_count, unsigned long long* total,
                         unsigned long long* mailbox, unsigned long long seq, hipStream_t st) {
    const uint32_t nwg = (uint32_t)((n + EXP_PER_WG - 1) / EXP_PER_WG);
    expand_count_kernel<<<nwg, EXP_THREADS, 0, st>>>(n, neural_opacity, wg_count);
    expand_scan_kernel<<<1, 1024, 0, st>>>(nwg, wg_count, total, mailbox, seq);
}

void launch_mask_count(int64_t n, const uint8_t* mask, uint32_t* wg_count, unsigned long long* total,
                       unsigned long long* mailbox, unsigned long long seq, hipStream_t st) {
    const uint32_t nwg = (uint32_t)((n + EXP_PER_WG - 1) / EXP_PER_WG);
    mask_count_kernel<<<nwg, EXP_THREADS, 0, st>>>(n, mask, wg_count);
    expand_scan_kernel<<<1, 1024, 0, st>>>(nwg, wg_count, total, mailbox, seq);
}
void launch_mask_index(int64_t n, const uint8_t* mask, const uint32_t* wg_offset, int64_t* index, int64_t* inverse,
                       hipStream_t st) {
    const uint32_t nwg = (uint32_t)((n + EXP_PER_WG - 1) / EXP_PER_WG);
    mask_index_kernel<<<nwg, EXP_THREADS, 0, st>>>(n, mask, wg_offset, index, inverse);
}

void launch_expand_run(int64_t n, int k, const float* neural_opacity, const float* color, const float* scale_rot,
                       const float* offsets, int ldo, const float* grid_scaling, const float* anchor,
                       const uint32_t* wg_offset, int32_t* out_index, uint8_t* mask_out, float* xyz,
                       float* color_out, float* opacity, float* scaling, float* rot, hipStream_t st) {
    const uint32_t nwg = (uint32_t)((n + EXP_PER_WG - 1) / EXP_PER_WG);
    expand_run_kernel<<<nwg, EXP_THREADS, 0, st>>>(n, k, neural_opacity, color, scale_rot, offsets, ldo, grid_scaling,
                                                   anchor, wg_offset, out_index, mask_out, xyz, color_out, opacity,
                                                   scaling, rot);
}

void launch_expand_backward(int64_t V, int k, const float* scale_rot, const float* offsets, int ldo,
                            const float* grid_scaling, const int32_t* out_index, const float* g_xyz,
                            const float* g_color, const float* g_opacity, const float* g_scaling,
                            const float* g_rot, float* d_neural_opacity, float* d_color, float* d_scale_rot,
                            float* d_offsets, float* d_grid_scaling, float* d_anchor, const float* g_reg, int64_t P,
                            hipStream_t st) {
    const int apw = max(1, min(32, (48 * 1024) / (k * 9 * (int)sizeof(float))));
    const int threads = min(1024, ((apw * k + 63) / 64) * 64);
    expand_backward_kernel<<<(unsigned)((V + apw - 1) / apw), threads, (size_t)apw * k * 9 * sizeof(float), st>>>(
        V, k, apw, scale_rot, offsets, ldo, grid_scaling, out_index, g_xyz, g_color, g_opacity, g_scaling, g_rot,
        d_neural_opacity, d_color, d_scale_rot, d_offsets, d_grid_scaling, d_anchor, g_reg,
        P > 0 ? (float)(1.0 / (double)P) : 0.0f);
}

}  // namespace scr
