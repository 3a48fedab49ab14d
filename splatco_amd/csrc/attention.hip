// splatco_amd/csrc/attention.hip -- TriPlaneAttention of the level-0 grid (scene/grids.py:22-64: ChannelAttention, then
// SpatialAttention, applied to the three planes stacked along the channels on EVERY call, scene/grids.py:166-168) as a
// handful of streaming passes for gfx950 instead of ~60 framework kernels per direction (two global pools, four 1x1
// "convolutions" on 15 numbers, broadcasts, a channel mean / max / cat, a 7x7 convolution through MIOpen with its layout
// changes, sigmoids, products, chunks and the cats that stack a plane on its attended twin) -- and without MIOpen, whose
// first process on a machine runs the 7x7 convolution on a fallback solver (measured: 34 instead of 27 ms per step).
//
//   x[c] (c = 3 planes x R channels, H x W each, read in place from the three parameters)
//   stats   : avg[c], max[c] (+ the pixel of the max) over the pixels                      (tpa_stats_*)
//   (host)  : ca = sigmoid(MLP(avg) + MLP(max))        -- 15 numbers, stays in the framework
//   reduce  : y[c] = ca[c] x[c];  s0 = mean_c y, s1 = max_c y (+ its channel)              (tpa_reduce_kernel)
//   apply   : sa = sigmoid(conv7x7(s, w));  out pair plane j = [ x of plane j | sa * y of plane j ]   (tpa_apply_kernel)
// The pair planes [2R, H, W] are what the tri-plane sampler wants (a plane stacked on its attended twin), so the
// framework's cats are gone too.  Backward (tpa_bwd_*): dL/dsa -> through the sigmoid -> transposed 7x7 -> mean / max
// routing -> dL/dy -> dL/dx and the per-channel sums dL/dca; the weight gradient of the 7x7 is a per-tile correlation.
// Every reduction is two-level with a fixed order (per-block partials, summed in block order): bit-reproducible.
#include "common.h"

namespace scr {

// up to 24 stacked channels (3 R with R <= 8; the reference: 15)
constexpr int TPA_TW = 64, TPA_TH = 16, TPA_HALO = 3, TPA_K = 7;   // pixel tile of the convolution passes, 7x7 window
constexpr int TPA_STAT_BLOCKS = 64;  // blocks per channel of the pooling pass

struct TpaPlanes { const float* p[3]; };
struct TpaOut { float* p[3]; };

__device__ __forceinline__ float block_sum_256(float v, float* lds) {   // lds: 4 floats; all threads get the sum
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// ---------------------------------------------------------------- pools over the pixels (ChannelAttention, :28-35)
__global__ void __launch_bounds__(256)
tpa_stats_partial_kernel(int R, int64_t HW, TpaPlanes x, float* __restrict__ psum, float* __restrict__ pmax,
                         int* __restrict__ parg) {
    const int c = blockIdx.y, b = blockIdx.x;
    const float* xc = x.p[c / R] + (int64_t)(c % R) * HW;
    const int64_t per = (HW + gridDim.x - 1) / gridDim.x, lo = b * per, hi = min(HW, lo + per);
    float s = 0.0f, m = -INFINITY;
    int64_t am = lo;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = xc[i];
        s += v;
        if (v > m) { m = v; am = i; }
    }
    __shared__ float ls[256], lm[256];
    __shared__ int la[256];
    ls[threadIdx.x] = s; lm[threadIdx.x] = m; la[threadIdx.x] = (int)am;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            ls[threadIdx.x] += ls[threadIdx.x + d];
            const float o = lm[threadIdx.x + d];
            const int oa = la[threadIdx.x + d];
            if (o > lm[threadIdx.x] || (o == lm[threadIdx.x] && oa < la[threadIdx.x])) { lm[threadIdx.x] = o; la[threadIdx.x] = oa; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        psum[c * gridDim.x + b] = ls[0];
        pmax[c * gridDim.x + b] = lm[0];
        parg[c * gridDim.x + b] = la[0];
    }
}
__global__ void tpa_stats_finish_kernel(int C, int nb, int64_t HW, const float* __restrict__ psum,
                                        const float* __restrict__ pmax, const int* __restrict__ parg,
                                        float* __restrict__ avg, float* __restrict__ mx, int* __restrict__ arg) {
    const int c = threadIdx.x;
    if (c >= C) return;
    float s = 0.0f, m = -INFINITY;
    int a = 0;
    for (int b = 0; b < nb; ++b) {
        s += psum[c * nb + b];
        const float o = pmax[c * nb + b];
        if (o > m) { m = o; a = parg[c * nb + b]; }   // blocks in pixel order: the first maximum wins
    }
    avg[c] = s / (float)HW;
    mx[c] = m;
    arg[c] = a;
}

// ---------------------------------------------------------------- channel mean / max of y = ca * x (SpatialAttention, :46-49)
__global__ void __launch_bounds__(256)
tpa_reduce_kernel(int R, int64_t HW, TpaPlanes x, const float* __restrict__ ca, float* __restrict__ s,
                  uint8_t* __restrict__ am) {
    const int C = 3 * R;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    float sum = 0.0f, m = -INFINITY;
    int a = 0;
    for (int c = 0; c < C; ++c) {
        const float y = ca[c] * x.p[c / R][(int64_t)(c % R) * HW + i];
        sum += y;
        if (y > m) { m = y; a = c; }
    }
    s[i] = sum / (float)C;
    s[HW + i] = m;
    am[i] = (uint8_t)a;
}

// the 2-channel map with its 3-pixel halo in LDS (zero outside the plane: the convolution's padding)
__device__ __forceinline__ void tpa_load_tile(int H, int W, int i0, int j0, const float* __restrict__ src, int nch,
                                              float* __restrict__ tile) {
    constexpr int LW = TPA_TW + 2 * TPA_HALO, LH = TPA_TH + 2 * TPA_HALO;
    for (int t = threadIdx.x; t < nch * LH * LW; t += 256) {
        const int ch = t / (LH * LW), r = (t / LW) % LH, q = t % LW;
        const int i = i0 + r - TPA_HALO, j = j0 + q - TPA_HALO;
        tile[t] = (i >= 0 && i < H && j >= 0 && j < W) ? src[(int64_t)ch * H * W + (int64_t)i * W + j] : 0.0f;
    }
}

// ---------------------------------------------------------------- sa = sigmoid(conv7x7(s)), pair planes written
__global__ void __launch_bounds__(256)
tpa_apply_kernel(int R, int H, int W, TpaPlanes x, const float* __restrict__ ca, const float* __restrict__ w,
                 const float* __restrict__ s, float* __restrict__ sa, TpaOut out) {
    constexpr int LW = TPA_TW + 2 * TPA_HALO, LH = TPA_TH + 2 * TPA_HALO;
    __shared__ float tile[2 * LH * LW];
    __shared__ float wk[2 * TPA_K * TPA_K];
    const int j0 = blockIdx.x * TPA_TW, i0 = blockIdx.y * TPA_TH;
    if (threadIdx.x < 2 * TPA_K * TPA_K) wk[threadIdx.x] = w[threadIdx.x];
    tpa_load_tile(H, W, i0, j0, s, 2, tile);
    __syncthreads();
    const int C = 3 * R;
    const int64_t HW = (int64_t)H * W;
    const int tj = threadIdx.x & (TPA_TW - 1), ti0 = threadIdx.x / TPA_TW;   // 4 rows of threads, 4 pixels each
    for (int ti = ti0; ti < TPA_TH; ti += 256 / TPA_TW) {
        const int i = i0 + ti, j = j0 + tj;
        if (i >= H || j >= W) continue;
        float z = 0.0f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int a = 0; a < TPA_K; ++a)
#pragma unroll
                for (int b = 0; b < TPA_K; ++b)
                    z = __builtin_fmaf(wk[(ch * TPA_K + a) * TPA_K + b], tile[(ch * LH + ti + a) * LW + tj + b], z);
        const float g = 1.0f / (1.0f + __expf(-z));
        const int64_t pix = (int64_t)i * W + j;
        sa[pix] = g;
        for (int c = 0; c < C; ++c) {
            const int pl = c / R, r = c % R;
            const float v = x.p[pl][(int64_t)r * HW + pix];
            out.p[pl][(int64_t)r * HW + pix] = v;                       // the plane itself
            out.p[pl][(int64_t)(R + r) * HW + pix] = g * (ca[c] * v);   // its attended twin
        }
    }
}

// ---------------------------------------------------------------- backward 1: dL/d(conv output)
// g[j] = gradient of pair plane j: channels 0..R-1 go straight to the plane, R..2R-1 to the attended twin
__global__ void __launch_bounds__(256)
tpa_bwd_pre_kernel(int R, int64_t HW, TpaPlanes x, const float* __restrict__ ca, const float* __restrict__ sa,
                   TpaPlanes g, float* __restrict__ dpre) {
    const int C = 3 * R;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    float dsa = 0.0f;
    for (int c = 0; c < C; ++c) {
        const int pl = c / R, r = c % R;
        dsa = __builtin_fmaf(g.p[pl][(int64_t)(R + r) * HW + i], ca[c] * x.p[pl][(int64_t)r * HW + i], dsa);
    }
    const float v = sa[i];
    dpre[i] = dsa * v * (1.0f - v);
}

// ---------------------------------------------------------------- backward 2: transposed 7x7, routing, dL/dx, dL/dca partials
template <int R>
__global__ void __launch_bounds__(256)
tpa_bwd_apply_kernel(int H, int W, TpaPlanes x, const float* __restrict__ ca, const float* __restrict__ w,
                     const float* __restrict__ sa, const uint8_t* __restrict__ am, const float* __restrict__ dpre,
                     TpaPlanes g, TpaOut dx, float* __restrict__ dca_part) {
    constexpr int LW = TPA_TW + 2 * TPA_HALO, LH = TPA_TH + 2 * TPA_HALO;
    __shared__ float tile[LH * LW];
    __shared__ float wk[2 * TPA_K * TPA_K];
    __shared__ float red[4];
    const int j0 = blockIdx.x * TPA_TW, i0 = blockIdx.y * TPA_TH;
    if (threadIdx.x < 2 * TPA_K * TPA_K) wk[threadIdx.x] = w[threadIdx.x];
    tpa_load_tile(H, W, i0, j0, dpre, 1, tile);
    __syncthreads();
    constexpr int C = 3 * R;
    const int64_t HW = (int64_t)H * W;
    const int tj = threadIdx.x & (TPA_TW - 1), ti0 = threadIdx.x / TPA_TW;
    float dca[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dca[c] = 0.0f;
    for (int ti = ti0; ti < TPA_TH; ti += 256 / TPA_TW) {
        const int i = i0 + ti, j = j0 + tj;
        if (i >= H || j >= W) continue;
        // ds[ch](i, j) = sum_ab w[ch][a][b] dpre(i - (a-3), j - (b-3))
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int a = 0; a < TPA_K; ++a)
#pragma unroll
            for (int b = 0; b < TPA_K; ++b) {
                const float v = tile[(ti + 2 * TPA_HALO - a) * LW + tj + 2 * TPA_HALO - b];
                d0 = __builtin_fmaf(wk[a * TPA_K + b], v, d0);
                d1 = __builtin_fmaf(wk[(TPA_K + a) * TPA_K + b], v, d1);
            }
        const int64_t pix = (int64_t)i * W + j;
        const float sv = sa[pix], dmean = d0 / (float)C;
        const int amax = am[pix];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int pl = c / R, r = c % R;
            const float xv = x.p[pl][(int64_t)r * HW + pix];
            const float dy = __builtin_fmaf(g.p[pl][(int64_t)(R + r) * HW + pix], sv, dmean) + (c == amax ? d1 : 0.0f);
            dx.p[pl][(int64_t)r * HW + pix] = __builtin_fmaf(dy, ca[c], g.p[pl][(int64_t)r * HW + pix]);
            dca[c] = __builtin_fmaf(dy, xv, dca[c]);
        }
    }
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float t = block_sum_256(dca[c], red);
        if (threadIdx.x == 0) dca_part[(int64_t)blk * C + c] = t;
    }
}

// ---------------------------------------------------------------- backward 3: weight gradient of the 7x7
// dw[ch][a][b] = sum_p dpre(p) s[ch](p + (a-3, b-3)): per tile, one thread per (tap, half of the tile's rows)
__global__ void __launch_bounds__(256)
tpa_bwd_weight_kernel(int H, int W, const float* __restrict__ s, const float* __restrict__ dpre, float* __restrict__ dw_part) {
    constexpr int LW = TPA_TW + 2 * TPA_HALO, LH = TPA_TH + 2 * TPA_HALO, NT = 2 * TPA_K * TPA_K;
    __shared__ float tile[2 * LH * LW];
    __shared__ float dp[TPA_TH * TPA_TW];
    __shared__ float half1[NT];
    const int j0 = blockIdx.x * TPA_TW, i0 = blockIdx.y * TPA_TH;
    tpa_load_tile(H, W, i0, j0, s, 2, tile);
    for (int t = threadIdx.x; t < TPA_TH * TPA_TW; t += 256) {
        const int i = i0 + t / TPA_TW, j = j0 + t % TPA_TW;
        dp[t] = (i < H && j < W) ? dpre[(int64_t)i * W + j] : 0.0f;
    }
    __syncthreads();
    float acc = 0.0f;
    const int tap = threadIdx.x % NT, half = threadIdx.x / NT;   // threads 196..255 idle
    if (half < 2) {
        const int ch = tap / (TPA_K * TPA_K), a = (tap / TPA_K) % TPA_K, b = tap % TPA_K;
        for (int ti = half * (TPA_TH / 2); ti < (half + 1) * (TPA_TH / 2); ++ti)
            for (int tj = 0; tj < TPA_TW; ++tj)
                acc = __builtin_fmaf(dp[ti * TPA_TW + tj], tile[(ch * LH + ti + a) * LW + tj + b], acc);
    }
    if (half == 1) half1[tap] = acc;
    __syncthreads();
    if (half == 0) dw_part[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * NT + tap] = acc + half1[tap];
}

// partial sums [nblk][n] -> out[n], in block order
__global__ void __launch_bounds__(256)
tpa_sum_partials_kernel(int nblk, int n, const float* __restrict__ part, float* __restrict__ out) {
    const int k = blockIdx.x;   // one block per output
    __shared__ float red[4];
    float s = 0.0f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[(int64_t)b * n + k];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[k] = s;
}

// ---------------------------------------------------------------- backward 4: the pools' gradients into dL/dx
// dx[c](p) += davg[c] / HW for every pixel, dx[c](arg[c]) += dmax[c]
__global__ void __launch_bounds__(256)
tpa_bwd_stats_kernel(int R, int64_t HW, const float* __restrict__ davg, const float* __restrict__ dmx,
                     const int* __restrict__ arg, TpaOut dx) {
    const int c = blockIdx.y;
    float* d = dx.p[c / R] + (int64_t)(c % R) * HW;
    const float add = davg[c] / (float)HW, m = dmx[c];
    const int64_t a = arg[c];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256)
        d[i] += add + (i == a ? m : 0.0f);
}

// ---------------------------------------------------------------- launchers
static dim3 tpa_tiles(int H, int W) { return dim3((W + TPA_TW - 1) / TPA_TW, (H + TPA_TH - 1) / TPA_TH); }

size_t tpa_scratch_bytes(int R, int H, int W) {
    const dim3 t = tpa_tiles(H, W);
    const size_t nblk = (size_t)t.x * t.y;
    const size_t stats = (size_t)3 * R * TPA_STAT_BLOCKS * 12;
    const size_t bwd = nblk * (size_t)(3 * R + 2 * TPA_K * TPA_K) * 4 + (size_t)H * W * 4;   // partials + dpre
    return align_up(stats > bwd ? stats : bwd);
}

void launch_tpa_stats(int R, int64_t HW, const float* p0, const float* p1, const float* p2, float* avg, float* mx,
                      int* arg, void* scratch, hipStream_t st) {
    const int C = 3 * R;
    float* psum = (float*)scratch;
    float* pmax = psum + (size_t)C * TPA_STAT_BLOCKS;
    int* parg = (int*)(pmax + (size_t)C * TPA_STAT_BLOCKS);
    const TpaPlanes x{{p0, p1, p2}};
    tpa_stats_partial_kernel<<<dim3(TPA_STAT_BLOCKS, C), 256, 0, st>>>(R, HW, x, psum, pmax, parg);
    tpa_stats_finish_kernel<<<1, 64, 0, st>>>(C, TPA_STAT_BLOCKS, HW, psum, pmax, parg, avg, mx, arg);
}

void launch_tpa_forward(int R, int H, int W, const float* p0, const float* p1, const float* p2, const float* ca,
                        const float* w, float* s, uint8_t* am, float* sa, float* o0, float* o1, float* o2, hipStream_t st) {
    const int64_t HW = (int64_t)H * W;
    const TpaPlanes x{{p0, p1, p2}};
    tpa_reduce_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, st>>>(R, HW, x, ca, s, am);
    tpa_apply_kernel<<<tpa_tiles(H, W), 256, 0, st>>>(R, H, W, x, ca, w, s, sa, TpaOut{{o0, o1, o2}});
}

void launch_tpa_backward(int R, int H, int W, const float* p0, const float* p1, const float* p2, const float* ca,
                         const float* w, const float* s, const uint8_t* am, const float* sa, const float* g0,
                         const float* g1, const float* g2, float* d0, float* d1, float* d2, float* dca, float* dw,
                         void* scratch, hipStream_t st) {
    const int64_t HW = (int64_t)H * W;
    const int C = 3 * R, NT = 2 * TPA_K * TPA_K;
    const dim3 t = tpa_tiles(H, W);
    const int nblk = (int)(t.x * t.y);
    float* dpre = (float*)scratch;
    float* dca_part = dpre + HW;
    float* dw_part = dca_part + (size_t)nblk * C;
    const TpaPlanes x{{p0, p1, p2}}, g{{g0, g1, g2}};
    tpa_bwd_pre_kernel<<<(unsigned)((HW + 255) / 256), 256, 0, st>>>(R, HW, x, ca, sa, g, dpre);
    const TpaOut dx{{d0, d1, d2}};
    switch (R) {   // the per-channel sums live in registers: the channel count is a template parameter
#define SCR_TPA_CASE(r) case r: tpa_bwd_apply_kernel<r><<<t, 256, 0, st>>>(H, W, x, ca, w, sa, am, dpre, g, dx, dca_part); break;
        SCR_TPA_CASE(1) SCR_TPA_CASE(2) SCR_TPA_CASE(3) SCR_TPA_CASE(4) SCR_TPA_CASE(5) SCR_TPA_CASE(6) SCR_TPA_CASE(7) SCR_TPA_CASE(8)
#undef SCR_TPA_CASE
    }
    tpa_bwd_weight_kernel<<<t, 256, 0, st>>>(H, W, s, dpre, dw_part);
    tpa_sum_partials_kernel<<<C, 256, 0, st>>>(nblk, C, dca_part, dca);
    tpa_sum_partials_kernel<<<NT, 256, 0, st>>>(nblk, NT, dw_part, dw);
}

void launch_tpa_backward_stats(int R, int64_t HW, const float* davg, const float* dmx, const int* arg, float* d0,
                               float* d1, float* d2, hipStream_t st) {
    tpa_bwd_stats_kernel<<<dim3(64, 3 * R), 256, 0, st>>>(R, HW, davg, dmx, arg, TpaOut{{d0, d1, d2}});
}

}  // namespace scr
