// splatco_amd/csrc/normlinear.hip -- BatchNorm1d (training mode) folded into the Linear that follows it (gfx950).
//
// FeaturePlanes (scene/gaussian_model.py:97-169) sends the tri-plane samples [V,60] and the anchor attributes
// [V,71] of every visible anchor through nn.Sequential(BatchNorm1d(d), Linear(d, 32)) once per active level and sums
// the results.  With batch statistics the chain is linear in the normalised input, so all levels fold into ONE
// matrix G [32,d] and one bias c (scene_model._norm_linear does the folding):
//
//     y = xhat G^T + c,    xhat = (x - mean(x)) / sqrt(var(x) + eps)      (column statistics over the V rows)
//
// x has millions of rows and d <= 80 columns: every op is a pass over [V,d] at HBM speed or it is wasted time.  torch
// needs var_mean + addmm (forward) and sum + a slab-split bmm + addmm + addcmul (backward): 8 ms of the 33 ms
// configs[2] step.  Here:
//   forward : nl_stats (one pass: shifted sums per column, fp64 combination) -> nl_forward (one pass: f32 MFMA with
//             the column-scaled weights held in registers)
//   backward: nl_bwd_reduce (one pass over dy and x: dy^T x and the column sums of dy on the MFMA, per-workgroup
//             partials summed in a fixed order) -> nl_bwd_finish (32 x d algebra: weight gradient and both BatchNorm
//             reduction terms) -> nl_bwd_dx (one pass: dx = k0 + dy (G inv) + x k1)
// MFMA v_mfma_f32_16x16x4_f32 throughout (fp32 FMA chains: no reduced precision anywhere); operand layout
// A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15], D[row = 4 (lane >> 4) + reg][col = lane & 15].
// Rows ("anchors") lie along N, features along M, so a lane's four accumulator registers are four consecutive
// features of one row: 16-byte stores.  A lane loads 16 bytes of a row (columns 16 q + 4 g .. + 3 for g = lane >> 4)
// and feeds them to four MFMA steps; the contraction index is permuted accordingly on the weight side.
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace scr {

typedef float nlf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ nlf4 nl_mfma(float a, float b, nlf4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int NL_OUT = 32;              // output features of the Linear (FeaturePlanes: out_dim = 32)
// (NL_DP = 80, common.h: columns of x supported = padded width of every coefficient row)
constexpr int NL_SLAB = 2048;           // rows per statistics workgroup, at least (sizes the scratch; see nl_stat_slab)
constexpr int NL_HSIZE = NL_OUT * NL_DP + NL_OUT;   // dy^T x (padded) + column sums of dy

// coefficient block in scratch (floats)
constexpr int NLC_GS = 0;                           // [32][80]  G * inv               (forward weights)
constexpr int NLC_CB = NLC_GS + NL_OUT * NL_DP;     // [32]      c - Gs mean           (forward bias)
constexpr int NLC_GI = NLC_CB + NL_OUT;             // [32][80]  G * inv               (backward)
constexpr int NLC_K0 = NLC_GI + NL_OUT * NL_DP;     // [80]
constexpr int NLC_K1 = NLC_K0 + NL_DP;              // [80]
constexpr int NLC_HRAW = NLC_K1 + NL_DP;            // [NL_HSIZE]
constexpr int NLC_END = NLC_HRAW + NL_HSIZE;

static inline int nl_stat_wgs(int64_t V) { return (int)((V + NL_SLAB - 1) / NL_SLAB); }
static inline int nl_bwd_wgs(int64_t V) {
    const int64_t blocks = (V + 63) / 64;           // a wave takes 16 rows per step, a workgroup 64
    return (int)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024);
}

size_t norm_linear_scratch_bytes(int64_t V) {
    return align_up((size_t)NLC_END * 4) + align_up((size_t)nl_stat_wgs(V) * 2 * NL_DP * 4) +
           align_up((size_t)nl_bwd_wgs(V) * NL_HSIZE * 4);
}

// ---- column statistics, pass 1: per-workgroup sums of (x - shift) and (x - shift)^2, shift = the first row
template <int CW>   // lanes per row: 64 (d <= 64) or 128
__global__ void __launch_bounds__(256)
nl_stats_partial_kernel(int64_t V, int d, int slab, const float* __restrict__ x, int ldx, float* __restrict__ partial) {
    constexpr int RG = 256 / CW;
    __shared__ float red[2][256];
    const int c = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int64_t r0 = (int64_t)blockIdx.x * slab;
    const int rows = (int)min((int64_t)slab, V - r0);
    float s = 0.0f, q = 0.0f;
    if (c < d) {
        const float shift = x[c];
        const float* p = x + (size_t)r0 * ldx + c;
        int r = rg;
        for (; r + 7 * RG < rows; r += 8 * RG) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u * RG) * ldx];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float t = v[u] - shift;
                s += t;
                q += t * t;
            }
        }
        for (; r < rows; r += RG) {
            const float t = p[(size_t)r * ldx] - shift;
            s += t;
            q += t * t;
        }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < CW && threadIdx.x < d) {
        float ss = 0.0f, qq = 0.0f;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            ss += red[0][g * CW + threadIdx.x];
            qq += red[1][g * CW + threadIdx.x];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * NL_DP + threadIdx.x] = ss;
        partial[((size_t)blockIdx.x * 2 + 1) * NL_DP + threadIdx.x] = qq;
    }
}

// ---- pass 2 (one workgroup): fp64 combination -> mean, biased variance, inv = rsqrt(var + eps); forward coefficients
__global__ void __launch_bounds__(1024)
nl_stats_finish_kernel(int64_t V, int d, int nwg, const float* __restrict__ x, const float* __restrict__ partial,
                       const float* __restrict__ G, const float* __restrict__ c, float eps, float* __restrict__ mean,
                       float* __restrict__ var, float* __restrict__ inv, float* __restrict__ coef) {
    __shared__ double rs[8][128], rq[8][128];
    __shared__ float s_mean[NL_DP], s_inv[NL_DP];
    const int col = threadIdx.x & 127, grp = threadIdx.x >> 7;
    double s = 0.0, q = 0.0;
    if (col < d) {
        int w = grp;
        for (; w + 56 < nwg; w += 64) {          // eight partials in flight (the loop is pure load latency)
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = partial[((size_t)(w + 8 * u) * 2 + 0) * NL_DP + col];
                b[u] = partial[((size_t)(w + 8 * u) * 2 + 1) * NL_DP + col];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s += (double)a[u];
                q += (double)b[u];
            }
        }
        for (; w < nwg; w += 8) {
            s += (double)partial[((size_t)w * 2 + 0) * NL_DP + col];
            q += (double)partial[((size_t)w * 2 + 1) * NL_DP + col];
        }
    }
    rs[grp][col] = s;
    rq[grp][col] = q;
    __syncthreads();
    if (threadIdx.x < d) {
        double ss = 0.0, qq = 0.0;
        for (int g = 0; g < 8; ++g) {
            ss += rs[g][threadIdx.x];
            qq += rq[g][threadIdx.x];
        }
        const double m = ss / (double)V;
        double v = qq / (double)V - m * m;
        if (v < 0.0) v = 0.0;
        const float mf = (float)((double)x[threadIdx.x] + m), vf = (float)v;
        const float iv = 1.0f / sqrtf(vf + eps);
        mean[threadIdx.x] = mf;
        var[threadIdx.x] = vf;
        inv[threadIdx.x] = iv;
        s_mean[threadIdx.x] = mf;
        s_inv[threadIdx.x] = iv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NL_OUT * NL_DP; i += 1024) {
        const int m = i / NL_DP, n = i % NL_DP;
        coef[NLC_GS + i] = n < d ? G[m * d + n] * s_inv[n] : 0.0f;
    }
    if (threadIdx.x < NL_OUT) {
        float acc = 0.0f;
        for (int n = 0; n < d; ++n) acc += G[threadIdx.x * d + n] * s_inv[n] * s_mean[n];
        coef[NLC_CB + threadIdx.x] = c[threadIdx.x] - acc;
    }
}

// 16 bytes of a row (or four dwords when rows are not 16-byte aligned), zero beyond column d.  The aligned load is
// UNCONDITIONAL -- the address is clamped into the row (the caller clamps the row into the matrix) and the components
// beyond d are zeroed with selects: a load under `if (k + 3 < d)` makes the compiler close every load with its own
// s_waitcnt vmcnt(0) at the join, i.e. the Q loads of a row are served one memory round trip after the other.
template <bool ALIGNED>
__device__ __forceinline__ nlf4 nl_raw4(const float* __restrict__ row, int k, int d, int ld) {     // the loads alone
    nlf4 v;
    if (ALIGNED) {
        v = *(const nlf4*)(row + min(k, ld - 4));      // k > ld - 4 means k >= ld >= d (both multiples of 4): all masked by nl_mask4
    } else {
        v.x = row[min(k, d - 1)];
        v.y = row[min(k + 1, d - 1)];
        v.z = row[min(k + 2, d - 1)];
        v.w = row[min(k + 3, d - 1)];
    }
    return v;
}
__device__ __forceinline__ nlf4 nl_mask4(nlf4 v, int k, int d) {      // columns past d read as zero
    v.x = k < d ? v.x : 0.0f;
    v.y = k + 1 < d ? v.y : 0.0f;
    v.z = k + 2 < d ? v.z : 0.0f;
    v.w = k + 3 < d ? v.w : 0.0f;
    return v;
}

// the same under its conditions (zero beyond column d / row V): what nl_bwd_dx_kernel uses -- there the unconditional
// form was measured SLOWER (d = 60: 0.71 against 0.60 ms at 4.6 M rows; same-box A/B), the forward kernel gains 20 %
template <bool ALIGNED>
__device__ __forceinline__ nlf4 nl_load4_if(const float* __restrict__ row, int k, int d, bool ok) {
    nlf4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!ok) return v;
    if (ALIGNED && k + 3 < d) return *(const nlf4*)(row + k);
    if (k < d) v.x = row[k];
    if (k + 1 < d) v.y = row[k + 1];
    if (k + 2 < d) v.z = row[k + 2];
    if (k + 3 < d) v.w = row[k + 3];
    return v;
}

// ---- forward: y[v][m] = sum_k x[v][k] Gs[m][k] + cb[m]
template <int Q, bool ALIGNED>   // Q = ceil(d / 16) sixteen-column groups
__global__ void __launch_bounds__(256)
nl_forward_kernel(int64_t V, int d, const float* __restrict__ x, int ldx, const float* __restrict__ coef,
                  float* __restrict__ y) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float a[2][4 * Q];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4 * Q; ++s) a[t][s] = coef[NLC_GS + (16 * t + r) * NL_DP + 16 * (s >> 2) + 4 * g + (s & 3)];
    nlf4 cb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) cb[t] = *(const nlf4*)(coef + NLC_CB + 16 * t + 4 * g);
    const int64_t blocks = (V + 15) / 16, stride = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < blocks; blk += stride) {
        const int64_t row = blk * 16 + r;
        const bool ok = row < V;
        const float* xr = x + (size_t)(ok ? row : V - 1) * ldx;       // rows past the end recompute the last row, nothing is stored
        nlf4 xq[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) xq[q] = nl_raw4<ALIGNED>(xr, 16 * q + 4 * g, d, ldx);
        // all loads of the step are issued before anything uses one: left alone, the scheduler sank three of the five
        // behind the MFMAs that use the first two and closed each with s_waitcnt vmcnt(0) -- four memory round trips
        // per step -- and with the column masks next to the loads it still waited for the first before issuing the third
        // (forward at cfg2 1.30 -> 1.22 ms).  The dx kernel below keeps its branchy loads: there the same treatment
        // (seven loads, one wait) was measured slower twice (rounds 2 and 3: 2.28 -> 2.40 ms per step at cfg2)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < Q; ++q) xq[q] = nl_mask4(xq[q], 16 * q + 4 * g, d);
        nlf4 acc[2] = {cb[0], cb[1]};
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = nl_mfma(a[t][4 * q + j], xq[q][j], acc[t]);
        if (ok) {
#pragma unroll
            for (int t = 0; t < 2; ++t) *(nlf4*)(y + (size_t)row * NL_OUT + 16 * t + 4 * g) = acc[t];
        }
    }
}

// ---- backward pass 1: Hraw[m][n] = sum_v dy[v][m] x[v][n], sdy[m] = sum_v dy[v][m]; per-workgroup partials
template <int NT>   // NT = ceil(d / 16) column tiles of x
__global__ void __launch_bounds__(256)
nl_bwd_reduce_kernel(int64_t V, int d, const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                     float* __restrict__ partial) {
    __shared__ float slot[4][NL_HSIZE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    nlf4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = nlf4{0.0f, 0.0f, 0.0f, 0.0f};
    float sdy[2] = {0.0f, 0.0f};
    bool colok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) colok[nt] = 16 * nt + r < d;
    const int64_t blocks = (V + 15) / 16, stride = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < blocks; blk += stride) {
        float av[4][2], bv[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {          // four MFMA steps of four rows each: 16 rows, all loads in flight
            const int64_t row = blk * 16 + 4 * u + g;
            const bool ok = row < V;
            const float* dr = dy + (size_t)row * lddy;
            const float* xr = x + (size_t)row * ldx;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) av[u][mt] = ok ? dr[16 * mt + r] : 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[u][nt] = ok && colok[nt] ? xr[16 * nt + r] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                sdy[mt] += av[u][mt];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = nl_mfma(av[u][mt], bv[u][nt], acc[mt][nt]);
            }
        }
    }
    // the workgroup's four waves -> one partial, summed in wave order
    for (int i = lane; i < NL_HSIZE; i += 64) slot[wave][i] = 0.0f;   // padded columns stay zero
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) slot[wave][(16 * mt + 4 * g + e) * NL_DP + 16 * nt + r] = acc[mt][nt][e];
        float s = sdy[mt];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0) slot[wave][NL_OUT * NL_DP + 16 * mt + r] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NL_HSIZE; i += 256)
        partial[(size_t)blockIdx.x * NL_HSIZE + i] = ((slot[0][i] + slot[1][i]) + slot[2][i]) + slot[3][i];
}

// ---- backward pass 2a: sum of the partials (fixed order), 32 outputs per workgroup
__global__ void __launch_bounds__(256)
nl_bwd_sum_kernel(int nwg, const float* __restrict__ partial, float* __restrict__ coef) {
    __shared__ float red[8][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
    float s = 0.0f;
    if (o < NL_HSIZE)
        for (int w = part; w < nwg; w += 8) s += partial[(size_t)w * NL_HSIZE + o];
    red[part][threadIdx.x & 31] = s;
    __syncthreads();
    if (threadIdx.x < 32 && o < NL_HSIZE) {
        float t = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) t += red[p][threadIdx.x];
        coef[NLC_HRAW + o] = t;
    }
}

// ---- backward pass 2b (one workgroup): dG = (Hraw - sdy mean^T) inv, dc = sdy; the two BatchNorm reduction terms
//   u[n] = sum_m G[m][n] sdy[m],  w[n] = sum_m G[m][n] dG[m][n],  k1 = -inv^2 w / V,  k0 = -inv u / V - mean k1
__global__ void __launch_bounds__(128)
nl_bwd_finish_kernel(int64_t V, int d, const float* __restrict__ G, const float* __restrict__ mean,
                     const float* __restrict__ inv, float* __restrict__ coef, float* __restrict__ dG, float* __restrict__ dc) {
    const int n = threadIdx.x;
    const float* hraw = coef + NLC_HRAW;
    const float* sdy = hraw + NL_OUT * NL_DP;
    if (n < NL_OUT) dc[n] = sdy[n];
    if (n < NL_DP) {
        float u = 0.0f, w = 0.0f;
        const float mn = n < d ? mean[n] : 0.0f, iv = n < d ? inv[n] : 0.0f;
        for (int m = 0; m < NL_OUT; ++m) {
            const float gm = n < d ? G[m * d + n] : 0.0f;
            const float h = (hraw[m * NL_DP + n] - sdy[m] * mn) * iv;
            if (n < d) dG[m * d + n] = h;
            u += gm * sdy[m];
            w += gm * h;
            coef[NLC_GI + m * NL_DP + n] = gm * iv;
        }
        const float k1 = -(iv * iv) * w / (float)V;
        coef[NLC_K1 + n] = k1;
        coef[NLC_K0 + n] = -iv * u / (float)V - mn * k1;
    }
}

// ---- backward pass 3: dx[v][n] = k0[n] + sum_m dy[v][m] Gi[m][n] + x[v][n] k1[n]
template <int NT, bool ALIGNED>
__global__ void __launch_bounds__(256)
nl_bwd_dx_kernel(int64_t V, int d, const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                 const float* __restrict__ coef, float* __restrict__ dx, int lddx) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float a[NT][8];     // A[m = feature 16 nt + r][k = dy column 16 q + 4 g + j], step s = 4 q + j
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 8; ++s) a[nt][s] = coef[NLC_GI + (16 * (s >> 2) + 4 * g + (s & 3)) * NL_DP + 16 * nt + r];
    const int64_t blocks = (V + 15) / 16, stride = (int64_t)gridDim.x * 4;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < blocks; blk += stride) {
        const int64_t row = blk * 16 + r;
        const bool ok = row < V;
        const float* dr = dy + (size_t)row * lddy;
        const float* xr = x + (size_t)row * ldx;
        nlf4 dq[2], xv[NT];
#pragma unroll
        for (int q = 0; q < 2; ++q) dq[q] = ok ? *(const nlf4*)(dr + 16 * q + 4 * g) : nlf4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xv[nt] = nl_load4_if<ALIGNED>(xr, 16 * nt + 4 * g, d, ok);
        float* outr = dx + (size_t)row * lddx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n0 = 16 * nt + 4 * g;
            const nlf4 k0 = *(const nlf4*)(coef + NLC_K0 + n0), k1 = *(const nlf4*)(coef + NLC_K1 + n0);
            nlf4 acc = k0 + xv[nt] * k1;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = nl_mfma(a[nt][s], dq[s >> 2][s & 3], acc);
            if (!ok) continue;
            if (ALIGNED && n0 + 3 < d) *(nlf4*)(outr + n0) = acc;
            else {
                if (n0 < d) outr[n0] = acc.x;
                if (n0 + 1 < d) outr[n0 + 1] = acc.y;
                if (n0 + 2 < d) outr[n0 + 2] = acc.z;
                if (n0 + 3 < d) outr[n0 + 3] = acc.w;
            }
        }
    }
}

// Grid of a grid-stride kernel = the workgroups the chip holds at once (occupancy x CUs), never a fixed number: with 2048
// workgroups the forward kernel (7 waves per SIMD for d = 60: 1792 resident) ran 1.14 rounds and the reduction (3
// workgroups per CU for its 41 KB of LDS: 768 resident) 1.33 with 1024 -- the last round at a fraction of the machine
// (reduction at cfg2: 0.92 -> 0.66 ms with 768).  Asked once per kernel, kept for the process (all GPUs of a node are alike).
static int nl_resident_lookup(const void* kernel, int threads, int fallback) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> known;
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(kernel);
    if (it != known.end()) return it->second;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1 || cus < 1)
        return fallback;          // not remembered: asked again next time
    return known[kernel] = per_cu * cus;
}
template <typename K>
static int nl_resident_wgs(K kernel, int threads, int fallback) { return nl_resident_lookup((const void*)kernel, threads, fallback); }
template <typename K>
static inline unsigned nl_grid(int64_t V, K kernel) {
    const int64_t wgs = ((V + 15) / 16 + 3) / 4;
    const int64_t cap = nl_resident_wgs(kernel, 256, 1024);
    return (unsigned)(wgs < cap ? (wgs > 0 ? wgs : 1) : cap);
}

// stats (may be NULL): column statistics the PRODUCER of x already formed -- stat_rows rows of [2][NL_DP] floats, sums of
// (x - x[0]) and (x - x[0])^2 over disjoint sets of rows that cover x (the anchor gather holds every 64 rows of its matrix
// in LDS anyway: csrc/anchor_gather.hip) -- then the statistics pass over x (1.3 GB at configs[2]) is skipped.
int launch_norm_linear_forward(int64_t V, int d, const float* x, int ldx, const float* G, const float* c, float eps, float* y,
                               float* mean, float* var, float* inv, void* scratch, const float* stats, int stat_rows,
                               hipStream_t st) {
    if (d < 1 || d > NL_DP) return 1;
    float* coef = (float*)scratch;
    float* spart = (float*)((char*)scratch + align_up((size_t)NLC_END * 4));
    if (stats && stat_rows > 0) {
        nl_stats_finish_kernel<<<1, 1024, 0, st>>>(V, d, stat_rows, x, stats, G, c, eps, mean, var, inv, coef);
    } else {
    // slabs of 2048 rows until they are more than the chip holds at once (8 workgroups per CU), then one round of larger
    // slabs: 2246 workgroups on 2048 places ran a second round at a tenth of the machine, and the single-workgroup finish
    // kernel read 9 k partials at 18 M rows
    const int resident = d <= 64 ? nl_resident_wgs(nl_stats_partial_kernel<64>, 256, 2048) : nl_resident_wgs(nl_stats_partial_kernel<128>, 256, 2048);
    int64_t slab = NL_SLAB;
    if ((V + slab - 1) / slab > resident) slab = ((V + resident - 1) / resident + 7) / 8 * 8;
    const int nwg = (int)((V + slab - 1) / slab);          // <= nl_stat_wgs(V): the scratch holds it
    if (d <= 64) nl_stats_partial_kernel<64><<<nwg, 256, 0, st>>>(V, d, (int)slab, x, ldx, spart);
    else nl_stats_partial_kernel<128><<<nwg, 256, 0, st>>>(V, d, (int)slab, x, ldx, spart);
    nl_stats_finish_kernel<<<1, 1024, 0, st>>>(V, d, nwg, x, spart, G, c, eps, mean, var, inv, coef);
    }
    const bool al = ldx % 4 == 0 && ((uintptr_t)x & 15) == 0;
#define SCR_NL_FWD(QQ)                                                                                                              \
    case QQ:                                                                                                                        \
        if (al) nl_forward_kernel<QQ, true><<<nl_grid(V, nl_forward_kernel<QQ, true>), 256, 0, st>>>(V, d, x, ldx, coef, y);        \
        else nl_forward_kernel<QQ, false><<<nl_grid(V, nl_forward_kernel<QQ, false>), 256, 0, st>>>(V, d, x, ldx, coef, y);         \
        break;
    switch ((d + 15) / 16) { SCR_NL_FWD(1) SCR_NL_FWD(2) SCR_NL_FWD(3) SCR_NL_FWD(4) SCR_NL_FWD(5) }
#undef SCR_NL_FWD
    return 0;
}

// coef_out (may be NULL): the coefficients of pass 3 -- Gi [32][NL_DP] | k0 [NL_DP] | k1 [NL_DP], contiguous -- copied out for a
// consumer that forms the rows of dx itself (csrc/anchor_gather.hip, csrc/triplane.hip pass 3); with dx == NULL pass 3 is
// not run here at all.
int launch_norm_linear_backward(int64_t V, int d, const float* x, int ldx, const float* dy, int lddy, const float* G,
                                const float* mean, const float* inv, float* dx, int lddx, float* dG, float* dc, void* scratch,
                                float* coef_out, hipStream_t st) {
    if (d < 1 || d > NL_DP) return 1;
    if (lddy % 4 != 0 || ((uintptr_t)dy & 15) != 0) return 2;
    float* coef = (float*)scratch;
    float* bpart = (float*)((char*)scratch + align_up((size_t)NLC_END * 4) + align_up((size_t)nl_stat_wgs(V) * 2 * NL_DP * 4));
    int nwg = nl_bwd_wgs(V);        // the scratch holds this many partials: an upper bound for the grid
#define SCR_NL_RED(NN)                                                                              \
    case NN:                                                                                        \
        nwg = min(nwg, nl_resident_wgs(nl_bwd_reduce_kernel<NN>, 256, 768));                        \
        nl_bwd_reduce_kernel<NN><<<nwg, 256, 0, st>>>(V, d, x, ldx, dy, lddy, bpart);               \
        break;
    switch ((d + 15) / 16) { SCR_NL_RED(1) SCR_NL_RED(2) SCR_NL_RED(3) SCR_NL_RED(4) SCR_NL_RED(5) }
#undef SCR_NL_RED
    nl_bwd_sum_kernel<<<(NL_HSIZE + 31) / 32, 256, 0, st>>>(nwg, bpart, coef);
    nl_bwd_finish_kernel<<<1, 128, 0, st>>>(V, d, G, mean, inv, coef, dG, dc);
    static_assert(NLC_K0 == NLC_GI + NL_OUT * NL_DP && NLC_K1 == NLC_K0 + NL_DP, "Gi | k0 | k1 are contiguous");
    if (coef_out && hipMemcpyAsync(coef_out, coef + NLC_GI, (size_t)(NL_OUT + 2) * NL_DP * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return 3;
    if (!dx) return 0;
    const bool al = ldx % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0;
#define SCR_NL_DX(NN)                                                                                                                          \
    case NN:                                                                                                                                   \
        if (al) nl_bwd_dx_kernel<NN, true><<<nl_grid(V, nl_bwd_dx_kernel<NN, true>), 256, 0, st>>>(V, d, x, ldx, dy, lddy, coef, dx, lddx);     \
        else nl_bwd_dx_kernel<NN, false><<<nl_grid(V, nl_bwd_dx_kernel<NN, false>), 256, 0, st>>>(V, d, x, ldx, dy, lddy, coef, dx, lddx);      \
        break;
    switch ((d + 15) / 16) { SCR_NL_DX(1) SCR_NL_DX(2) SCR_NL_DX(3) SCR_NL_DX(4) SCR_NL_DX(5) }
#undef SCR_NL_DX
    return 0;
}

// Pass 3 on its own, from a coefficient block a previous launch_norm_linear_backward(coef_out) left (Gi | k0 | k1): for the
// consumer of those coefficients that finds it cannot form the rows itself after all (layout outside its fused pass).
int launch_norm_linear_dx(int64_t V, int d, const float* x, int ldx, const float* dy, int lddy, const float* coef3, float* dx,
                          int lddx, hipStream_t st) {
    if (d < 1 || d > NL_DP) return 1;
    if (lddy % 4 != 0 || ((uintptr_t)dy & 15) != 0) return 2;
    const float* coef = coef3 - NLC_GI;          // the kernel reads coef + NLC_GI / NLC_K0 / NLC_K1 only: contiguous, in this order
    const bool al = ldx % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0;
#define SCR_NL_DX(NN)                                                                                                                          \
    case NN:                                                                                                                                   \
        if (al) nl_bwd_dx_kernel<NN, true><<<nl_grid(V, nl_bwd_dx_kernel<NN, true>), 256, 0, st>>>(V, d, x, ldx, dy, lddy, coef, dx, lddx);     \
        else nl_bwd_dx_kernel<NN, false><<<nl_grid(V, nl_bwd_dx_kernel<NN, false>), 256, 0, st>>>(V, d, x, ldx, dy, lddy, coef, dx, lddx);      \
        break;
    switch ((d + 15) / 16) { SCR_NL_DX(1) SCR_NL_DX(2) SCR_NL_DX(3) SCR_NL_DX(4) SCR_NL_DX(5) }
#undef SCR_NL_DX
    return 0;
}

// ------------------------------------------------------------------ folding the BatchNorm / Linear parameters
// FeaturePlanes sums L <= 4 pairs Linear_i(BatchNorm_i(.)) over its active levels.  With batch statistics the pairs fold
// into ONE weight matrix G [32, d] and one bias c [32] (scene_model._norm_linear):
//   pair i sees columns [col_i, col_i + d_i) of the input (the plane branch: the levels' samples side by side) or the
//   whole input (col_i = 0, d_i = d: the attribute branch, every level reads g_fea):
//     G[r, col_i + j] += W_i[r, j] * gamma_i[j]          c[r] = sum_i ( sum_j W_i[r, j] * beta_i[j] + b_i[r] )
// In the framework that is ~15 kernels on 32 x 71 numbers per branch and twice that on the way back, each a launch of
// its own (4.5 us of GPU time apiece: 0.45 ms of a 23.5 ms cfg2 step).  Here: one single-workgroup kernel per direction;
// fixed order of additions (pair 0 first): deterministic.
struct NlFoldArgs {
    int L, d;
    unsigned char at[NL_DP];   // column of G (and of the input matrix) that holds reference column j: identity unless the caller
                               // keeps its input columns in another order (FeaturePlanes stacks two grids' planes)
    int dd[4], col[4];
    const float* W[4];      // [32, d_i]
    const float* b[4];      // [32]
    const float* gamma[4];  // [d_i]
    const float* beta[4];   // [d_i]
    float* dW[4];           // backward outputs (same shapes)
    float* db[4];
    float* dgamma[4];
    float* dbeta[4];
};

__global__ void __launch_bounds__(256) nl_fold_kernel(NlFoldArgs a, float* __restrict__ G, float* __restrict__ c) {
    // grid: one workgroup per 256 elements of G; the last workgroup also forms c (8 lanes per row, fixed-order shuffle sum)
    const int e = (int)blockIdx.x * 256 + threadIdx.x;
    if (e < 32 * a.d) {
        const int r = e / a.d, col = e - r * a.d;      // col: reference column
        float g = 0.0f;
        for (int i = 0; i < a.L; ++i) {
            const int j = col - a.col[i];
            if (j >= 0 && j < a.dd[i]) g += a.W[i][r * a.dd[i] + j] * a.gamma[i][j];
        }
        G[r * a.d + a.at[col]] = g;
    }
    if (blockIdx.x == gridDim.x - 1) {
        const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
        float acc = 0.0f;
        for (int i = 0; i < a.L; ++i) {
            float dot = 0.0f;
            for (int j = sub; j < a.dd[i]; j += 8) dot += a.W[i][r * a.dd[i] + j] * a.beta[i][j];
            dot += __shfl_xor(dot, 1, 64);
            dot += __shfl_xor(dot, 2, 64);
            dot += __shfl_xor(dot, 4, 64);
            acc += dot + a.b[i][r];
        }
        if (sub == 0) c[r] = acc;
    }
}

// dW_i[r, j] = dG[r, col_i + j] * gamma_i[j] + dc[r] * beta_i[j];  db_i = dc;
// dgamma_i[j] = sum_r dG[r, col_i + j] * W_i[r, j];  dbeta_i[j] = sum_r dc[r] * W_i[r, j]
__global__ void __launch_bounds__(256) nl_fold_backward_kernel(NlFoldArgs a, const float* __restrict__ dG,
                                                               const float* __restrict__ dc) {
    // grid: (ceil(32 d_max / 256), L): workgroup (x, i) writes 256 elements of dW_i; x == 0 also the column sums of pair i
    const int i = (int)blockIdx.y, di = a.dd[i];
    const int e = (int)blockIdx.x * 256 + threadIdx.x;
    if (e < 32 * di) {
        const int r = e / di, j = e - r * di;
        a.dW[i][e] = dG[r * a.d + a.at[a.col[i] + j]] * a.gamma[i][j] + dc[r] * a.beta[i][j];
    }
    if (blockIdx.x == 0) {
        for (int j = threadIdx.x; j < di; j += 256) {
            float sg = 0.0f, sb = 0.0f;
            for (int r = 0; r < 32; ++r) {
                const float w = a.W[i][r * di + j];
                sg += dG[r * a.d + a.at[a.col[i] + j]] * w;
                sb += dc[r] * w;
            }
            a.dgamma[i][j] = sg;
            a.dbeta[i][j] = sb;
        }
        if (threadIdx.x < 32) a.db[i][threadIdx.x] = dc[threadIdx.x];
    }
}

// running statistics of the L BatchNorms after a training-mode forward (nn.BatchNorm1d with a momentum):
//   running_mean = (1 - m) running_mean + m mean;  running_var = (1 - m) running_var + m var n / (n - 1);  batches += 1
struct NlRunArgs {
    int L;
    unsigned char at[NL_DP];
    int dd[4], col[4];
    float momentum[4];
    float* run_mean[4];
    float* run_var[4];
    long long* batches[4];
};
__global__ void __launch_bounds__(256) nl_running_stats_kernel(NlRunArgs a, const float* __restrict__ mean,
                                                               const float* __restrict__ var, float unbias) {
    for (int i = 0; i < a.L; ++i) {
        const float m = a.momentum[i];
        for (int j = threadIdx.x; j < a.dd[i]; j += 256) {
            const int at = a.at[a.col[i] + j];
            a.run_mean[i][j] = a.run_mean[i][j] * (1.0f - m) + m * mean[at];
            a.run_var[i][j] = a.run_var[i][j] * (1.0f - m) + m * (var[at] * unbias);
        }
        if (threadIdx.x == 0 && a.batches[i]) *a.batches[i] += 1;
    }
}

static bool nl_set_perm(unsigned char (&at)[NL_DP], int d, const unsigned char* col_at) {
    bool seen[NL_DP] = {};
    for (int j = 0; j < d; ++j) {
        const int t = col_at ? col_at[j] : j;
        if (t >= d || seen[t]) return false;       // not a permutation of 0 .. d-1
        seen[t] = true;
        at[j] = (unsigned char)t;
    }
    return true;
}

int launch_nl_fold(int L, int d, const int* dd, const int* col, const unsigned char* col_at, const float* const* W,
                   const float* const* b, const float* const* gamma, const float* const* beta, float* G, float* c,
                   hipStream_t st) {
    if (L < 1 || L > 4 || d < 1 || d > NL_DP) return 1;
    NlFoldArgs a = {};
    a.L = L; a.d = d;
    if (!nl_set_perm(a.at, d, col_at)) return 1;
    for (int i = 0; i < L; ++i) {
        if (dd[i] < 1 || col[i] < 0 || col[i] + dd[i] > d) return 1;
        a.dd[i] = dd[i]; a.col[i] = col[i]; a.W[i] = W[i]; a.b[i] = b[i]; a.gamma[i] = gamma[i]; a.beta[i] = beta[i];
    }
    nl_fold_kernel<<<(32 * d + 255) / 256, 256, 0, st>>>(a, G, c);
    return 0;
}

int launch_nl_fold_backward(int L, int d, const int* dd, const int* col, const unsigned char* col_at, const float* const* W,
                            const float* const* gamma, const float* const* beta, const float* dG, const float* dc,
                            float* const* dW, float* const* db, float* const* dgamma, float* const* dbeta, hipStream_t st) {
    if (L < 1 || L > 4 || d < 1 || d > NL_DP) return 1;
    NlFoldArgs a = {};
    a.L = L; a.d = d;
    if (!nl_set_perm(a.at, d, col_at)) return 1;
    for (int i = 0; i < L; ++i) {
        if (dd[i] < 1 || col[i] < 0 || col[i] + dd[i] > d) return 1;
        a.dd[i] = dd[i]; a.col[i] = col[i]; a.W[i] = W[i]; a.gamma[i] = gamma[i]; a.beta[i] = beta[i];
        a.dW[i] = dW[i]; a.db[i] = db[i]; a.dgamma[i] = dgamma[i]; a.dbeta[i] = dbeta[i];
    }
    nl_fold_backward_kernel<<<dim3((32 * d + 255) / 256, L), 256, 0, st>>>(a, dG, dc);
    return 0;
}

int launch_nl_running_stats(int L, int d, const int* dd, const int* col, const unsigned char* col_at, const float* momentum,
                            float* const* run_mean, float* const* run_var, long long* const* batches, const float* mean,
                            const float* var, int64_t n, hipStream_t st) {
    if (L < 1 || L > 4 || d < 1 || d > NL_DP) return 1;
    NlRunArgs a = {};
    a.L = L;
    if (!nl_set_perm(a.at, d, col_at)) return 1;
    for (int i = 0; i < L; ++i)
        if (dd[i] < 1 || col[i] < 0 || col[i] + dd[i] > d) return 1;
    for (int i = 0; i < L; ++i) {
        a.dd[i] = dd[i]; a.col[i] = col[i]; a.momentum[i] = momentum[i];
        a.run_mean[i] = run_mean[i]; a.run_var[i] = run_var[i]; a.batches[i] = batches ? batches[i] : nullptr;
    }
    nl_running_stats_kernel<<<1, 256, 0, st>>>(a, mean, var, (float)((double)n / (double)(n > 1 ? n - 1 : 1)));
    return 0;
}

}  // namespace scr
