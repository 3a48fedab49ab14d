// splatco_amd/csrc/ssim.hip -- fused L1 + SSIM image loss, forward and backward (gfx950).
//
// The per-view training loss right after the rasterizer (train.py:192-196) is
// 0.8 * L1 + 0.2 * (1 - SSIM) with SSIM = utils/loss_utils.py:34-63: five grouped 11x11 conv2d
// (zero padding 5, Gaussian window sigma 1.5) + elementwise maps.  In PyTorch that costs 11 ms
// forward+backward at 1080p -- 6x the whole rasterizer step.  Here each 16x16 output tile stages
// its 26x26 halo of both images in LDS once, runs the separable window (11 + 11 taps) for the
// five moments, evaluates the SSIM map and its three partial derivatives in registers, and
// reduces the tile's L1 / SSIM sums deterministically (per-tile partials, fixed-order final sum).
// The backward is the transposed window applied to the three derivative maps.
//
//   mu1 = G*x, mu2 = G*y, E11 = G*x^2, E22 = G*y^2, E12 = G*xy
//   ssim = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s11 + s22 + C2)),  s.. = E.. - mu mu
//   dL/dx(p) = sum_q G(q-p) [ dmu1(q) + 2 x(p) dE11(q) + y(p) dE12(q) ]
#include "common.h"

namespace scr {

constexpr int SS_T = 16;             // output tile edge
constexpr int SS_R = 5;              // window radius (11 taps)
constexpr int SS_H = SS_T + 2 * SS_R;  // 26: tile edge with halo

struct SsimWindow { float g[11]; };

__device__ __forceinline__ float ss_load(const float* __restrict__ img, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.0f;  // conv2d zero padding
}

// grid: (ceil(W/16), ceil(H/16), C), 256 threads.  partial[block] = (sum |x-y|, sum ssim) of the tile.
__global__ void __launch_bounds__(256)
l1_ssim_forward_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                       SsimWindow win, float* __restrict__ dmaps /*[3][C][H][W] or NULL*/,
                       float2* __restrict__ partial) {
    __shared__ float t1[SS_H][SS_H + 1], t2[SS_H][SS_H + 1];
    __shared__ float hz[5][SS_H][SS_T + 1];  // horizontal pass: 5 moments, 26 rows x 16 columns
    __shared__ float2 wsum[4];
    const int c = blockIdx.z, C = gridDim.z;
    const size_t plane = (size_t)H * W;
    const float* x1 = img1 + c * plane;
    const float* x2 = img2 + c * plane;
    const int ox = blockIdx.x * SS_T, oy = blockIdx.y * SS_T;
    for (int i = threadIdx.x; i < SS_H * SS_H; i += 256) {
        const int ly = i / SS_H, lx = i % SS_H;
        t1[ly][lx] = ss_load(x1, H, W, oy + ly - SS_R, ox + lx - SS_R);
        t2[ly][lx] = ss_load(x2, H, W, oy + ly - SS_R, ox + lx - SS_R);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_H * SS_T; i += 256) {
        const int ly = i / SS_T, lx = i % SS_T;
        float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float a = t1[ly][lx + k], b = t2[ly][lx + k], g = win.g[k];
            m1 += g * a; m2 += g * b; e11 += g * (a * a); e22 += g * (b * b); e12 += g * (a * b);
        }
        hz[0][ly][lx] = m1; hz[1][ly][lx] = m2; hz[2][ly][lx] = e11; hz[3][ly][lx] = e22; hz[4][ly][lx] = e12;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = ox + lx, py = oy + ly;
    float l1 = 0.0f, ss = 0.0f;
    if (px < W && py < H) {
        float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = win.g[k];
            mu1 += g * hz[0][ly + k][lx]; mu2 += g * hz[1][ly + k][lx]; e11 += g * hz[2][ly + k][lx];
            e22 += g * hz[3][ly + k][lx]; e12 += g * hz[4][ly + k][lx];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        const float s11 = e11 - mu1s, s22 = e22 - mu2s, s12 = e12 - mu12;
        const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2, B1 = mu1s + mu2s + C1, B2 = s11 + s22 + C2;
        const float iB1 = 1.0f / B1, iB2 = 1.0f / B2;
        ss = (A1 * A2) * (iB1 * iB2);
        l1 = fabsf(t1[ly + SS_R][lx + SS_R] - t2[ly + SS_R][lx + SS_R]);
        if (dmaps) {
            // partial derivatives w.r.t. the three window outputs that depend on img1
            const float dE11 = -ss * iB2;                       // d/ds11 = -A1 A2 / (B1 B2^2)
            const float dE12 = 2.0f * A1 * (iB1 * iB2);         // d/ds12
            const float dmu1 = 2.0f * mu2 * A2 * (iB1 * iB2)    // through A1
                               - 2.0f * mu1 * ss * iB1          // through B1
                               - mu2 * dE12                     // s12 = E12 - mu1 mu2
                               - 2.0f * mu1 * dE11;             // s11 = E11 - mu1^2
            const size_t o = (size_t)py * W + px;
            dmaps[(0 * (size_t)C + c) * plane + o] = dmu1;
            dmaps[(1 * (size_t)C + c) * plane + o] = dE11;
            dmaps[(2 * (size_t)C + c) * plane + o] = dE12;
        }
    }
    // deterministic tile sums: wave shuffles, then 4 wave partials
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        l1 += __shfl_down(l1, d, WAVE);
        ss += __shfl_down(ss, d, WAVE);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = make_float2(l1, ss);
    __syncthreads();
    if (threadIdx.x == 0) {
        float2 r = wsum[0];
        for (int w = 1; w < 4; ++w) { r.x += wsum[w].x; r.y += wsum[w].y; }
        partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = r;
    }
}

// one workgroup: fixed-order sum of the tile partials -> out[0] = mean |x-y|, out[1] = mean ssim
__global__ void __launch_bounds__(1024)
l1_ssim_reduce_kernel(int nblocks, const float2* __restrict__ partial, double inv_n, float* __restrict__ out) {
    __shared__ double sa[1024], sb[1024];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) { a += partial[i].x; b += partial[i].y; }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(sa[0] * inv_n); out[1] = (float)(sb[0] * inv_n); }
}

// backward: dL/dimg1 = g_l1/N * sign(x - y) + g_ssim/N * [ G*dmu1 + 2 x G*dE11 + y G*dE12 ]
__global__ void __launch_bounds__(256)
l1_ssim_backward_kernel(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                        SsimWindow win, const float* __restrict__ dmaps, const float* __restrict__ g_l1,
                        const float* __restrict__ g_ssim, float inv_n, float* __restrict__ dimg1) {
    __shared__ float t[3][SS_H][SS_H + 1];
    __shared__ float hz[3][SS_H][SS_T + 1];
    const int c = blockIdx.z, C = gridDim.z;
    const size_t plane = (size_t)H * W;
    const int ox = blockIdx.x * SS_T, oy = blockIdx.y * SS_T;
    for (int i = threadIdx.x; i < SS_H * SS_H; i += 256) {
        const int ly = i / SS_H, lx = i % SS_H;
#pragma unroll
        for (int m = 0; m < 3; ++m)
            t[m][ly][lx] = ss_load(dmaps + (m * (size_t)C + c) * plane, H, W, oy + ly - SS_R, ox + lx - SS_R);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SS_H * SS_T; i += 256) {
        const int ly = i / SS_T, lx = i % SS_T;
        float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = win.g[k];
            a0 += g * t[0][ly][lx + k]; a1 += g * t[1][ly][lx + k]; a2 += g * t[2][ly][lx + k];
        }
        hz[0][ly][lx] = a0; hz[1][ly][lx] = a1; hz[2][ly][lx] = a2;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = ox + lx, py = oy + ly;
    if (px >= W || py >= H) return;
    float c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float g = win.g[k];
        c0 += g * hz[0][ly + k][lx]; c1 += g * hz[1][ly + k][lx]; c2 += g * hz[2][ly + k][lx];
    }
    const size_t o = c * plane + (size_t)py * W + px;
    const float x = img1[o], y = img2[o];
    const float d = x - y;
    const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    dimg1[o] = (g_l1[0] * inv_n) * sgn + (g_ssim[0] * inv_n) * (c0 + 2.0f * x * c1 + y * c2);
}

static SsimWindow make_window() {
    SsimWindow w;
    double g[11], s = 0;
    for (int i = 0; i < 11; ++i) { g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    // the reference builds the 1-D window in fp32 (torch.Tensor of python floats) and normalises in fp32
    float gf[11], sf = 0.0f;
    for (int i = 0; i < 11; ++i) { gf[i] = (float)g[i]; sf += gf[i]; }
    for (int i = 0; i < 11; ++i) w.g[i] = gf[i] / sf;
    (void)s;
    return w;
}

size_t l1_ssim_scratch_bytes(int C, int H, int W, int with_grad) {
    const size_t nb = (size_t)((W + SS_T - 1) / SS_T) * ((H + SS_T - 1) / SS_T) * C;
    return align_up(nb * sizeof(float2)) + (with_grad ? align_up((size_t)3 * C * H * W * 4) : 0);
}

void launch_l1_ssim_forward(int C, int H, int W, const float* img1, const float* img2, void* scratch,
                            int with_grad, float* out2, hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
    const size_t nb = (size_t)grid.x * grid.y * grid.z;
    float2* partial = (float2*)scratch;
    float* dmaps = with_grad ? (float*)((char*)scratch + align_up(nb * sizeof(float2))) : nullptr;
    l1_ssim_forward_kernel<<<grid, 256, 0, st>>>(H, W, img1, img2, make_window(), dmaps, partial);
    l1_ssim_reduce_kernel<<<1, 1024, 0, st>>>((int)nb, partial, 1.0 / ((double)C * H * W), out2);
}

void launch_l1_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const void* scratch,
                             const float* g_l1, const float* g_ssim, float* dimg1, hipStream_t st) {
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, C);
    const size_t nb = (size_t)grid.x * grid.y * grid.z;
    const float* dmaps = (const float*)((const char*)scratch + align_up(nb * sizeof(float2)));
    l1_ssim_backward_kernel<<<grid, 256, 0, st>>>(H, W, img1, img2, make_window(), dmaps, g_l1, g_ssim,
                                                  (float)(1.0 / ((double)C * H * W)), dimg1);
}

// ------------------------------------------------------------------ scaling regulariser of the per-view loss
// mean_p(s[p,0] s[p,1] s[p,2]) (train.py:192-196: scaling.prod(dim=1).mean()) and its gradient.  torch's prod backward
// counts the zeros of its input first -- a compare, an int64 reduction at 60 GB/s and a host read -- before a division
// pass; here: one streaming pass per direction, partial sums per workgroup added in order by one more workgroup.
constexpr int SREG_ROWS = 2048;      // rows per workgroup
__global__ void __launch_bounds__(256)
scaling_reg_partial_kernel(int64_t P, const float* __restrict__ s, double* __restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t r0 = (int64_t)blockIdx.x * SREG_ROWS;
    for (int64_t r = r0 + threadIdx.x; r < min(P, r0 + SREG_ROWS); r += 256)
        acc += (double)((s[3 * r] * s[3 * r + 1]) * s[3 * r + 2]);      // the product in binary32, as torch.prod forms it
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(1024)
scaling_reg_finish_kernel(int nb, const double* __restrict__ partial, double inv_n, float* __restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nb; b += 1024) acc += partial[b];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = (float)(t * inv_n);
    }
}
__global__ void __launch_bounds__(256)
scaling_reg_backward_kernel(int64_t P, const float* __restrict__ s, const float* __restrict__ g, float inv_n,
                            float* __restrict__ ds) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= P) return;
    const float a = s[3 * r], b = s[3 * r + 1], c = s[3 * r + 2], w = g[0] * inv_n;
    ds[3 * r] = w * (b * c);
    ds[3 * r + 1] = w * (a * c);
    ds[3 * r + 2] = w * (a * b);
}

// ---- the L1 of the cross-view consistency term (train.py:208-217): mean | (real1 - real2) - (gen1 - gen2) | over the n elements of
// four equally shaped images, and its gradient with respect to gen1 / gen2 (+- sign / n).  The framework spends a dozen
// elementwise kernels per view pair on it (0.3 ms at 1080p; six pairs per --mv 4 step); here one pass per direction.  The
// differences are formed in binary32 in the reference's order, the sum in binary64 per workgroup and in a fixed order.
constexpr int PL1_PER_WG = 256 * 16;
__global__ void __launch_bounds__(256)
pair_l1_partial_kernel(int64_t n, const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ r1,
                       const float* __restrict__ r2, double* __restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t e0 = (int64_t)blockIdx.x * PL1_PER_WG;
    for (int64_t e = e0 + threadIdx.x; e < min(n, e0 + PL1_PER_WG); e += 256)
        acc += (double)fabsf((r1[e] - r2[e]) - (g1[e] - g2[e]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256)
pair_l1_backward_kernel(int64_t n, const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ r1,
                        const float* __restrict__ r2, const float* __restrict__ g, float inv_n, float* __restrict__ d1,
                        float* __restrict__ d2) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float d = (r1[e] - r2[e]) - (g1[e] - g2[e]);       // d |d| / d gen1 = -sign(d), / d gen2 = +sign(d); sign(0) = 0 as torch.abs
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f), w = g[0] * inv_n;
    if (d1) d1[e] = -sg * w;
    if (d2) d2[e] = sg * w;
}
size_t pair_l1_scratch_bytes(int64_t n) { return align_up((size_t)((n + PL1_PER_WG - 1) / PL1_PER_WG + 1) * 8); }
void launch_pair_l1_forward(int64_t n, const float* g1, const float* g2, const float* r1, const float* r2, void* scratch, float* out,
                            hipStream_t st) {
    const int nb = (int)((n + PL1_PER_WG - 1) / PL1_PER_WG);
    pair_l1_partial_kernel<<<nb, 256, 0, st>>>(n, g1, g2, r1, r2, (double*)scratch);
    scaling_reg_finish_kernel<<<1, 1024, 0, st>>>(nb, (const double*)scratch, 1.0 / (double)n, out);
}
void launch_pair_l1_backward(int64_t n, const float* g1, const float* g2, const float* r1, const float* r2, const float* g, float* d1,
                             float* d2, hipStream_t st) {
    pair_l1_backward_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, g1, g2, r1, r2, g, (float)(1.0 / (double)n), d1, d2);
}

size_t scaling_reg_scratch_bytes(int64_t P) { return align_up((size_t)((P + SREG_ROWS - 1) / SREG_ROWS + 1) * 8); }
void launch_scaling_reg_forward(int64_t P, const float* s, void* scratch, float* out, hipStream_t st) {
    const int nb = (int)((P + SREG_ROWS - 1) / SREG_ROWS);
    scaling_reg_partial_kernel<<<nb, 256, 0, st>>>(P, s, (double*)scratch);
    scaling_reg_finish_kernel<<<1, 1024, 0, st>>>(nb, (const double*)scratch, 1.0 / (double)P, out);
}
void launch_scaling_reg_backward(int64_t P, const float* s, const float* g, float* ds, hipStream_t st) {
    scaling_reg_backward_kernel<<<(unsigned)((P + 255) / 256), 256, 0, st>>>(P, s, g, (float)(1.0 / (double)P), ds);
}

}  // namespace scr
