// splatco_amd/csrc/blend.hip -- front-to-back tile alpha blending, forward and backward (gfx950).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile, one pixel per lane; a wave owns a 16x4
// pixel strip.  The tile's depth-sorted splat list is streamed in batches of 256: each lane
// gathers one 48-byte splat record (three float4) into LDS, then all lanes walk the batch with
// conflict-free broadcast ds_read_b128.
//
// Backward: per (tile, Gaussian) gradients are reduced on chip -- DPP row/bank shifts inside the
// wave, one LDS slot per (wave, splat), a fixed-order 4-way add -- and written ONCE as a 48-byte
// record into the slot the instance occupied before the depth sort.  No floating-point atomics:
// results are bit-reproducible.  The per-Gaussian sum over tiles happens in
// preprocess_backward_kernel.
//
// Blend arithmetic is normative where a decision hangs on it: power is evaluated as
// fma(dx, fma(A,dx,B*dy), (C*dy)*dy) with A=-Qxx/2, B=-Qxy, C=-Qyy/2; exp() may differ from the
// oracle's libm by 2 ulp (v_exp_f32).  Compiled with -ffp-contract=off; FMAs only where written.
#include "common.h"

namespace scr {

constexpr int BATCH = 256;

__device__ __forceinline__ float fast_exp(float x) {  // v_exp_f32(x * log2 e)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

// ------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256)
blend_forward_kernel(int W, int H, int gx, int tiles, const uint32_t* __restrict__ ranges,
                     const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                     const float* __restrict__ bg, float* __restrict__ out_color,
                     float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ float4 s0[BATCH], s1[BATCH];
    __shared__ float s2[BATCH];
    int t = xcd_tile(blockIdx.x, tiles);
    if (t < 0) return;
    const int tx = t % gx, ty = t / gx;
    // lane -> pixel: wave w covers rows 4w..4w+3 of the tile
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t lo = ranges[2 * t], hi = ranges[2 * t + 1];
    bool done = !inside;
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t contributor = 0, last = 0;
    for (uint32_t base = lo; base < hi; base += BATCH) {
        if (__syncthreads_count(done) == 256) break;
        uint32_t idx = base + threadIdx.x;
        if (idx < hi) {
            uint32_t g = point_list[idx];
            s0[threadIdx.x] = rec[3 * (size_t)g];
            s1[threadIdx.x] = rec[3 * (size_t)g + 1];
            s2[threadIdx.x] = rec[3 * (size_t)g + 2].x;
        }
        __syncthreads();
        const int cnt = min((uint32_t)BATCH, hi - base);
        for (int j = 0; !done && j < cnt; ++j) {
            ++contributor;
            float4 a = s0[j], b = s1[j];
            float dx = a.x - pxf, dy = a.y - pyf;
            float power = __builtin_fmaf(dx, __builtin_fmaf(a.z, dx, a.w * dy), (b.x * dy) * dy);
            if (power > 0.0f) continue;
            float alpha = fminf(0.99f, b.y * fast_exp(power));
            if (alpha < 1.0f / 255.0f) continue;
            float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            float w = alpha * T;
            C0 = __builtin_fmaf(b.z, w, C0);
            C1 = __builtin_fmaf(b.w, w, C1);
            C2 = __builtin_fmaf(s2[j], w, C2);
            T = test_T;
            last = contributor;
        }
    }
    if (inside) {
        size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
        out_color[pix] = __builtin_fmaf(T, bg[0], C0);
        out_color[hw + pix] = __builtin_fmaf(T, bg[1], C1);
        out_color[2 * hw + pix] = __builtin_fmaf(T, bg[2], C2);
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ------------------------------------------------------------------ wave64 sum via DPP
// Inclusive row scans (row_shr 1,2,4,8) then row_bcast15 / row_bcast31: lane 63 ends up with the
// sum over the wave, always added in the same order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, /*bound_ctrl=*/true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row = row sum
    v = dpp_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return v;
}

// ------------------------------------------------------------------ backward
__global__ void __launch_bounds__(256)
blend_backward_kernel(int W, int H, int gx, int tiles, const uint32_t* __restrict__ ranges,
                      const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ orig_slot,
                      const float4* __restrict__ rec, const float* __restrict__ bg,
                      const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                      const float* __restrict__ dL_dpix, float4* __restrict__ grad_rec) {
    __shared__ float4 s0[BATCH], s1[BATCH];
    __shared__ float s2[BATCH];
    __shared__ float acc[4][BATCH][9];  // per-wave partial sums for the current batch
    __shared__ uint32_t wave_max[4];
    int t = xcd_tile(blockIdx.x, tiles);
    if (t < 0) return;
    const int tx = t % gx, ty = t / gx;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = tx * TILE + lx, py = ty * TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t lo = ranges[2 * t], hi = ranges[2 * t + 1];
    const uint32_t n = hi - lo;
    if (n == 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    const float T_final = inside ? final_T[pix] : 0.0f;
    float T = T_final;
    float dLp0 = 0, dLp1 = 0, dLp2 = 0;
    if (inside) {
        dLp0 = dL_dpix[pix];
        dLp1 = dL_dpix[hw + pix];
        dLp2 = dL_dpix[2 * hw + pix];
    }
    const float bg_dot = (bg[0] * dLp0 + bg[1] * dLp1) + bg[2] * dLp2;
    float ac0 = 0, ac1 = 0, ac2 = 0, last_alpha = 0, lc0 = 0, lc1 = 0, lc2 = 0;
    // workgroup-wide largest contributor count: batches past it hold only zero gradients
    uint32_t wm = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, d, WAVE));
    if (lane == 0) wave_max[wave] = wm;
    __syncthreads();
    const uint32_t max_last = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    const uint32_t wave_last = wave_max[wave];
    const int nbatch = (int)((n + BATCH - 1) / BATCH);
    for (int bi = nbatch - 1; bi >= 0; --bi) {
        const uint32_t base = (uint32_t)bi * BATCH;  // list position of the batch's first splat
        const int cnt = (int)min((uint32_t)BATCH, n - base);
        const bool live = base < max_last;
        __syncthreads();  // previous batch's acc / staging fully consumed
        if (live) {
            if (threadIdx.x < cnt) {
                uint32_t g = point_list[lo + base + threadIdx.x];
                s0[threadIdx.x] = rec[3 * (size_t)g];
                s1[threadIdx.x] = rec[3 * (size_t)g + 1];
                s2[threadIdx.x] = rec[3 * (size_t)g + 2].x;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int c = 0; c < 9; ++c) acc[w][threadIdx.x][c] = 0.0f;
            __syncthreads();
            // back to front inside the batch; a wave skips splats behind all of its pixels
            int jstart = cnt - 1;
            if (base + (uint32_t)cnt > wave_last) jstart = (int)wave_last - (int)base - 1;
            for (int j = jstart; j >= 0; --j) {
                const uint32_t q = base + (uint32_t)j;  // list position; contributor number q+1
                float4 a = s0[j], b = s1[j];
                float cb = s2[j];
                float dx = a.x - pxf, dy = a.y - pyf;
                float power = __builtin_fmaf(dx, __builtin_fmaf(a.z, dx, a.w * dy), (b.x * dy) * dy);
                float G = fast_exp(power);
                float alpha = fminf(0.99f, b.y * G);
                bool hit = (q < last) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;
                float g_mx = 0, g_my = 0, g_qxx = 0, g_qxy = 0, g_qyy = 0, g_o = 0, g_c0 = 0, g_c1 = 0, g_c2 = 0;
                if (hit) {
                    T = T * __builtin_amdgcn_rcpf(1.0f - alpha);
                    const float dchan = alpha * T;
                    ac0 = last_alpha * lc0 + (1.0f - last_alpha) * ac0;
                    ac1 = last_alpha * lc1 + (1.0f - last_alpha) * ac1;
                    ac2 = last_alpha * lc2 + (1.0f - last_alpha) * ac2;
                    lc0 = b.z; lc1 = b.w; lc2 = cb;
                    float dL_dalpha = ((b.z - ac0) * dLp0 + (b.w - ac1) * dLp1) + (cb - ac2) * dLp2;
                    g_c0 = dchan * dLp0; g_c1 = dchan * dLp1; g_c2 = dchan * dLp2;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(1.0f - alpha)) * bg_dot;
                    const float dL_dG = b.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    // Q = (-2A, -B, -2C)
                    const float Qxx = -2.0f * a.z, Qxy = -a.w, Qyy = -2.0f * b.x;
                    g_mx = dL_dG * (-gdx * Qxx - gdy * Qxy);
                    g_my = dL_dG * (-gdy * Qyy - gdx * Qxy);
                    g_qxx = -0.5f * gdx * dx * dL_dG;
                    g_qxy = -gdx * dy * dL_dG;
                    g_qyy = -0.5f * gdy * dy * dL_dG;
                    g_o = G * dL_dalpha;
                }
                g_mx = wave_sum_to_lane63(g_mx);
                g_my = wave_sum_to_lane63(g_my);
                g_qxx = wave_sum_to_lane63(g_qxx);
                g_qxy = wave_sum_to_lane63(g_qxy);
                g_qyy = wave_sum_to_lane63(g_qyy);
                g_o = wave_sum_to_lane63(g_o);
                g_c0 = wave_sum_to_lane63(g_c0);
                g_c1 = wave_sum_to_lane63(g_c1);
                g_c2 = wave_sum_to_lane63(g_c2);
                if (lane == 63) {
                    float* d = acc[wave][j];
                    d[0] = g_mx; d[1] = g_my; d[2] = g_qxx; d[3] = g_qxy; d[4] = g_qyy;
                    d[5] = g_o; d[6] = g_c0; d[7] = g_c1; d[8] = g_c2;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x < cnt) {
            float r[9];
#pragma unroll
            for (int c = 0; c < 9; ++c)
                r[c] = live ? ((acc[0][threadIdx.x][c] + acc[1][threadIdx.x][c]) + acc[2][threadIdx.x][c]) +
                                  acc[3][threadIdx.x][c]
                            : 0.0f;
            size_t slot = orig_slot[lo + base + threadIdx.x];
            grad_rec[3 * slot + 0] = make_float4(r[0], r[1], r[2], r[3]);
            grad_rec[3 * slot + 1] = make_float4(r[4], r[5], r[6], r[7]);
            grad_rec[3 * slot + 2] = make_float4(r[8], 0.0f, 0.0f, 0.0f);
        }
    }
}

// ------------------------------------------------------------------ launchers
void launch_blend_forward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                          float* out_color, hipStream_t st) {
    Grid g(ks.H, ks.W);
    blend_forward_kernel<<<(unsigned)xcd_grid(g.tiles), 256, 0, st>>>(
        ks.W, ks.H, g.gx, g.tiles, gv.ranges, bv.point_list, gv.rec, ks.bg, out_color, iv.final_T, iv.n_contrib);
}

void launch_blend_backward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                           const float* dL_dcolor, float4* grad_rec, hipStream_t st) {
    Grid g(ks.H, ks.W);
    blend_backward_kernel<<<(unsigned)xcd_grid(g.tiles), 256, 0, st>>>(
        ks.W, ks.H, g.gx, g.tiles, gv.ranges, bv.point_list, bv.orig_slot, gv.rec, ks.bg, iv.final_T,
        iv.n_contrib, dL_dcolor, grad_rec);
}

}  // namespace scr
