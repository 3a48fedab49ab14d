// splatco_amd/csrc/mlp_heads.hip -- the three MLP heads of generate_neural_gaussians as ONE kernel per direction
// (gfx950, fp32 MFMA).  Reference: gaussian_renderer/__init__.py:58-93 with the default flags
// (add_*_dist False, appearance_dim 0) and scene/gaussian_model.py:315-337:
//
//     x = cat(feat[V,32], ob_view[V,3], geo_fea[V,64])                 ob_view = (anchor - campos) / |anchor - campos|
//     (geo_fea arrives as its two halves geo_a | geo_b [V,32] each: the outputs of FeaturePlanes' two GEMMs, never concatenated)
//     neural_opacity = tanh   (W2o relu(W1o x + b1o) + b2o)   [V,10]
//     color          = sigmoid(W2c relu(W1c x + b1c) + b2c)   [V,30]
//     scale_rot      =         W2v relu(W1v x + b1v) + b2v    [V,70]
//
// In PyTorch this is cat + 4 GEMMs + 4 pointwise kernels forward and twice that backward, every one of them a pass
// over [V, 32..110] matrices with V in the millions (one row per visible anchor).  Here x is never materialised, the
// hidden layer stays in registers, and the weights sit in LDS.
//
// It IS a dense contraction (26 kflop per anchor against 0.84 KB of traffic), so it runs on the matrix cores:
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, 32 cycles per instruction and SIMD).  Everything is computed
// TRANSPOSED -- features along M, the 16 anchors of a wave's tile along N:
//     H^T [96 x 16] = W1 [96 x 112] . X^T [112 x 16]           (K = 32 feat + 64 geo + (ob 3, bias 1, 12 zero))
//     O^T [128 x 16] = W2 [128 x 32 per head] . H^T
// because then (a) the B operand of layer 1 is what a lane loads anyway: lane (n = l & 15, g = l >> 4) reads the
// float4 X[anchor n][16 blk + 4 g ..+3] -- 64 contiguous bytes per anchor row and instruction -- and feeds its four
// floats to four K-steps (the K order of a contraction is free, the weights are laid out to match); and (b) the
// accumulator layout of layer 1 (lane (n, g), register r  <->  hidden row 16 mt + 4 g + r, anchor n) IS the
// B-operand layout of layer 2 with K-step (mt, r): the hidden layer never moves between the layers.
// The backward pass is the same trick three times (dH^T = W2^T dZ^T, dX^T = W1^T dPre^T) plus the two weight
// gradients, whose contraction runs over the anchors: their operands go through a wave-private LDS transpose.
// Weight-gradient partial sums stay in registers for all tiles of a wave, are written once per wave and summed in
// wave order by a second kernel: no atomics, bit-reproducible.
#include "common.h"

namespace scr {


typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int MH_FEAT = 32, MH_GEO = 64, MH_HID = 32, MH_IN = 99;   // x = feat | ob_view(3) | geo
constexpr int MH_NO = 10, MH_NC = 30, MH_NV = 70;                     // outputs of the opacity / colour / cov heads
constexpr int MH_KB = 7;          // 16-wide K blocks of layer 1: feat 0-1, geo 2-5, (ob, bias, zeros) 6
constexpr int MH_MT = 6;          // 16-row tiles of the stacked hidden layer (3 heads x 32)
constexpr int MH_OT = 8;          // 16-row tiles of the outputs: opacity 1, colour 2, cov 5
constexpr int MH_WAVES = 4;       // waves per workgroup, one 16-anchor tile each per iteration
// partial weight gradients per wave: dW1 [96][112] (in this file's K order), dW2 [128][32], db2 [128]
constexpr int MH_PART = MH_MT * 16 * MH_KB * 16 + MH_OT * 16 * MH_HID + MH_OT * 16;

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// column of W1 (and of x) behind K index 16 blk + j of this file's order; -1 = bias, -2 = zero padding
__device__ __forceinline__ int mh_kmap(int blk, int j) {
    if (blk < 2) return 16 * blk + j;                  // feat
    if (blk < 6) return 35 + 16 * (blk - 2) + j;       // geo_fea
    // ob_view and the bias sit at j = 0, 4, 8 and 12: K-step t of an MFMA block takes the indices 4 g + t, so all four fall
    // into K-step 0 and the other three K-steps of the block, all zero, are not issued (forward: 18 of 232 MFMAs per tile)
    return (j & 3) ? -2 : (j < 12 ? 32 + (j >> 2) : -1);
}
// head (0 opacity, 1 colour, 2 cov), first output tile and number of valid outputs of output tile ot
__device__ __forceinline__ void mh_head_of(int ot, int& head, int& ot0, int& nout) {
    if (ot == 0) { head = 0; ot0 = 0; nout = MH_NO; }
    else if (ot < 3) { head = 1; ot0 = 1; nout = MH_NC; }
    else { head = 2; ot0 = 3; nout = MH_NV; }
}

struct MhWeights {
    const float* w1;   // [96][99] stacked first layers (opacity, colour, cov)
    const float* b1;   // [96]
    const float* w2[3];  // [10][32], [30][32], [70][32]
    const float* b2[3];
};

// Output activations through v_exp_f32 + v_rcp_f32 (1 ulp each): a wave spends a third of the forward kernel in them and
// in the output stores (phase-stamped build), and libm's tanhf alone is ~50 instructions.  |error| <= 2e-7 absolute;
// exp overflow / underflow give exactly +-1 / 0 / 1.
__device__ __forceinline__ void mh_store16(float* p, float a, float b, float c, float d) {     // dword-aligned address
    const f4 q = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(q) : "memory");
}
// (__frcp_rn is the correctly rounded reciprocal: a dozen instructions of division sequence each; the instruction itself)
__device__ __forceinline__ float mh_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float mh_tanh(float z) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * z)); }

// ---------------------------------------------------------------- forward
__global__ void __launch_bounds__(64 * MH_WAVES, 2)
mlp_heads_forward_kernel(int64_t V, const float* __restrict__ feat, int ldf, const float* __restrict__ anchor,
                         const float* __restrict__ campos, const float* __restrict__ geo_a,
                         const float* __restrict__ geo_b, MhWeights w, f4* __restrict__ hidden_save, float* __restrict__ out_o, float* __restrict__ out_c,
                         float* __restrict__ out_v) {
    __shared__ f4 A1[MH_MT][MH_KB][64];    // layer-1 A operands: 4 K-steps per ds_read_b128
    __shared__ f4 A2[MH_OT][2][64];        // layer-2 A operands
    __shared__ float B2[MH_OT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < MH_MT * MH_KB * 64; e += blockDim.x) {
        const int l = e & 63, blk = (e >> 6) % MH_KB, mt = e / (64 * MH_KB);
        const int row = 16 * mt + (l & 15), g = l >> 4;
        f4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = mh_kmap(blk, 4 * g + t);
            v[t] = c >= 0 ? w.w1[row * MH_IN + c] : (c == -1 ? w.b1[row] : 0.0f);
        }
        A1[mt][blk][l] = v;
    }
    for (int e = tid; e < MH_OT * 2 * 64; e += blockDim.x) {
        const int l = e & 63, ml = (e >> 6) & 1, ot = e >> 7;
        int head, ot0, nout;
        mh_head_of(ot, head, ot0, nout);
        const int orow = 16 * (ot - ot0) + (l & 15), g = l >> 4;
        f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = orow < nout ? w.w2[head][orow * MH_HID + 16 * ml + 4 * g + r] : 0.0f;
        A2[ot][ml][l] = v;
    }
    for (int e = tid; e < MH_OT * 16; e += blockDim.x) {
        int head, ot0, nout;
        mh_head_of(e >> 4, head, ot0, nout);
        const int orow = 16 * ((e >> 4) - ot0) + (e & 15);
        B2[e] = orow < nout ? w.b2[head][orow] : 0.0f;
    }
    __syncthreads();
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    const int n = lane & 15, g = lane >> 4;
    const int64_t tiles = (V + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * MH_WAVES + wave; tile < tiles; tile += (int64_t)gridDim.x * MH_WAVES) {
        const int64_t v = tile * 16 + n;
        const bool valid = v < V;
        const int64_t vc = valid ? v : V - 1;       // loads stay in bounds, results of padding anchors are dropped
        f4 xb[MH_KB];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) xb[blk] = *(const f4*)(feat + vc * ldf + 16 * blk + 4 * g);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)   // geo_fea = (geo_a | geo_b): the two halves come from two GEMMs, never concatenated
            xb[2 + blk] = *(const f4*)((blk < 2 ? geo_a : geo_b) + vc * (MH_GEO / 2) + 16 * (blk & 1) + 4 * g);
        {
            const float ox = anchor[3 * vc] - cx, oy = anchor[3 * vc + 1] - cy, oz = anchor[3 * vc + 2] - cz;
            const float inv = 1.0f / sqrtf((ox * ox + oy * oy) + oz * oz);
            xb[6] = f4{g == 0 ? ox * inv : (g == 1 ? oy * inv : (g == 2 ? oz * inv : 1.0f)), 0.0f, 0.0f, 0.0f};
        }
        f4 h[MH_MT];
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) h[mt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        // consecutive MFMAs go to DIFFERENT accumulators: the same accumulator can be re-issued after 40 cycles, an
        // independent one after 32
        // (A operands one block ahead; the scheduling barriers keep the compiler from hoisting all 42 LDS reads)
        f4 a1[2][MH_MT];
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) a1[0][mt] = A1[mt][0][lane];
#pragma unroll
        for (int blk = 0; blk < MH_KB; ++blk) {
            if (blk + 1 < MH_KB) {
#pragma unroll
                for (int mt = 0; mt < MH_MT; ++mt) a1[(blk + 1) & 1][mt] = A1[mt][blk + 1][lane];
            }
#pragma unroll
            for (int t = 0; t < (blk == 6 ? 1 : 4); ++t)      // block 6: (ob, bias) in K-step 0, zeros behind (mh_kmap)
#pragma unroll
                for (int mt = 0; mt < MH_MT; ++mt) h[mt] = mfma4(a1[blk & 1][mt][t], xb[blk][t], h[mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(h[mt][r], 0.0f);
            hidden_save[(tile * MH_MT + mt) * 64 + lane] = h[mt];     // register layout, coalesced 1 KB
        }
        f4 o[MH_OT];
#pragma unroll
        for (int ot = 0; ot < MH_OT; ++ot) o[ot] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ml = 0; ml < 2; ++ml) {
            f4 a[MH_OT];
#pragma unroll
            for (int ot = 0; ot < MH_OT; ++ot) a[ot] = A2[ot][ml][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ot = 0; ot < MH_OT; ++ot) {
                    const int head = ot == 0 ? 0 : (ot < 3 ? 1 : 2);
                    o[ot] = mfma4(a[ot][r], h[2 * head + ml][r], o[ot]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (valid) {
#pragma unroll
            for (int ot = 0; ot < MH_OT; ++ot) {
                const int head = ot == 0 ? 0 : (ot < 3 ? 1 : 2), ot0 = ot == 0 ? 0 : (ot < 3 ? 1 : 3);
                const int col0 = 16 * (ot - ot0) + 4 * g;     // first of this lane's four output columns
                // the lane's four columns in one 16-byte store (rows are only 8-byte aligned: gfx950 needs dword alignment
                // for multi-dword accesses, the compiler does not know), two columns where the head ends mid-way: as
                // 8-byte stores a store instruction left 64 fragments 16 bytes apart and the kernel queued behind them
                float z[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zr = o[ot][r] + B2[16 * ot + 4 * g + r];
                    z[r] = head == 0 ? mh_tanh(zr) : (head == 1 ? mh_sigmoid(zr) : zr);
                }
                const int nout = head == 0 ? MH_NO : (head == 1 ? MH_NC : MH_NV);
                float* dst = (head == 0 ? out_o : (head == 1 ? out_c : out_v)) + v * nout + col0;
                if (col0 + 3 < nout) mh_store16(dst, z[0], z[1], z[2], z[3]);
                else if (col0 + 1 < nout) *(float2*)dst = make_float2(z[0], z[1]);
            }
        }
    }
}

// ---------------------------------------------------------------- backward
constexpr int MH_AS = 20;     // row stride (floats) of the wave's A staging buffer  [128 rows][16 anchors + pad]
constexpr int MH_XS = 116;    // row stride of the X staging buffer  [16 anchors][112 + pad]
constexpr int MH_STAGE_A = 128 * MH_AS;                                    // dZ^T [128][.], then dPre^T [96][.]
constexpr int MH_STAGE_B = 96 * MH_AS > 16 * MH_XS ? 96 * MH_AS : 16 * MH_XS;   // H^T [96][.], then X [16][.]

__global__ void __launch_bounds__(64 * MH_WAVES, 1)
mlp_heads_backward_kernel(int64_t V, const float* __restrict__ feat, int ldf, const float* __restrict__ anchor,
                          const float* __restrict__ campos, const float* __restrict__ geo_a,
                          const float* __restrict__ geo_b, MhWeights w, const f4* __restrict__ hidden_save, const float* __restrict__ out_o,
                          const float* __restrict__ out_c, const float* __restrict__ g_o,
                          const float* __restrict__ g_c, const float* __restrict__ g_v,
                          float* __restrict__ d_feat, float* __restrict__ d_anchor, float* __restrict__ d_geo_a,
                          float* __restrict__ d_geo_b, float* __restrict__ partial) {
    __shared__ f4 A2T[MH_OT][2][64];           // dH = W2^T dZ : A[i = hidden][k = output]
    __shared__ f4 A1T[MH_KB][MH_MT][64];       // dX = W1^T dPre: A[i = input feature][k = hidden]
    __shared__ __attribute__((aligned(16))) float stA[MH_WAVES][MH_STAGE_A];
    __shared__ __attribute__((aligned(16))) float stB[MH_WAVES][MH_STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < MH_OT * 2 * 64; e += blockDim.x) {
        const int l = e & 63, ml = (e >> 6) & 1, q = e >> 7;
        int head, ot0, nout;
        mh_head_of(q, head, ot0, nout);
        const int hid = 16 * ml + (l & 15), g = l >> 4;
        f4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int orow = 16 * (q - ot0) + 4 * g + t;
            v[t] = orow < nout ? w.w2[head][orow * MH_HID + hid] : 0.0f;
        }
        A2T[q][ml][l] = v;
    }
    for (int e = tid; e < MH_KB * MH_MT * 64; e += blockDim.x) {
        const int l = e & 63, mt = (e >> 6) % MH_MT, ft = e / (64 * MH_MT);
        const int c = mh_kmap(ft, l & 15), g = l >> 4;
        f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = c >= 0 ? w.w1[(16 * mt + 4 * g + r) * MH_IN + c] : 0.0f;
        A1T[ft][mt][l] = v;
    }
    __syncthreads();
    float* sa = stA[wave];
    float* sb = stB[wave];
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    const int n = lane & 15, g = lane >> 4;
    f4 aW1[MH_MT][MH_KB], aW2[MH_OT][2];
    float db2[2] = {0.0f, 0.0f};
#pragma unroll
    for (int mt = 0; mt < MH_MT; ++mt)
#pragma unroll
        for (int ft = 0; ft < MH_KB; ++ft) aW1[mt][ft] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ot = 0; ot < MH_OT; ++ot) aW2[ot][0] = aW2[ot][1] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t tiles = (V + 15) / 16;
    // dZ = upstream gradient x activation derivative, in B-operand layout (lane (n, g): outputs 16 q + 4 g + t), and the
    // saved hidden layer of one tile.  Software pipeline: the loads of tile i+1 are issued in the middle of tile i,
    // into the registers of dz / h that have just died (one wave per SIMD: nobody else would hide the latency).
    // load_tile only ISSUES the loads (raw upstream gradients ru, raw outputs ry of the two heads with an activation,
    // hidden layer h); the activation derivatives are applied at the top of the next iteration (finish_tile).  Doing the
    // arithmetic inside load_tile made the compiler wait for every load pair right there -- ten exposed memory
    // latencies per tile with one wave per SIMD, as much time as all the MFMAs of the tile.
    f4 dz[MH_OT], h[MH_MT], ru[MH_OT], ry[3];
    auto load_tile = [&](int64_t tile) {
        const int64_t v = tile * 16 + n;
        const bool valid = v < V;
#pragma unroll
        for (int q = 0; q < MH_OT; ++q) {
            const int head = q == 0 ? 0 : (q < 3 ? 1 : 2), ot0 = q == 0 ? 0 : (q < 3 ? 1 : 3);
#pragma unroll
            for (int t = 0; t < 4; t += 2) {     // rows are 8-byte aligned and the output counts even: float2 loads
                const int col = 16 * (q - ot0) + 4 * g + t;
                float2 u = make_float2(0.0f, 0.0f), y = make_float2(0.0f, 0.0f);
                if (valid) {
                    if (head == 0) {
                        if (col < MH_NO) { y = *(const float2*)(out_o + v * MH_NO + col); u = *(const float2*)(g_o + v * MH_NO + col); }
                    } else if (head == 1) {
                        if (col < MH_NC) { y = *(const float2*)(out_c + v * MH_NC + col); u = *(const float2*)(g_c + v * MH_NC + col); }
                    } else if (col < MH_NV) {
                        u = *(const float2*)(g_v + v * MH_NV + col);
                    }
                }
                ru[q][t] = u.x;
                ru[q][t + 1] = u.y;
                if (q < 3) {
                    ry[q][t] = y.x;
                    ry[q][t + 1] = y.y;
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) {     // predicated load, not load + select: a select would wait for the load here
            f4 hv = f4{0.0f, 0.0f, 0.0f, 0.0f};
            if (valid) hv = hidden_save[(tile * MH_MT + mt) * 64 + lane];
            h[mt] = hv;
        }
    };
    // dZ = upstream gradient x activation derivative (tanh: 1 - y^2, sigmoid: y (1 - y), none for the covariance head)
    auto finish_tile = [&]() {
#pragma unroll
        for (int q = 0; q < MH_OT; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (q == 0) dz[q][t] = ru[q][t] * (1.0f - ry[q][t] * ry[q][t]);
                else if (q < 3) dz[q][t] = ru[q][t] * (ry[q][t] * (1.0f - ry[q][t]));
                else dz[q][t] = ru[q][t];
            }
    };
    const int64_t tile0 = (int64_t)blockIdx.x * MH_WAVES + wave, tstep = (int64_t)gridDim.x * MH_WAVES;
    if (tile0 < tiles) load_tile(tile0);
    // The phases of a tile are fenced with sched_barrier(0): left alone, the scheduler (one wave per SIMD, 500 registers)
    // moves pieces of one phase into another and ends up 12 % slower than the fenced order (found in round 2 with a
    // phase-stamped developer build: the stamps' fences alone made the kernel faster; profiles/HISTORY.md).
    for (int64_t tile = tile0; tile < tiles; tile += tstep) {
        const int64_t v = tile * 16 + n;
        const bool valid = v < V;
        const int64_t vc = valid ? v : V - 1;
        finish_tile();
        // ---- stage dZ^T [output][anchor] and H^T [hidden][anchor] for the dW2 contraction over the anchors
#pragma unroll
        for (int q = 0; q < MH_OT; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) sa[(16 * q + 4 * g + t) * MH_AS + n] = dz[q][t];
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sb[(16 * mt + 4 * g + r) * MH_AS + n] = h[mt][r];
        __builtin_amdgcn_sched_barrier(0);
        // ---- dH^T = W2^T dZ^T per head; dPre = dH where the hidden unit was active
        f4 dpre[MH_MT];
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) dpre[mt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < MH_OT; ++q) {       // the two hidden tiles of the block's head alternate: no dependent issue
            const int head = q == 0 ? 0 : (q < 3 ? 1 : 2);
            const f4 a0 = A2T[q][0][lane], a1 = A2T[q][1][lane];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dpre[2 * head] = mfma4(a0[t], dz[q][t], dpre[2 * head]);
                dpre[2 * head + 1] = mfma4(a1[t], dz[q][t], dpre[2 * head + 1]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dpre[mt][r] = h[mt][r] > 0.0f ? dpre[mt][r] : 0.0f;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's staging writes have landed (wave-private buffers)
        // ---- dW2[out][hidden] += dZ^T H ; db2[out] += sum over the anchors
#pragma unroll
        for (int ot = 0; ot < MH_OT; ++ot) {
            const int head = ot == 0 ? 0 : (ot < 3 ? 1 : 2);
            const f4 a = *(const f4*)&sa[(16 * ot + n) * MH_AS + 4 * g];        // A[i = output][k = anchor 4 g + s]
            const f4 b0 = *(const f4*)&sb[(16 * (2 * head) + n) * MH_AS + 4 * g];       // B[k = anchor][j = hidden]
            const f4 b1 = *(const f4*)&sb[(16 * (2 * head + 1) + n) * MH_AS + 4 * g];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                aW2[ot][0] = mfma4(a[s], b0[s], aW2[ot][0]);
                aW2[ot][1] = mfma4(a[s], b1[s], aW2[ot][1]);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float* row = &sa[(lane + 64 * half) * MH_AS];
            float sum = 0.0f;
#pragma unroll
            for (int c = 0; c < 16; c += 4) { const f4 x = *(const f4*)&row[c]; sum += (x[0] + x[1]) + (x[2] + x[3]); }
            db2[half] += sum;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tile + tstep < tiles) load_tile(tile + tstep);      // dz and h are dead from here on: next tile's loads
        // ---- dX^T = W1^T dPre^T -> d feat, d geo_fea, d ob_view
        f4 xb[MH_KB];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) xb[blk] = *(const f4*)(feat + vc * ldf + 16 * blk + 4 * g);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
            xb[2 + blk] = *(const f4*)((blk < 2 ? geo_a : geo_b) + vc * (MH_GEO / 2) + 16 * (blk & 1) + 4 * g);
        const float ax = anchor[3 * vc], ay = anchor[3 * vc + 1], az = anchor[3 * vc + 2];   // used after the dX MFMAs
        __builtin_amdgcn_sched_barrier(0);
        f4 dx[MH_KB];
#pragma unroll
        for (int ft = 0; ft < MH_KB; ++ft) dx[ft] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) {
            f4 a[MH_KB];
#pragma unroll
            for (int ft = 0; ft < MH_KB; ++ft) a[ft] = A1T[ft][mt][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ft = 0; ft < MH_KB; ++ft) dx[ft] = mfma4(a[ft][r], dpre[mt][r], dx[ft]);
        }
        // Nothing that needs one of the loads above may be scheduled before the 168 MFMAs above: vmcnt retires in order,
        // so the first use waits for ALL of them (the compiler had hoisted this 1 / |o| arithmetic over the MFMAs: one
        // exposed memory round trip per tile).
        __builtin_amdgcn_sched_barrier(0);
        const float ox = ax - cx, oy = ay - cy, oz = az - cz;
        const float inv = 1.0f / sqrtf((ox * ox + oy * oy) + oz * oz);
        xb[6] = f4{g == 0 ? ox * inv : (g == 1 ? oy * inv : (g == 2 ? oz * inv : 1.0f)), 0.0f, 0.0f, 0.0f};
        // d ob sits in rows 0, 4, 8 of block 6 (mh_kmap): register 0 of the lanes (n, 0), (n, 1), (n, 2)
        const float dob1 = __shfl(dx[6][0], n + 16, 64), dob2 = __shfl(dx[6][0], n + 32, 64);
        if (valid) {
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) *(f4*)(d_feat + v * MH_FEAT + 16 * ft + 4 * g) = dx[ft];
#pragma unroll
            for (int ft = 2; ft < 6; ++ft) *(f4*)((ft < 4 ? d_geo_a : d_geo_b) + v * (MH_GEO / 2) + 16 * (ft & 1) + 4 * g) = dx[ft];
            if (g == 0) {   // ob = o / |o|:  d o = (d ob - ob <ob, d ob>) / |o|
                const float ux = ox * inv, uy = oy * inv, uz = oz * inv;
                const float dot = (ux * dx[6][0] + uy * dob1) + uz * dob2;
                d_anchor[3 * v] = (dx[6][0] - ux * dot) * inv;
                d_anchor[3 * v + 1] = (dob1 - uy * dot) * inv;
                d_anchor[3 * v + 2] = (dob2 - uz * dot) * inv;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- dW1[hidden][k] += dPre^T X : restage (the dW2 reads above are done: same wave, program order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sa[(16 * mt + 4 * g + r) * MH_AS + n] = dpre[mt][r];
#pragma unroll
        for (int blk = 0; blk < MH_KB; ++blk) *(f4*)&sb[n * MH_XS + 16 * blk + 4 * g] = valid ? xb[blk] : f4{0.0f, 0.0f, 0.0f, 0.0f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float bx[MH_KB][4];
#pragma unroll
        for (int ft = 0; ft < MH_KB; ++ft)
#pragma unroll
            for (int s = 0; s < 4; ++s) bx[ft][s] = sb[(4 * g + s) * MH_XS + 16 * ft + n];     // B[k = anchor 4 g + s][j = feature]
#pragma unroll
        for (int mt = 0; mt < MH_MT; ++mt) {
            const f4 a = *(const f4*)&sa[(16 * mt + n) * MH_AS + 4 * g];                        // A[i = hidden][k = anchor]
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ft = 0; ft < MH_KB; ++ft) aW1[mt][ft] = mfma4(a[s], bx[ft][s], aW1[mt][ft]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads finished before the next tile's staging writes
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- this wave's partial sums: accumulator (lane (j, gi), register r) = row 4 gi + r, column j of its 16x16 tile
    // (the lane's coordinates are re-derived behind an opaque copy: computed from the loop's own n / g the compiler forms the
    // store addresses BEFORE the anchor loop and, all 512 registers being taken inside it, spills them to scratch)
    int le = threadIdx.x;
    asm volatile("" : "+v"(le));
    const int ne = le & 15, ge = (le >> 4) & 3, we = le >> 6, lane_e = le & 63;
    float* p = partial + ((size_t)blockIdx.x * MH_WAVES + we) * MH_PART;
#pragma unroll
    for (int mt = 0; mt < MH_MT; ++mt)
#pragma unroll
        for (int ft = 0; ft < MH_KB; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(16 * mt + 4 * ge + r) * (MH_KB * 16) + 16 * ft + ne] = aW1[mt][ft][r];
    float* p2 = p + MH_MT * 16 * MH_KB * 16;
#pragma unroll
    for (int ot = 0; ot < MH_OT; ++ot)
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int r = 0; r < 4; ++r) p2[(16 * ot + 4 * ge + r) * MH_HID + 16 * ml + ne] = aW2[ot][ml][r];
    float* p3 = p2 + MH_OT * 16 * MH_HID;
    p3[lane_e] = db2[0];
    p3[lane_e + 64] = db2[1];
}

// sums the per-wave partials in wave order and un-permutes them into the parameter layouts
__global__ void __launch_bounds__(256)
mlp_heads_reduce_kernel(int nparts, const float* __restrict__ partial, float* __restrict__ d_w1, float* __restrict__ d_b1,
                        float* __restrict__ d_w2o, float* __restrict__ d_b2o, float* __restrict__ d_w2c,
                        float* __restrict__ d_b2c, float* __restrict__ d_w2v, float* __restrict__ d_b2v) {
    // 32 outputs per workgroup x 8 groups of partials (group g takes partials g, g + 8, ...; eight loads in flight), the
    // groups' sums added in group order: still one fixed order, but 0.03 instead of 0.24 ms for the 1024 partials of a
    // full grid (one thread walking all of them was pure load latency)
    __shared__ float red[8][32];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float acc = 0.0f;
    if (e < MH_PART) {
        int p = grp;
        for (; p + 56 < nparts; p += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(p + 8 * u) * MH_PART + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; p < nparts; p += 8) acc += partial[(size_t)p * MH_PART + e];
    }
    red[grp][threadIdx.x & 31] = acc;
    __syncthreads();
    if (threadIdx.x >= 32 || e >= MH_PART) return;
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][threadIdx.x];
    const int n1 = MH_MT * 16 * MH_KB * 16, n2 = MH_OT * 16 * MH_HID;
    if (e < n1) {
        const int row = e / (MH_KB * 16), k = e % (MH_KB * 16);
        const int c = mh_kmap(k >> 4, k & 15);
        if (c >= 0) d_w1[row * MH_IN + c] = s;
        else if (c == -1) d_b1[row] = s;
    } else if (e < n1 + n2) {
        const int orow16 = (e - n1) / MH_HID, hid = (e - n1) % MH_HID;
        int head, ot0, nout;
        mh_head_of(orow16 >> 4, head, ot0, nout);
        const int orow = orow16 - 16 * ot0;
        float* dst = head == 0 ? d_w2o : (head == 1 ? d_w2c : d_w2v);
        if (orow < nout) dst[orow * MH_HID + hid] = s;
    } else {
        const int orow16 = e - n1 - n2;
        int head, ot0, nout;
        mh_head_of(orow16 >> 4, head, ot0, nout);
        const int orow = orow16 - 16 * ot0;
        float* dst = head == 0 ? d_b2o : (head == 1 ? d_b2c : d_b2v);
        if (orow < nout) dst[orow] = s;
    }
}

// ---------------------------------------------------------------- launchers
static int mh_grid(int64_t V, int per_cu = 1) {
    const int64_t tiles = (V + 15) / 16;
    const int64_t want = (tiles + MH_WAVES - 1) / MH_WAVES, cap = 256 * per_cu;
    return (int)(want < cap ? (want > 0 ? want : 1) : cap);      // `per_cu` workgroups per CU, grid-stride over the tiles
}

size_t mlp_heads_hidden_bytes(int64_t V) { return align_up((size_t)((V + 15) / 16) * MH_MT * 64 * sizeof(f4)); }
size_t mlp_heads_partial_bytes(int64_t V) { return align_up((size_t)mh_grid(V) * MH_WAVES * MH_PART * sizeof(float)); }

void launch_mlp_heads_forward(int64_t V, const float* feat, int ldf, const float* anchor, const float* campos, const float* geo_a,
                              const float* geo_b, const float* w1, const float* b1, const float* w2o, const float* b2o, const float* w2c,
                              const float* b2c, const float* w2v, const float* b2v, void* hidden_save, float* out_o,
                              float* out_c, float* out_v, hipStream_t st) {
    MhWeights w{w1, b1, {w2o, w2c, w2v}, {b2o, b2c, b2v}};
    mlp_heads_forward_kernel<<<mh_grid(V, 2), 64 * MH_WAVES, 0, st>>>(V, feat, ldf, anchor, campos, geo_a, geo_b, w, (f4*)hidden_save, out_o,
                                                                   out_c, out_v);
}

void launch_mlp_heads_backward(int64_t V, const float* feat, int ldf, const float* anchor, const float* campos, const float* geo_a,
                               const float* geo_b, const float* w1, const float* w2o, const float* w2c, const float* w2v,
                               const void* hidden_save, const float* out_o, const float* out_c, const float* g_o,
                               const float* g_c, const float* g_v, void* partial, float* d_feat, float* d_anchor,
                               float* d_geo_a, float* d_geo_b, float* d_w1, float* d_b1, float* d_w2o, float* d_b2o, float* d_w2c,
                               float* d_b2c, float* d_w2v, float* d_b2v, hipStream_t st) {
    MhWeights w{w1, nullptr, {w2o, w2c, w2v}, {nullptr, nullptr, nullptr}};
    const int grid = mh_grid(V);
    mlp_heads_backward_kernel<<<grid, 64 * MH_WAVES, 0, st>>>(V, feat, ldf, anchor, campos, geo_a, geo_b, w, (const f4*)hidden_save,
                                                              out_o, out_c, g_o, g_c, g_v, d_feat, d_anchor, d_geo_a, d_geo_b,
                                                              (float*)partial);
    mlp_heads_reduce_kernel<<<(MH_PART + 31) / 32, 256, 0, st>>>(grid * MH_WAVES, (const float*)partial, d_w1, d_b1, d_w2o,
                                                                  d_b2o, d_w2c, d_b2c, d_w2v, d_b2v);
}


}  // namespace scr
