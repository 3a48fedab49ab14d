// splatco_amd/csrc/adam.hip -- the optimizer step of the sharded training step as one streaming pass (gfx950).
//
// Reference: train.py:310-312 (`gaussians.optimizer.step()`), the optimizer is torch.optim.Adam(l, lr=0.0, eps=1e-15)
// over one group per per-anchor parameter plus the MLP / plane groups (scene/gaussian_model.py:520-575); no weight decay,
// no amsgrad.  At 20 M anchors the four per-anchor parameters are 5.7 GB: the step reads parameter, gradient and both
// moments and writes three of them back -- 40 GB, a tenth of the cfg4 step -- so it is priced against the copy probe,
// not against anything arithmetic.  Shape: one float4 of each of the four streams per lane and nothing else in flight
// (tools/exp/copy_probe.hip: the flat shape reaches 6.3 TB/s, eight loads per lane 3.6); up to ADAM_MAX tensors share a
// launch (the MLP / plane group is dozens of small tensors), a workgroup finds its tensor in a prefix table that lives
// in the kernel arguments.
//
// Arithmetic = torch's (fused_adam_utils.cuh; the single-tensor path gives the same numbers):
//   m  = m + (1 - beta1) (g - m)                      v = beta2 v + ((1 - beta2) g) g          (1 - beta in double, then fp32)
//   p -= (lr / bias1) * m / (sqrt(v) / sqrt(bias2) + eps)          bias_i = 1 - beta_i^step  (the two scalars formed in double by
//   the caller, as torch's Python does, and rounded to fp32 once)
#include "adam.h"

namespace scr {

typedef float adam_f4 __attribute__((ext_vector_type(4)));

struct AdamArgs {
    int n;
    float beta1, beta2, omb1, omb2, eps;    // omb = 1 - beta, rounded ONCE from double (1 - 0.999f is 4.7e-5 off)
    uint32_t first_block[ADAM_MAX + 1];     // tensor t owns workgroups first_block[t] .. first_block[t + 1] - 1
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int64_t numel[ADAM_MAX];
    float step_size[ADAM_MAX];              // lr / bias1 (the caller's double quotient, rounded once)
    float bias2_sqrt[ADAM_MAX];             // torch divides by it (no reciprocal)
    uint8_t aligned[ADAM_MAX];              // all four pointers on 16 bytes
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2, float eps,
                                         float step_size, float bias2_sqrt) {
    m = m + omb1 * (g - m);
    v = b2 * v + (omb2 * g) * g;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a) {
    int t = 0;
    while (t + 1 < a.n && blockIdx.x >= a.first_block[t + 1]) ++t;      // uniform: scalar registers
    const int64_t n = a.numel[t];
    const int64_t e0 = ((int64_t)(blockIdx.x - a.first_block[t]) * 256 + threadIdx.x) * 4;
    if (e0 >= n) return;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const float o1 = a.omb1, b2 = a.beta2, o2 = a.omb2, eps = a.eps, ss = a.step_size[t], bs = a.bias2_sqrt[t];
    if (a.aligned[t] && e0 + 3 < n) {
        adam_f4 P = *(const adam_f4*)(p + e0), M = *(const adam_f4*)(m + e0), V = *(const adam_f4*)(v + e0);
        const adam_f4 G = __builtin_nontemporal_load((const adam_f4*)(g + e0));      // the gradient is dead after this pass
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = P[j], mj = M[j], vj = V[j];
            adam_one(pj, G[j], mj, vj, o1, b2, o2, eps, ss, bs);
            P[j] = pj;
            M[j] = mj;
            V[j] = vj;
        }
        *(adam_f4*)(p + e0) = P;
        *(adam_f4*)(m + e0) = M;
        *(adam_f4*)(v + e0) = V;
    } else {
        for (int64_t e = e0; e < min(e0 + 4, n); ++e) {
            float P = p[e], M = m[e], V = v[e];
            adam_one(P, g[e], M, V, o1, b2, o2, eps, ss, bs);
            p[e] = P;
            m[e] = M;
            v[e] = V;
        }
    }
}

int launch_adam(int n, const scr_adam_tensor* ts, double beta1, double beta2, double eps, hipStream_t st) {
    for (int t0 = 0; t0 < n; t0 += ADAM_MAX) {
        AdamArgs a;
        a.n = min(ADAM_MAX, n - t0);
        a.beta1 = (float)beta1;
        a.beta2 = (float)beta2;
        a.omb1 = (float)(1.0 - beta1);
        a.omb2 = (float)(1.0 - beta2);
        a.eps = (float)eps;
        uint64_t blocks = 0;
        for (int t = 0; t < a.n; ++t) {
            const scr_adam_tensor& x = ts[t0 + t];
            a.first_block[t] = (uint32_t)blocks;
            blocks += (uint64_t)((x.numel + 1023) / 1024);     // 256 lanes x one float4
            if (blocks > 0x7fffffffull) return 1;
            a.p[t] = x.param;
            a.g[t] = x.grad;
            a.m[t] = x.exp_avg;
            a.v[t] = x.exp_avg_sq;
            a.numel[t] = x.numel;
            a.step_size[t] = (float)x.step_size;
            a.bias2_sqrt[t] = (float)x.bias_correction2_sqrt;
            a.aligned[t] = ((((uintptr_t)x.param | (uintptr_t)x.grad | (uintptr_t)x.exp_avg | (uintptr_t)x.exp_avg_sq) & 15u) == 0);
        }
        a.first_block[a.n] = (uint32_t)blocks;
        if (blocks) adam_kernel<<<(unsigned)blocks, 256, 0, st>>>(a);
    }
    return 0;
}

}  // namespace scr
