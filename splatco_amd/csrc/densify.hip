// splatco_amd/csrc/densify.hip -- densification statistics on the device (gfx950): the consumer of the
// rasterizer's dL/dmeans2D output, GaussianModel.training_statis (scene/gaussian_model.py:761-782).
//
// The reference updates four accumulators with a chain of boolean-mask scatters over all N*k offsets.  Here the
// update is split in two streaming kernels over the V VISIBLE anchors of the view, so that in the sharded --mv
// step the rank that rendered the last view computes the increments once, broadcasts V*(k+2) words, and every
// rank applies them (train_step.sync_densification_stats):
//
//   compute:  inc_opacity[v] = sum_slot max(neural_opacity[v,slot], 0)
//             inc_grad[v,slot] = |dL/dmeans2D[p, :2]|  when candidate (v,slot) was selected (out_index = p >= 0,
//                                the expansion kernel's compaction index) and Gaussian p was rendered
//                                (update_filter[p]); -1 otherwise (a norm is never negative)
//   apply:    a = visible_index[v]:  opacity_accum[a] += inc_opacity[v];  anchor_demon[a] += 1;
//             offset_gradient_accum[a*k+slot] += inc_grad, offset_denom[a*k+slot] += 1  where inc_grad >= 0
//
// Every (anchor, slot) is touched by exactly one thread: no atomics, bit-reproducible.  HBM-bound:
// compute reads 8 B per candidate + 9 B per rendered Gaussian, apply 8 B per candidate read-modify-write.
#include "common.h"

namespace scr {

__global__ void __launch_bounds__(256)
statis_compute_kernel(int64_t V, int k, const float* __restrict__ neural_opacity, const int32_t* __restrict__ out_index,
                      const uint8_t* __restrict__ update_filter, const float* __restrict__ grad, int gstride,
                      float* __restrict__ inc_opacity, float* __restrict__ inc_grad) {
    // one thread per candidate; the k candidates of an anchor are summed by the thread of slot 0
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= V * k) return;
    const int32_t p = out_index[c];
    float g = -1.0f;
    if (p >= 0 && update_filter[p]) {
        const float gx = grad[(size_t)p * gstride], gy = grad[(size_t)p * gstride + 1];
        g = sqrtf(gx * gx + gy * gy);
    }
    inc_grad[c] = g;
    if (c % k == 0) {
        float s = 0.0f;
        for (int j = 0; j < k; ++j) s += fmaxf(neural_opacity[c + j], 0.0f);
        inc_opacity[c / k] = s;
    }
}

__global__ void __launch_bounds__(256)
statis_apply_kernel(int64_t V, int k, const int64_t* __restrict__ visible_index, const float* __restrict__ inc_opacity,
                    const float* __restrict__ inc_grad, float* __restrict__ opacity_accum,
                    float* __restrict__ anchor_demon, float* __restrict__ offset_gradient_accum,
                    float* __restrict__ offset_denom) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= V * k) return;
    const int64_t v = c / k;
    const int slot = (int)(c - v * k);
    const int64_t a = visible_index[v];
    const float g = inc_grad[c];
    if (g >= 0.0f) {
        offset_gradient_accum[a * k + slot] += g;
        offset_denom[a * k + slot] += 1.0f;
    }
    if (slot == 0) {
        opacity_accum[a] += inc_opacity[v];
        anchor_demon[a] += 1.0f;
    }
}

void launch_statis_compute(int64_t V, int k, const float* neural_opacity, const int32_t* out_index,
                           const uint8_t* update_filter, const float* grad, int gstride, float* inc_opacity,
                           float* inc_grad, hipStream_t st) {
    const int64_t n = V * k;
    if (n <= 0) return;
    statis_compute_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(V, k, neural_opacity, out_index, update_filter,
                                                                     grad, gstride, inc_opacity, inc_grad);
}

void launch_statis_apply(int64_t V, int k, const int64_t* visible_index, const float* inc_opacity, const float* inc_grad,
                         float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                         hipStream_t st) {
    const int64_t n = V * k;
    if (n <= 0) return;
    statis_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(V, k, visible_index, inc_opacity, inc_grad,
                                                                   opacity_accum, anchor_demon, offset_gradient_accum,
                                                                   offset_denom);
}

}  // namespace scr
