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
statis_compute_kernel(int64_t V, int k, int per, const float* __restrict__ neural_opacity, const int32_t* __restrict__ out_index,
                      const uint8_t* __restrict__ update_filter, const float* __restrict__ grad, int gstride,
                      float* __restrict__ inc_opacity, float* __restrict__ inc_grad) {
    // one thread per candidate; a workgroup takes per = (256 / k) * k candidates -- whole anchors -- so that every load is
    // coalesced and the k opacities of an anchor meet in LDS (a thread that walked its anchor's k values alone read
    // 40 bytes per lane with six lanes of a wave active)
    __shared__ float op[256];
    const int64_t c = (int64_t)blockIdx.x * per + threadIdx.x;
    const bool in = (int)threadIdx.x < per && c < V * k;
    float g = -1.0f, o = 0.0f;
    if (in) {
        const int32_t p = out_index[c];
        o = fmaxf(neural_opacity[c], 0.0f);
        if (p >= 0 && update_filter[p]) {
            const float gx = grad[(size_t)p * gstride], gy = grad[(size_t)p * gstride + 1];
            g = sqrtf(gx * gx + gy * gy);
        }
        inc_grad[c] = g;
    }
    op[threadIdx.x] = o;
    __syncthreads();
    if (in && threadIdx.x % k == 0) {
        float s = 0.0f;
        for (int j = 0; j < k; ++j) s += op[threadIdx.x + j];     // slot order, as before
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

// k even (the reference's n_offsets = 10): two candidates of one anchor per lane, 8-byte accesses -- the 4-byte version
// above moves 3.1 TB/s at 184 M candidates (cfg4), the same lesson as the anchor gather's
__global__ void __launch_bounds__(256)
statis_compute2_kernel(int64_t V, int k, int per, const float* __restrict__ neural_opacity, const int32_t* __restrict__ out_index,
                       const uint8_t* __restrict__ update_filter, const float* __restrict__ grad, int gstride,
                       float* __restrict__ inc_opacity, float* __restrict__ inc_grad) {
    __shared__ float op[512];
    const int64_t c = (int64_t)blockIdx.x * per + 2 * threadIdx.x;       // per = (512 / k) * k candidates: whole anchors
    const bool in = 2 * (int)threadIdx.x < per && c < V * k;
    float o0 = 0.0f, o1 = 0.0f;
    if (in) {
        const int2 p = *(const int2*)(out_index + c);
        const float2 no = *(const float2*)(neural_opacity + c);
        o0 = fmaxf(no.x, 0.0f);
        o1 = fmaxf(no.y, 0.0f);
        // filter byte and gradient of both candidates asked for together, from clamped indices (an unselected candidate
        // reads Gaussian 0's): as `p >= 0 && filter[p]` then `if (...) grad[p]` they were two more dependent round trips
        const size_t q0 = (size_t)max(p.x, 0), q1 = (size_t)max(p.y, 0);
        const uint8_t f0 = update_filter[q0], f1 = update_filter[q1];
        const float gx0 = grad[q0 * gstride], gy0 = grad[q0 * gstride + 1];
        const float gx1 = grad[q1 * gstride], gy1 = grad[q1 * gstride + 1];
        float2 g = make_float2(-1.0f, -1.0f);
        if (p.x >= 0 && f0) g.x = sqrtf(gx0 * gx0 + gy0 * gy0);
        if (p.y >= 0 && f1) g.y = sqrtf(gx1 * gx1 + gy1 * gy1);
        *(float2*)(inc_grad + c) = g;
    }
    op[2 * threadIdx.x] = o0;
    op[2 * threadIdx.x + 1] = o1;
    __syncthreads();
    if (in && (2 * threadIdx.x) % k == 0) {
        float s = 0.0f;
        for (int j = 0; j < k; ++j) s += op[2 * threadIdx.x + j];     // slot order, as in the one-candidate kernel
        inc_opacity[c / k] = s;
    }
}

__global__ void __launch_bounds__(256)
statis_apply2_kernel(int64_t V, int k, const int64_t* __restrict__ visible_index, const float* __restrict__ inc_opacity,
                     const float* __restrict__ inc_grad, float* __restrict__ opacity_accum,
                     float* __restrict__ anchor_demon, float* __restrict__ offset_gradient_accum,
                     float* __restrict__ offset_denom) {
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (c >= V * k) return;
    const int64_t v = c / k;
    const int slot = (int)(c - v * k);          // even; slot + 1 < k belongs to the same anchor
    const int64_t a = visible_index[v];
    const float2 g = *(const float2*)(inc_grad + c);
    if (g.x >= 0.0f || g.y >= 0.0f) {
        float2* pa = (float2*)(offset_gradient_accum + a * k + slot);
        float2* pd = (float2*)(offset_denom + a * k + slot);
        float2 acc = *pa, den = *pd;
        if (g.x >= 0.0f) { acc.x += g.x; den.x += 1.0f; }
        if (g.y >= 0.0f) { acc.y += g.y; den.y += 1.0f; }
        *pa = acc;
        *pd = den;
    }
    if (slot == 0) {
        opacity_accum[a] += inc_opacity[v];
        anchor_demon[a] += 1.0f;
    }
}

static inline bool statis_pairs_ok(int k, const void* a, const void* b, const void* c, const void* d) {
    return k % 2 == 0 && k <= 512 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 7u) == 0);
}

void launch_statis_compute(int64_t V, int k, const float* neural_opacity, const int32_t* out_index,
                           const uint8_t* update_filter, const float* grad, int gstride, float* inc_opacity,
                           float* inc_grad, hipStream_t st) {
    const int64_t n = V * k;
    if (n <= 0) return;
    // (the pair kernel reads entry 0 of update_filter / grad for unselected candidates: both must exist)
    if (statis_pairs_ok(k, neural_opacity, out_index, inc_grad, inc_grad) && update_filter && grad) {
        const int per2 = (512 / k) * k;
        statis_compute2_kernel<<<(unsigned)((n + per2 - 1) / per2), 256, 0, st>>>(V, k, per2, neural_opacity, out_index, update_filter,
                                                                                 grad, gstride, inc_opacity, inc_grad);
        return;
    }
    const int per = k <= 256 ? (256 / k) * k : 0;
    if (!per) return;      // scr_statis_compute rejects k > 256
    statis_compute_kernel<<<(unsigned)((n + per - 1) / per), 256, 0, st>>>(V, k, per, neural_opacity, out_index, update_filter,
                                                                         grad, gstride, inc_opacity, inc_grad);
}

void launch_statis_apply(int64_t V, int k, const int64_t* visible_index, const float* inc_opacity, const float* inc_grad,
                         float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                         hipStream_t st) {
    const int64_t n = V * k;
    if (n <= 0) return;
    if (statis_pairs_ok(k, inc_grad, offset_gradient_accum, offset_denom, inc_grad)) {
        statis_apply2_kernel<<<(unsigned)((n / 2 + 255) / 256), 256, 0, st>>>(V, k, visible_index, inc_opacity, inc_grad,
                                                                           opacity_accum, anchor_demon, offset_gradient_accum,
                                                                           offset_denom);
        return;
    }
    statis_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(V, k, visible_index, inc_opacity, inc_grad,
                                                                   opacity_accum, anchor_demon, offset_gradient_accum,
                                                                   offset_denom);
}

}  // namespace scr

// ------------------------------------------------------------------ k nearest neighbours (compute_curvature)
// GaussianModel.compute_curvature (scene/gaussian_model.py:1092-1110) asks sklearn for the k+1 nearest anchors of every
// anchor on the HOST and loops over the anchors in Python.  Here: the anchors are bucketed into a uniform grid (cell
// keys sorted by the caller with torch.sort, cell_start from the sorted keys), one thread per query walks the cells of
// growing cubes around its own cell keeping its k best candidates in registers (insertion into a sorted list), and
// stops as soon as the k-th best distance is inside the cube already searched.  Exact (same neighbour SET as a brute
// force search; ties at equal distance aside).  knn_curvature_kernel then forms the neighbours' covariance and its
// eigenvalues (closed form for a symmetric 3x3, in fp64) and returns lambda_min / sum(lambda).
namespace scr {

constexpr int KNN_MAX_K = 16;

struct KnnGrid {
    float x0, y0, z0, inv_h, h;
    int nx, ny, nz;
};

__device__ __forceinline__ int knn_cell(float v, float lo, float inv_h, int n) {
    const int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// sorted_pts: points in cell order (float x,y,z per point), sorted_id: original index of every sorted point,
// cell_start[ncells + 1].  out_idx[q * k + j] = original index of the j-th nearest OTHER point of query q (by
// original index q), nearest first.
template <int K>
__global__ void __launch_bounds__(256)
knn_kernel(int64_t N, int k, KnnGrid gr, const float* __restrict__ sorted_pts, const int64_t* __restrict__ sorted_id,
           const int32_t* __restrict__ cell_start, int64_t* __restrict__ out_idx) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;      // position in cell order: neighbours in memory
    if (s >= N) return;
    const float qx = sorted_pts[3 * s], qy = sorted_pts[3 * s + 1], qz = sorted_pts[3 * s + 2];
    const int cx = knn_cell(qx, gr.x0, gr.inv_h, gr.nx), cy = knn_cell(qy, gr.y0, gr.inv_h, gr.ny),
              cz = knn_cell(qz, gr.z0, gr.inv_h, gr.nz);
    float bd[K];          // the K best so far, ascending (K >= k; registers: every index below is a compile-time constant)
    int64_t bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = 3.0e38f; bi[j] = -1; }
    float kth = 3.0e38f;  // bd[k - 1]
    const int rmax = max(gr.nx, max(gr.ny, gr.nz));
    // distance from the query to the faces of its own cell: the searched cube of radius r reaches at least r*h + that
    const float fx = (qx - gr.x0) * gr.inv_h - (float)cx, fy = (qy - gr.y0) * gr.inv_h - (float)cy,
                fz = (qz - gr.z0) * gr.inv_h - (float)cz;
    const float margin = gr.h * fminf(fminf(fminf(fx, 1.0f - fx), fminf(fy, 1.0f - fy)), fminf(fz, 1.0f - fz));
    for (int r = 0; r <= rmax; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= gr.nz) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= gr.ny) continue;
                const bool shell_yz = (dz == -r || dz == r || dy == -r || dy == r);
                for (int dx = -r; dx <= r; dx += (shell_yz || r == 0 ? 1 : 2 * r)) {     // only the cube's surface
                    const int x = cx + dx;
                    if (x < 0 || x >= gr.nx) continue;
                    const int cell = (z * gr.ny + y) * gr.nx + x;
                    for (int p = cell_start[cell]; p < cell_start[cell + 1]; ++p) {
                        if (p == s) continue;
                        const float ex = sorted_pts[3 * (int64_t)p] - qx, ey = sorted_pts[3 * (int64_t)p + 1] - qy,
                                    ez = sorted_pts[3 * (int64_t)p + 2] - qz;
                        const float d = (ex * ex + ey * ey) + ez * ez;
                        if (d < kth) {
                            float cd = d;                    // insertion into the ascending list
                            int64_t ci = sorted_id[p];
#pragma unroll
                            for (int j = 0; j < K; ++j) {
                                const bool sw = cd < bd[j];
                                const float td = bd[j];
                                const int64_t ti = bi[j];
                                bd[j] = sw ? cd : td; bi[j] = sw ? ci : ti;
                                cd = sw ? td : cd; ci = sw ? ti : ci;
                            }
#pragma unroll
                            for (int j = 0; j < K; ++j) if (j == k - 1) kth = bd[j];
                        }
                    }
                }
            }
        }
        const float reach = fmaxf(0.0f, (float)r * gr.h + margin);      // everything closer than this has been seen
        if (kth <= reach * reach) break;
    }
    const int64_t q = sorted_id[s];
#pragma unroll
    for (int j = 0; j < K; ++j) if (j < k) out_idx[q * k + j] = bi[j];
}

// curvature[q] = lambda_min / (lambda_0 + lambda_1 + lambda_2) of cov = C^T C / (k - 1), C = neighbours - their mean
__global__ void __launch_bounds__(256)
knn_curvature_kernel(int64_t N, int k, const float* __restrict__ pts, const int64_t* __restrict__ idx,
                     float* __restrict__ curvature) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= N) return;
    double m[3] = {0, 0, 0};
    for (int j = 0; j < k; ++j) {
        const int64_t p = idx[q * k + j];
        for (int c = 0; c < 3; ++c) m[c] += (double)pts[3 * p + c];
    }
    for (int c = 0; c < 3; ++c) m[c] /= k;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    for (int j = 0; j < k; ++j) {
        const int64_t p = idx[q * k + j];
        const double x = pts[3 * p] - m[0], y = pts[3 * p + 1] - m[1], z = pts[3 * p + 2] - m[2];
        a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
    }
    const double s = 1.0 / (k - 1);
    a00 *= s; a01 *= s; a02 *= s; a11 *= s; a12 *= s; a22 *= s;
    // smallest eigenvalue of a symmetric 3x3 (trigonometric closed form), fp64
    const double tr = a00 + a11 + a22, mq = tr / 3.0;
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double b00 = a00 - mq, b11 = a11 - mq, b22 = a22 - mq;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    double lmin;
    if (p2 <= 0.0) {
        lmin = mq;
    } else {
        const double p = sqrt(p2 / 6.0);
        const double det = (b00 * (b11 * b22 - a12 * a12) - a01 * (a01 * b22 - a12 * a02) + a02 * (a01 * a12 - b11 * a02)) / (p * p * p);
        const double rr = fmin(1.0, fmax(-1.0, 0.5 * det));
        const double phi = acos(rr) / 3.0;
        lmin = mq + 2.0 * p * cos(phi + 2.0943951023931953);     // + 2 pi / 3: the smallest root
    }
    curvature[q] = (float)(lmin / tr);
}

void launch_knn(int64_t N, int k, const float* grid9, const float* sorted_pts, const int64_t* sorted_id,
                const int32_t* cell_start, int64_t* out_idx, hipStream_t st) {
    KnnGrid g;
    g.x0 = grid9[0]; g.y0 = grid9[1]; g.z0 = grid9[2]; g.h = grid9[3]; g.inv_h = 1.0f / grid9[3];
    g.nx = (int)grid9[4]; g.ny = (int)grid9[5]; g.nz = (int)grid9[6];
    const unsigned grid = (unsigned)((N + 255) / 256);
    if (k <= 10) knn_kernel<10><<<grid, 256, 0, st>>>(N, k, g, sorted_pts, sorted_id, cell_start, out_idx);   // the reference's k
    else knn_kernel<KNN_MAX_K><<<grid, 256, 0, st>>>(N, k, g, sorted_pts, sorted_id, cell_start, out_idx);
}

void launch_knn_curvature(int64_t N, int k, const float* pts, const int64_t* idx, float* curvature, hipStream_t st) {
    knn_curvature_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(N, k, pts, idx, curvature);
}

}  // namespace scr
