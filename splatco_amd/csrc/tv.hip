// splatco_amd/csrc/tv.hip -- the tri-plane total-variation term of the training step, added straight into the plane
// gradients (gfx950).
//
// Reference: train.py:242-243 (every 4th iteration, between backward() and optimizer.step()):
//   gaussians.feat_planes.tv_loss(opt.tv_weight_a)  ->  scene/gaussian_model.py:217-220 (grid `level` gets the weight
//   w * 0.5^(2 - level))  ->  scene/grids.py:240-250 PlaneGrid.total_variation_add_grad: for each of the three planes
//   [1, R, A, B] the smooth-L1 (beta = 1, reduction 'sum') of the differences of neighbours along A and along B, the six
//   sums weighted by w, divided by 6, .backward() -- i.e. the derivative is ACCUMULATED into the planes' .grad.
//
// The derivative in closed form, with h(d) = c * clamp(d, -1, 1) and c = fp32(1/6) * fp32(w) (the two factors autograd
// multiplies: DivBackward's 1/6 and MulBackward's scalar w):
//   grad[r, a, b] += [a > 0] h(p[a,b] - p[a-1,b]) - [a < A-1] h(p[a+1,b] - p[a,b])
//                  + [b > 0] h(p[a,b] - p[a,b-1]) - [b < B-1] h(p[a,b+1] - p[a,b])
// (smooth_l1_loss_backward gives x*c for |x| < 1 and +-c otherwise, for the minuend slice, and its negation for the
// subtrahend slice.)  No autograd graph, no temporaries: torch materialises 12 difference / mask / padded-slice tensors
// per plane for this.
//
// Roofline: HBM.  Algorithmic bytes = 12 per plane element (read p, read grad, write grad).  One wave owns a tile of
// 64 x VEC columns and walks down a strip of TV_ROWS rows keeping the previous / current / next row in registers, so every
// plane element is loaded once per strip plus the two halo rows (2 / TV_ROWS = 6 %, mostly L2 hits); the left / right
// neighbours of a lane's columns come from the neighbouring lanes' cache lines (L1 hits).  Four rows of loads are in flight
// per wave.  Consecutive units of the same XCD are neighbouring column tiles and strips, so the halos meet in that XCD's L2.
#include "tv.h"

namespace scr {

constexpr int TV_ROWS = 32;       // rows per strip
constexpr int TV_BATCH = 4;       // rows whose loads are issued together

typedef float tv_f4 __attribute__((ext_vector_type(4)));

struct TvArgs {
    int n;
    uint32_t first_unit[TV_MAX + 1];   // plane t owns wave-units first_unit[t] .. first_unit[t + 1] - 1
    const float* p[TV_MAX];
    float* g[TV_MAX];
    int C[TV_MAX], A[TV_MAX], B[TV_MAX];
    int strips[TV_MAX], ctiles[TV_MAX];
    float coef[TV_MAX];
    uint8_t vec4[TV_MAX];              // B % 4 == 0 and both pointers on 16 bytes
};

__device__ __forceinline__ float tv_h(float d, float c) { return fminf(fmaxf(d, -1.0f), 1.0f) * c; }

template <int VEC> struct TvRow;
template <> struct TvRow<4> {
    typedef tv_f4 T;
    static __device__ __forceinline__ T load(const float* q) { return *(const tv_f4*)q; }
    static __device__ __forceinline__ void store(float* q, T v) { *(tv_f4*)q = v; }
    static __device__ __forceinline__ float get(const T& v, int k) { return v[k]; }
    static __device__ __forceinline__ void set(T& v, int k, float x) { v[k] = x; }
};
template <> struct TvRow<1> {
    typedef float T;
    static __device__ __forceinline__ T load(const float* q) { return *q; }
    static __device__ __forceinline__ void store(float* q, T v) { *q = v; }
    static __device__ __forceinline__ float get(const T& v, int) { return v; }
    static __device__ __forceinline__ void set(T& v, int, float x) { v = x; }
};

// one wave: channel plane `p` / `g` (A rows of B floats), rows a0 .. a1 - 1, columns j0 .. j0 + VEC - 1 of this lane
template <int VEC>
__device__ __forceinline__ void tv_strip(const float* __restrict__ p, float* __restrict__ g, int A, int B, int a0, int a1, int j0,
                                         float c) {
    typedef TvRow<VEC> R;
    typedef typename R::T T;
    if (j0 >= B) return;
    const bool has_l = j0 > 0, has_r = j0 + VEC < B;
    const float* col = p + j0;
    T up = R::load(col + (size_t)max(a0 - 1, 0) * B);        // a0 == 0: the row itself, masked below
    T cur = R::load(col + (size_t)a0 * B);
    for (int a = a0; a < a1; a += TV_BATCH) {
        T nx[TV_BATCH], gr[TV_BATCH];
        float lf[TV_BATCH], rt[TV_BATCH];
#pragma unroll
        for (int k = 0; k < TV_BATCH; ++k) {
            const int row = a + k;
            if (row < a1) {
                nx[k] = R::load(col + (size_t)min(row + 1, A - 1) * B);
                gr[k] = R::load(g + (size_t)row * B + j0);
                lf[k] = has_l ? col[(size_t)row * B - 1] : 0.0f;
                rt[k] = has_r ? col[(size_t)row * B + VEC] : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < TV_BATCH; ++k) {
            const int row = a + k;
            if (row < a1) {
                const bool has_u = row > 0, has_d = row + 1 < A;
                T out = gr[k];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float x = R::get(cur, e);
                    const float tu = has_u ? tv_h(x - R::get(up, e), c) : 0.0f;
                    const float td = has_d ? tv_h(R::get(nx[k], e) - x, c) : 0.0f;
                    const bool el = e > 0 || has_l, er = e + 1 < VEC || has_r;
                    const float xl = e > 0 ? R::get(cur, e > 0 ? e - 1 : 0) : lf[k];
                    const float xr = e + 1 < VEC ? R::get(cur, e + 1 < VEC ? e + 1 : 0) : rt[k];
                    const float tl = el ? tv_h(x - xl, c) : 0.0f;
                    const float tr = er ? tv_h(xr - x, c) : 0.0f;
                    R::set(out, e, R::get(out, e) + ((tu - td) + (tl - tr)));
                }
                R::store(g + (size_t)row * B + j0, out);
                up = cur;
                cur = nx[k];
            }
        }
    }
}

__global__ void __launch_bounds__(256) tv_add_grad_kernel(const TvArgs a) {
    // workgroups go round-robin over the 8 XCDs: give each XCD a contiguous eighth of the units (the grid is a multiple of 8)
    const unsigned per = gridDim.x >> 3;
    const unsigned vb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const unsigned u = vb * 4u + (threadIdx.x >> 6);
    if (u >= a.first_unit[a.n]) return;
    int t = 0;
    while (t + 1 < a.n && u >= a.first_unit[t + 1]) ++t;
    unsigned r = u - a.first_unit[t];
    const int ct = (int)(r % (unsigned)a.ctiles[t]);
    r /= (unsigned)a.ctiles[t];
    const int strip = (int)(r % (unsigned)a.strips[t]);
    const int ch = (int)(r / (unsigned)a.strips[t]);
    const int A = a.A[t], B = a.B[t];
    const size_t plane = (size_t)ch * A * B;
    const int a0 = strip * TV_ROWS, a1 = min(a0 + TV_ROWS, A);
    const int lane = threadIdx.x & 63;
    if (a.vec4[t]) tv_strip<4>(a.p[t] + plane, a.g[t] + plane, A, B, a0, a1, (ct * 64 + lane) * 4, a.coef[t]);
    else           tv_strip<1>(a.p[t] + plane, a.g[t] + plane, A, B, a0, a1, ct * 64 + lane, a.coef[t]);
}

int launch_tv_add_grad(int n, const scr_tv_plane* planes, hipStream_t st) {
    for (int t0 = 0; t0 < n; t0 += TV_MAX) {
        TvArgs a;
        a.n = 0;
        uint64_t units = 0;
        for (int t = t0; t < min(t0 + TV_MAX, n); ++t) {
            const scr_tv_plane& x = planes[t];
            if ((int64_t)x.channels * x.rows * x.cols == 0) continue;
            const int k = a.n++;
            a.first_unit[k] = (uint32_t)units;
            a.p[k] = x.plane;
            a.g[k] = x.grad;
            a.C[k] = x.channels;
            a.A[k] = x.rows;
            a.B[k] = x.cols;
            a.coef[k] = x.coef;
            a.vec4[k] = (x.cols % 4 == 0) && ((((uintptr_t)x.plane | (uintptr_t)x.grad) & 15u) == 0);
            const int per_lane = a.vec4[k] ? 4 : 1;
            a.strips[k] = (x.rows + TV_ROWS - 1) / TV_ROWS;
            a.ctiles[k] = (x.cols + 64 * per_lane - 1) / (64 * per_lane);
            units += (uint64_t)x.channels * a.strips[k] * a.ctiles[k];
            if (units > 0x7fffffffull) return 1;
        }
        if (!a.n) continue;
        a.first_unit[a.n] = (uint32_t)units;
        const uint64_t blocks = ((units + 3) / 4 + 7) / 8 * 8;
        tv_add_grad_kernel<<<(unsigned)blocks, 256, 0, st>>>(a);
    }
    return 0;
}

}  // namespace scr
