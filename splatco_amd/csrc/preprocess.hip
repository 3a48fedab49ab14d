// splatco_amd/csrc/preprocess.hip -- per-Gaussian kernels (gfx950):
//   visible_filter / markVisible        (reference call site gaussian_renderer/__init__.py:239-242)
//   forward preprocess                  (projection, 2D covariance, radius, tile rect, colour;
//                                        fused per-tile counting + per-block partial sums)
//   backward reduce + preprocess-backward (deterministic gather of the per-instance gradient
//                                        records, then the chain back to the operator inputs)
//
// All of these stream P records once: they are HBM-bound.  One thread per Gaussian, fully
// coalesced struct-of-arrays reads (12/12/16/4/12 B per lane), one 48-byte record written.
//
// NORMATIVE ARITHMETIC: the integer outputs (radii, tile rect, depth bits) must be bit-identical
// to the CPU oracle, so this file is compiled with -ffp-contract=off and every expression below
// is written in the evaluation order of DESIGN.md "Normative arithmetic" (no FMA, IEEE sqrt/div).
#include "common.h"

namespace scr {

// ------------------------------------------------------------------ shared projection maths
struct Proj {
    float t[3];         // view-space position
    float hx, hy, hw;   // homogeneous clip x, y, w
    float cov[6];       // Sigma3D (xx,xy,xz,yy,yz,zz)
    float txc, tyc;     // clamped t.x, t.y
    bool clx, cly;
    float T[2][3];      // J * W
    float a, b, c;      // dilated 2D covariance
};

__device__ __forceinline__ void xform3(const float* __restrict__ M, float x, float y, float z, int i,
                                       float& out) {
    out = ((M[i] * x + M[4 + i] * y) + M[8 + i] * z) + M[12 + i];
}

__device__ __forceinline__ void rotmat(const float q[4], float R[3][3]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];  // used as given (not re-normalised)
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - r * z);
    R[0][2] = 2.0f * (x * z + r * y);
    R[1][0] = 2.0f * (x * y + r * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - r * x);
    R[2][0] = 2.0f * (x * z - r * y);
    R[2][1] = 2.0f * (y * z + r * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float s[3], float mod, const float q[4],
                                                     float cov[6]) {
    float R[3][3];
    rotmat(q, R);
    float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float L[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) L[i][j] = R[i][j] * sc[j];
#define SIG(i, j) ((L[i][0] * L[j][0] + L[i][1] * L[j][1]) + L[i][2] * L[j][2])
    cov[0] = SIG(0, 0);
    cov[1] = SIG(0, 1);
    cov[2] = SIG(0, 2);
    cov[3] = SIG(1, 1);
    cov[4] = SIG(1, 2);
    cov[5] = SIG(2, 2);
#undef SIG
}

// A.2 step 4: cov2D = J W Sigma W^T J^T + 0.3 I
__device__ __forceinline__ void cov2d(Proj& ps, float fx, float fy, float tanfovx, float tanfovy,
                                      const float* __restrict__ V) {
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float tz = ps.t[2];
    float txtz = ps.t[0] / tz, tytz = ps.t[1] / tz;
    float cx = fminf(limx, fmaxf(-limx, txtz));
    float cy = fminf(limy, fmaxf(-limy, tytz));
    ps.clx = (txtz < -limx) || (txtz > limx);
    ps.cly = (tytz < -limy) || (tytz > limy);
    float tx = cx * tz, ty = cy * tz;
    ps.txc = tx;
    ps.tyc = ty;
    float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    // W[i][c] = V[c*4+i]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ps.T[0][c] = J00 * V[c * 4 + 0] + J02 * V[c * 4 + 2];
        ps.T[1][c] = J11 * V[c * 4 + 1] + J12 * V[c * 4 + 2];
    }
    const float* cv = ps.cov;
    float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
    float U[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            U[r][c] = (ps.T[r][0] * S[0][c] + ps.T[r][1] * S[1][c]) + ps.T[r][2] * S[2][c];
    ps.a = ((U[0][0] * ps.T[0][0] + U[0][1] * ps.T[0][1]) + U[0][2] * ps.T[0][2]) + 0.3f;
    ps.b = ((U[0][0] * ps.T[1][0] + U[0][1] * ps.T[1][1]) + U[0][2] * ps.T[1][2]);
    ps.c = ((U[1][0] * ps.T[1][0] + U[1][1] * ps.T[1][1]) + U[1][2] * ps.T[1][2]) + 0.3f;
}

struct Foot {  // screen-space footprint
    float mx, my, det_inv;
    int radius;
    int rminx, rminy, rmaxx, rmaxy;
};

// A.2 steps 1-8.  Returns false when the Gaussian is culled.
__device__ __forceinline__ bool project(int64_t i, const float* __restrict__ means3D,
                                        const float* __restrict__ scales,
                                        const float* __restrict__ rotations,
                                        const float* __restrict__ cov3D, const KSettings& ks, Proj& ps,
                                        Foot& ft) {
    float x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
    xform3(ks.view, x, y, z, 0, ps.t[0]);
    xform3(ks.view, x, y, z, 1, ps.t[1]);
    xform3(ks.view, x, y, z, 2, ps.t[2]);
    if (!(ps.t[2] > 0.2f)) return false;
    xform3(ks.proj, x, y, z, 0, ps.hx);
    xform3(ks.proj, x, y, z, 1, ps.hy);
    xform3(ks.proj, x, y, z, 3, ps.hw);
    float p_w = 1.0f / (ps.hw + 0.0000001f);
    float px = ps.hx * p_w, py = ps.hy * p_w;
    if (cov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ps.cov[k] = cov3D[6 * i + k];
    } else {
        float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        cov3d_from_scale_rot(s, ks.scale_modifier, q, ps.cov);
    }
    cov2d(ps, ks.fx, ks.fy, ks.tanfovx, ks.tanfovy, ks.view);
    float a = ps.a, b = ps.b, c = ps.c;
    float det = a * c - b * b;
    if (det == 0.0f || det != det) return false;
    ft.det_inv = 1.0f / det;
    float mid = 0.5f * (a + c);
    float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    float l1 = mid + sq, l2 = mid - sq;
    float rad = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
    float Wf = (float)ks.W, Hf = (float)ks.H;
    float mx = ((px + 1.0f) * Wf - 1.0f) * 0.5f;
    float my = ((py + 1.0f) * Hf - 1.0f) * 0.5f;
    int gx = (ks.W + TILE - 1) / TILE, gy = (ks.H + TILE - 1) / TILE;
    ft.rminx = (int)fminf((float)gx, fmaxf(0.0f, truncf((mx - rad) / 16.0f)));
    ft.rminy = (int)fminf((float)gy, fmaxf(0.0f, truncf((my - rad) / 16.0f)));
    ft.rmaxx = (int)fminf((float)gx, fmaxf(0.0f, truncf(((mx + rad) + 15.0f) / 16.0f)));
    ft.rmaxy = (int)fminf((float)gy, fmaxf(0.0f, truncf(((my + rad) + 15.0f) / 16.0f)));
    if ((ft.rmaxx - ft.rminx) * (ft.rmaxy - ft.rminy) <= 0) return false;
    ft.radius = (int)fminf(rad, 1073741824.0f);
    ft.mx = mx;
    ft.my = my;
    return true;
}

// ------------------------------------------------------------------ visible_filter / markVisible
__global__ void __launch_bounds__(256) filter_kernel(int64_t P, const float* __restrict__ means3D,
                                                     const float* __restrict__ scales,
                                                     const float* __restrict__ rotations,
                                                     const float* __restrict__ cov3D, KSettings ks,
                                                     int32_t* __restrict__ radii) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    Proj ps;
    Foot ft;
    radii[i] = project(i, means3D, scales, rotations, cov3D, ks, ps, ft) ? ft.radius : 0;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int64_t P, const float* __restrict__ means3D,
                                                           const float* __restrict__ V,
                                                           uint8_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float tz;
    xform3(V, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], 2, tz);
    out[i] = tz > 0.2f;
}

// ------------------------------------------------------------------ SH (utils/sh_utils.py:57-112)
__device__ static const float SH_C0 = 0.28209479177387814f;
__device__ static const float SH_C1 = 0.4886025119029199f;
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

__device__ __forceinline__ void sh_to_rgb(int deg, const float* __restrict__ sh, float x, float y,
                                          float z, float rgb[3], uint8_t& clampbits) {
    clampbits = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#define SHV(k) sh[(k)*3 + c]
        float res = SH_C0 * SHV(0);
        if (deg > 0) {
            res = res - SH_C1 * y * SHV(1) + SH_C1 * z * SHV(2) - SH_C1 * x * SHV(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * SHV(4) + SH_C2[1] * yz * SHV(5) +
                      SH_C2[2] * (2.0f * zz - xx - yy) * SHV(6) + SH_C2[3] * xz * SHV(7) +
                      SH_C2[4] * (xx - yy) * SHV(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * SHV(9) + SH_C3[1] * xy * z * SHV(10) +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * SHV(11) +
                          SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHV(12) +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * SHV(13) + SH_C3[5] * z * (xx - yy) * SHV(14) +
                          SH_C3[6] * x * (xx - 3.0f * yy) * SHV(15);
                }
            }
        }
#undef SHV
        res += 0.5f;
        if (res < 0.0f) {
            clampbits |= (uint8_t)(1u << c);
            res = 0.0f;
        }
        rgb[c] = res;
    }
}

// ------------------------------------------------------------------ forward preprocess
// Writes radii, the 48-byte splat record, tiles_touched; counts instances per tile; leaves the
// workgroup's tiles_touched sum in block_sums.
//
// Per-tile counting: device integer atomics run at ~28 G/s on MI355X whatever their scope
// (tools/exp/atomics_probe.hip), so a workgroup of 1024 threads x 4 Gaussians first counts into an
// LDS histogram of all tiles (ds_add_u32) and then issues ONE global atomic per tile it touched
// (LDS_HIST; ~2.5x fewer global atomics at the benchmark density, far fewer for coherent scenes
// and large splats).  Images with more than LDS_HIST_MAX_TILES tiles count directly in global memory.
template <bool LDS_HIST>
__global__ void __launch_bounds__(BIN_THREADS)
preprocess_kernel(int64_t P, int M, const float* __restrict__ means3D, const float* __restrict__ scales,
                  const float* __restrict__ rotations, const float* __restrict__ cov3D,
                  const float* __restrict__ opacities, const float* __restrict__ shs,
                  const float* __restrict__ colors, KSettings ks, int tiles, float4* __restrict__ rec,
                  uint32_t* __restrict__ tiles_touched, uint8_t* __restrict__ clamped,
                  uint32_t* __restrict__ block_sums, uint32_t* __restrict__ tile_count,
                  int32_t* __restrict__ radii, unsigned long long* __restrict__ plan_flags) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];  // [tiles] when LDS_HIST
    __shared__ uint32_t wave_sum[BIN_THREADS / WAVE];
    if (LDS_HIST) {
        for (int t = threadIdx.x; t < tiles; t += BIN_THREADS) hist[t] = 0;
        __syncthreads();
    }
    const int gx = (ks.W + TILE - 1) / TILE;
    uint32_t tsum = 0;
    for (int r = 0; r < BIN_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * BIN_GPW + r * BIN_THREADS + threadIdx.x;
        if (i >= P) break;
        uint32_t tt = 0;
        Proj ps;
        Foot ft;
        bool vis = project(i, means3D, scales, rotations, cov3D, ks, ps, ft);
        if (vis) {
            float a = ps.a, b = ps.b, c = ps.c;
            float Qxx = c * ft.det_inv, Qxy = -b * ft.det_inv, Qyy = a * ft.det_inv;
            float rgb[3];
            uint8_t cb = 0;
            if (colors) {
                rgb[0] = colors[3 * i];
                rgb[1] = colors[3 * i + 1];
                rgb[2] = colors[3 * i + 2];
            } else {
                float dx = means3D[3 * i] - ks.campos[0], dy = means3D[3 * i + 1] - ks.campos[1],
                      dz = means3D[3 * i + 2] - ks.campos[2];
                float n = sqrtf((dx * dx + dy * dy) + dz * dz);
                sh_to_rgb(ks.sh_degree, shs + (size_t)i * M * 3, dx / n, dy / n, dz / n, rgb, cb);
            }
            if (clamped) clamped[i] = cb;
            tt = (uint32_t)((ft.rmaxx - ft.rminx) * (ft.rmaxy - ft.rminy));
            // plan flags (rare; one atomic per wave that has something to report):
            //  * a colour that is not finite (NaN / Inf input): the blend kernels must keep it away from the pixels its splat
            //    does not contribute to -- the host launches their SAFE instantiations for this call (blend.hip)
            //  * a rect of more than 32 tiles: the per-tile record verdicts of its tiles 32.. do not fit live_bits; the
            //    backward zeroes those records first and sums them unconditionally (zero_far_records_kernel)
            const unsigned long long w_nf = lanes(nonfinite3(rgb[0], rgb[1], rgb[2])), w_big = lanes(tt > 32u);
            if ((w_nf | w_big) != 0ull) {      // the first ACTIVE lane reports (lane 0 may be culled or past the end)
                const unsigned long long active = lanes(true);
                if ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __builtin_ctzll(active))
                    atomicOr(plan_flags, (unsigned long long)((w_nf ? SCR_PLAN_NONFINITE_COLOUR : 0) | (w_big ? SCR_PLAN_LARGE_RECTS : 0)));
            }
            uint32_t rlo = (uint32_t)ft.rminx | ((uint32_t)ft.rminy << 16);
            uint32_t rhi = (uint32_t)ft.rmaxx | ((uint32_t)ft.rmaxy << 16);
            // blend-ready conic: A = -Qxx/2, B = -Qxy, C = -Qyy/2 (exact scalings of Q)
            rec[3 * i + 0] = make_float4(ft.mx, ft.my, -0.5f * Qxx, -Qxy);
            rec[3 * i + 1] = make_float4(-0.5f * Qyy, opacities[i], rgb[0], rgb[1]);
            rec[3 * i + 2] = make_float4(rgb[2], ps.t[2], __uint_as_float(rlo), __uint_as_float(rhi));
            radii[i] = ft.radius;
            for (int ty = ft.rminy; ty < ft.rmaxy; ++ty)
                for (int tx = ft.rminx; tx < ft.rmaxx; ++tx) {
                    if (LDS_HIST) atomicAdd(&hist[ty * gx + tx], 1u);
                    else atomicAdd(&tile_count[ty * gx + tx], 1u);
                }
        } else {
            radii[i] = 0;
            if (clamped) clamped[i] = 0;
        }
        tiles_touched[i] = tt;
        tsum = sat_add_u32(tsum, tt);
    }
    // workgroup sum of tiles_touched (wave reduction, then the wave partials through LDS).  SATURATING: 4096 Gaussians
    // times a million-tile image passes 2^32, and a wrapped sum would slip under the host's num_rendered guard
    // (min(sum, 2^32 - 1) is the same in any order of additions)
    uint32_t s = tsum;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) s = sat_add_u32(s, (uint32_t)__shfl_down(s, d, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0) wave_sum[threadIdx.x / WAVE] = s;
    __syncthreads();  // also: every LDS histogram update of the workgroup is done
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < BIN_THREADS / WAVE; ++w) tot = sat_add_u32(tot, wave_sum[w]);
        block_sums[blockIdx.x] = tot;
    }
    if (LDS_HIST)
        for (int t = threadIdx.x; t < tiles; t += BIN_THREADS) {
            const uint32_t c = hist[t];
            if (c) atomicAdd(&tile_count[t], c);
        }
}

// ------------------------------------------------------------------ backward: reduce + chain
// One thread per Gaussian.  Sums its per-instance gradient records in tile order (fixed order ->
// bit-reproducible), then differentiates the projection (recomputed from the inputs).
__global__ void __launch_bounds__(PRE_BLOCK, 1)
preprocess_backward_kernel(int64_t P, int M, const float* __restrict__ means3D,
                           const float* __restrict__ scales, const float* __restrict__ rotations,
                           const float* __restrict__ cov3D, const float* __restrict__ shs, KSettings ks,
                           const int32_t* __restrict__ radii, const uint32_t* __restrict__ tiles_touched,
                           const uint32_t* __restrict__ point_offsets, const uint32_t* __restrict__ live_bits,
                           const uint8_t* __restrict__ has_rec, const uint8_t* __restrict__ clamped,
                           const float4* __restrict__ rec, const GradRec* __restrict__ grad_rec,
                           const unsigned long long* __restrict__ cut_key, unsigned long long stamp, int tiles,
                           float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
                           float* __restrict__ dL_dcolors, float* __restrict__ dL_dsh,
                           float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales,
                           float* __restrict__ dL_drotations, float* __restrict__ dL_dcov3D) {
    int64_t i = (int64_t)blockIdx.x * PRE_BLOCK + threadIdx.x;
    if (i >= P) return;
    float gm[3] = {0, 0, 0}, gm2[3] = {0, 0, 0}, gcol[3] = {0, 0, 0}, gop = 0, gs[3] = {0, 0, 0},
          gq[4] = {0, 0, 0, 0}, g6[6] = {0, 0, 0, 0, 0, 0};
    bool vis = radii[i] > 0;
    // deep lists (kernel-uniform pointer): a Gaussian without a single gradient record -- behind every pixel's last contributor
    // in all of its tiles, most of them at 20 M anchors -- has zero gradients: it is treated like an invisible one (its
    // inputs are not even read; the chain below would multiply them by sums that are all zero)
    if (has_rec && !has_rec[i]) vis = false;
    // depth and tile rect as the forward stored them: the record loop below needs them for its cut_key addresses, and taking
    // them from here instead of from project() lets those loads go out while project() is still waiting for its inputs
    const float4 r2 = rec[3 * i + 2];
    Proj ps;
    Foot ft;
    if (vis) vis = project(i, means3D, scales, rotations, cov3D, ks, ps, ft);
    if (vis) {
        // ---- deterministic reduction of the per-(tile, Gaussian) records
        uint32_t n = tiles_touched[i];
        uint32_t off = point_offsets[i] - n;
        float sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0;  // moments of Y = opacity G dL/dalpha over the footprint
        // this Gaussian's records are contiguous, in tile order (the row-major walk of its tile rect); four at a time so
        // that twelve record loads are in flight per thread -- the loop is otherwise one memory latency per record --
        // summed in order.  A tile's list entries behind every pixel's last contributor have NO record: the blend backward
        // left the sort key of the first such entry in cut_key[tile], and this Gaussian's key (depth bits, index) says on
        // which side it lies.  The 8-byte keys (a 64 KB table at 1080p: cache hits) are loaded WITH the records -- waiting
        // for the verdict before asking for the record would put two dependent round trips where there is one -- and the
        // bytes of missing records are dropped by a select, never used in arithmetic.
        const GradRec* gr = grad_rec + off;
        const unsigned long long mykey = ((unsigned long long)__float_as_uint(r2.y) << 32) | (uint32_t)i;
        const uint32_t rlo = __float_as_uint(r2.z), rhi = __float_as_uint(r2.w);
        const int gxt = (ks.W + TILE - 1) / TILE, rx0 = (int)(rlo & 0xffffu), rx1 = (int)(rhi & 0xffffu);
        int tcx = rx0, tcy = (int)(rlo >> 16);      // tile of record k
        // An instance no quadrant of its tile can reach (quadrant mask 0: the splat's rect covers the tile, its ellipse at
        // the alpha >= 1/255 level does not -- a third of the instances at the benchmark density) has NO record: the blend
        // backward skipped it.  The scatter kernel left the verdicts of the first 32 tiles of the walk in live_bits (the
        // records of a larger rect's further tiles: see below).
        const uint32_t lbits = live_bits[i];
        uint32_t kwalk = 0;
        auto next_tile = [&](bool& live) {      // the first 32 tiles of the walk: a bit test, nothing that branches
            const int t = tcy * gxt + tcx;
            live = ((lbits >> (kwalk & 31u)) & 1u) != 0u;
            ++kwalk;
            if (++tcx == rx1) { tcx = rx0; ++tcy; }
            return t;
        };
        auto add = [&](const GradRec& q, bool ok) {
            sx += ok ? q.a.x : 0.0f; sy += ok ? q.a.y : 0.0f; sxx += ok ? q.a.z : 0.0f; sxy += ok ? q.a.w : 0.0f;
            syy += ok ? q.b.x : 0.0f; gop += ok ? q.b.y : 0.0f; gcol[0] += ok ? q.b.z : 0.0f; gcol[1] += ok ? q.b.w : 0.0f;
            gcol[2] += ok ? q.c : 0.0f;
        };
        // The loads stay unconditional and four in flight per thread; the bytes of a record that was never written (dead
        // instance, or behind its tile's cut) are dropped by the select in add(), never used in arithmetic.  Measured at
        // cfg1 against the 0.0735 ms of summing every record: a predicated load per record 0.103 ms (the compiler closes
        // each with its own wait), dead loads redirected to one shared address 0.081 ms (a hot spot on one L2 channel), to
        // the Gaussian's own first record 0.079 - 0.089 ms (the address select sits in front of every load); as below
        // 0.075 ms -- what the verdicts buy is the blend backward's stores (a third of its 36-byte records), not reads.
#define SCR_PB_SRC(live_, p_) (p_)
        uint32_t k = 0;
        const uint32_t nb = n < 32u ? n : 32u;      // records whose verdict is in lbits
        const bool any_cut = cut_key[tiles] == stamp;
        if (any_cut) {                      // wave-uniform: some tile of this call left entries without records
            for (; k + 4 <= nb; k += 4) {
                GradRec q[4];
                unsigned long long ck[4];
                bool live[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { ck[j] = cut_key[next_tile(live[j])]; q[j] = *SCR_PB_SRC(live[j], gr + k + j); }
#pragma unroll
                for (int j = 0; j < 4; ++j) add(q[j], live[j] && mykey < ck[j]);
            }
            for (; k < nb; ++k) {
                bool live;
                const unsigned long long ck = cut_key[next_tile(live)];
                const GradRec q = *SCR_PB_SRC(live, gr + k);
                add(q, live && mykey < ck);
            }
        } else {                            // every live instance has its record (the benchmark density): no look-ups
            for (; k + 4 <= nb; k += 4) {
                GradRec q[4];
                bool live[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { next_tile(live[j]); q[j] = *SCR_PB_SRC(live[j], gr + k + j); }
#pragma unroll
                for (int j = 0; j < 4; ++j) add(q[j], live[j]);
            }
            for (; k < nb; ++k) {
                bool live;
                next_tile(live);
                add(*SCR_PB_SRC(live, gr + k), live);
            }
        }
        // tiles 32.. of a large rect: their records were zeroed before the blend backward ran (zero_far_records_kernel; the
        // plan flag SCR_PLAN_LARGE_RECTS told the host) and are summed as they are -- no second evaluation of the blend
        // backward's "does this instance get a record" decision that would have to agree with it bit for bit
        for (; k < n; ++k) add(gr[k], true);
        // the per-splat constants the blend kernel left out.  The moments are of Y = opacity * G * dL/dalpha, i.e.
        // dL/dG already: dG/dmean = -G Q d with Q = (-2A, -B, -2C), dG/dQ = -G/2 d d^T (Qxy counted once: factor 1);
        // dL/dopacity = sum G dL/dalpha = (sum Y) / opacity -- a non-zero sum means some pixel passed
        // alpha >= 1/255 with G <= 1, so the opacity is at least 1/255 there
        const float4 ra = rec[3 * i], rb = rec[3 * i + 1];
        const float cA = ra.z, cB = ra.w, cC = rb.x, op = rb.y;
        gop = gop != 0.0f ? gop / op : 0.0f;
        const float gmx = 2.0f * cA * sx + cB * sy, gmy = 2.0f * cC * sy + cB * sx;
        const float gQxx = -0.5f * sxx, gQxy = -sxy, gQyy = -0.5f * syy;
        const float* V = ks.view;
        const float* Pm = ks.proj;
        float a = ps.a, b = ps.b, c = ps.c;
        float det = a * c - b * b;
        float d2 = 1.0f / (det * det);
        float dL_da = d2 * (-c * c * gQxx + b * c * gQxy - b * b * gQyy);
        float dL_db = d2 * (2.0f * b * c * gQxx - (det + 2.0f * b * b) * gQxy + 2.0f * a * b * gQyy);
        float dL_dc = d2 * (-b * b * gQxx + a * b * gQxy - a * a * gQyy);
        const float* cv = ps.cov;
        float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
        float gS[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                gS[j][k] = dL_da * ps.T[0][j] * ps.T[0][k] + dL_db * ps.T[0][j] * ps.T[1][k] +
                           dL_dc * ps.T[1][j] * ps.T[1][k];
        g6[0] = gS[0][0]; g6[1] = gS[0][1] + gS[1][0]; g6[2] = gS[0][2] + gS[2][0];
        g6[3] = gS[1][1]; g6[4] = gS[1][2] + gS[2][1]; g6[5] = gS[2][2];
        float ST0[3], ST1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            ST0[j] = S[j][0] * ps.T[0][0] + S[j][1] * ps.T[0][1] + S[j][2] * ps.T[0][2];
            ST1[j] = S[j][0] * ps.T[1][0] + S[j][1] * ps.T[1][1] + S[j][2] * ps.T[1][2];
        }
        float gT[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gT[0][j] = 2.0f * dL_da * ST0[j] + dL_db * ST1[j];
            gT[1][j] = 2.0f * dL_dc * ST1[j] + dL_db * ST0[j];
        }
        float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            gJ00 += gT[0][cc] * V[cc * 4 + 0];
            gJ02 += gT[0][cc] * V[cc * 4 + 2];
            gJ11 += gT[1][cc] * V[cc * 4 + 1];
            gJ12 += gT[1][cc] * V[cc * 4 + 2];
        }
        float tz = ps.t[2], tx = ps.txc, ty = ps.tyc;
        float tz2 = tz * tz, tz3 = tz2 * tz;
        float fx = ks.fx, fy = ks.fy;
        float g_tx = -fx / tz2 * gJ02;
        float g_ty = -fy / tz2 * gJ12;
        float g_tz = -fx / tz2 * gJ00 - fy / tz2 * gJ11 + 2.0f * fx * tx / tz3 * gJ02 + 2.0f * fy * ty / tz3 * gJ12;
        float g_t[3] = {ps.clx ? 0.0f : g_tx, ps.cly ? 0.0f : g_ty, g_tz};  // A.5 (ii)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            gm[cc] = g_t[0] * V[cc * 4 + 0] + g_t[1] * V[cc * 4 + 1] + g_t[2] * V[cc * 4 + 2];
        // screen-space mean path
        float p_w = 1.0f / (ps.hw + 0.0000001f);
        float gpx = gmx * 0.5f * (float)ks.W, gpy = gmy * 0.5f * (float)ks.H;
        gm2[0] = gpx;
        gm2[1] = gpy;
        float g_hx = gpx * p_w, g_hy = gpy * p_w;
        float g_hw = -(gpx * ps.hx + gpy * ps.hy) * p_w * p_w;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            gm[cc] += g_hx * Pm[cc * 4 + 0] + g_hy * Pm[cc * 4 + 1] + g_hw * Pm[cc * 4 + 3];
        // SH colour path
        if (shs && dL_dsh) {
            float d0[3] = {means3D[3 * i] - ks.campos[0], means3D[3 * i + 1] - ks.campos[1],
                           means3D[3 * i + 2] - ks.campos[2]};
            float n2 = (d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2];
            float nn = sqrtf(n2);
            float x = d0[0] / nn, y = d0[1] / nn, z = d0[2] / nn;
            const float* sh = shs + (size_t)i * M * 3;
            float* gsh = dL_dsh + (size_t)i * M * 3;
            int deg = ks.sh_degree;
            uint8_t cb = clamped[i];
            float gdir[3] = {0, 0, 0};
            for (int k = (deg + 1) * (deg + 1) * 3; k < 3 * M; ++k) gsh[k] = 0.0f;  // unused bands
            for (int ch = 0; ch < 3; ++ch) {
                float gc = ((cb >> ch) & 1) ? 0.0f : gcol[ch];
#define SHV(k) sh[(k)*3 + ch]
                gsh[0 * 3 + ch] = SH_C0 * gc;
                if (deg > 0) {
                    gsh[1 * 3 + ch] = -SH_C1 * y * gc;
                    gsh[2 * 3 + ch] = SH_C1 * z * gc;
                    gsh[3 * 3 + ch] = -SH_C1 * x * gc;
                    float dx_ = -SH_C1 * SHV(3), dy_ = -SH_C1 * SHV(1), dz_ = SH_C1 * SHV(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        gsh[4 * 3 + ch] = SH_C2[0] * xy * gc;
                        gsh[5 * 3 + ch] = SH_C2[1] * yz * gc;
                        gsh[6 * 3 + ch] = SH_C2[2] * (2.0f * zz - xx - yy) * gc;
                        gsh[7 * 3 + ch] = SH_C2[3] * xz * gc;
                        gsh[8 * 3 + ch] = SH_C2[4] * (xx - yy) * gc;
                        dx_ += SH_C2[0] * y * SHV(4) + SH_C2[2] * 2.0f * -x * SHV(6) + SH_C2[3] * z * SHV(7) +
                               SH_C2[4] * 2.0f * x * SHV(8);
                        dy_ += SH_C2[0] * x * SHV(4) + SH_C2[1] * z * SHV(5) + SH_C2[2] * 2.0f * -y * SHV(6) +
                               SH_C2[4] * 2.0f * -y * SHV(8);
                        dz_ += SH_C2[1] * y * SHV(5) + SH_C2[2] * 4.0f * z * SHV(6) + SH_C2[3] * x * SHV(7);
                        if (deg > 2) {
                            gsh[9 * 3 + ch] = SH_C3[0] * y * (3.0f * xx - yy) * gc;
                            gsh[10 * 3 + ch] = SH_C3[1] * xy * z * gc;
                            gsh[11 * 3 + ch] = SH_C3[2] * y * (4.0f * zz - xx - yy) * gc;
                            gsh[12 * 3 + ch] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * gc;
                            gsh[13 * 3 + ch] = SH_C3[4] * x * (4.0f * zz - xx - yy) * gc;
                            gsh[14 * 3 + ch] = SH_C3[5] * z * (xx - yy) * gc;
                            gsh[15 * 3 + ch] = SH_C3[6] * x * (xx - 3.0f * yy) * gc;
                            dx_ += SH_C3[0] * SHV(9) * 6.0f * xy + SH_C3[1] * SHV(10) * yz +
                                   SH_C3[2] * SHV(11) * -2.0f * xy + SH_C3[3] * SHV(12) * -6.0f * xz +
                                   SH_C3[4] * SHV(13) * (-3.0f * xx + 4.0f * zz - yy) +
                                   SH_C3[5] * SHV(14) * 2.0f * xz + SH_C3[6] * SHV(15) * 3.0f * (xx - yy);
                            dy_ += SH_C3[0] * SHV(9) * 3.0f * (xx - yy) + SH_C3[1] * SHV(10) * xz +
                                   SH_C3[2] * SHV(11) * (-3.0f * yy + 4.0f * zz - xx) +
                                   SH_C3[3] * SHV(12) * -6.0f * yz + SH_C3[4] * SHV(13) * -2.0f * xy +
                                   SH_C3[5] * SHV(14) * -2.0f * yz + SH_C3[6] * SHV(15) * -6.0f * xy;
                            dz_ += SH_C3[1] * SHV(10) * xy + SH_C3[2] * SHV(11) * 8.0f * yz +
                                   SH_C3[3] * SHV(12) * 3.0f * (2.0f * zz - xx - yy) +
                                   SH_C3[4] * SHV(13) * 8.0f * xz + SH_C3[5] * SHV(14) * (xx - yy);
                        }
                    }
                    gdir[0] += dx_ * gc;
                    gdir[1] += dy_ * gc;
                    gdir[2] += dz_ * gc;
                }
#undef SHV
            }
            float inv3 = 1.0f / (n2 * nn);
            float dotv = d0[0] * gdir[0] + d0[1] * gdir[1] + d0[2] * gdir[2];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) gm[cc] += (n2 * gdir[cc] - d0[cc] * dotv) * inv3;
        }
        // Sigma3D -> scale, quaternion
        if (!cov3D) {
            float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
            float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
            float mod = ks.scale_modifier;
            float R[3][3];
            rotmat(q, R);
            float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
            float gL[3][3], gR[3][3];
#pragma unroll
            for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 3; ++b_) {
                    float acc = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc += (gS[a_][k] + gS[k][a_]) * (R[k][b_] * sc[b_]);
                    gL[a_][b_] = acc;
                }
#pragma unroll
            for (int b_ = 0; b_ < 3; ++b_) {
                float acc = 0;
#pragma unroll
                for (int a_ = 0; a_ < 3; ++a_) {
                    acc += gL[a_][b_] * R[a_][b_];
                    gR[a_][b_] = gL[a_][b_] * sc[b_];
                }
                gs[b_] = acc * mod;
            }
            float r = q[0], x = q[1], y = q[2], z = q[3];
            gq[0] = 2.0f * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
            gq[1] = 2.0f * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - 2.0f * x * gR[1][1] - r * gR[1][2] +
                            z * gR[2][0] + r * gR[2][1] - 2.0f * x * gR[2][2]);
            gq[2] = 2.0f * (-2.0f * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] -
                            r * gR[2][0] + z * gR[2][1] - 2.0f * y * gR[2][2]);
            gq[3] = 2.0f * (-2.0f * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] -
                            2.0f * z * gR[1][1] + y * gR[1][2] + x * gR[2][0] + y * gR[2][1]);
        }
    } else if (shs && dL_dsh) {
        for (int k = 0; k < 3 * M; ++k) dL_dsh[(size_t)i * 3 * M + k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dL_dmeans3D[3 * i + k] = gm[k];
        dL_dmeans2D[3 * i + k] = gm2[k];
        if (dL_dcolors) dL_dcolors[3 * i + k] = gcol[k];
        if (dL_dscales) dL_dscales[3 * i + k] = gs[k];
    }
    dL_dopacity[i] = gop;
    if (dL_drotations) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dL_drotations[4 * i + k] = gq[k];
    }
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = g6[k];
    }
}

// ------------------------------------------------------------------ launchers
static inline unsigned nblk(int64_t P, int per) { return (unsigned)((P + per - 1) / per); }

void launch_filter(int64_t P, const float* means3D, const float* scales, const float* rotations,
                   const float* cov3D, const KSettings& ks, int32_t* radii, hipStream_t st) {
    if (P <= 0) return;
    filter_kernel<<<nblk(P, 256), 256, 0, st>>>(P, means3D, scales, rotations, cov3D, ks, radii);
}

void launch_mark_visible(int64_t P, const float* means3D, const float* view, uint8_t* out, hipStream_t st) {
    if (P <= 0) return;
    mark_visible_kernel<<<nblk(P, 256), 256, 0, st>>>(P, means3D, view, out);
}

void launch_preprocess(int64_t P, int M, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D, const float* opacities, const float* shs, const float* colors,
                       const KSettings& ks, const GeomView& gv, int32_t* radii, hipStream_t st) {
    if (P <= 0) return;
    Grid g(ks.H, ks.W);
    // histograms beyond the default 64 KB dynamic-LDS limit (gfx950 has 160 KB per CU): the attribute is per device
    static bool big_lds[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !big_lds[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)preprocess_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 LDS_HIST_MAX_TILES * 4);
        if (e == hipSuccess && dev >= 0 && dev < 64) big_lds[dev] = true;
        (void)hipGetLastError();      // a refused attribute shows up as a launch error below
    }
    if (g.tiles <= LDS_HIST_MAX_TILES)
        preprocess_kernel<true><<<nblk(P, BIN_GPW), BIN_THREADS, (size_t)g.tiles * 4, st>>>(
            P, M, means3D, scales, rotations, cov3D, opacities, shs, colors, ks, g.tiles, gv.rec, gv.tiles_touched,
            shs ? gv.clamped : nullptr, gv.block_sums, gv.tile_count, radii, gv.total + 3);
    else
        preprocess_kernel<false><<<nblk(P, BIN_GPW), BIN_THREADS, 0, st>>>(
            P, M, means3D, scales, rotations, cov3D, opacities, shs, colors, ks, g.tiles, gv.rec, gv.tiles_touched,
            shs ? gv.clamped : nullptr, gv.block_sums, gv.tile_count, radii, gv.total + 3);
}

void launch_preprocess_backward(int64_t P, int M, const float* means3D, const float* scales,
                                const float* rotations, const float* cov3D, const float* shs,
                                const KSettings& ks, const int32_t* radii, const GeomView& gv,
                                const BinView& bv, const GradRec* grad_rec, const unsigned long long* cut_key,
                                unsigned long long stamp, bool deep, float* dL_dmeans3D,
                                float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh, float* dL_dopacity,
                                float* dL_dscales, float* dL_drotations, float* dL_dcov3D, hipStream_t st) {
    if (P <= 0) return;
    preprocess_backward_kernel<<<nblk(P, PRE_BLOCK), PRE_BLOCK, 0, st>>>(
        P, M, means3D, scales, rotations, cov3D, shs, ks, radii, gv.tiles_touched, gv.point_offsets, gv.live_bits,
        deep ? gv.has_rec : nullptr, gv.clamped, gv.rec, grad_rec, cut_key, stamp, Grid(ks.H, ks.W).tiles, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dsh, dL_dopacity,
        dL_dscales, dL_drotations, dL_dcov3D);
}

// ---- gradient records 32.. of every Gaussian whose rect has more than 32 tiles: cleared before the blend backward writes
// the ones it has something for.  One wave per 64 Gaussians; the (rare) large ones are cleared by the whole wave, 256 bytes
// per store instruction.  Launched by every backward; leaves at once unless the forward raised SCR_PLAN_LARGE_RECTS.
__global__ void __launch_bounds__(256) zero_far_records_kernel(int64_t P, const unsigned long long* __restrict__ total,
                                                               const uint32_t* __restrict__ tiles_touched,
                                                               const uint32_t* __restrict__ point_offsets, uint32_t* __restrict__ rec_words) {
    // the forward's own verdict, where it left it on the device (geom_buf): nothing here depends on what the caller passed
    // back as plan_flags -- a stale or zero argument cannot leave uninitialised records for preprocess_backward to sum
    if ((total[3] & (unsigned long long)SCR_PLAN_LARGE_RECTS) == 0ull) return;
    const int lane = threadIdx.x & 63;
    // (a capped grid walking the Gaussians: every backward launches this kernel, and 61 000 workgroups that leave at once
    // still cost 15 us at 15.6 M Gaussians)
    for (int64_t base = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); base < P; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + lane;
        const uint32_t n = i < P ? tiles_touched[i] : 0u;
        const uint32_t end = (i < P && n > 32u) ? point_offsets[i] : 0u;      // inclusive scan: one past the Gaussian's last record
        unsigned long long big = lanes(n > 32u);
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const uint32_t n_s = (uint32_t)__shfl((int)n, src, WAVE), end_s = (uint32_t)__shfl((int)end, src, WAVE);
            const size_t w0 = (size_t)(end_s - n_s + 32u) * GRAD_F, w1 = (size_t)end_s * GRAD_F;      // dwords
            for (size_t w = w0 + lane; w < w1; w += WAVE) rec_words[w] = 0u;
        }
    }
}
void launch_zero_far_records(int64_t P, const GeomView& gv, GradRec* grad_rec, hipStream_t st) {
    if (P <= 0) return;
    const int64_t want = (P + 255) / 256;
    zero_far_records_kernel<<<(unsigned)(want < 2048 ? want : 2048), 256, 0, st>>>(P, gv.total, gv.tiles_touched, gv.point_offsets, (uint32_t*)grad_rec);
}

// ---- ZeroList: blockIdx.y = buffer, blockIdx.x = 16 KB piece of it
__global__ void __launch_bounds__(256) zero_list_kernel(ZeroList z) {
    const unsigned long long bytes = z.n[blockIdx.y];
    char* base = (char*)z.p[blockIdx.y];
    const unsigned long long lo = (unsigned long long)blockIdx.x * 16384ull;
    if (lo >= bytes) return;
    const unsigned long long hi = lo + 16384ull < bytes ? lo + 16384ull : bytes;
    if (((uintptr_t)base & 15) == 0) {
        for (unsigned long long o = lo + 16ull * threadIdx.x; o + 16 <= hi; o += 16ull * 256) *(float4*)(base + o) = make_float4(0, 0, 0, 0);
        for (unsigned long long o = lo + ((hi - lo) & ~15ull) + 4ull * threadIdx.x; o < hi; o += 4ull * 256) *(uint32_t*)(base + o) = 0u;
    } else {
        for (unsigned long long o = lo + 4ull * threadIdx.x; o < hi; o += 4ull * 256) *(uint32_t*)(base + o) = 0u;
    }
}
void launch_zero(const ZeroList& z, hipStream_t st) {
    if (z.count <= 0) return;
    unsigned long long mx = 0;
    for (int i = 0; i < z.count; ++i) mx = z.n[i] > mx ? z.n[i] : mx;
    zero_list_kernel<<<dim3((unsigned)((mx + 16383) / 16384), (unsigned)z.count), 256, 0, st>>>(z);
}

}  // namespace scr
