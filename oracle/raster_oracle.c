/*
 * oracle/raster_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement (plain C, scalar, single thread) of the differentiable Gaussian-splatting
 * tile rasterizer that SplatCo's gaussian_renderer.render() calls through the pip package
 * `diff_gaussian_rasterization` (reference call sites: gaussian_renderer/__init__.py:15,
 * :145-171 forward, :208-242 visible_filter).
 *
 * PARITY UNPINNED at the rasterizer boundary: the reference's rasterizer source lives in
 * submodules.zip, which is absent from /root/reference (.MISSING_LARGE_BLOBS:1,
 * environment.yml:26, no version pin), and the reference ships no test / golden vector for
 * it.  This file therefore restates the published algorithm of the 3DGS / Scaffold-GS
 * rasterizer family and is the *normative spec* the HIP kernels in
 * splatco_amd/csrc are checked against.  Conventions that ARE pinned in-tree are cited:
 *   - row-vector matrices  p_view = [x,y,z,1] * viewmatrix     utils/graphics_utils.py:22-29
 *   - viewmatrix = W2C^T, projmatrix = W2C^T * P^T             scene/cameras.py:54-56
 *   - quaternion (w,x,y,z) -> rotation matrix                   utils/general_utils.py:78-99
 *   - Sigma3D = (R diag(s)) (R diag(s))^T                        utils/general_utils.py:101-110
 *   - cov3D 6-vector (xx,xy,xz,yy,yz,zz)                         utils/general_utils.py:64-73
 *   - SH basis / constants                                       utils/sh_utils.py:57-112
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Arithmetic is NORMATIVE (DESIGN.md "Normative arithmetic"): every integer decision
 * (radius, tile rect, depth-bit sort key, power>0 test) is a fixed sequence of IEEE-754
 * binary32 operations with no implicit contraction; fused multiply-adds appear only where
 * written as fmaf().  Build with -ffp-contract=off (oracle/Makefile).  With -DORC_F64 the
 * same code runs in binary64 (used only to bound the fp32 error of images / gradients).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_F64
typedef double real;
#define R(x) x
#define r_sqrt sqrt
#define r_exp exp
#define r_ceil ceil
#define r_trunc trunc
#define r_fmin fmin
#define r_fmax fmax
#define r_fma fma
#else
typedef float real;
#define R(x) x##f
#define r_sqrt sqrtf
#define r_exp expf
#define r_ceil ceilf
#define r_trunc truncf
#define r_fmin fminf
#define r_fmax fmaxf
#define r_fma fmaf
#endif

#define TILE 16

typedef struct {
    int32_t image_height, image_width;
    float tanfovx, tanfovy;
    float bg[3];
    float scale_modifier;
    float viewmatrix[16]; /* flattened row-major [4,4] tensor as stored by scene/cameras.py:54 */
    float projmatrix[16]; /* scene/cameras.py:55-56 */
    int32_t sh_degree;
    float campos[3];
    int32_t prefiltered;
    int32_t debug;
} orc_settings;

int orc_real_size(void) { return (int)sizeof(real); }

/* Optional multi-threaded build (-fopenmp -DORC_OMP, libraster_oracle_f32_omp.so): used ONLY for the timed
   cpu_baseline of bench.py, so that the baseline runs on all host cores.  Loops over Gaussians / tiles are
   split across threads; the per-Gaussian sums of the blend backward become atomic adds (their order, and so
   the last bits, then vary from run to run -- the parity tests use the serial build, where these macros
   expand to nothing and the code is the scalar restatement as before). */
#ifdef ORC_OMP
#include <omp.h>
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(dynamic, 8)")
#define ORC_ATOMIC _Pragma("omp atomic")
int orc_threads(void) { return omp_get_max_threads(); }
#else
#define ORC_PARALLEL_FOR
#define ORC_ATOMIC
int orc_threads(void) { return 1; }
#endif

/* ---------------------------------------------------------------- SH (utils/sh_utils.py:57-112) */
static const real SH_C0 = R(0.28209479177387814);
static const real SH_C1 = R(0.4886025119029199);
static const real SH_C2[5] = {R(1.0925484305920792), R(-1.0925484305920792), R(0.31539156525252005),
                              R(-1.0925484305920792), R(0.5462742152960396)};
static const real SH_C3[7] = {R(-0.5900435899266435), R(2.890611442640554), R(-0.4570457994644658),
                              R(0.3731763325901154), R(-0.4570457994644658), R(1.445305721320277),
                              R(-0.5900435899266435)};

/* colour = max(0, SH(dir) + 0.5); clamped[c] = 1 when the max() clipped (gradient is zero there) */
static void sh_to_rgb(int deg, int M, const float* sh /*[M][3]*/, const real dir[3], real rgb[3],
                      int32_t clamped[3]) {
    real x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; ++c) {
#define SHV(k) ((real)sh[(k)*3 + c])
        real res = SH_C0 * SHV(0);
        if (deg > 0) {
            res = res - SH_C1 * y * SHV(1) + SH_C1 * z * SHV(2) - SH_C1 * x * SHV(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * SHV(4) + SH_C2[1] * yz * SHV(5) +
                      SH_C2[2] * (R(2.0) * zz - xx - yy) * SHV(6) + SH_C2[3] * xz * SHV(7) +
                      SH_C2[4] * (xx - yy) * SHV(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (R(3.0) * xx - yy) * SHV(9) + SH_C3[1] * xy * z * SHV(10) +
                          SH_C3[2] * y * (R(4.0) * zz - xx - yy) * SHV(11) +
                          SH_C3[3] * z * (R(2.0) * zz - R(3.0) * xx - R(3.0) * yy) * SHV(12) +
                          SH_C3[4] * x * (R(4.0) * zz - xx - yy) * SHV(13) +
                          SH_C3[5] * z * (xx - yy) * SHV(14) + SH_C3[6] * x * (xx - R(3.0) * yy) * SHV(15);
                }
            }
        }
#undef SHV
        res += R(0.5);
        clamped[c] = res < 0;
        rgb[c] = res < 0 ? 0 : res;
    }
    (void)M;
}

/* ---------------------------------------------------------------- A.2 helpers */
static void cov3d_from_scale_rot(const float* s, real mod, const float* q, real cov[6]) {
    real r = q[0], x = q[1], y = q[2], z = q[3]; /* used as given, not re-normalised */
    real Rm[3][3];
    Rm[0][0] = R(1.0) - R(2.0) * (y * y + z * z);
    Rm[0][1] = R(2.0) * (x * y - r * z);
    Rm[0][2] = R(2.0) * (x * z + r * y);
    Rm[1][0] = R(2.0) * (x * y + r * z);
    Rm[1][1] = R(1.0) - R(2.0) * (x * x + z * z);
    Rm[1][2] = R(2.0) * (y * z - r * x);
    Rm[2][0] = R(2.0) * (x * z - r * y);
    Rm[2][1] = R(2.0) * (y * z + r * x);
    Rm[2][2] = R(1.0) - R(2.0) * (x * x + y * y);
    real sc[3] = {mod * (real)s[0], mod * (real)s[1], mod * (real)s[2]};
    real L[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) L[i][j] = Rm[i][j] * sc[j];
#define SIG(i, j) ((L[i][0] * L[j][0] + L[i][1] * L[j][1]) + L[i][2] * L[j][2])
    cov[0] = SIG(0, 0);
    cov[1] = SIG(0, 1);
    cov[2] = SIG(0, 2);
    cov[3] = SIG(1, 1);
    cov[4] = SIG(1, 2);
    cov[5] = SIG(2, 2);
#undef SIG
}

typedef struct {
    real t[3];          /* view-space position */
    real txc, tyc;      /* clamped t.x, t.y (after re-multiplying by t.z) */
    int clamp_x, clamp_y;
    real J00, J02, J11, J12;
    real T[2][3];       /* J * W */
    real a, b, c;       /* dilated cov2D */
} proj_state;

/* steps 4 of A.2: cov2D = J W Sigma W^T J^T + 0.3 I */
static void cov2d(const real t[3], real fx, real fy, real tanfovx, real tanfovy, const float* V,
                  const real cov[6], proj_state* ps) {
    real limx = R(1.3) * tanfovx, limy = R(1.3) * tanfovy;
    real txtz = t[0] / t[2], tytz = t[1] / t[2];
    real cx = r_fmin(limx, r_fmax(-limx, txtz));
    real cy = r_fmin(limy, r_fmax(-limy, tytz));
    ps->clamp_x = (txtz < -limx) || (txtz > limx);
    ps->clamp_y = (tytz < -limy) || (tytz > limy);
    real tx = cx * t[2], ty = cy * t[2], tz = t[2];
    ps->txc = tx;
    ps->tyc = ty;
    real J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    real J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    ps->J00 = J00; ps->J02 = J02; ps->J11 = J11; ps->J12 = J12;
    /* W[i][c] : view = W * world ;  p_view.i = sum_c V[c*4+i] * world.c  */
#define Wm(i, c) ((real)V[(c)*4 + (i)])
    for (int c = 0; c < 3; ++c) {
        ps->T[0][c] = J00 * Wm(0, c) + J02 * Wm(2, c);
        ps->T[1][c] = J11 * Wm(1, c) + J12 * Wm(2, c);
    }
#undef Wm
    real S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
    real U[2][3];
    for (int r_ = 0; r_ < 2; ++r_)
        for (int c = 0; c < 3; ++c)
            U[r_][c] = (ps->T[r_][0] * S[0][c] + ps->T[r_][1] * S[1][c]) + ps->T[r_][2] * S[2][c];
    ps->a = ((U[0][0] * ps->T[0][0] + U[0][1] * ps->T[0][1]) + U[0][2] * ps->T[0][2]) + R(0.3);
    ps->b = ((U[0][0] * ps->T[1][0] + U[0][1] * ps->T[1][1]) + U[0][2] * ps->T[1][2]);
    ps->c = ((U[1][0] * ps->T[1][0] + U[1][1] * ps->T[1][1]) + U[1][2] * ps->T[1][2]) + R(0.3);
}

static void xform4(const float* M, const float* p, real out[4]) {
    real x = p[0], y = p[1], z = p[2];
    for (int i = 0; i < 4; ++i)
        out[i] = (((real)M[i] * x + (real)M[4 + i] * y) + (real)M[8 + i] * z) + (real)M[12 + i];
}

/*
 * A.2 steps 1-9.  One routine serves visible_filter (only radii requested) and the forward
 * preprocess.  Any output pointer may be NULL.
 *   rect[4] = (min.x, min.y, max.x, max.y) in tiles;  tiles_touched = area, 0 when culled.
 */
void orc_preprocess(int P, int M, const float* means3D, const float* scales, const float* rotations,
                    const float* cov3D_precomp, const float* opacities, const float* shs,
                    const float* colors_precomp, const orc_settings* st, int32_t* radii, real* xy,
                    real* depth, real* cov3D, real* conic_opacity, real* rgb, int32_t* clamped,
                    uint32_t* tiles_touched, int32_t* rect) {
    const int W = st->image_width, H = st->image_height;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const real tanfovx = st->tanfovx, tanfovy = st->tanfovy;
    const real fx = (real)W / (R(2.0) * tanfovx), fy = (real)H / (R(2.0) * tanfovy);
    ORC_PARALLEL_FOR
    for (int i = 0; i < P; ++i) {
        if (radii) radii[i] = 0;
        if (tiles_touched) tiles_touched[i] = 0;
        if (rect) rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        if (xy) xy[2 * i] = xy[2 * i + 1] = 0;
        if (depth) depth[i] = 0;
        if (conic_opacity) for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0;
        if (rgb) for (int k = 0; k < 3; ++k) rgb[3 * i + k] = 0;
        if (clamped) for (int k = 0; k < 3; ++k) clamped[3 * i + k] = 0;
        if (cov3D) for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0;

        const float* p = means3D + 3 * i;
        real t4[4], h[4];
        xform4(st->viewmatrix, p, t4);
        if (!(t4[2] > R(0.2))) continue; /* step 1: near cull (NaN culls too) */
        xform4(st->projmatrix, p, h);
        real p_w = R(1.0) / (h[3] + R(0.0000001));
        real px = h[0] * p_w, py = h[1] * p_w;

        real cov[6];
        if (cov3D_precomp) {
            for (int k = 0; k < 6; ++k) cov[k] = cov3D_precomp[6 * i + k];
        } else {
            cov3d_from_scale_rot(scales + 3 * i, st->scale_modifier, rotations + 4 * i, cov);
        }
        proj_state ps;
        cov2d(t4, fx, fy, tanfovx, tanfovy, st->viewmatrix, cov, &ps);
        real a = ps.a, b = ps.b, c = ps.c;
        real det = a * c - b * b;
        if (det == 0 || det != det) continue; /* step 5 */
        real det_inv = R(1.0) / det;
        real mid = R(0.5) * (a + c);
        real sq = r_sqrt(r_fmax(R(0.1), mid * mid - det));
        real l1 = mid + sq, l2 = mid - sq;
        real rad = r_ceil(R(3.0) * r_sqrt(r_fmax(l1, l2)));
        real mx = ((px + R(1.0)) * (real)W - R(1.0)) * R(0.5);
        real my = ((py + R(1.0)) * (real)H - R(1.0)) * R(0.5);
        /* step 8: truncation then clamp, carried out in the float domain (identical to C
           float->int truncation for every in-range value, and well defined outside it) */
        int rminx = (int)r_fmin((real)gx, r_fmax(0, r_trunc((mx - rad) / R(16.0))));
        int rminy = (int)r_fmin((real)gy, r_fmax(0, r_trunc((my - rad) / R(16.0))));
        int rmaxx = (int)r_fmin((real)gx, r_fmax(0, r_trunc(((mx + rad) + R(15.0)) / R(16.0))));
        int rmaxy = (int)r_fmin((real)gy, r_fmax(0, r_trunc(((my + rad) + R(15.0)) / R(16.0))));
        int area = (rmaxx - rminx) * (rmaxy - rminy);
        if (area <= 0) continue;

        if (radii) radii[i] = (int32_t)r_fmin(rad, R(1073741824.0));
        if (tiles_touched) tiles_touched[i] = (uint32_t)area;
        if (rect) { rect[4 * i] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy; }
        if (xy) { xy[2 * i] = mx; xy[2 * i + 1] = my; }
        if (depth) depth[i] = t4[2];
        if (cov3D) for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = cov[k];
        if (conic_opacity) {
            conic_opacity[4 * i + 0] = c * det_inv;
            conic_opacity[4 * i + 1] = -b * det_inv;
            conic_opacity[4 * i + 2] = a * det_inv;
            conic_opacity[4 * i + 3] = opacities ? (real)opacities[i] : 0;
        }
        if (rgb) {
            if (colors_precomp) {
                for (int k = 0; k < 3; ++k) rgb[3 * i + k] = colors_precomp[3 * i + k];
            } else if (shs) {
                real d[3] = {(real)p[0] - (real)st->campos[0], (real)p[1] - (real)st->campos[1],
                             (real)p[2] - (real)st->campos[2]};
                real n = r_sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
                d[0] /= n; d[1] /= n; d[2] /= n;
                int32_t cl[3];
                sh_to_rgb(st->sh_degree, M, shs + (size_t)i * M * 3, d, rgb + 3 * i, cl);
                if (clamped) for (int k = 0; k < 3; ++k) clamped[3 * i + k] = cl[k];
            }
        }
    }
}

/* markVisible: in-frustum test of the rasterizer family = view-space z > 0.2 */
void orc_mark_visible(int P, const float* means3D, const orc_settings* st, uint8_t* present) {
    for (int i = 0; i < P; ++i) {
        real t4[4];
        xform4(st->viewmatrix, means3D + 3 * i, t4);
        present[i] = t4[2] > R(0.2);
    }
}

/* ---------------------------------------------------------------- A.3 binning */
typedef struct { uint64_t key; uint32_t id; } kv_t;

static void merge_sort_kv(kv_t* a, kv_t* tmp, size_t n) { /* stable, bottom-up */
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(kv_t));
    }
}

#ifdef ORC_OMP
/* Threaded build: the same stable order, found per tile -- a stable counting pass groups the pairs by
   tile (the high key word), then every tile's segment is merge-sorted on its own thread. */
static void sort_kv_by_tile(kv_t* a, kv_t* tmp, size_t n, int tiles) {
    size_t* start = (size_t*)calloc((size_t)tiles + 1, sizeof(size_t));
    for (size_t k = 0; k < n; ++k) ++start[(a[k].key >> 32) + 1];
    for (int t = 0; t < tiles; ++t) start[t + 1] += start[t];
    size_t* cur = (size_t*)malloc((size_t)tiles * sizeof(size_t));
    memcpy(cur, start, (size_t)tiles * sizeof(size_t));
    for (size_t k = 0; k < n; ++k) tmp[cur[a[k].key >> 32]++] = a[k];
    ORC_PARALLEL_FOR
    for (int t = 0; t < tiles; ++t) merge_sort_kv(tmp + start[t], a + start[t], start[t + 1] - start[t]);
    memcpy(a, tmp, n * sizeof(kv_t));
    free(cur);
    free(start);
}
#endif

/* pass 1: offsets = inclusive scan(tiles_touched); returns I = offsets[P-1] */
int64_t orc_scan(int P, const uint32_t* tiles_touched, uint64_t* offsets) {
    uint64_t s = 0;
    for (int i = 0; i < P; ++i) { s += tiles_touched[i]; offsets[i] = s; }
    return (int64_t)s;
}

/* pass 2: emit (key,id), stable sort by key, per-tile ranges.  depth32 = binary32 depth bits. */
void orc_bin(int P, const uint32_t* tiles_touched, const int32_t* rect, const float* depth32,
             int gx, int gy, int64_t I, uint64_t* keys_sorted, uint32_t* ids_sorted,
             uint32_t* ranges /*[gx*gy][2]*/) {
    kv_t* kv = (kv_t*)malloc((size_t)(I > 0 ? I : 1) * sizeof(kv_t));
    kv_t* tmp = (kv_t*)malloc((size_t)(I > 0 ? I : 1) * sizeof(kv_t));
    /* emit in Gaussian order: Gaussian i owns the slots [first[i], first[i] + tiles_touched[i]) */
    size_t* first = (size_t*)malloc((size_t)(P > 0 ? P : 1) * sizeof(size_t));
    size_t n = 0;
    for (int i = 0; i < P; ++i) { first[i] = n; n += tiles_touched[i]; }
    ORC_PARALLEL_FOR
    for (int i = 0; i < P; ++i) {
        if (!tiles_touched[i]) continue;
        uint32_t dbits;
        memcpy(&dbits, depth32 + i, 4);
        size_t k = first[i];
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
                kv[k].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                kv[k].id = (uint32_t)i;
                ++k;
            }
    }
    free(first);
#ifdef ORC_OMP
    sort_kv_by_tile(kv, tmp, n, gx * gy);
#else
    merge_sort_kv(kv, tmp, n);
#endif
    memset(ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
    ORC_PARALLEL_FOR
    for (long long kk = 0; kk < (long long)n; ++kk) {
        const size_t k = (size_t)kk;
        keys_sorted[k] = kv[k].key;
        ids_sorted[k] = kv[k].id;
        uint32_t tile = (uint32_t)(kv[k].key >> 32);
        if (k == 0 || tile != (uint32_t)(kv[k - 1].key >> 32)) ranges[2 * tile] = (uint32_t)k;
        if (k + 1 == n || tile != (uint32_t)(kv[k + 1].key >> 32)) ranges[2 * tile + 1] = (uint32_t)(k + 1);
    }
    free(kv);
    free(tmp);
    (void)gy;
}

/* ---------------------------------------------------------------- A.4 blend forward */
/* margin[pix] (optional): smallest relative distance of any evaluated decision that depends on
   exp() (alpha vs 1/255, T' vs 1e-4) from its threshold; tests use it to excuse n_contrib
   differences that come only from the 2-ulp freedom the spec gives exp(). */
void orc_blend_forward(int H, int W, const uint32_t* ranges, const uint32_t* ids_sorted,
                       const real* xy, const real* conic_opacity, const real* rgb, const float* bg,
                       real* out_color /*[3][H][W]*/, real* final_T /*[H][W]*/,
                       uint32_t* n_contrib /*[H][W]*/, real* margin /*[H][W] or NULL*/) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    ORC_PARALLEL_FOR
    for (int tile = 0; tile < gx * gy; ++tile) {
        {
            const int ty = tile / gx, tx = tile % gx;
            uint32_t lo = ranges[2 * (ty * gx + tx)], hi = ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < TILE; ++ly)
                for (int lx = 0; lx < TILE; ++lx) {
                    int px = tx * TILE + lx, py = ty * TILE + ly;
                    if (px >= W || py >= H) continue;
                    real pxf = (real)px, pyf = (real)py;
                    real T = R(1.0), C[3] = {0, 0, 0};
                    uint32_t contributor = 0, last = 0;
                    real mg = R(1e30);
                    for (uint32_t k = lo; k < hi; ++k) {
                        uint32_t g = ids_sorted[k];
                        ++contributor;
                        real dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
                        real A = R(-0.5) * conic_opacity[4 * g], B = -conic_opacity[4 * g + 1],
                             Cc = R(-0.5) * conic_opacity[4 * g + 2];
                        real power = r_fma(dx, r_fma(A, dx, B * dy), (Cc * dy) * dy);
                        if (power > 0) continue;
                        real al = conic_opacity[4 * g + 3] * r_exp(power);
                        real alpha = r_fmin(R(0.99), al);
                        real m1 = (alpha - R(1.0) / R(255.0)) * R(255.0);
                        if (m1 < 0) m1 = -m1;
                        if (m1 < mg) mg = m1;
                        if (alpha < R(1.0) / R(255.0)) continue;
                        real test_T = T * (R(1.0) - alpha);
                        real m2 = (test_T - R(0.0001)) * R(10000.0);
                        if (m2 < 0) m2 = -m2;
                        if (m2 < mg) mg = m2;
                        if (test_T < R(0.0001)) break;
                        real w = alpha * T;
                        for (int ch = 0; ch < 3; ++ch) C[ch] = r_fma(rgb[3 * g + ch], w, C[ch]);
                        T = test_T;
                        last = contributor;
                    }
                    size_t pix = (size_t)py * W + px;
                    for (int ch = 0; ch < 3; ++ch)
                        out_color[(size_t)ch * H * W + pix] = r_fma(T, (real)bg[ch], C[ch]);
                    final_T[pix] = T;
                    n_contrib[pix] = last;
                    if (margin) margin[pix] = mg;
                }
        }
    }
}

/* ---------------------------------------------------------------- A.5 blend backward */
/* Per pixel, back to front over the first n_contrib splats of its tile list.  Gradients w.r.t.
   per-Gaussian screen-space quantities:  dL_dmean2D[P][2] (PIXEL units), dL_dconic[P][3]
   (true partials w.r.t. Qxx,Qxy,Qyy), dL_dopacity[P], dL_dcolor[P][3].  Accumulated in `real`
   in tile-major, pixel-row-major order. */
void orc_blend_backward(int H, int W, const uint32_t* ranges, const uint32_t* ids_sorted,
                        const real* xy, const real* conic_opacity, const real* rgb, const float* bg,
                        const real* final_T, const uint32_t* n_contrib, const real* dL_dpix /*[3][H][W]*/,
                        int P, real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolor) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(dL_dmean2D, 0, sizeof(real) * 2 * (size_t)P);
    memset(dL_dconic, 0, sizeof(real) * 3 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(real) * (size_t)P);
    memset(dL_dcolor, 0, sizeof(real) * 3 * (size_t)P);
    ORC_PARALLEL_FOR
    for (int tile = 0; tile < gx * gy; ++tile) {
        {
            const int ty = tile / gx, tx = tile % gx;
            uint32_t lo = ranges[2 * (ty * gx + tx)];
            for (int ly = 0; ly < TILE; ++ly)
                for (int lx = 0; lx < TILE; ++lx) {
                    int px = tx * TILE + lx, py = ty * TILE + ly;
                    if (px >= W || py >= H) continue;
                    size_t pix = (size_t)py * W + px;
                    real pxf = (real)px, pyf = (real)py;
                    real T_final = final_T[pix];
                    real T = T_final;
                    uint32_t last = n_contrib[pix];
                    real dLp[3];
                    for (int ch = 0; ch < 3; ++ch) dLp[ch] = dL_dpix[(size_t)ch * H * W + pix];
                    real bg_dot = ((real)bg[0] * dLp[0] + (real)bg[1] * dLp[1]) + (real)bg[2] * dLp[2];
                    real accum[3] = {0, 0, 0}, last_alpha = 0, last_col[3] = {0, 0, 0};
                    for (uint32_t kk = last; kk-- > 0;) {
                        uint32_t g = ids_sorted[lo + kk];
                        real dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
                        real Qxx = conic_opacity[4 * g], Qxy = conic_opacity[4 * g + 1],
                             Qyy = conic_opacity[4 * g + 2], o = conic_opacity[4 * g + 3];
                        real A = R(-0.5) * Qxx, B = -Qxy, Cc = R(-0.5) * Qyy;
                        real power = r_fma(dx, r_fma(A, dx, B * dy), (Cc * dy) * dy);
                        if (power > 0) continue;
                        real G = r_exp(power);
                        real alpha = r_fmin(R(0.99), o * G);
                        if (alpha < R(1.0) / R(255.0)) continue;
                        T = T / (R(1.0) - alpha); /* transmittance in front of this splat */
                        real dchan = alpha * T;
                        real dL_dalpha = 0;
                        for (int ch = 0; ch < 3; ++ch) {
                            real c = rgb[3 * g + ch];
                            accum[ch] = last_alpha * last_col[ch] + (R(1.0) - last_alpha) * accum[ch];
                            last_col[ch] = c;
                            dL_dalpha += (c - accum[ch]) * dLp[ch];
                            ORC_ATOMIC
                            dL_dcolor[3 * g + ch] += dchan * dLp[ch];
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (R(1.0) - alpha)) * bg_dot;
                        /* straight-through min(0.99, .): alpha ~ o*G */
                        real dL_dG = o * dL_dalpha;
                        real gdx = G * dx, gdy = G * dy;
                        real dG_ddx = -gdx * Qxx - gdy * Qxy;
                        real dG_ddy = -gdy * Qyy - gdx * Qxy;
                        ORC_ATOMIC
                        dL_dmean2D[2 * g] += dL_dG * dG_ddx;
                        ORC_ATOMIC
                        dL_dmean2D[2 * g + 1] += dL_dG * dG_ddy;
                        ORC_ATOMIC
                        dL_dconic[3 * g] += R(-0.5) * gdx * dx * dL_dG;
                        ORC_ATOMIC
                        dL_dconic[3 * g + 1] += -gdx * dy * dL_dG;
                        ORC_ATOMIC
                        dL_dconic[3 * g + 2] += R(-0.5) * gdy * dy * dL_dG;
                        ORC_ATOMIC
                        dL_dopacity[g] += G * dL_dalpha;
                    }
                }
        }
    }
}

/* The same sums in the FORMULATION of the device's blend backward (splatco_amd/csrc/blend.hip, "moments about the
   quadrant's origin -> about the splat's centre"): per 8x8 pixel quadrant of a tile the raw moments
   M0 = sum Y, Mx = sum Y x, My, Mxx, Mxy, Myy of Y = (o G) dL/dalpha in the pixel's integer coordinates inside the quadrant,
   shifted once per (quadrant, splat) to the splat's centre with a = mean.x - x0, b = mean.y - y0:
     sum Y dx = a M0 - Mx          sum Y dx^2  = a (sum Y dx - Mx) + Mxx          sum Y dx dy = (a sum Y dy - b Mx) + Mxy
   summed over quadrants and tiles per Gaussian, then factored into dL/dmean2D, dL/dconic, dL/dopacity.  A STUDY leg
   (tools/exp/shift_study.py): run in binary32 next to orc_blend_backward it isolates what the raw-moment shift costs
   against sums taken directly about the centre, everything else (per-pixel values, visit order) being equal. */
void orc_blend_backward_raw_moments(int H, int W, const uint32_t* ranges, const uint32_t* ids_sorted,
                                    const real* xy, const real* conic_opacity, const real* rgb, const float* bg,
                                    const real* final_T, const uint32_t* n_contrib, const real* dL_dpix /*[3][H][W]*/,
                                    int P, real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolor,
                                    int centred /* 1: the per-quadrant sums are taken of dx, dy directly (no shift): the
                                                   re-association alone, as a control */) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(dL_dcolor, 0, sizeof(real) * 3 * (size_t)P);
    real* S = (real*)calloc((size_t)P * 6 + 1, sizeof(real));      /* per Gaussian: sum Y dx, Y dy, Y dx^2, Y dx dy, Y dy^2, Y */
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t lo = ranges[2 * tile], hi = ranges[2 * tile + 1];
        if (hi <= lo) continue;
        const size_t n = hi - lo;
        real* mom = (real*)calloc(n * 4 * 6, sizeof(real));        /* [entry][quadrant][M0, Mx, My, Mxx, Mxy, Myy] */
        for (int ly = 0; ly < TILE; ++ly)
            for (int lx = 0; lx < TILE; ++lx) {
                int px = tx * TILE + lx, py = ty * TILE + ly;
                if (px >= W || py >= H) continue;
                size_t pix = (size_t)py * W + px;
                real pxf = (real)px, pyf = (real)py;
                const int q = (ly >> 3) * 2 + (lx >> 3);
                const real xq = (real)(lx & 7), yq = (real)(ly & 7);
                real T_final = final_T[pix];
                real T = T_final;
                uint32_t last = n_contrib[pix];
                real dLp[3];
                for (int ch = 0; ch < 3; ++ch) dLp[ch] = dL_dpix[(size_t)ch * H * W + pix];
                real bg_dot = ((real)bg[0] * dLp[0] + (real)bg[1] * dLp[1]) + (real)bg[2] * dLp[2];
                real accum[3] = {0, 0, 0}, last_alpha = 0, last_col[3] = {0, 0, 0};
                for (uint32_t kk = last; kk-- > 0;) {
                    uint32_t g = ids_sorted[lo + kk];
                    real dx = xy[2 * g] - pxf, dy = xy[2 * g + 1] - pyf;
                    real Qxx = conic_opacity[4 * g], Qxy = conic_opacity[4 * g + 1],
                         Qyy = conic_opacity[4 * g + 2], o = conic_opacity[4 * g + 3];
                    real A = R(-0.5) * Qxx, B = -Qxy, Cc = R(-0.5) * Qyy;
                    real power = r_fma(dx, r_fma(A, dx, B * dy), (Cc * dy) * dy);
                    if (power > 0) continue;
                    real G = r_exp(power);
                    real alpha = r_fmin(R(0.99), o * G);
                    if (alpha < R(1.0) / R(255.0)) continue;
                    T = T / (R(1.0) - alpha);
                    real dchan = alpha * T;
                    real dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ++ch) {
                        real c = rgb[3 * g + ch];
                        accum[ch] = last_alpha * last_col[ch] + (R(1.0) - last_alpha) * accum[ch];
                        last_col[ch] = c;
                        dL_dalpha += (c - accum[ch]) * dLp[ch];
                        dL_dcolor[3 * g + ch] += dchan * dLp[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (R(1.0) - alpha)) * bg_dot;
                    const real Y = (o * G) * dL_dalpha;
                    real* m = mom + ((size_t)kk * 4 + q) * 6;
                    const real ux = centred ? dx : xq, uy = centred ? dy : yq;
                    const real Yx = Y * ux, Yy = Y * uy;
                    m[0] += Y;
                    m[1] += Yx;
                    m[2] += Yy;
                    m[3] += Yx * ux;
                    m[4] += Yx * uy;
                    m[5] += Yy * uy;
                }
            }
        for (size_t kk = 0; kk < n; ++kk) {
            const uint32_t g = ids_sorted[lo + kk];
            real part[6] = {0, 0, 0, 0, 0, 0};
            for (int q = 0; q < 4; ++q) {
                const real* m = mom + (kk * 4 + q) * 6;
                const real a = xy[2 * g] - (real)(tx * TILE + (q & 1) * 8), b = xy[2 * g + 1] - (real)(ty * TILE + (q >> 1) * 8);
                const real M0 = m[0], Mx = m[1], My = m[2], Mxx = m[3], Mxy = m[4], Myy = m[5];
                if (centred) {
                    for (int v = 0; v < 5; ++v) part[v] += m[v + 1];
                    part[5] += M0;
                    continue;
                }
                const real t1 = a * M0 - Mx, t2 = b * M0 - My;
                part[0] += t1;
                part[1] += t2;
                part[2] += a * (t1 - Mx) + Mxx;
                part[3] += (a * t2 - b * Mx) + Mxy;
                part[4] += b * (t2 - My) + Myy;
                part[5] += M0;
            }
            for (int v = 0; v < 6; ++v) S[(size_t)g * 6 + v] += part[v];
        }
        free(mom);
    }
    for (int g = 0; g < P; ++g) {
        const real* s = S + (size_t)g * 6;
        const real Qxx = conic_opacity[4 * g], Qxy = conic_opacity[4 * g + 1], Qyy = conic_opacity[4 * g + 2], o = conic_opacity[4 * g + 3];
        dL_dmean2D[2 * g] = -(Qxx * s[0]) - Qxy * s[1];
        dL_dmean2D[2 * g + 1] = -(Qyy * s[1]) - Qxy * s[0];
        dL_dconic[3 * g] = R(-0.5) * s[2];
        dL_dconic[3 * g + 1] = -s[3];
        dL_dconic[3 * g + 2] = R(-0.5) * s[4];
        dL_dopacity[g] = s[5] != 0 ? s[5] / o : 0;
    }
    free(S);
}

/* ---------------------------------------------------------------- A.5 preprocess backward */
/* Chain from (dL_dmean2D [pixel units], dL_dconic, dL_dopacity passthrough, dL_dcolor) to the
   operator inputs.  Recomputes the forward projection.  dL_dmeans2D_out is the extra output
   (vi) of A.5: (dL/dm.x * 0.5 W, dL/dm.y * 0.5 H, 0). */
void orc_preprocess_backward(int P, int M, const float* means3D, const float* scales,
                             const float* rotations, const float* cov3D_precomp, const float* shs,
                             const orc_settings* st, const int32_t* radii, const int32_t* clamped,
                             const real* dL_dmean2D, const real* dL_dconic, const real* dL_dcolor,
                             real* dL_dmeans3D, real* dL_dmeans2D_out, real* dL_dscales,
                             real* dL_drotations, real* dL_dcov3D, real* dL_dsh) {
    const int W = st->image_width, H = st->image_height;
    const real tanfovx = st->tanfovx, tanfovy = st->tanfovy;
    const real fx = (real)W / (R(2.0) * tanfovx), fy = (real)H / (R(2.0) * tanfovy);
    const float* V = st->viewmatrix;
    const float* Pm = st->projmatrix;
    ORC_PARALLEL_FOR
    for (int i = 0; i < P; ++i) {
        for (int k = 0; k < 3; ++k) { dL_dmeans3D[3 * i + k] = 0; dL_dmeans2D_out[3 * i + k] = 0; }
        if (dL_dscales) for (int k = 0; k < 3; ++k) dL_dscales[3 * i + k] = 0;
        if (dL_drotations) for (int k = 0; k < 4; ++k) dL_drotations[4 * i + k] = 0;
        if (dL_dcov3D) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0;
        if (dL_dsh) for (int k = 0; k < 3 * M; ++k) dL_dsh[(size_t)i * 3 * M + k] = 0;
        if (radii[i] <= 0) continue;
        const float* p = means3D + 3 * i;
        real t4[4], h[4];
        xform4(V, p, t4);
        xform4(Pm, p, h);
        real cov[6];
        if (cov3D_precomp) for (int k = 0; k < 6; ++k) cov[k] = cov3D_precomp[6 * i + k];
        else cov3d_from_scale_rot(scales + 3 * i, st->scale_modifier, rotations + 4 * i, cov);
        proj_state ps;
        cov2d(t4, fx, fy, tanfovx, tanfovy, V, cov, &ps);
        real a = ps.a, b = ps.b, c = ps.c;
        real det = a * c - b * b;
        real d2 = R(1.0) / (det * det);
        real gQxx = dL_dconic[3 * i], gQxy = dL_dconic[3 * i + 1], gQyy = dL_dconic[3 * i + 2];
        /* Q = (c, -b, a)/det */
        real dL_da = d2 * (-c * c * gQxx + b * c * gQxy - b * b * gQyy);
        real dL_db = d2 * (R(2.0) * b * c * gQxx - (det + R(2.0) * b * b) * gQxy + R(2.0) * a * b * gQyy);
        real dL_dc = d2 * (-b * b * gQxx + a * b * gQxy - a * a * gQyy);
        /* cov2D = T S T^T ; a = T0 S T0^T, b = T0 S T1^T, c = T1 S T1^T */
        real S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
        real (*T)[3] = ps.T;
        /* dL/dS[j][k] (full, symmetric contributions) = dL_da T0j T0k + dL_db T0j T1k + dL_dc T1j T1k */
        real gS[3][3];
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k)
                gS[j][k] = dL_da * T[0][j] * T[0][k] + dL_db * T[0][j] * T[1][k] + dL_dc * T[1][j] * T[1][k];
        /* 6-vector parametrisation: off-diagonals appear twice */
        real g6[6] = {gS[0][0], gS[0][1] + gS[1][0], gS[0][2] + gS[2][0], gS[1][1], gS[1][2] + gS[2][1], gS[2][2]};
        if (dL_dcov3D && cov3D_precomp) for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = g6[k];
        /* dL/dT */
        real ST0[3], ST1[3];
        for (int j = 0; j < 3; ++j) {
            ST0[j] = S[j][0] * T[0][0] + S[j][1] * T[0][1] + S[j][2] * T[0][2];
            ST1[j] = S[j][0] * T[1][0] + S[j][1] * T[1][1] + S[j][2] * T[1][2];
        }
        real gT[2][3];
        for (int j = 0; j < 3; ++j) {
            gT[0][j] = R(2.0) * dL_da * ST0[j] + dL_db * ST1[j];
            gT[1][j] = R(2.0) * dL_dc * ST1[j] + dL_db * ST0[j];
        }
        /* T = J W : T0c = J00 W0c + J02 W2c ; T1c = J11 W1c + J12 W2c */
#define Wm(i_, c_) ((real)V[(c_)*4 + (i_)])
        real gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
        for (int cc = 0; cc < 3; ++cc) {
            gJ00 += gT[0][cc] * Wm(0, cc);
            gJ02 += gT[0][cc] * Wm(2, cc);
            gJ11 += gT[1][cc] * Wm(1, cc);
            gJ12 += gT[1][cc] * Wm(2, cc);
        }
        real tz = t4[2], tx = ps.txc, ty = ps.tyc;
        real tz2 = tz * tz, tz3 = tz2 * tz;
        /* J00 = fx/tz ; J02 = -fx tx / tz^2 ; J11 = fy/tz ; J12 = -fy ty / tz^2 */
        real g_tx = -fx / tz2 * gJ02;
        real g_ty = -fy / tz2 * gJ12;
        real g_tz = -fx / tz2 * gJ00 - fy / tz2 * gJ11 + R(2.0) * fx * tx / tz3 * gJ02 + R(2.0) * fy * ty / tz3 * gJ12;
        /* A.5 (ii): where t.x/t.z (t.y/t.z) was clamped, the clamped value is a constant of the
           backward pass: its Jacobian-path gradient is dropped (family convention). */
        real g_t[3];
        g_t[0] = ps.clamp_x ? 0 : g_tx;
        g_t[1] = ps.clamp_y ? 0 : g_ty;
        g_t[2] = g_tz;
        /* t = W p + trans */
        real gm[3];
        for (int cc = 0; cc < 3; ++cc) gm[cc] = g_t[0] * Wm(0, cc) + g_t[1] * Wm(1, cc) + g_t[2] * Wm(2, cc);
#undef Wm
        /* mean2D path: m.x = ((hx*p_w + 1) W - 1)/2 */
        real p_w = R(1.0) / (h[3] + R(0.0000001));
        real gpx = dL_dmean2D[2 * i] * R(0.5) * (real)W;     /* dL/dp_proj.x */
        real gpy = dL_dmean2D[2 * i + 1] * R(0.5) * (real)H;
        dL_dmeans2D_out[3 * i] = gpx;
        dL_dmeans2D_out[3 * i + 1] = gpy;
        real g_hx = gpx * p_w, g_hy = gpy * p_w;
        real g_hw = -(gpx * h[0] + gpy * h[1]) * p_w * p_w;
        for (int cc = 0; cc < 3; ++cc)
            gm[cc] += g_hx * (real)Pm[cc * 4 + 0] + g_hy * (real)Pm[cc * 4 + 1] + g_hw * (real)Pm[cc * 4 + 3];
        /* SH colour path */
        if (shs && dL_dsh) {
            real d0[3] = {(real)p[0] - (real)st->campos[0], (real)p[1] - (real)st->campos[1],
                          (real)p[2] - (real)st->campos[2]};
            real n2 = (d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2];
            real n = r_sqrt(n2);
            real x = d0[0] / n, y = d0[1] / n, z = d0[2] / n;
            const float* sh = shs + (size_t)i * M * 3;
            real* gsh = dL_dsh + (size_t)i * M * 3;
            int deg = st->sh_degree;
            real gdir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
                real gc = clamped[3 * i + ch] ? 0 : dL_dcolor[3 * i + ch];
#define SHV(k) ((real)sh[(k)*3 + ch])
                gsh[0 * 3 + ch] = SH_C0 * gc;
                if (deg > 0) {
                    gsh[1 * 3 + ch] = -SH_C1 * y * gc;
                    gsh[2 * 3 + ch] = SH_C1 * z * gc;
                    gsh[3 * 3 + ch] = -SH_C1 * x * gc;
                    real dx_ = -SH_C1 * SHV(3), dy_ = -SH_C1 * SHV(1), dz_ = SH_C1 * SHV(2);
                    if (deg > 1) {
                        real xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
                        gsh[4 * 3 + ch] = SH_C2[0] * xy_ * gc;
                        gsh[5 * 3 + ch] = SH_C2[1] * yz * gc;
                        gsh[6 * 3 + ch] = SH_C2[2] * (R(2.0) * zz - xx - yy) * gc;
                        gsh[7 * 3 + ch] = SH_C2[3] * xz * gc;
                        gsh[8 * 3 + ch] = SH_C2[4] * (xx - yy) * gc;
                        dx_ += SH_C2[0] * y * SHV(4) + SH_C2[2] * R(2.0) * -x * SHV(6) + SH_C2[3] * z * SHV(7) + SH_C2[4] * R(2.0) * x * SHV(8);
                        dy_ += SH_C2[0] * x * SHV(4) + SH_C2[1] * z * SHV(5) + SH_C2[2] * R(2.0) * -y * SHV(6) + SH_C2[4] * R(2.0) * -y * SHV(8);
                        dz_ += SH_C2[1] * y * SHV(5) + SH_C2[2] * R(2.0) * R(2.0) * z * SHV(6) + SH_C2[3] * x * SHV(7);
                        if (deg > 2) {
                            gsh[9 * 3 + ch] = SH_C3[0] * y * (R(3.0) * xx - yy) * gc;
                            gsh[10 * 3 + ch] = SH_C3[1] * xy_ * z * gc;
                            gsh[11 * 3 + ch] = SH_C3[2] * y * (R(4.0) * zz - xx - yy) * gc;
                            gsh[12 * 3 + ch] = SH_C3[3] * z * (R(2.0) * zz - R(3.0) * xx - R(3.0) * yy) * gc;
                            gsh[13 * 3 + ch] = SH_C3[4] * x * (R(4.0) * zz - xx - yy) * gc;
                            gsh[14 * 3 + ch] = SH_C3[5] * z * (xx - yy) * gc;
                            gsh[15 * 3 + ch] = SH_C3[6] * x * (xx - R(3.0) * yy) * gc;
                            dx_ += SH_C3[0] * SHV(9) * R(3.0) * R(2.0) * xy_ + SH_C3[1] * SHV(10) * yz +
                                   SH_C3[2] * SHV(11) * -R(2.0) * xy_ + SH_C3[3] * SHV(12) * -R(3.0) * R(2.0) * xz +
                                   SH_C3[4] * SHV(13) * (-R(3.0) * xx + R(4.0) * zz - yy) +
                                   SH_C3[5] * SHV(14) * R(2.0) * xz + SH_C3[6] * SHV(15) * R(3.0) * (xx - yy);
                            dy_ += SH_C3[0] * SHV(9) * R(3.0) * (xx - yy) + SH_C3[1] * SHV(10) * xz +
                                   SH_C3[2] * SHV(11) * (-R(3.0) * yy + R(4.0) * zz - xx) +
                                   SH_C3[3] * SHV(12) * -R(3.0) * R(2.0) * yz + SH_C3[4] * SHV(13) * -R(2.0) * xy_ +
                                   SH_C3[5] * SHV(14) * -R(2.0) * yz + SH_C3[6] * SHV(15) * -R(3.0) * R(2.0) * xy_;
                            dz_ += SH_C3[1] * SHV(10) * xy_ + SH_C3[2] * SHV(11) * R(4.0) * R(2.0) * yz +
                                   SH_C3[3] * SHV(12) * R(3.0) * (R(2.0) * zz - xx - yy) +
                                   SH_C3[4] * SHV(13) * R(4.0) * R(2.0) * xz + SH_C3[5] * SHV(14) * (xx - yy);
                        }
                    }
                    gdir[0] += dx_ * gc; gdir[1] += dy_ * gc; gdir[2] += dz_ * gc;
                }
#undef SHV
            }
            /* d(normalize(v))/dv applied to gdir */
            real inv3 = R(1.0) / (n2 * n);
            real dotv = d0[0] * gdir[0] + d0[1] * gdir[1] + d0[2] * gdir[2];
            for (int cc = 0; cc < 3; ++cc) gm[cc] += (n2 * gdir[cc] - d0[cc] * dotv) * inv3;
        }
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = gm[k];
        /* Sigma3D -> scale, quaternion */
        if (!cov3D_precomp && dL_dscales && dL_drotations) {
            const float* s = scales + 3 * i;
            const float* q = rotations + 4 * i;
            real mod = st->scale_modifier;
            real r = q[0], x = q[1], y = q[2], z = q[3];
            real Rm[3][3];
            Rm[0][0] = R(1.0) - R(2.0) * (y * y + z * z); Rm[0][1] = R(2.0) * (x * y - r * z); Rm[0][2] = R(2.0) * (x * z + r * y);
            Rm[1][0] = R(2.0) * (x * y + r * z); Rm[1][1] = R(1.0) - R(2.0) * (x * x + z * z); Rm[1][2] = R(2.0) * (y * z - r * x);
            Rm[2][0] = R(2.0) * (x * z - r * y); Rm[2][1] = R(2.0) * (y * z + r * x); Rm[2][2] = R(1.0) - R(2.0) * (x * x + y * y);
            real sc[3] = {mod * (real)s[0], mod * (real)s[1], mod * (real)s[2]};
            /* S = L L^T, L = Rm diag(sc);  dL/dL = (gS + gS^T) L */
            real gL[3][3];
            for (int a_ = 0; a_ < 3; ++a_)
                for (int b_ = 0; b_ < 3; ++b_) {
                    real acc = 0;
                    for (int k = 0; k < 3; ++k) acc += (gS[a_][k] + gS[k][a_]) * (Rm[k][b_] * sc[b_]);
                    gL[a_][b_] = acc;
                }
            real gR[3][3];
            for (int b_ = 0; b_ < 3; ++b_) {
                real acc = 0;
                for (int a_ = 0; a_ < 3; ++a_) {
                    acc += gL[a_][b_] * Rm[a_][b_];
                    gR[a_][b_] = gL[a_][b_] * sc[b_];
                }
                dL_dscales[3 * i + b_] = acc * mod;
            }
            real gq_r = R(2.0) * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
            real gq_x = R(2.0) * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - R(2.0) * x * gR[1][1] - r * gR[1][2] + z * gR[2][0] + r * gR[2][1] - R(2.0) * x * gR[2][2]);
            real gq_y = R(2.0) * (-R(2.0) * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] - r * gR[2][0] + z * gR[2][1] - R(2.0) * y * gR[2][2]);
            real gq_z = R(2.0) * (-R(2.0) * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] - R(2.0) * z * gR[1][1] + y * gR[1][2] + x * gR[2][0] + y * gR[2][1]);
            dL_drotations[4 * i] = gq_r; dL_drotations[4 * i + 1] = gq_x;
            dL_drotations[4 * i + 2] = gq_y; dL_drotations[4 * i + 3] = gq_z;
        }
    }
}
