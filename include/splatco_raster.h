/*
 * include/splatco_raster.h  --  C-ABI of the MI355X-native differentiable Gaussian rasterizer.
 *
 * This is the drop-in boundary for SplatCo's gaussian_renderer.render() path.  The reference
 * binds the same functionality through the pybind module `diff_gaussian_rasterization._C`
 * (absent from /root/reference); the entry points below are what a
 * binding for the three reference call sites needs:
 *
 *   scr_visible_filter   <- GaussianRasterizer.visible_filter   gaussian_renderer/__init__.py:239-242
 *   scr_forward_plan/run <- GaussianRasterizer.forward          gaussian_renderer/__init__.py:163-171
 *   scr_backward         <- autograd backward of the above      train.py:240 (total_loss.backward())
 *   scr_mark_visible     <- GaussianRasterizer.markVisible      (operator family API; unused by SplatCo)
 *   scr_settings         <- GaussianRasterizationSettings       gaussian_renderer/__init__.py:145-158
 *
 * Conventions
 *   - Plain C, no exceptions, no torch types.  Every pointer is a DEVICE pointer unless the
 *     name ends in `_host`.  All arrays are contiguous fp32 / int32 / uint32 as stated.
 *   - `stream` is a hipStream_t passed as void*.  Nothing synchronises the device; the only host waits
 *     are the ones that return a data-dependent count (scr_forward_plan: num_rendered; scr_expand_plan; scr_mask_index_plan):
 *     the kernel posts the count to a pinned mailbox the calling thread polls, with a copy +
 *     hipStreamSynchronize as fallback (and a sync + error check after every kernel when settings.debug != 0).
 *   - The caller allocates every buffer (sizes from the scr_*_bytes queries), so several forward graphs
 *     (the mv views of train.py:171-240) can be alive at once.  Process-lifetime state of the library:
 *     one 64-byte pinned mailbox per calling host thread (created on first use), a per-device flag that
 *     the >64 KB dynamic-LDS attribute has been set, and the opt-in profiling event pool (scr_profile_*).
 *     One host thread per process/GPU.
 *   - Return value: 0 = ok, non-zero = error; scr_last_error() returns a thread-local message.
 */
#ifndef SPLATCO_RASTER_H
#define SPLATCO_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCR_ABI_VERSION 27
#define SCR_TILE 16 /* 16x16-pixel tiles: part of the result contract (tile rects, ranges, sort keys) */

/* The 12 fields of GaussianRasterizationSettings, same order (gaussian_renderer/__init__.py:145-158).
 * viewmatrix / projmatrix are the flattened row-major [4,4] tensors of scene/cameras.py:54-56
 * (row-vector convention: p_view = [x,y,z,1] * viewmatrix). */
typedef struct scr_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    const float* bg;         /* device [3] */
    float scale_modifier;
    const float* viewmatrix; /* device [16] */
    const float* projmatrix; /* device [16] */
    int32_t sh_degree;
    const float* campos;     /* device [3] */
    int32_t prefiltered;
    int32_t debug;
} scr_settings;

int scr_abi_version(void);
const char* scr_last_error(void);

/* ---- buffer sizes (bytes).  Buffers must be 256-byte aligned (any hipMalloc / torch allocation is). */
size_t scr_geom_bytes(int64_t P, int32_t image_height, int32_t image_width); /* per-Gaussian state + per-tile counters */
size_t scr_binning_bytes(int64_t num_rendered, int64_t max_tile_instances);  /* per tile-instance lists */
size_t scr_image_bytes(int32_t image_height, int32_t image_width);           /* final_T + n_contrib */
size_t scr_backward_scratch_bytes(int64_t num_rendered);                     /* per-instance gradient records */

/* ---- visible_filter: radii_out[P] int32 (> 0 <=> visible).  Either (scales, rotations) or cov3D_precomp. */
int scr_visible_filter(int64_t P, const float* means3D, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const scr_settings* settings, int32_t* radii_out,
                       void* stream);

/* ---- markVisible: present_out[P] uint8 (view-space z > 0.2). */
int scr_mark_visible(int64_t P, const float* means3D, const float* viewmatrix, uint8_t* present_out,
                     void* stream);

/* ---- forward, phase 1: projection / culling / tile counting / offsets.
 * Exactly one of (shs, colors_precomp) and one of ((scales, rotations), cov3D_precomp) non-NULL.
 * M = SH coefficients per Gaussian (shs is [P, M, 3]); opacities is [P] (or [P,1]).
 * Writes radii_out[P] and geom_buf; returns through plan_host[4] (host pointer; the call waits for
 * the numbers -- the scan kernel posts them to a pinned mailbox this thread polls, with a stream
 * synchronisation as fallback): [0] the number of (Gaussian, tile) instances, [1] the largest per-tile instance
 * count (it sizes the sort's grid) -- both go to scr_binning_bytes / scr_forward_run --, [2] unused here (see
 * scr_forward_plan_run), [3] the plan flags: an opaque word the caller hands back to scr_forward_run and scr_backward.
 * SCR_PLAN_NONFINITE_COLOUR: a visible Gaussian carries a colour that is NaN or +-Inf.  The reference
 * SKIPS a splat at every pixel it does not contribute to, so such a colour reaches only the pixels the splat does
 * contribute to; the fast blend kernels carry non-contributing splats with alpha 0 (0 * NaN would spread), so calls
 * with this bit run the kernels' select-based instantiations and give the reference's result, NaN for NaN. */
enum { SCR_PLAN_NONFINITE_COLOUR = 1,
       SCR_PLAN_LARGE_RECTS = 2 };  /* some Gaussian's tile rect has more than 32 tiles: scr_backward clears the gradient
                                     * records of its further tiles before the blend backward fills in the ones it writes.
                                     * The backward takes THIS bit from geom_buf (where the forward left it), not from its
                                     * plan_flags argument: a stale argument cannot leave uninitialised records in the sums.
                                     * SCR_PLAN_NONFINITE_COLOUR selects a kernel instantiation on the host and is taken
                                     * from the argument (a wrong value there only changes how NaN colours spread); with
                                     * settings->debug the argument is compared with geom_buf's word and a mismatch fails. */
int scr_forward_plan(int64_t P, int32_t M, const float* means3D, const float* scales,
                     const float* rotations, const float* cov3D_precomp, const float* opacities,
                     const float* shs, const float* colors_precomp, const scr_settings* settings,
                     void* geom_buf, int32_t* radii_out, int64_t* plan_host, void* stream);

/* ---- forward, phase 2: per-tile bucketing, depth sort, front-to-back blend.
 * out_color is [3, H, W] fp32.  geom_buf / binning_buf / image_buf must be kept for scr_backward. */
int scr_forward_run(int64_t P, int64_t num_rendered, int64_t max_tile_instances, int64_t plan_flags,
                    const scr_settings* settings, void* geom_buf, void* binning_buf, void* image_buf, float* out_color,
                    void* stream);

/* ---- forward, both phases in one call when the caller's guess of the binning size was good enough.
 * Same as scr_forward_plan; then, if binning_buf is not NULL and binning_capacity_bytes >= scr_binning_bytes(I, max tile),
 * scr_forward_run on it without returning to the caller in between (the GPU otherwise idles for the caller's allocation and
 * second call: 20 us of a 1 ms step).  plan_host[4]: instances, largest tile, 1 if phase 2 ran (0: allocate and call
 * scr_forward_run), plan flags.  A training loop's instance count moves by a few per cent per step: last step's count plus slack. */
int scr_forward_plan_run(int64_t P, int32_t M, const float* means3D, const float* scales, const float* rotations,
                         const float* cov3D_precomp, const float* opacities, const float* shs, const float* colors_precomp,
                         const scr_settings* settings, void* geom_buf, int32_t* radii_out, int64_t* plan_host,
                         void* binning_buf, size_t binning_capacity_bytes, void* image_buf, float* out_color, void* stream);

/* ---- backward.  dL_dcolor is [3,H,W].  Outputs (each may be NULL when its input was NULL):
 * dL_dmeans3D[P,3], dL_dmeans2D[P,3] (d/d NDC position, z = 0: the gradient SplatCo reads back
 * at scene/gaussian_model.py:779), dL_dcolors[P,3], dL_dsh[P,M,3], dL_dopacity[P], dL_dscales[P,3],
 * dL_drotations[P,4], dL_dcov3D[P,6].  Every output element is written (zeros for culled Gaussians).
 * Deterministic: bit-identical results run to run (no floating-point atomics).  scratch: scr_backward_scratch_bytes
 * (one 36-byte gradient record per (Gaussian, tile) instance; the entries of a tile's list behind every pixel's last
 * contributor get none -- the tile's cut key in the image buffer, written by the backward, tells which).
 * geom_buf and image_buf are WRITTEN (per-Gaussian record flags, per-tile cut keys): one backward at a time per forward
 * state; the forward's results in them are left intact, so the backward can be repeated.  plan_flags: plan_host[3]. */
int scr_backward(int64_t P, int32_t M, int64_t num_rendered, int64_t plan_flags, const float* means3D, const float* scales,
                 const float* rotations, const float* cov3D_precomp, const float* shs,
                 const scr_settings* settings, const int32_t* radii, void* geom_buf,
                 const void* binning_buf, void* image_buf, const float* dL_dcolor, void* scratch,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dsh,
                 float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                 void* stream);

/* ---- debug getters: copy the integer / float intermediates out of the opaque buffers (parity tests).
 * which: see SCR_DBG_*.  `out` is a device buffer of the stated element count. */
enum {
    SCR_DBG_TILES_TOUCHED = 0, /* uint32[P]          geom   */
    SCR_DBG_POINT_OFFSETS = 1, /* uint32[P] inclusive scan of tiles_touched  (geom, valid after forward_run) */
    SCR_DBG_RANGES = 2,        /* uint32[tiles][2]   geom   */
    SCR_DBG_POINT_LIST = 3,    /* uint32[I] sorted Gaussian ids  binning */
    SCR_DBG_N_CONTRIB = 4,     /* uint32[H*W]        image  */
    SCR_DBG_FINAL_T = 5,       /* float[H*W]         image  */
    SCR_DBG_SPLAT_RECORDS = 6, /* float[P][12]: mx,my,-Qxx/2,-Qxy,-Qyy/2,opacity,r,g,b,depth,rect bits x2   geom */
    SCR_DBG_QMASK = 7,         /* uint8[I]  quadrant mask of every sorted list entry (bit q: the splat can reach quadrant q)  binning */
    SCR_DBG_GM_INDEX = 8       /* uint32[I] Gaussian-major index of every sorted list entry (where its gradient record goes)   binning */
};
/* Test hook: force (1) or forbid (0) the deep-tile-list variants of the forward's sort and of the backward (no gm_index
 * array, per-Gaussian record flags), which the library otherwise selects when the lists average more than 8192 entries per
 * tile (-1: automatic, the default).  Process-global; set it before a forward call and keep it for that call's backward. */
int scr_debug_force_deep_lists(int mode);
int scr_debug_get(int which, int64_t P, int64_t num_rendered, int32_t image_height, int32_t image_width,
                  const void* geom_buf, const void* binning_buf, const void* image_buf, void* out,
                  void* stream);

/* ---- TriPlaneAttention of the level-0 grid (scene/grids.py:22-64; applied to cat(xy, xz, yz planes) on every call,
 * scene/grids.py:166-168) without the framework's ~60 small kernels per direction and without MIOpen's 7x7 convolution.
 * The three planes are [R, H, W] each (equal sizes), read in place; C = 3 R <= 24 stacked channels.
 *   scr_tpa_stats          : avg[C], max[C], arg[C] (pixel of the maximum) over the pixels (ChannelAttention's pools, :28-35);
 *                            the 15-number MLP + sigmoid that turns them into ca[C] stays with the caller.
 *   scr_tpa_forward        : y = ca x; s = (mean_c y, max_c y) [2,H,W], am = channel of the max [H*W] uint8;
 *                            sa = sigmoid(conv7x7(s, w[2,7,7], zero padding 3)) [H*W]; pair plane j [2R,H,W] =
 *                            (plane j | sa * y of plane j) -- the stacked planes the tri-plane sampler takes.
 *   scr_tpa_backward       : g_j = dL/d(pair plane j) -> d_j = dL/d(plane j) [R,H,W] (both halves, written, not added),
 *                            dca[C], dw[2,7,7].  Needs s, am, sa of the forward.  Deterministic (ordered partial sums).
 *   scr_tpa_backward_stats : adds the pools' gradients, d[c] += davg[c] / (H W) everywhere and dmax[c] at arg[c].
 * scratch: scr_tpa_scratch_bytes(R, H, W) bytes for every call. */
size_t scr_tpa_scratch_bytes(int32_t R, int32_t H, int32_t W);
int scr_tpa_stats(int32_t R, int32_t H, int32_t W, const float* p0, const float* p1, const float* p2, float* avg,
                  float* mx, int32_t* arg, void* scratch, void* stream);
int scr_tpa_forward(int32_t R, int32_t H, int32_t W, const float* p0, const float* p1, const float* p2, const float* ca,
                    const float* w, float* s, uint8_t* am, float* sa, float* pair0, float* pair1, float* pair2, void* stream);
int scr_tpa_backward(int32_t R, int32_t H, int32_t W, const float* p0, const float* p1, const float* p2, const float* ca,
                     const float* w, const float* s, const uint8_t* am, const float* sa, const float* g0, const float* g1,
                     const float* g2, float* d0, float* d1, float* d2, float* dca, float* dw, void* scratch, void* stream);
int scr_tpa_backward_stats(int32_t R, int32_t H, int32_t W, const float* davg, const float* dmax, const int32_t* arg,
                           float* d0, float* d1, float* d2, void* stream);

/* ---- fused neural-Gaussian expansion + opacity-mask compaction: the op chain of
 * gaussian_renderer/__init__.py:68-111 (mask = neural_opacity > 0; repeat / cat / boolean index /
 * split; sigmoid, normalize, FMA) as one streaming pass.  Candidate c = anchor v * k + slot; kept
 * candidates keep their order.  V anchors, k offsets per anchor, n = V*k candidates:
 *   neural_opacity[n], color[n,3], scale_rot[n,7], offsets (= _offset[V,k,3]: the 3 k floats of an anchor contiguous, the
 *   anchors' rows offsets_ld floats apart -- 3 k when packed, 72 when they are columns 35.. of scr_anchor_gather's g_fea),
 *   grid_scaling[V,6], anchor[V,3]  ->  xyz[P,3], color_out[P,3], opacity[P], scaling[P,3], rot[P,4],
 *   out_index[n] int32 (compacted index or -1), mask_out[n] uint8 (may be NULL).
 * scr_expand_plan counts the kept candidates (stream-synchronises once, like the boolean index it
 * replaces) and leaves workgroup offsets in `scratch` for scr_expand_run.  The backward writes every
 * element of its outputs and uses no atomics (deterministic). */
size_t scr_expand_scratch_bytes(int64_t n_candidates);
int scr_expand_plan(int64_t n_candidates, const float* neural_opacity, void* scratch,
                    int64_t* num_selected_host, void* stream);
/* mask -> ascending index list (the `t[visible_mask]` index of gaussian_renderer/__init__.py:23-29; what
 * torch.nonzero returns for a 1-D mask): plan counts the set bytes (one stream synchronisation, like scr_expand_plan),
 * run writes index[num_set] int64 and, when `inverse` is not NULL, inverse[n] int64 = the position of entry i in the
 * index list, -1 where the mask is clear (what scr_anchor_gather_backward takes as inverse_index: the framework builds
 * it with a fill, an arange and an index_put).  scratch: scr_expand_scratch_bytes(n). */
int scr_mask_index_plan(int64_t n, const uint8_t* mask, void* scratch, int64_t* num_set_host, void* stream);
int scr_mask_index_run(int64_t n, const uint8_t* mask, const void* scratch, int64_t* index, int64_t* inverse, void* stream);
int scr_expand_run(int64_t V, int32_t k, const float* neural_opacity, const float* color,
                   const float* scale_rot, const float* offsets, int32_t offsets_ld, const float* grid_scaling,
                   const float* anchor, const void* scratch, int32_t* out_index, uint8_t* mask_out,
                   float* xyz, float* color_out, float* opacity, float* scaling, float* rot, void* stream);
int scr_expand_backward(int64_t V, int32_t k, const float* scale_rot, const float* offsets, int32_t offsets_ld,
                        const float* grid_scaling, const int32_t* out_index, const float* g_xyz,
                        const float* g_color, const float* g_opacity, const float* g_scaling,
                        const float* g_rot, float* d_neural_opacity, float* d_color, float* d_scale_rot,
                        float* d_offsets, float* d_grid_scaling, float* d_anchor, const float* g_reg, int64_t P,
                        void* stream);
/* g_reg (device scalar, may be NULL) + P (the number of selected candidates = rows of g_scaling): the upstream gradient
 * of the view loss's regulariser mean(prod(scaling, 1)) (train.py:192-196).  When given, the kernel adds
 * g_reg / P * (product of the candidate's other two scaling components) to g_scaling on the fly -- what autograd would
 * otherwise materialise as a [P,3] tensor and add to the rasterizer's dL/dscales in a pass of its own. */

/* ---- tri-plane bilinear feature sampling (scene/grids.py:146-182).  One plane sample is
 * out[v, r] = grid_sample(plane[1,R,A,B], (gx, gy) in [-1,1], bilinear, align_corners=True, zeros
 * padding) -- gx indexes the last plane dimension (B), gy the dimension A.
 *
 * scr_triplane_forward: coords[V][cstride] holds the normalised (x, y, z) of every anchor in its first
 * three columns; the three planes xy[R,X,Y], xz[R,X,Z], yz[R,Y,Z] are sampled with the coordinate pairs
 * of scene/grids.py:148-150 ((y,x), (z,x), (z,y)) and written to out[v*ld + col_xy/col_xz/col_yz + r], i.e.
 * straight into the concatenated feature matrix the reference builds with torch.cat (:165,:181).
 * channel_last = 0: planes in the reference's layout [R,A,B]; 1: caller passes [A,B,R] copies (one or two
 * cache lines per sampled row instead of R; pays off when the planes exceed the L2); 2: caller passes the row-pair
 * layout [A-1,B,2,R] that scr_plane_row_pairs builds from [R,A,B] (the four corners of a sample are 4 R consecutive
 * floats: 1.6 instead of 2.6 random cache lines per sample at R = 5; twice the plane's size).
 *
 * scr_plane_sample_backward: `planes` (1 or 2) planes [R,A,B] that are sampled at the same positions (the plain and
 * the attended plane of the attention grid, scene/grids.py:174-181) receive the scatter-add of the four corner weights
 * times their block of R gradient columns (grad_out0 / grad_out1: the column-offset pointers into the gradient matrix of
 * row stride ld); (cx, cy) are the columns of coords that hold (gx, gy).  Both grad_plane arrays are overwritten.
 * The points' coordinates and gradient pieces are first moved into per-tile runs of records (32x32-cell tiles), then
 * summed per node in registers; the nodes on a tile's border go through a per-tile halo block and are added across the
 * (up to four) tiles that share them in a fixed order by a last small pass: no floating-point atomics anywhere, every element
 * of the plane gradient is written exactly once.  The cell sums are EXACT: every term g * w is rounded to a fixed-point
 * grid of its tile (2^-29 of the tile's largest |gradient value|) and summed in 64-bit integers, so the plane gradient
 * is a function of the SET of points -- the same bits in any order, run after run (a tile that holds a non-finite value
 * is summed in fp32 instead).  A crowded tile (more than 16 K records and 1/256 of the plane's: a flat or a
 * contracted scene) is cut into segments summed by a workgroup each; their integer sums add up to the same bits.
 * R <= 16 (with planes = 2: two passes above R = 5); scratch from
 * scr_plane_sample_scratch_bytes(V, A, B, R * planes).  The sample positions get no gradient (the reference detaches
 * them, scene/gaussian_model.py:210).
 *
 * scr_triplane_backward: the same for the three projections of a grid in one pass over the points (the backward of
 * scr_triplane_forward): grad_out is the gradient of the concatenated matrix (row stride ld); cols[3 * planes] (host
 * array) holds the first column of xy, xz, yz and, with planes = 2, of the second triple sampled at the same
 * positions; grad_planes[3 * planes] (host array of device pointers, same order) are overwritten.  Scratch from
 * scr_triplane_backward_scratch_bytes(V, X, Y, Z, R * planes). */
/* scr_triplane_backward_multi: the backward of up to three grids that were sampled at the SAME coordinates into one
 * matrix (FeaturePlanes: the attention grid with every plane stacked on its attended twin, R = 2 r, then one or two plain
 * grids, R = r) in one pass over the points: R/X/Y/Z/col[ngrids] (host arrays) give the channels per plane, the plane
 * sizes and the first gradient column of every grid (grid g owns columns col[g] .. col[g] + 3 R[g], xy | xz | yz, the
 * grids back to back); grad_planes[3 * ngrids] (host array of device pointers, grid-major) are overwritten.  Returns 3
 * WITHOUT touching anything that matters when the layout is not one the fused pass is built for -- the caller then calls
 * scr_triplane_backward per grid.  Scratch from scr_triplane_backward_multi_scratch_bytes.
 * Fused dx (ABI 27; nl_coef NULL: off): the sampled matrix's one consumer is the plane branch's BatchNorm-Linear; given
 * nl_coef = Gi [32][80] | k0 [80] | k1 [80] (scr_norm_linear_backward: coef_out), nl_dy [V,32] and nl_x = the sampled matrix
 * itself, the pass over the points forms every point's gradient row on the way (dx = k0 + x k1 + dy Gi) instead of reading a
 * [V, 3 R ngrids] matrix another kernel wrote; grad_out may then be NULL (if not, it is added). */
size_t scr_triplane_backward_multi_scratch_bytes(int64_t V, int32_t ngrids, const int32_t* R, const int32_t* X, const int32_t* Y,
                                                 const int32_t* Z);
int scr_triplane_backward_multi(int64_t V, const float* coords, int32_t cstride, int32_t ngrids, const int32_t* R,
                                const int32_t* X, const int32_t* Y, const int32_t* Z, const int32_t* col, const float* grad_out,
                                int32_t ld, float* const* grad_planes, void* scratch, const float* nl_coef, const float* nl_dy,
                                int32_t nl_lddy, const float* nl_x, int32_t nl_ldx, void* stream);
int scr_plane_row_pairs(int32_t R, int32_t A, int32_t B, const float* plane, float* pairs, void* stream);
int scr_triplane_forward(int64_t V, const float* coords, int32_t cstride, const float* xy, const float* xz,
                         const float* yz, int32_t R, int32_t X, int32_t Y, int32_t Z, int32_t channel_last, float* out,
                         int32_t ld, int32_t col_xy, int32_t col_xz, int32_t col_yz, void* stream);
size_t scr_plane_sample_scratch_bytes(int64_t V, int32_t A, int32_t B, int32_t channels);
size_t scr_triplane_backward_scratch_bytes(int64_t V, int32_t X, int32_t Y, int32_t Z, int32_t channels);
int scr_triplane_backward(int64_t V, const float* coords, int32_t cstride, int32_t R, int32_t X, int32_t Y, int32_t Z,
                          int32_t planes, const float* grad_out, int32_t ld, const int32_t* cols, float* const* grad_planes,
                          void* scratch, void* stream);
int scr_plane_sample_backward(int64_t V, const float* coords, int32_t cstride, int32_t cx, int32_t cy, int32_t R,
                              int32_t A, int32_t B, int32_t planes, const float* grad_out0, const float* grad_out1,
                              int32_t ld, float* grad_plane0, float* grad_plane1, void* scratch, void* stream);

/* ---- fused L1 + SSIM image loss (train.py:192-196, utils/loss_utils.py:17-63): img1 = rendered
 * image [C,H,W] (gets the gradient), img2 = ground truth.  Forward writes out2[0] = mean |img1-img2|
 * and out2[1] = mean SSIM (11x11 Gaussian window, sigma 1.5, zero padding) to DEVICE memory; with
 * with_grad != 0 it also leaves the three derivative maps in `scratch` for the backward, which
 * computes dimg1 = g_l1 * dL1/dimg1 + g_ssim * dSSIM/dimg1 with the upstream scalars read from
 * device memory (no host synchronisation anywhere).  Deterministic (fixed-order reductions). */
size_t scr_l1_ssim_scratch_bytes(int32_t C, int32_t H, int32_t W, int32_t with_grad);
int scr_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, void* scratch,
                        int32_t with_grad, float* out2, void* stream);
int scr_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                         const void* scratch, const float* g_l1, const float* g_ssim, float* dimg1,
                         void* stream);

/* The loss's scaling regulariser, mean_p(scaling[p,0] scaling[p,1] scaling[p,2]) (train.py:192-196:
 * `scaling.prod(dim=1).mean()`) -> out[1], and its gradient dscaling[P,3] = g[0] / P * (products of the other two).
 * One streaming pass per direction, ordered partial sums (deterministic); no host read. */
size_t scr_scaling_reg_scratch_bytes(int64_t P);
int scr_scaling_reg_forward(int64_t P, const float* scaling, void* scratch, float* out, void* stream);
int scr_scaling_reg_backward(int64_t P, const float* scaling, const float* g, float* dscaling, void* stream);

/* The L1 of the cross-view consistency term of the mv loop (train.py:208-217: `l1_loss(real_img1 - real_img2, gen_img1 - gen_img2)`):
 * out[0] = mean over the n elements of | (real1 - real2) - (gen1 - gen2) | for four equally shaped images, and its gradient
 * d_gen1 = -g[0] sign(.) / n, d_gen2 = +g[0] sign(.) / n (either may be NULL: in the sharded step a rank differentiates its own
 * image only).  One streaming pass per direction, ordered partial sums (deterministic); no host read. */
size_t scr_pair_l1_scratch_bytes(int64_t n);
int scr_pair_l1_forward(int64_t n, const float* gen1, const float* gen2, const float* real1, const float* real2, void* scratch,
                        float* out, void* stream);
int scr_pair_l1_backward(int64_t n, const float* gen1, const float* gen2, const float* real1, const float* real2, const float* g,
                         float* d_gen1, float* d_gen2, void* stream);

/* ---- the head of generate_neural_gaussians (gaussian_renderer/__init__.py:23-31) for the reference's sizes (feat 32, 10 offsets):
 * the four visible-anchor gathers, exp(_scaling) and the [V,71] concatenation in one pass.  visible_index[V] int64 = the
 * visible anchors in order; outputs feat[V,32], anchor[V,3], offsets[V,30], grid_scaling[V,6] = exp(scaling) and
 * g_fea[V,71] = cat of the four, with row stride g_fea_ld = 71 (packed) or 72 (16-byte aligned rows, the pad column
 * written as 0 / ignored on the way back: what the fused BatchNorm-Linear wants).  feat_out and offsets_out may be NULL
 * (with g_fea_ld = 72): their consumers then read columns 0..31 / 35..64 of g_fea through a row stride (scr_mlp_heads_*:
 * feat_ld, scr_expand_*: offsets_ld) and 62 of the 143 floats per anchor are not written twice.  The backward takes V (rows of the upstream gradients), inverse_index[N] int64 (row of every anchor, -1 = not visible) and
 * the upstream gradients of the five outputs (any may be NULL) and overwrites EVERY element of the four parameter
 * gradients [N,32] / [N,3] / [N,30] / [N,6] (zeros for invisible anchors; d exp applied): no atomics, no memset.
 * accumulate != 0: ADDS to what the four arrays hold instead (a further view of the same step writing into the same
 * gradient buffers: the caller hands the parameters' .grad memory itself, multiview.GradArena.sink).
 * A RANGE of anchors [n0, n0 + N) with n0 a multiple of 64 is the same call on offset pointers (inverse_index + n0, the
 * four gradient arrays + n0 rows; the upstream arrays and V -- their total row count -- unchanged): the sharded step
 * finishes the per-anchor gradients range by range so that a range's exchange overlaps the next range's kernel. */
/* col_stats_out (may be NULL): a buffer of scr_anchor_gather_stat_buffer_rows(V) rows of [2][80] floats whose first
 * scr_anchor_gather_stat_rows(V) rows receive the sums of (x - x[0]) and (x - x[0])^2 over disjoint row sets of g_fea, column
 * by column (the rest is per-tile scratch): the BatchNorm that reads g_fea takes them (scr_norm_linear_forward: col_stats)
 * instead of making a statistics pass of its own over the matrix. */
int32_t scr_anchor_gather_stat_rows(int64_t V);
int64_t scr_anchor_gather_stat_buffer_rows(int64_t V);
int scr_anchor_gather(int64_t V, const int64_t* visible_index, const float* anchor_feat, const float* anchor,
                      const float* offset, const float* scaling, float* feat_out, float* anchor_out, float* offsets_out,
                      float* grid_scaling_out, float* g_fea_out, int32_t g_fea_ld, float* col_stats_out, void* stream);
/* Fused dx (ABI 27; nl_coef NULL: off).  g_fea's one consumer is the attribute branch's BatchNorm-Linear, whose backward ends in
 *     d g_fea[v][n] = k0[n] + g_fea[v][n] k1[n] + sum_m dy[v][m] Gi[m][n]
 * -- a [V,72] matrix written by one kernel and read back by this one.  Given nl_coef = Gi [32][80] | k0 [80] | k1 [80] (what
 * scr_norm_linear_backward leaves in coef_out), nl_dy [V,32] (row stride nl_lddy, 16-byte aligned) and nl_x = g_fea [V,71]
 * (row stride nl_ldx) the backward forms those rows itself, workgroup by workgroup, and ADDS them to whatever d_g_fea
 * carries (normally NULL then): 1.15 GB less written and read at configs[2], and the whole tail of a step's backward pass
 * -- dx and the gather -- runs per anchor RANGE, so that a range's exchange overlaps the next range's kernels. */
int scr_anchor_gather_backward(int64_t N, int64_t V, const int64_t* inverse_index, const float* grid_scaling, const float* d_feat,
                               const float* d_anchor, const float* d_offsets, const float* d_grid_scaling,
                               const float* d_g_fea, int32_t g_fea_ld, float* g_anchor_feat, float* g_anchor,
                               float* g_offset, float* g_scaling, int32_t accumulate, const float* nl_coef, const float* nl_dy,
                               int32_t nl_lddy, const float* nl_x, int32_t nl_ldx, void* stream);

/* ---- BatchNorm1d in training mode folded into the Linear(d, 32) that follows it: the two nn.Sequential(BatchNorm1d,
 * Linear) stacks of FeaturePlanes (scene/gaussian_model.py:118-124,160-166) for all active levels at once.  The caller
 * folds the BatchNorm affine parameters and the levels into G [32,d] and c [32] (scene_model._norm_linear):
 *     y = xhat G^T + c,  xhat = (x - mean) * inv,  inv = 1 / sqrt(var + eps),  mean / var (biased) over the V rows.
 * forward writes y [V,32] and mean, var, inv [d] (the caller updates the running statistics from mean / var).
 * backward, given dy [V,32] (row stride lddy, 16-byte aligned), writes dG [32,d], dc [32] and, unless dx is NULL,
 * dx [V,d] (row stride lddx) including both BatchNorm reduction terms.  d <= 80; x rows may have any stride (16-byte
 * aligned rows take the wide-load path).  Sums over the rows are formed per workgroup and combined in a fixed order:
 * bit-reproducible.  Scratch from scr_norm_linear_scratch_bytes(V), the same block for forward and backward. */
size_t scr_norm_linear_scratch_bytes(int64_t V);
/* Normalised tri-plane sample coordinates of contiguous xyz[V,3] in the box [lo, hi] (host floats):
 * out = (xyz - lo) / (hi - lo) * 2 - 1 (scene/grids.py:146), the same IEEE operations in the same order as the four
 * elementwise passes of the reference, in one. */
int scr_box_coords(int64_t V, const float* xyz, const float* lo_host, const float* hi_host, float* out, void* stream);
/* The parameter side of the same fold.  FeaturePlanes sums L <= 4 pairs Linear_i(BatchNorm_i(.)) (scene/gaussian_model.py:
 * 149-169): pair i reads columns [cols[i], cols[i] + widths[i]) of the d-column input (the plane branch) or all of it
 * (cols[i] = 0, widths[i] = d: the attribute branch).  scr_norm_fold builds G[32, d] and c[32] from the pairs' parameters
 * (G[r, cols_i + j] += W_i[r, j] gamma_i[j];  c[r] = sum_i (W_i[r, :] . beta_i + b_i[r])), scr_norm_fold_backward turns
 * dG / dc into the gradients of the 4 L parameter tensors, scr_norm_running_stats applies nn.BatchNorm1d's running-statistics
 * update (momentum form) for all pairs from the batch mean / biased variance and n rows.  A few small launches each;
 * the `_host` arguments are HOST arrays of L widths / column offsets / DEVICE pointers.  col_at_host (NULL = identity):
 * d bytes, col_at[j] = the column of the input matrix (hence of G, mean, var) that holds reference column j -- for a
 * caller that keeps its input columns in another order than the reference's concatenation (two grids' planes stacked
 * and sampled together land interleaved). */
int scr_norm_fold(int32_t L, int32_t d, const int32_t* widths_host, const int32_t* cols_host, const uint8_t* col_at_host,
                  const void* const* lin_weight_host, const void* const* lin_bias_host, const void* const* bn_weight_host,
                  const void* const* bn_bias_host, float* G, float* c, void* stream);
int scr_norm_fold_backward(int32_t L, int32_t d, const int32_t* widths_host, const int32_t* cols_host, const uint8_t* col_at_host,
                           const void* const* lin_weight_host, const void* const* bn_weight_host, const void* const* bn_bias_host,
                           const float* dG, const float* dc, void* const* d_lin_weight_host, void* const* d_lin_bias_host,
                           void* const* d_bn_weight_host, void* const* d_bn_bias_host, void* stream);
int scr_norm_running_stats(int32_t L, int32_t d, const int32_t* widths_host, const int32_t* cols_host, const uint8_t* col_at_host,
                           const float* momentum_host, void* const* running_mean_host, void* const* running_var_host,
                           void* const* num_batches_host, const float* mean, const float* var, int64_t n, void* stream);
/* col_stats / col_stat_rows (NULL / 0: the statistics pass runs here): column statistics the producer of x already formed --
 * col_stat_rows rows of [2][80] floats, the sums of (x - x[0]) and (x - x[0])^2 over disjoint row sets that cover x
 * (scr_anchor_gather: col_stats_out); combined in fp64 like the partial sums of the built-in pass. */
int scr_norm_linear_forward(int64_t V, int32_t d, const float* x, int32_t ldx, const float* G, const float* c, float eps,
                            float* y, float* mean, float* var, float* inv, void* scratch, const float* col_stats,
                            int32_t col_stat_rows, void* stream);
/* coef_out (ABI 27; may be NULL): [32 + 2][80] floats that receive Gi = G * inv | k0 | k1, the coefficients of
 * dx[v][n] = k0[n] + x[v][n] k1[n] + sum_m dy[v][m] Gi[m][n], for a consumer that forms the rows of dx itself
 * (scr_anchor_gather_backward: nl_coef); with dx NULL the [V,d] matrix is then never written. */
/* Pass 3 of the backward on its own, from a coefficient block scr_norm_linear_backward left in coef_out: for a consumer of
 * the coefficients that cannot form the rows itself after all (a layout outside its fused pass). */
int scr_norm_linear_dx(int64_t V, int32_t d, const float* x, int32_t ldx, const float* dy, int32_t lddy, const float* coef,
                       float* dx, int32_t lddx, void* stream);
int scr_norm_linear_backward(int64_t V, int32_t d, const float* x, int32_t ldx, const float* dy, int32_t lddy, const float* G,
                             const float* mean, const float* inv, float* dx, int32_t lddx, float* dG, float* dc,
                             void* scratch, float* coef_out, void* stream);

/* ---- the three MLP heads of generate_neural_gaussians (gaussian_renderer/__init__.py:58-93 with the default flags,
 * scene/gaussian_model.py:315-337) as one fp32-MFMA kernel per direction, for the reference's layer sizes
 * (feat 32, geo_fea 64, hidden 32 per head, n_offsets 10):
 *   x = cat(feat[V,32], (anchor - campos) / |anchor - campos|, geo_fea[V,64])       (never materialised; geo_fea is passed as
 *   its two halves geo_a | geo_b [V,32], the outputs of FeaturePlanes' two GEMMs, scene/gaussian_model.py:160-168)
 *   out_opacity[V,10] = tanh(W2o relu(W1[0:32] x + b1[0:32]) + b2o);  out_color[V,30] = sigmoid(... [32:64] ...);
 *   out_cov[V,70] = W2v relu(W1[64:96] x + b1[64:96]) + b2v
 * feat rows are feat_ld floats apart (32 when packed; 16-byte aligned rows, feat_ld a multiple of 4).
 * w1[96,99] / b1[96] are the three first layers stacked (opacity, colour, cov).  hidden_save (opaque,
 * scr_mlp_heads_hidden_bytes) keeps the hidden layer for the backward pass.  The backward overwrites every output:
 * d_feat[V,32], d_anchor[V,3] (through ob_view), d_geo_a / d_geo_b [V,32] and the parameter gradients; weight-gradient partial
 * sums go through `partial` (scr_mlp_heads_partial_bytes) and are added in a fixed order (bit-reproducible). */
size_t scr_mlp_heads_hidden_bytes(int64_t V);
size_t scr_mlp_heads_partial_bytes(int64_t V);
int scr_mlp_heads_forward(int64_t V, const float* feat, int32_t feat_ld, const float* anchor, const float* campos, const float* geo_a, const float* geo_b,
                          const float* w1, const float* b1, const float* w2o, const float* b2o, const float* w2c,
                          const float* b2c, const float* w2v, const float* b2v, void* hidden_save, float* out_opacity,
                          float* out_color, float* out_cov, void* stream);
int scr_mlp_heads_backward(int64_t V, const float* feat, int32_t feat_ld, const float* anchor, const float* campos, const float* geo_a, const float* geo_b,
                           const float* w1, const float* w2o, const float* w2c, const float* w2v, const void* hidden_save,
                           const float* out_opacity, const float* out_color, const float* g_opacity, const float* g_color,
                           const float* g_cov, void* partial, float* d_feat, float* d_anchor, float* d_geo_a, float* d_geo_b, float* d_w1,
                           float* d_b1, float* d_w2o, float* d_b2o, float* d_w2c, float* d_b2c, float* d_w2v, float* d_b2v,
                           void* stream);

/* ---- densification statistics: GaussianModel.training_statis (scene/gaussian_model.py:761-782), the consumer of
 * dL_dmeans2D (train.py:264-266), over the V visible anchors of one view (k offsets each), in two steps so that the
 * sharded --mv step can broadcast the increments between them:
 *   scr_statis_compute: neural_opacity[V*k], out_index[V*k] (scr_expand_run's compaction index, -1 = not selected),
 *     update_filter[P] uint8 (radii > 0), viewspace_grad[P][grad_stride] (columns 0,1 = dL_dmeans2D.xy)
 *     -> inc_opacity[V] = sum_slot max(neural_opacity, 0);  inc_grad[V*k] = |grad.xy| of the rendered selected
 *     candidates, -1 elsewhere.
 *   scr_statis_apply: visible_index[V] int64 (anchor of every visible row) + the increments -> the four accumulators
 *     opacity_accum[N], anchor_demon[N], offset_gradient_accum[N*k], offset_denom[N*k] are updated in place.
 * No atomics: every (anchor, slot) belongs to one thread; bit-reproducible. */
int scr_statis_compute(int64_t V, int32_t k, const float* neural_opacity, const int32_t* out_index,
                       const uint8_t* update_filter, const float* viewspace_grad, int32_t grad_stride,
                       float* inc_opacity, float* inc_grad, void* stream);
int scr_statis_apply(int64_t V, int32_t k, const int64_t* visible_index, const float* inc_opacity, const float* inc_grad,
                     float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                     void* stream);

/* ---- the optimizer step of the training step BASELINE.json configs[3] / configs[4] time: train.py:310-312
 * `gaussians.optimizer.step()` with torch.optim.Adam(l, lr=0.0, eps=1e-15) (scene/gaussian_model.py:575; one group per
 * per-anchor parameter, lr set per group by the schedulers), no weight decay, no amsgrad.  One streaming pass over
 * parameter, gradient and the two moments of every tensor in the HOST table `tensors` (read during the call only):
 *   exp_avg    += (1 - beta1) (grad - exp_avg)            exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) grad^2
 *   param      -= step_size * exp_avg / (sqrt(exp_avg_sq) / bias_correction2_sqrt + eps)
 * with step_size = lr / (1 - beta1^step) and bias_correction2_sqrt = sqrt(1 - beta2^step) of the tensor's own step count,
 * both formed by the caller in DOUBLE (as torch does for non-capturable steps: Python floats) and rounded to binary32 once,
 * here, where torch's kernels round their scalar arguments; beta1, beta2, eps are doubles because 1 - beta is
 * formed in double and rounded once, as torch does (1 - 0.999f is 4.7e-5 off 0.001).  fp32, contiguous; any 4-byte alignment (16-byte
 * aligned tensors take the vector path).  Elementwise: bit-reproducible. */
typedef struct scr_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    double step_size, bias_correction2_sqrt;
} scr_adam_tensor;
int scr_adam_step(int32_t n_tensors, const scr_adam_tensor* tensors, double beta1, double beta2, double eps, void* stream);

/* ---- the tri-plane total-variation term of the training step: train.py:242-243 (`iteration % 4 == 0`, after
 * backward(), before optimizer.step()) -> scene/gaussian_model.py:217-220 (grid `level` of the active levels gets the
 * weight w * 0.5^(2 - level)) -> scene/grids.py:240-250 PlaneGrid.total_variation_add_grad(w): smooth-L1 (beta 1, 'sum')
 * of the neighbour differences along both axes of each of the grid's three planes, the six sums times w, / 6,
 * .backward().  Here: the closed-form derivative ADDED into `grad` in place, one streaming pass, no temporaries:
 *   grad[r,a,b] += h(p[a,b] - p[a-1,b]) - h(p[a+1,b] - p[a,b]) + h(p[a,b] - p[a,b-1]) - h(p[a,b+1] - p[a,b])
 * with h(d) = coef * clamp(d, -1, 1), terms with a neighbour outside the plane dropped.  coef = fp32(1/6) * fp32(w_level)
 * (the caller forms it; that is the product autograd forms).  `planes` is a HOST table read during the call only;
 * plane / grad: [channels, rows, cols] fp32 contiguous device buffers (the reference's [1, R, X, Y] parameters), distinct.
 * In a sharded step the term depends on the parameters only: add it ONCE, after the gradient SUM, on every rank.
 * Elementwise: bit-reproducible. */
typedef struct scr_tv_plane {
    const float* plane;
    float* grad;
    int32_t channels, rows, cols;
    float coef;
} scr_tv_plane;
int scr_tv_add_grad(int32_t n_planes, const scr_tv_plane* planes, void* stream);

/* ---- k nearest neighbours for GaussianModel.compute_curvature (scene/gaussian_model.py:1092-1110: sklearn on the host +
 * a Python loop over the anchors there).  The caller buckets the N points into a uniform grid (grid_host[7] = x0, y0, z0,
 * cell size h, nx, ny, nz -- HOST floats), sorts them by cell (sorted_pts[N,3], sorted_id[N] = original index,
 * cell_start[nx*ny*nz + 1]); scr_knn writes out_idx[N,k]: the k nearest OTHER points of every point, nearest first,
 * indexed by original point (k <= 16; exact).  scr_knn_curvature: lambda_min / trace of the covariance of those
 * neighbours (mean-centred, / (k-1)), per point. */
int scr_knn(int64_t N, int32_t k, const float* grid_host, const float* sorted_pts, const int64_t* sorted_id,
            const int32_t* cell_start, int64_t* out_idx, void* stream);
int scr_knn_curvature(int64_t N, int32_t k, const float* points, const int64_t* idx, float* curvature, void* stream);

/* ---- measurement aid: a float4 grid-stride copy of `bytes` bytes (src -> dst, device pointers).  bench.py times it to
 * quote the HBM bandwidth a streaming kernel reaches on the box next to the 8 TB/s datasheet figure. */
int scr_copy_probe(const void* src, void* dst, size_t bytes, void* stream);

/* ---- opt-in kernel timing (bench / profiling only; process-global, off by default).
 * scr_profile_enable(mask): bit i of mask selects kernel class i (SCR_PROF_*); -1 = all, 0 = off.
 * Launches of the selected classes are bracketed by hipEventRecord on the launch stream (each
 * event pair costs a few microseconds of stream time, so a benchmark times only the class it
 * reports).  scr_profile_read() waits for the recorded events, ADDS the elapsed time and
 * launch count of each kernel class since the last read into total_ms[SCR_PROF_COUNT] /
 * launches[SCR_PROF_COUNT], and resets.
 * NOT thread-safe: the event pool is process-global and unlocked.  Enable / read it from the one host thread that
 * makes the library's calls (the product never enables it; bench.py and tools/ do). */
enum {
    SCR_PROF_FILTER = 0, SCR_PROF_PREPROCESS = 1, SCR_PROF_PLAN_SCAN = 2, SCR_PROF_SCATTER = 3,
    SCR_PROF_TILE_SORT = 4, SCR_PROF_BLEND_FORWARD = 5, SCR_PROF_BLEND_BACKWARD = 6,
    SCR_PROF_PREPROCESS_BACKWARD = 7, SCR_PROF_EXPAND = 8, SCR_PROF_EXPAND_BACKWARD = 9, SCR_PROF_PLANE_BACKWARD = 10,
    SCR_PROF_L1_SSIM = 11, SCR_PROF_L1_SSIM_BACKWARD = 12, SCR_PROF_TRIPLANE_FORWARD = 13, SCR_PROF_MLP_HEADS = 14,
    SCR_PROF_MLP_HEADS_BACKWARD = 15, SCR_PROF_NORM_LINEAR = 16, SCR_PROF_NORM_LINEAR_BACKWARD = 17, SCR_PROF_PLANE_ATTENTION = 18,
    SCR_PROF_COUNT = 19
};
int scr_profile_enable(int mask);
/* Bracket only every `every`-th launch of a selected class (default 1 = every launch): a benchmark that times K steps
 * takes its dominant kernel's average from K / every samples and leaves the other steps undisturbed. */
int scr_profile_stride(int every);
int scr_profile_read(double* total_ms, int64_t* launches);
const char* scr_profile_kernel_name(int idx);

/* ---- opt-in stage markers (profiling only; process-global, off by default; ABI 27).
 * The reference brackets a whole training iteration with a pair of events (train.py:136-137,163,245); a profile of this
 * library wants the stages inside it.  scr_markers_enable(1) loads the roctx library on first use (librocprofiler-sdk-roctx.so,
 * then libroctx64.so; nothing is linked or loaded before) and from then on every C-ABI entry point and every kernel class
 * it launches (the SCR_PROF_* names) opens a roctx range on the calling thread, so that `rocprofv3 --marker-trace
 * --kernel-trace` attributes a step to scr_forward_plan / scr_forward_run / scr_backward / the anchor-path operators
 * and, inside them, to scatter / tile sort / blend / preprocess.  Off, a marker is one load and one branch.
 * scr_marker_push / scr_marker_pop let the host side add its own ranges (splatco_amd/_C.py stage(): prefilter, neural
 * Gaussians, rasterize, loss, backward, exchange, optimizer); they do nothing while markers are off.
 * Returns 0, or non-zero with scr_last_error() when no roctx library can be loaded. */
int scr_markers_enable(int on);
int scr_marker_push(const char* name);
int scr_marker_pop(void);

#ifdef __cplusplus
}
#endif
#endif /* SPLATCO_RASTER_H */
