// splatco_amd/csrc/blend.hip -- front-to-back tile alpha blending, forward and backward (gfx950).
//
// Decomposition (wave64-first): a 16x16 tile is four 8x8-pixel quadrants, ONE WAVE PER QUADRANT,
// one pixel per lane.  The tile's depth-sorted splat list carries a 4-bit quadrant mask per
// instance (computed by the scatter kernel, common.h quadrant_mask, carried in the sort key): a wave walks
// only the splats that can reach its quadrant (about a third of the list at the benchmark density), compacted with
// wave ballots + mbcnt prefix counts.  Splat records (48 B) are gathered once per surviving
// (wave, splat) into LDS and then read back with conflict-free broadcast ds_read_b128.
//
// Forward: waves are independent -> one-wave workgroups, no barriers; the next chunk's masks,
// ids and records are in flight while the current chunk is blended.
// Backward: the four waves of a tile share one workgroup.  Per (tile, Gaussian) gradient moments are
// reduced on chip -- inside the wave through a wave-private LDS transposition (pixels x values -> columns per
// value), a per-lane weighted contraction over eight pixels and one bank-masked DPP fold + two quad butterflies
// over the eight pixel columns (column_fold), one LDS slot per (wave, splat), a fixed-order add over the waves
// that took part -- and written ONCE as a 36-byte record at the instance's Gaussian-major index (so the
// per-Gaussian reduction reads its records contiguously).  Rounds behind every pixel's last contributor get no records: the tile leaves the
// sort key of its first entry without one (cut_key).  No floating-point atomics: results are
// bit-reproducible.  The per-Gaussian sum over tiles happens in preprocess_backward_kernel.
//
// Blend arithmetic is normative where a decision hangs on it: power is evaluated as
// fma(dx, fma(A,dx,B*dy), (C*dy)*dy) with A=-Qxx/2, B=-Qxy, C=-Qyy/2; exp() may differ from the
// oracle's libm by 2 ulp (v_exp_f32).  Compiled with -ffp-contract=off; FMAs only where written.
#include <type_traits>

#include "common.h"

namespace scr {

typedef float v2f __attribute__((ext_vector_type(2)));  // two fp32 lanes of a packed VALU op

// SCR_BLEND_COUNT (tools/blend_lane_use.py builds a variant library with it; never defined in the product): the blend
// kernels count how many of the 64 lanes of every staged (wave, splat) pair do useful work.
//   forward : [0] staged pairs  [1] lanes that blend the splat (hit, pixel not finished)  [2] lanes whose pixel is already finished
//             [3] pairs of groups that were skipped whole (no lane hit)
//   backward: [8] staged pairs  [9] lanes that contribute (alpha >= 1/255, power <= 0, not behind the pixel's last contributor)
//             [10] lanes behind their pixel's last contributor  [11] lanes inside the alpha >= 1/255 ellipse, whatever the order
//             [12] pairs in groups whose reduction was skipped (no lane hit)  [13] empty slots of partial groups
//   4x4-pixel sub-blocks (what a finer mask could pack, four sub-blocks per wave, each walking its own list):
//             [4] / [14] (pair, sub-block) combinations with a lane at work, forward / backward;
//             [5] / [15] sum over waves of the LONGEST of the wave's four sub-block lists (the rounds such a wave would take
//             if its sub-blocks never waited for each other);  [6] / [7] the same maximum taken per chunk of 64 list entries
//             and summed (sub-blocks in step chunk by chunk)
#ifdef SCR_BLEND_COUNT
// lanes of the four 4x4 sub-blocks of a quadrant.  forward: lane = y * 8 + x; backward: lane = x * 8 + y -- the same masks
__device__ __forceinline__ void count_sub_blocks(unsigned long long m, unsigned long long (&len)[4]) {
    len[0] += (m & 0x000000000f0f0f0full) != 0ull;
    len[1] += (m & 0x00000000f0f0f0f0ull) != 0ull;
    len[2] += (m & 0x0f0f0f0f00000000ull) != 0ull;
    len[3] += (m & 0xf0f0f0f000000000ull) != 0ull;
}
#define SCR_COUNT_SUB(m) (count_sub_blocks((m), sub_), count_sub_blocks((m), subc_))
// end of a chunk / round of 64 list entries: the longest of the four sub-block lists of THIS chunk (a packed wave that
// finishes every chunk before it starts the next takes the sum of these maxima)
#define SCR_COUNT_CHUNK_END(i)                                                                             \
    do {                                                                                                   \
        const unsigned long long a_ = subc_[0] > subc_[1] ? subc_[0] : subc_[1], b_ = subc_[2] > subc_[3] ? subc_[2] : subc_[3]; \
        cnt_[(i)] += a_ > b_ ? a_ : b_;                                                                    \
        subc_[0] = subc_[1] = subc_[2] = subc_[3] = 0;                                                     \
    } while (0)
#define SCR_COUNT_SUB_FLUSH(i_sum, i_max)                                                                  \
    do {                                                                                                   \
        cnt_[(i_sum)] += sub_[0] + sub_[1] + sub_[2] + sub_[3];                                            \
        const unsigned long long a_ = sub_[0] > sub_[1] ? sub_[0] : sub_[1], b_ = sub_[2] > sub_[3] ? sub_[2] : sub_[3]; \
        cnt_[(i_max)] += a_ > b_ ? a_ : b_;                                                                \
    } while (0)
#endif
#ifdef SCR_BLEND_COUNT
__device__ unsigned long long g_blend_counters[16];
#define SCR_COUNT(i, v) (cnt_[(i)] += (unsigned long long)(v))
#define SCR_COUNT_DECL unsigned long long cnt_[16] = {}, sub_[4] = {}, subc_[4] = {}
#define SCR_COUNT_FLUSH(lane)                                                        \
    if ((lane) == 0)                                                                 \
        for (int q_ = 0; q_ < 16; ++q_)                                              \
            if (cnt_[q_]) atomicAdd(&g_blend_counters[q_], cnt_[q_])
#else
#define SCR_COUNT(i, v) ((void)0)
#define SCR_COUNT_SUB(m) ((void)0)
#define SCR_COUNT_CHUNK_END(i) ((void)0)
#define SCR_COUNT_SUB_FLUSH(a, b) ((void)0)
#define SCR_COUNT_DECL ((void)0)
#define SCR_COUNT_FLUSH(lane) ((void)0)
#endif

// exp(x) = 2^(x log2 e) on the v_exp_f32 unit.  The product x * log2 e is rounded to 24 bits, an error of up to
// |x| 2^-24 in the exponent -- 3e-7 relative on G at the alpha = 1/255 end.  That looks harmless, but the
// transmittance is a product of (1 - alpha) factors, which amplifies a relative error of alpha by
// alpha / (1 - alpha) (up to 99): in opaque scenes with thousand-entry tile lists the gradients ended up
// 1e-4 off where libm's expf gives 1e-5 (tools/exp/diag_stress.py 32 18).
// What the rounding dropped is x - p ln 2 (one fma, exact up to its own rounding) and exp of that is 1 + it to
// first order, so two more VALU operations restore the accuracy of expf.
__device__ __forceinline__ float fast_exp(float x) {
    const float p = x * 1.4426950408889634f;
    const float g = __builtin_amdgcn_exp2f(p);
    const float r = __builtin_fmaf(-p, 0.6931471805599453f, x);  // x - p ln 2: the part of x that 2^p misses
    return __builtin_fmaf(g, r, g);
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long ballot) {  // popcount of lower lanes
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

// ------------------------------------------------------------------ longest tiles first, XCDs balanced
// A blend launch is a few rounds of tiles over the chip's workgroup slots; blocks are handed out in order, round-robin
// over the 8 XCDs.  Two things go wrong on a scene that is not uniform: the last, partial round lasts as long as the
// longest tile in it, and with the contiguous bands of xcd_tile the XCDs of the image centre hold most of the work
// while the others wait for their turn.  When the largest tile holds more than 1.5 x the mean the launcher runs this
// kernel first: every XCD gets the tiles of 4x4-tile blocks dealt round-robin (xcd_of_tile) and walks them longest list
// first.  One workgroup per XCD: counting sort of its tiles by list length (1024 bins), descending; ties and the order
// inside a bin follow LDS atomics -- it only schedules, results do not depend on it.  The lists lie back to back in
// `order` (the consumed tile_count array), their first entries in first[0..8]; *ordered = 1 tells the blend kernels.
// Uniform cfg1 scene: not run (its 6 us would not be repaid: 1.168 vs 1.165 ms per step with it).
__global__ void __launch_bounds__(1024)
tile_order_kernel(int tiles, int gx, const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order,
                  unsigned long long* __restrict__ ordered, unsigned long long* __restrict__ first) {
    __shared__ uint32_t hist[1024], start[1024], wsum[16];
    __shared__ uint32_t s_max, cnt[NUM_XCD];
    const int me = blockIdx.x;
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_max = 1;
    if (threadIdx.x < NUM_XCD) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mx = 0, mine[NUM_XCD] = {};
    for (int t = threadIdx.x; t < tiles; t += 1024) {
        const int x = xcd_of_tile(t, gx);
#pragma unroll
        for (int q = 0; q < NUM_XCD; ++q) mine[q] += x == q;
        if (x == me) mx = max(mx, ranges[2 * t + 1] - ranges[2 * t]);
    }
#pragma unroll
    for (int q = 0; q < NUM_XCD; ++q)
        if (mine[q]) atomicAdd(&cnt[q], mine[q]);
    if (mx) atomicMax(&s_max, mx);
    __syncthreads();
    uint32_t off = 0;
    for (int q = 0; q < me; ++q) off += cnt[q];
    const float scale = 1023.0f / (float)s_max;
    for (int t = threadIdx.x; t < tiles; t += 1024) {
        if (xcd_of_tile(t, gx) != me) continue;
        const uint32_t n = ranges[2 * t + 1] - ranges[2 * t];
        atomicAdd(&hist[1023 - min(1023u, (uint32_t)((float)n * scale))], 1u);      // bin 0 = the longest lists
    }
    __syncthreads();
    {   // exclusive scan of the 1024 bins
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t v = hist[threadIdx.x], inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int q = 0; q < w; ++q) base += wsum[q];
        start[threadIdx.x] = base + inc - v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += 1024) {
        if (xcd_of_tile(t, gx) != me) continue;
        const uint32_t n = ranges[2 * t + 1] - ranges[2 * t];
        order[off + atomicAdd(&start[1023 - min(1023u, (uint32_t)((float)n * scale))], 1u)] = (uint32_t)t;
    }
    if (threadIdx.x == 0) {
        first[me] = off;
        if (me == NUM_XCD - 1) first[NUM_XCD] = off + cnt[me];
        if (me == 0) *ordered = 1;
    }
}

// block -> tile of a blend launch: the XCD's list when the tiles were ordered, its contiguous band otherwise
__device__ __forceinline__ int blend_tile(int slot, int tiles, const uint32_t* __restrict__ order,
                                          const unsigned long long* __restrict__ total) {
    const int xcd = slot % NUM_XCD, j = slot / NUM_XCD;
    if (total[2]) {
        const uint32_t f = (uint32_t)total[4 + xcd], e = (uint32_t)total[5 + xcd];
        return f + j < e ? (int)order[f + j] : -1;
    }
    const int chunk = (tiles + NUM_XCD - 1) / NUM_XCD;
    const int t = xcd * chunk + j;
    return j < chunk && t < tiles ? t : -1;
}

// ------------------------------------------------------------------ forward
constexpr int FCHUNK = 64;  // list entries examined per round (one per lane)

// SAFE: some visible Gaussian of this call carries a colour that is not finite (NaN / Inf input; preprocess_kernel raised
// SCR_PLAN_NONFINITE_COLOUR and the host picked this instantiation).  Such a colour must reach only the pixels its splat
// contributes to, as it does where non-contributing splats are SKIPPED; blended with alpha 0 it would turn 0 * colour into
// NaN for every pixel of every quadrant that stages the record.  The SAFE instantiation selects the colour sums instead.
template <bool SAFE>
__global__ void __launch_bounds__(64)
blend_forward_kernel(int W, int H, int gx, int tiles, const uint32_t* __restrict__ order, const unsigned long long* __restrict__ total,
                     const uint32_t* __restrict__ ranges,
                     const uint32_t* __restrict__ point_list, const uint8_t* __restrict__ qmask,
                     const float4* __restrict__ rec, const float* __restrict__ bg,
                     float* __restrict__ out_color, float* __restrict__ final_T,
                     uint32_t* __restrict__ n_contrib) {
    // pair-interleaved staging: entry q holds splats 2q and 2q+1 field by field, so that one
    // ds_read_b128 lands (field of splat 2q, field of splat 2q+1) in adjacent registers and the
    // per-pixel arithmetic runs as packed fp32 (v_pk_fma_f32 & co: two splats per instruction)
    __shared__ float4 sp[5][FCHUNK / 2];  // (mx,mx',my,my') (A,A',B,B') (C,C',o,o') (r,r',g,g') (b,b',j,j')
    // the four quadrant waves of a tile are consecutive slots of one XCD
    const int quad = (blockIdx.x / NUM_XCD) & 3;
    const int t = blend_tile((int)(blockIdx.x % NUM_XCD + (blockIdx.x / (4 * NUM_XCD)) * NUM_XCD), tiles, order, total);
    if (t < 0) return;
    const int lane = threadIdx.x;
    const int px = (t % gx) * TILE + (quad & 1) * 8 + (lane & 7);
    const int py = (t / gx) * TILE + (quad >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t lo = ranges[2 * t], n = ranges[2 * t + 1] - lo;
    unsigned long long done = lanes(!inside);  // lane mask of finished pixels
    SCR_COUNT_DECL;
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    float c099 = 0.99f;
    asm volatile("" : "+v"(c099));  // keep the clamp in a VGPR: VOP2 with a literal issues slower
    uint32_t last = 0;

    // software pipeline: (mask, id) two chunks ahead, gathered records one chunk ahead
    uint32_t m_next = 0, id_next = 0;           // chunk c+1's mask bit / id for this lane
    bool sel_cur = false;                       // chunk c: this lane holds a surviving splat
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
    float r2x = 0.0f;
    auto load_mask_id = [&](uint32_t base, uint32_t& m, uint32_t& id) {
        uint32_t i = base + lane;
        bool have = i < n;
        m = have ? ((qmask[lo + i] >> quad) & 1u) : 0u;
        id = have ? point_list[lo + i] : 0u;
    };
    auto gather = [&](uint32_t m, uint32_t id) {
        sel_cur = m != 0;
        if (sel_cur) {
            r0 = rec[3 * (size_t)id];
            r1 = rec[3 * (size_t)id + 1];
            r2x = rec[3 * (size_t)id + 2].x;
        }
    };
    // slots past a chunk's count are blended with alpha 0: they must hold finite numbers
    for (int i = lane; i < 5 * (FCHUNK / 2); i += WAVE) (&sp[0][0])[i] = make_float4(0, 0, 0, 0);
    uint32_t m0, id0;
    load_mask_id(0, m0, id0);
    gather(m0, id0);
    load_mask_id(FCHUNK, m_next, id_next);

    for (uint32_t base = 0; base < n; base += FCHUNK) {
        // ---- stage chunk `base` (already in registers), compacted in list order
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(sel_cur);
        const int cnt = __builtin_popcountll(bal);
        if (sel_cur) {
            const uint32_t pos = lanes_below(bal), q = pos >> 1, h = pos & 1;
            float* f0 = (float*)&sp[0][q];
            float* f1 = (float*)&sp[1][q];
            float* f2 = (float*)&sp[2][q];
            float* f3 = (float*)&sp[3][q];
            float* f4 = (float*)&sp[4][q];
            f0[h] = r0.x; f0[2 + h] = r0.y;
            f1[h] = r0.z; f1[2 + h] = r0.w;
            f2[h] = r1.x; f2[2 + h] = r1.y;
            f3[h] = r1.z; f3[2 + h] = r1.w;
            f4[h] = r2x;  f4[2 + h] = __uint_as_float(base + lane + 1);  // contributor number
        }
        // ---- put the next chunk's gathers and the one after's mask/id loads in flight
        gather(m_next, id_next);
        load_mask_id(base + 2 * FCHUNK, m_next, id_next);
        __syncthreads();  // one-wave workgroup: orders the LDS writes above before the reads below
        // two packed pairs per group: all LDS reads and both exponent chains are in flight before the sequential
        // (front-to-back) transmittance updates.  A chunk's last group holds one pair only in half of the chunks: NP = 1
        // skips the missing pair (wave-uniform) instead of blending it with alpha 0.
        auto group = [&](const int k, auto npairs) {
            constexpr int NP = decltype(npairs)::value, NS = 2 * NP;
            v2f power[NP], al[NP];
            float4 p3[NP], p4[NP];
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                const int q = (k >> 1) + h;  // pair index; slots past cnt hold stale data and are masked below
                const float4 p0 = sp[0][q], p1 = sp[1][q], p2 = sp[2][q];
                p3[h] = sp[3][q];
                p4[h] = sp[4][q];
                const v2f dx = v2f{p0.x, p0.y} - pxf, dy = v2f{p0.z, p0.w} - pyf;
                const v2f A = {p1.x, p1.y}, B = {p1.z, p1.w}, Cq = {p2.x, p2.y}, o = {p2.z, p2.w};
                // normative order: fma(dx, fma(A,dx,B*dy), (C*dy)*dy), both splats per instruction
                power[h] = __builtin_elementwise_fma(dx, __builtin_elementwise_fma(A, dx, B * dy), (Cq * dy) * dy);
                al[h] = o * v2f{fast_exp(power[h].x), fast_exp(power[h].y)};
            }
            float alpha[NS], pw[NS];
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                alpha[2 * h] = vmin(c099, al[h].x); alpha[2 * h + 1] = vmin(c099, al[h].y);
                pw[2 * h] = power[h].x; pw[2 * h + 1] = power[h].y;
            }
            // lane masks (SGPR pairs): which pixels does splat u touch
            unsigned long long hit[NS], anyh = 0ull;
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                hit[u] = (k + u < cnt) ? (lanes(!(pw[u] > 0.0f)) & lanes(!(alpha[u] < 1.0f / 255.0f))) : 0ull;
                anyh |= hit[u];
            }
#ifdef SCR_BLEND_COUNT
            for (int u = 0; u < NS; ++u)
                if (k + u < cnt) {
                    SCR_COUNT(0, 1);
                    SCR_COUNT(2, __builtin_popcountll(done & lanes(inside)));
                    SCR_COUNT(3, (anyh & ~done) == 0ull);
                }
#endif
            if ((anyh & ~done) == 0ull) return;
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int h = u >> 1;
                const float cr = (u & 1) ? p3[h].y : p3[h].x, cg = (u & 1) ? p3[h].w : p3[h].z;
                const float cb = (u & 1) ? p4[h].y : p4[h].x, cj = (u & 1) ? p4[h].w : p4[h].z;
                // A pixel the splat does not touch, a finished pixel, and the pixel this very splat would finish (the
                // reference tests T (1 - alpha) < 1e-4 BEFORE blending and drops the splat) all blend it with alpha 0,
                // which leaves T and C bit-for-bit unchanged (T * (1 - 0), fma(c, 0 * T, C)): ONE select on alpha does
                // the work of three (alpha, the weight, the new T) -- selects issue at half the rate of plain fp32.
                // The stop test runs on the unselected alpha and is masked afterwards; T never drops below 1e-4 (the
                // update that would do so is the stop).
                const unsigned long long hm = hit[u] & ~done;
                const unsigned long long stop = lanes(T * (1.0f - alpha[u]) < 0.0001f) & hm;
                done |= stop;
                const unsigned long long live = hm & ~stop;
                SCR_COUNT(1, __builtin_popcountll(live));
                SCR_COUNT_SUB(live);
                const float a = sel(live, alpha[u], 0.0f);
                const float w = a * T;
                if (SAFE) {     // 0 * colour is not 0 for a colour that is not finite
                    C0 = sel(live, __builtin_fmaf(cr, w, C0), C0);
                    C1 = sel(live, __builtin_fmaf(cg, w, C1), C1);
                    C2 = sel(live, __builtin_fmaf(cb, w, C2), C2);
                } else {
                    C0 = __builtin_fmaf(cr, w, C0);
                    C1 = __builtin_fmaf(cg, w, C1);
                    C2 = __builtin_fmaf(cb, w, C2);
                }
                T = T * (1.0f - a);
                last = sel(live, __float_as_uint(cj), last);
            }
        };
        int k = 0;
        for (; k + 2 < cnt; k += 4) group(k, std::integral_constant<int, 2>{});
        if (k < cnt) group(k, std::integral_constant<int, 1>{});
        SCR_COUNT_CHUNK_END(6);
        if (~done == 0ull) break;
        __syncthreads();  // reads of this chunk finished before the next chunk overwrites LDS
    }
    SCR_COUNT_SUB_FLUSH(4, 5);
    SCR_COUNT_FLUSH(lane);
    if (inside) {
        size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
        out_color[pix] = __builtin_fmaf(T, bg[0], C0);
        out_color[hw + pix] = __builtin_fmaf(T, bg[1], C1);
        out_color[2 * hw + pix] = __builtin_fmaf(T, bg[2], C2);
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ------------------------------------------------------------------ wave64 sums, four splats at a time
// What a wave owes per splat are nine sums over its 64 pixels: the six moments of Y = opacity G dL/dalpha about the
// QUADRANT's origin (weights 1, x, y, x^2, xy, y^2 of the lane's pixel (x, y) = (lane >> 3, lane & 7)) and the three colour
// gradients sum_pixel (alpha T) dL/dpixel_c.  Reduced IN PLACE (round 2-5: v_permlane32/16_swap folds, then bank-masked
// DPP adds) that costs 18 two-pass swaps + 17 DPP adds + 20 adds per four splats, and three products per pixel and splat
// for the colour terms.  Since round 6 the eight values of four splats (Y and alpha T each) go through a wave-private LDS
// transposition instead: lane (x, y) stores them as eight columns [column][pixel] (eight ds_write_b32, conflict-free),
// lane (column c = lane >> 3, x = lane & 7) reads back the eight pixels (x, y = 0..7) of its column (two ds_read_b128)
// and contracts them with per-lane weights that are constants of the tile --
//     Y columns (c < 4):       (1, y, y^2)                  ->  S0 = sum_y Y,  S1 = sum_y y Y,  S2 = sum_y y^2 Y
//     alpha T columns (c >= 4): dL/dpixel_0..2 at (x, y)    ->  the three colour sums of the pixel column x
// -- 24 multiply-adds (two half-chains of four per sum, added: 27 issues), the same instruction stream for both kinds.  Three
// more products (x S0, x^2 S0, x S1; the weight x is 0 in the
// colour lanes) and the sum over the eight x lanes of a column is left: one bank-masked DPP level that folds the six values
// into three registers (the 4-lane banks of even parity keep (S0, S1, S2), the odd ones (x S0, x^2 S0, x S1)) and two
// quad_perm butterflies -- 12 DPP adds.  Per four splats: 30 plain + 12 DPP issues where the folds took 18 swaps + 37;
// always added in the same order (bit-reproducible).  The DPP part is one asm block: hipcc splits the builtin DPP form into
// v_mov_dpp + v_add pairs padded with s_nop; here the three chains are interleaved so that every DPP source was written at
// least two instructions earlier (the 2-wait-state VALU-write -> DPP-read hazard); the leading s_nop 1 covers the
// instruction before the block.
#define SCR_DPP(d, s, ctrl) "v_add_f32_dpp " d ", " s ", " s " " ctrl "\n\t"
// in: s0..s2 = the lane's three direct sums, t0..t2 = their x-weighted companions; out, by parity of the lane's 4-lane bank:
// s0 = (sum s0 | sum t0), s1 = (sum s1 | sum t1), s2 = (sum s2 | sum t2) over the eight lanes of the column
__device__ __forceinline__ void column_fold(float& s0, float& s1, float& s2, float t0, float t1, float t2) {
    asm volatile("s_nop 1\n\t"
                 SCR_DPP("%0", "%0", "row_shl:4 row_mask:0xf bank_mask:0x5")
                 SCR_DPP("%1", "%1", "row_shl:4 row_mask:0xf bank_mask:0x5")
                 SCR_DPP("%2", "%2", "row_shl:4 row_mask:0xf bank_mask:0x5")
                 SCR_DPP("%0", "%3", "row_shr:4 row_mask:0xf bank_mask:0xa")
                 SCR_DPP("%1", "%4", "row_shr:4 row_mask:0xf bank_mask:0xa")
                 SCR_DPP("%2", "%5", "row_shr:4 row_mask:0xf bank_mask:0xa")
                 SCR_DPP("%0", "%0", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 SCR_DPP("%1", "%1", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 SCR_DPP("%2", "%2", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 SCR_DPP("%0", "%0", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 SCR_DPP("%1", "%1", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 SCR_DPP("%2", "%2", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(s0), "+v"(s1), "+v"(s2)
                 : "v"(t0), "v"(t1), "v"(t2));
}
constexpr int TCS = WAVE + 4;   // words per column of a wave's transposition block (the four extra words keep the two
                                // columns of a 16-lane row on different banks in the ds_read_b128 above)

// Per-pixel gradient terms of one splat (back-to-front recurrences).  Not decision bearing, so the
// compiler may contract mul+add pairs here (fewer VALU issues); the tolerance is the gradient bar of
// the parity tests (rel-L2 <= 1e-4 vs the oracle).
//
// Everything that is constant per splat is moved out of the per-pixel work (A.5 restated):
//   * colour enters dL/dalpha only through d = <colour, dL/dpixel>, so the "colour behind this splat"
//     accumulator of the reference (3 channels) becomes ONE scalar recurrence on
//     behind = <accumulated colour behind, dL/dpixel>, started at <bg, dL/dpixel> (the background is
//     the last layer: T_final * bg), which also absorbs the separate background term;
//   * with Y = opacity * G * dL/dalpha, the position / conic gradients are conic-weighted
//     combinations of the five moments  sum Y dx, sum Y dy, sum Y dx^2, sum Y dx dy, sum Y dy^2;
//     the weights are applied once per Gaussian in preprocess_backward_kernel.
// A wave therefore reduces, per splat: the moments of Y (six: 1, x, y, x^2, xy, y^2 -- taken about the QUADRANT's
// origin, so that their weights are constants of the tile; the wave shifts them to the splat's centre once per entry) and
// the three colour gradients  sum alpha T dL/dpixel_c: a pixel hands over Y and w = alpha T (column_fold and above).
struct PixState {
    float T, dLp0, dLp1, dLp2;
    float behind, last_alpha, d_last;
};
// SAFE (a colour that is not finite is among the call's Gaussians, see blend_forward_kernel): the colour term of a splat
// that does not contribute to this pixel (`hit` clear; G = alpha = 0) is dropped by a select -- 0 * NaN is not 0, and the
// reference skips such a splat altogether.
template <bool SAFE>
__device__ __forceinline__ void splat_pixel_grad(PixState& s, float T, float4 b, float cb, float G, float alpha,
                                                 unsigned long long hit, float& Y, float& w) {
#pragma clang fp contract(fast)
    // T = transmittance in front of this splat (group_transmittance)
    w = alpha * T;
    // (SAFE: the reference's form of the same interpolation -- it keeps an infinite accumulator infinite where a (d - b) + b
    // makes Inf - Inf of it)
    s.behind = SAFE ? s.last_alpha * s.d_last + (1.0f - s.last_alpha) * s.behind : s.last_alpha * (s.d_last - s.behind) + s.behind;
    float d = b.z * s.dLp0 + b.w * s.dLp1 + cb * s.dLp2;
    if (SAFE) d = sel(hit, d, 0.0f);
    Y = G * (T * (d - s.behind));  // G = opacity * exp(power) here: the unclamped alpha times dL/dalpha (straight-through min(0.99, .))
    if (SAFE) Y = sel(hit, Y, 0.0f);     // `behind` is NaN once a NaN colour contributed to this pixel; a skipped splat takes nothing from it
    s.last_alpha = alpha;
    s.d_last = d;
}
// Transmittance in front of each of the four splats of a group, walked back to front: T_u = T_(u-1) / (1 - alpha_u).
// ONE division per group instead of one per splat: the front-most value is T / (om0 om1 om2 om3), correctly rounded
// as the division of the reference arithmetic is -- v_rcp_f32, one Newton step, one residual correction of the quotient
// (four fmas instead of the ten-instruction IEEE sequence) -- and the three behind it follow by multiplication.  The
// recurrence that runs over the hundreds to thousands of splats of a pixel (whose accumulated rounding the conic
// gradients of needle-like or deeply buried Gaussians amplify, tools/exp/diag_stress.py) now takes one step per group:
// three roundings in the product and half an ulp in the quotient per FOUR splats, where the per-splat division took
// half an ulp per splat; the other three values of a group are off that chain.  Saves 9 VALU operations and three
// v_rcp_f32 per group.
__device__ __forceinline__ void group_transmittance(float& T, const float (&om)[4], float (&Tu)[4]) {
#pragma clang fp contract(off)
    const float P = ((om[0] * om[1]) * om[2]) * om[3];
    float r = __builtin_amdgcn_rcpf(P);
    r = __builtin_fmaf(__builtin_fmaf(-P, r, 1.0f), r, r);
    const float q = T * r;
    Tu[3] = __builtin_fmaf(__builtin_fmaf(-q, P, T), r, q);
    Tu[2] = Tu[3] * om[3];
    Tu[1] = Tu[2] * om[2];
    Tu[0] = Tu[1] * om[1];
    T = Tu[3];
}

// 16-byte store to a dword-aligned address (the 36-byte gradient records): global memory on gfx950 only needs dword
// alignment for multi-dword accesses, but the compiler splits a packed 16-byte store into four instructions
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_dword_aligned(void* p, float4 v) {
    const f4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(q) : "memory");
}

// ------------------------------------------------------------------ backward
constexpr int BCH = 64;  // list entries per round: one per lane of each wave
constexpr int BWD_MIN_WAVES = 4;   // 128 VGPRs, 39.3 KB of LDS: four workgroups per CU; no spills
constexpr int ACC_BUFS = 2;        // per-round sums double-buffered by round parity (one barrier per round)

template <bool SAFE>       // see blend_forward_kernel / splat_pixel_grad
__global__ void __launch_bounds__(256, BWD_MIN_WAVES)
blend_backward_kernel(int W, int H, int gx, int tiles, const uint32_t* __restrict__ order, const unsigned long long* __restrict__ total,
                      const uint32_t* __restrict__ ranges,
                      const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ gm_index,
                      const uint2* __restrict__ gm_base,
                      const uint8_t* __restrict__ qmask, const float4* __restrict__ rec,
                      const float* __restrict__ bg, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                      GradRec* __restrict__ grad_rec, unsigned long long* __restrict__ cut_key, unsigned long long stamp,
                      uint8_t* __restrict__ has_rec) {
    // wave-private compacted records of the round: [wave][field group][3 pad + position]; group 0/1 =
    // the first 32 bytes of the splat record, group 2 = (blue, position in round, -, -).  A group of four
    // reads slots k .. k+3 of each field group: one address register and immediate offsets.  The three
    // pad slots in front stay zero (a partial last group blends them with alpha 0).
    __shared__ float4 st[4][3][BCH + 4];
    // [round parity][wave]: the nine sums per position, as 16 + 16 + 4 bytes: (M0, Mx, My, Mxx | Mxy, Myy, c0, c1 | c2)
    // until the wave has shifted them to the splat's centre, the record layout afterwards
    __shared__ float4 accA[ACC_BUFS][4][BCH], accB[ACC_BUFS][4][BCH];
    __shared__ float accC[ACC_BUFS][4][BCH];
    __shared__ __attribute__((aligned(16))) float tr[4][8 * TCS];    // the waves' transposition blocks (column_fold)
    __shared__ uint32_t wave_max[4];
    const int t = blend_tile((int)blockIdx.x, tiles, order, total);
    if (t < 0) return;
    const uint32_t lo = ranges[2 * t], n = ranges[2 * t + 1] - lo;
    if (n == 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;  // wave == quadrant
    // x in the HIGH lane bits, y in the low ones: lane p stores pixel p = 8 x + y of its columns, lane (c, x) reads (x, 0..7)
    const int qx0 = (t % gx) * TILE + (wave & 1) * 8, qy0 = (t / gx) * TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane >> 3), py = qy0 + (lane & 7);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float qx0f = (float)qx0, qy0f = (float)qy0;
    const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    PixState ps;
    ps.T = inside ? final_T[pix] : 0.0f;
    ps.dLp0 = ps.dLp1 = ps.dLp2 = 0.0f;
    if (inside) {
        ps.dLp0 = dL_dpix[pix];
        ps.dLp1 = dL_dpix[hw + pix];
        ps.dLp2 = dL_dpix[2 * hw + pix];
    }
    ps.behind = (bg[0] * ps.dLp0 + bg[1] * ps.dLp1) + bg[2] * ps.dLp2;
    ps.last_alpha = ps.d_last = 0.0f;
    float c099 = 0.99f;
    asm volatile("" : "+v"(c099));  // keep the clamp in a VGPR: VOP2 with a literal issues slower
    SCR_COUNT_DECL;
    // the reducing role of this lane: column rcol of the transposition block (0-3: Y of the group's splats, 4-7: alpha T),
    // pixel column rx of the quadrant; its weights for the eight pixels (rx, y)
    const int rcol = lane >> 3, rx = lane & 7;
    float wgt[3][8];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        wgt[0][y] = 1.0f; wgt[1][y] = (float)y; wgt[2][y] = (float)(y * y);
        if (rcol >= 4) {
            const bool in = qx0 + rx < W && qy0 + y < H;
            const size_t q = (size_t)(qy0 + y) * W + (size_t)(qx0 + rx);
#pragma unroll
            for (int c = 0; c < 3; ++c) wgt[c][y] = in ? dL_dpix[c * hw + q] : 0.0f;
        }
    }
    const float xw = rcol < 4 ? (float)rx : 0.0f;
    float* const trw = &tr[wave][0];
    const float* const trr = trw + rcol * TCS + 8 * rx;
    // where the lane's three sums go (lanes 0 and 4 of every column store): per sum the address of entry 0 in parity 0 and
    // the entry stride, in bytes -- (M0, My, Myy) from a Y column's even bank, (Mx, Mxx, Mxy) from its odd one, (c0, c1, c2)
    // from a colour column's even bank
    const int rkind = rcol >= 4 ? 2 : (lane >> 2) & 1;
    const bool rstore = (lane & 3) == 0 && (rcol < 4 || ((lane >> 2) & 1) == 0);
    const int rcu = rcol & 3;                                    // the column's splat of the group
    // (lane masks in SGPR pairs: the list position of the column's splat is selected from the four the group's record reads
    // broadcast anyway -- an LDS read of its own would be a second exposed round trip per group)
    const unsigned long long rcu1 = lanes(rcu == 1), rcu2 = lanes(rcu == 2), rcu3 = lanes(rcu == 3);
    char* const accA0 = (char*)&accA[0][wave][0];
    char* const accB0 = (char*)&accB[0][wave][0];
    char* const rdst0 = rkind == 0 ? accA0 : rkind == 1 ? accA0 + 4 : accB0 + 8;
    char* const rdst1 = rkind == 0 ? accA0 + 8 : rkind == 1 ? accA0 + 12 : accB0 + 12;
    char* const rdst2 = rkind == 0 ? accB0 + 4 : rkind == 1 ? accB0 : (char*)&accC[0][wave][0];
    const uint32_t rstride2 = rkind == 2 ? 4u : 16u;
    constexpr uint32_t PAR_A = sizeof(float4) * 4 * BCH, PAR_C = sizeof(float) * 4 * BCH;     // bytes from parity 0 to parity 1
    const uint32_t rpar2 = rkind == 2 ? PAR_C : PAR_A;
    // per-wave largest contributor count: list positions >= it cannot matter to the wave
    uint32_t wm = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, d, WAVE));
    if (lane == 0) wave_max[wave] = wm;
    if (lane < 9) st[wave][lane / 3][lane % 3] = make_float4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t wmax0 = wave_max[0], wmax1 = wave_max[1], wmax2 = wave_max[2], wmax3 = wave_max[3];
    const uint32_t max_last = max(max(wmax0, wmax1), max(wmax2, wmax3));
    const uint32_t wave_last = wave == 0 ? wmax0 : wave == 1 ? wmax1 : wave == 2 ? wmax2 : wmax3;
    const int nround = (int)((n + BCH - 1) / BCH);
    const int live_top = max_last ? (int)((max_last - 1) / BCH) : -1;  // last round with any work

    // The rounds behind every pixel's last contributor get NO records (at 20 M anchors a tile's list holds 24 k entries
    // and its pixels are opaque after a few hundred: they used to be written as zeros, 36 scattered bytes per instance,
    // and read back).  The tile publishes the sort key (depth bits, id) of its first entry WITHOUT a record instead:
    // the list is sorted by that key, so preprocess_backward_kernel knows from a Gaussian's own depth and index whether
    // its instance in this tile has a record -- one 8-byte store per tile, no per-instance flags.
    // A tile with only a round or two to spare (the benchmark density) fills them with zero records as before and
    // publishes "no cut"; when no tile of a call cuts, the reader skips the look-ups altogether: the word behind the table
    // takes this call's stamp as soon as one tile does.  (A stale or uninitialised word that happened to equal the stamp
    // would cost the reader its look-ups, never the result.)
    constexpr int CUT_MIN_ROUNDS = 3;
    const bool cut = nround - 1 - live_top >= CUT_MIN_ROUNDS;    // workgroup-uniform
    if (threadIdx.x == 0) {
        unsigned long long ck = ~0ull;      // every entry has a record
        if (cut) {
            const uint32_t idc = point_list[lo + (uint32_t)(live_top + 1) * BCH];
            ck = ((unsigned long long)__float_as_uint(rec[3 * (size_t)idc + 2].y) << 32) | idc;
            cut_key[tiles] = stamp;
        }
        cut_key[t] = ck;
    }
    // Gaussian-major index of a list entry = where its gradient record goes
    const uint32_t tile_x = (uint32_t)(t % gx), tile_y = (uint32_t)(t / gx);
    auto gm_of = [&](uint32_t pos, uint32_t id) -> uint32_t {
        if (gm_index) return gm_index[pos];           // kernel-uniform: the tile sort materialised it
        const uint2 b = gm_base[id];      // deep lists: 8 B per Gaussian (scatter_kernel), index of its rect walk's origin + rect width
        return b.x + tile_y * b.y + tile_x;
    };
    if (!cut && wave < 3)
        for (int ci = nround - 1; ci > live_top; --ci) {     // all-zero records (waves 0..2 write one part each)
            const uint32_t i = (uint32_t)ci * BCH + lane;
            if (i < n && qmask[lo + i] != 0) {     // mask 0: no record (see the combine below)
                GradRec& gr = grad_rec[gm_of(lo + i, point_list[lo + i])];
                if (wave == 0) store16_dword_aligned(&gr.a, make_float4(0, 0, 0, 0));
                else if (wave == 1) store16_dword_aligned(&gr.b, make_float4(0, 0, 0, 0));
                else gr.c = 0.0f;
            }
        }

    // ---- software pipeline over the live rounds, back to front:
    // (mask, id, slot) two rounds ahead, gathered records one round ahead
    uint32_t m_next = 0, id_next = 0, slot_next = 0;  // round ci-1
    uint32_t m_cur = 0, slot_cur = 0, id_cur = 0;     // round ci (records in r0/r1/r2x)
    bool sel_cur = false;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
    float r2x = 0.0f;
    auto load_meta = [&](int ci, uint32_t& m, uint32_t& id, uint32_t& slot) {
        const uint32_t i = (uint32_t)ci * BCH + lane;
        const bool have = ci >= 0 && i < n;
        m = have ? qmask[lo + i] : 0u;
        id = have ? point_list[lo + i] : 0u;
        slot = have ? gm_of(lo + i, id) : 0u;
    };
    auto gather = [&](int ci, uint32_t m, uint32_t id, uint32_t slot) {
        const uint32_t i = (uint32_t)ci * BCH + lane;
        sel_cur = ((m >> wave) & 1u) && i < wave_last;
        m_cur = m;
        slot_cur = slot;
        id_cur = id;
        if (sel_cur) {
            r0 = rec[3 * (size_t)id];
            r1 = rec[3 * (size_t)id + 1];
            r2x = rec[3 * (size_t)id + 2].x;
        }
    };
    if (live_top >= 0) {
        uint32_t m0, id0, sl0;
        load_meta(live_top, m0, id0, sl0);
        gather(live_top, m0, id0, sl0);
        load_meta(live_top - 1, m_next, id_next, slot_next);
    }
    for (int ci = live_top; ci >= 0; --ci) {
        const uint32_t base = (uint32_t)ci * BCH;
        const int last_rel = (int)last - (int)base;  // entry j of the round is contributor base + j + 1
        // ---- stage this round's surviving records (wave-private, list order)
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(sel_cur);
        const int cnt = __builtin_popcountll(bal);
        if (sel_cur) {
            const uint32_t pos = lanes_below(bal);
            st[wave][0][pos + 3] = r0;
            st[wave][1][pos + 3] = r1;
            st[wave][2][pos + 3] = make_float4(r2x, __uint_as_float((uint32_t)lane), 0.0f, 0.0f);
        }
        const uint32_t m_this = m_cur, slot_this = slot_cur, id_this = id_cur;
        gather(ci - 1, m_next, id_next, slot_next);
        load_meta(ci - 2, m_next, id_next, slot_next);
        // acc is double-buffered by round parity: this round's writes cannot collide with the
        // previous round's combine, so ONE barrier per round suffices (a wave reaches the writes of
        // round r+2 only after barrier B of round r+1, which every wave passes after its combine of r)
        const int par = ci & 1;
        // back to front, four splats per reduction.  A round's first group (the last one walked) is partial in three
        // rounds of four: its missing splats are skipped with wave-uniform branches (TAIL) instead of being blended
        // with alpha 0 -- about 1.5 of the ~22 entries a wave holds per round; the full groups stay branch-free.
        auto group = [&](const int k, auto tail) {
            constexpr bool TAIL = decltype(tail)::value;
            float Y[4], Wt[4];  // Y and alpha T of each splat
            uint32_t jj[4] = {0u, 0u, 0u, 0u};
            unsigned long long hits[4] = {0ull, 0ull, 0ull, 0ull};
            // all four records first (one LDS round trip per group instead of four)
            float4 ra[4], rb[4];
            float2 rc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // entry k - u lives in slot k - u + 3
                ra[u] = st[wave][0][k + 3 - u];
                rb[u] = st[wave][1][k + 3 - u];
                rc[u] = *(const float2*)&st[wave][2][k + 3 - u];
            }
            float Gs[4], al[4], om[4], Tu[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (TAIL && k - u < 0) {  // wave-uniform
                    Gs[u] = al[u] = 0.0f;
                    om[u] = 1.0f;
                    SCR_COUNT(13, 1);
                    continue;
                }
                const float4 a = ra[u], b = rb[u];
                const uint32_t j = __float_as_uint(rc[u].y);
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = __builtin_fmaf(dx, __builtin_fmaf(a.z, dx, a.w * dy), (b.x * dy) * dy);
                const float ar = b.y * fast_exp(power);   // alpha before the clamp: min(0.99, ar) < 1/255 <=> ar < 1/255
                const unsigned long long hit = lanes((int)j < last_rel) & lanes(!(power > 0.0f)) & lanes(!(ar < 1.0f / 255.0f));
                // Branch-free: a splat that does not contribute to this pixel is carried through the
                // back-to-front recurrences with alpha = 0, which leaves T and the colour-behind
                // accumulator exactly as skipping it would (T / (1 - 0) = T; the accumulator folds
                // 0 * d).  ONE select zeroes both the alpha and the weight of the moments: the moments are taken
                // of Y = opacity G dL/dalpha (the unclamped alpha times dL/dalpha, straight through the clamp), which
                // is what every consumer but dL/dopacity wants anyway (preprocess_backward_kernel divides that one).
                SCR_COUNT(8, 1);
                SCR_COUNT(9, __builtin_popcountll(hit));
                SCR_COUNT_SUB(hit);
                SCR_COUNT(10, __builtin_popcountll(lanes(!((int)j < last_rel)) & lanes(inside)));
                SCR_COUNT(11, __builtin_popcountll(lanes(!(power > 0.0f)) & lanes(!(ar < 1.0f / 255.0f)) & lanes(inside)));
                Gs[u] = sel(hit, ar, 0.0f);
                al[u] = vmin(c099, Gs[u]);
                om[u] = 1.0f - al[u];
                if (SAFE) hits[u] = hit;
                jj[u] = j;
            }
            group_transmittance(ps.T, om, Tu);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (TAIL && k - u < 0) {
                    Y[u] = Wt[u] = 0.0f;
                    continue;
                }
                splat_pixel_grad<SAFE>(ps, Tu[u], rb[u], rc[u].x, Gs[u], al[u], hits[u], Y[u], Wt[u]);
            }
            // (a group none of whose splats reaches any pixel of the quadrant -- 0.1 % of them at the benchmark density -- goes
            // through the same path with zeros: a wave-uniform skip cost every group five scalar instructions)
            float s0, s1, s2;
            {
                // the block's stores and loads are LDS operations of ONE wave: they execute in issue order; the compiler
                // must not move them across each other (it sees per-thread addresses only)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    trw[u * TCS + lane] = Y[u];
                    trw[(4 + u) * TCS + lane] = Wt[u];
                }
                asm volatile("" ::: "memory");
                const float4 va = *(const float4*)trr, vb = *(const float4*)(trr + 4);
                asm volatile("" ::: "memory");
                const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
                // two chains of four per sum (even / odd pixel rows), added at the end: half the rounding depth of one chain
                // of eight -- the conic gradients of needle-like Gaussians amplify the moments' last bits -- and six
                // independent fma chains instead of three
                float e0 = wgt[0][0] * v[0], e1 = wgt[1][0] * v[0], e2 = wgt[2][0] * v[0];
                float o0 = wgt[0][1] * v[1], o1 = wgt[1][1] * v[1], o2 = wgt[2][1] * v[1];
#pragma unroll
                for (int y = 2; y < 8; y += 2) {
                    e0 = __builtin_fmaf(wgt[0][y], v[y], e0);
                    e1 = __builtin_fmaf(wgt[1][y], v[y], e1);
                    e2 = __builtin_fmaf(wgt[2][y], v[y], e2);
                    o0 = __builtin_fmaf(wgt[0][y + 1], v[y + 1], o0);
                    o1 = __builtin_fmaf(wgt[1][y + 1], v[y + 1], o1);
                    o2 = __builtin_fmaf(wgt[2][y + 1], v[y + 1], o2);
                }
                s0 = e0 + o0; s1 = e1 + o1; s2 = e2 + o2;
                const float t0 = xw * s0, t1 = xw * t0, t2 = xw * s1;      // x S0, x^2 S0, x S1 (colour lanes: 0)
                column_fold(s0, s1, s2, t0, t1, t2);
            }
            // column rcol & 3 is splat u of the group, entry k - u of the wave's list; every listed position is written
            if (rstore && (!TAIL || k - rcu >= 0)) {
                const uint32_t jw = sel(rcu3, jj[3], sel(rcu2, jj[2], sel(rcu1, jj[1], jj[0])));
                *(float*)(rdst0 + par * PAR_A + jw * 16u) = s0;
                *(float*)(rdst1 + par * PAR_A + jw * 16u) = s1;
                *(float*)(rdst2 + par * rpar2 + jw * rstride2) = s2;
            }
        };
        int k = cnt - 1;
        for (; k >= 3; k -= 4) group(k, std::false_type{});
        if (k >= 0) group(k, std::true_type{});
        SCR_COUNT_CHUNK_END(7);
        // ---- the wave's entries: moments about the quadrant's origin -> about the splat's centre, in the record layout
        // the combine below and preprocess_backward_kernel read: (sum Y dx, sum Y dy, sum Y dx^2, sum Y dx dy |
        // sum Y dy^2, sum Y, c0, c1 | c2) with d = mean - pixel = (mean - origin) - (x, y).  One lane per entry.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's result stores have landed (same wave: in order)
        if (lane < cnt) {
            const float4 ra = st[wave][0][lane + 3];
            const uint32_t j = __float_as_uint(st[wave][2][lane + 3].y);
            const float a = ra.x - qx0f, b = ra.y - qy0f;
            const float4 A = accA[par][wave][j], B = accB[par][wave][j];
            const float M0 = A.x, Mx = A.y, My = A.z, Mxx = A.w, Mxy = B.x, Myy = B.y;
            const float t1 = a * M0 - Mx, t2 = b * M0 - My;      // sum Y dx, sum Y dy
            accA[par][wave][j] = make_float4(t1, t2, a * (t1 - Mx) + Mxx, (a * t2 - b * Mx) + Mxy);
            accB[par][wave][j] = make_float4(b * (t2 - My) + Myy, M0, B.z, B.w);
        }
        __syncthreads();  // B: every wave's sums for this round are in acc
        // ---- combine: wave p (< 3) writes part p of the 36-byte gradient record of position `lane`,
        // adding the waves that took part in a fixed order
        // Every entry of a live round that SOME quadrant can reach gets its record (zeros if no wave took part).  An entry
        // with quadrant mask 0 -- the splat's rect covers the tile, its alpha >= 1/255 ellipse does not: a third of the
        // instances at the benchmark density -- gets none: the scatter kernel left that verdict per Gaussian in live_bits
        // and preprocess_backward_kernel does not read what was not written
        if (wave < 3 && base + lane < n && m_this != 0) {
            const uint32_t i = base + lane;
            float4 r = make_float4(0, 0, 0, 0);
            auto part = [&](int w) {
                return wave == 0 ? accA[par][w][lane] : wave == 1 ? accB[par][w][lane]
                                                                  : make_float4(accC[par][w][lane], 0.0f, 0.0f, 0.0f);
            };
            if (((m_this >> 0) & 1u) && i < wmax0) { const float4 x = part(0); r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w; }
            if (((m_this >> 1) & 1u) && i < wmax1) { const float4 x = part(1); r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w; }
            if (((m_this >> 2) & 1u) && i < wmax2) { const float4 x = part(2); r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w; }
            if (((m_this >> 3) & 1u) && i < wmax3) { const float4 x = part(3); r.x += x.x; r.y += x.y; r.z += x.z; r.w += x.w; }
            GradRec& gr = grad_rec[slot_this];
            if (wave == 0) store16_dword_aligned(&gr.a, r);
            else if (wave == 1) store16_dword_aligned(&gr.b, r);
            else {
                gr.c = r.x;
                if (has_rec) has_rec[id_this] = 1;      // deep lists (kernel-uniform): this Gaussian has something to sum
            }
        }
    }
    SCR_COUNT_SUB_FLUSH(14, 15);
    SCR_COUNT_FLUSH(lane);
}

// ------------------------------------------------------------------ launchers
// blocks per XCD that cover either mapping: the contiguous bands or the round-robin 4x4-tile blocks
static int blend_slots_per_xcd(const Grid& g) {
    int cnt[NUM_XCD] = {};
    for (int t = 0; t < g.tiles; ++t) ++cnt[xcd_of_tile(t, g.gx)];
    int m = (g.tiles + NUM_XCD - 1) / NUM_XCD;
    for (int q = 0; q < NUM_XCD; ++q) m = cnt[q] > m ? cnt[q] : m;
    return m;
}

void launch_blend_forward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                          float* out_color, bool longest_first, bool safe, hipStream_t st) {
    Grid g(ks.H, ks.W);
    // gv.tile_count has been consumed by the plan scan: with longest_first it holds from here on every XCD's tiles,
    // longest list first, and total[2] says so to this launch and to the backward one
    if (longest_first)
        tile_order_kernel<<<NUM_XCD, 1024, 0, st>>>(g.tiles, g.gx, gv.ranges, gv.tile_count, gv.total + 2, gv.total + 4);
    auto kernel = safe ? blend_forward_kernel<true> : blend_forward_kernel<false>;
    kernel<<<(unsigned)blend_slots_per_xcd(g) * NUM_XCD * 4, 64, 0, st>>>(
        ks.W, ks.H, g.gx, g.tiles, gv.tile_count, gv.total, gv.ranges, bv.point_list, bv.qmask, gv.rec, ks.bg, out_color, iv.final_T,
        iv.n_contrib);
}

void launch_blend_backward(const KSettings& ks, const GeomView& gv, const BinView& bv, const ImgView& iv,
                           const float* dL_dcolor, GradRec* grad_rec, unsigned long long stamp, bool deep, bool flags,
                           bool safe, hipStream_t st) {
    Grid g(ks.H, ks.W);
    const bool gm_from_base = deep;
    if (flags) { ZeroList z; z.add(gv.has_rec, ((size_t)gv.P + 3) / 4 * 4, st); launch_zero(z, st); }   // (the array is padded to 256 bytes)
    auto kernel = safe ? blend_backward_kernel<true> : blend_backward_kernel<false>;
    kernel<<<(unsigned)blend_slots_per_xcd(g) * NUM_XCD, 256, 0, st>>>(
        ks.W, ks.H, g.gx, g.tiles, gv.tile_count, gv.total, gv.ranges, bv.point_list, gm_from_base ? nullptr : bv.gm_index, gv.gm_base,
        bv.qmask, gv.rec, ks.bg,
        iv.final_T, iv.n_contrib, dL_dcolor, grad_rec, iv.cut_key, stamp, flags ? gv.has_rec : nullptr);
}

}  // namespace scr

#ifdef SCR_BLEND_COUNT
extern "C" int scr_tool_blend_counters(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(scr::g_blend_counters), 16 * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(scr::g_blend_counters), z, 16 * 8) != hipSuccess) return 1;
    }
    return 0;
}
#endif
