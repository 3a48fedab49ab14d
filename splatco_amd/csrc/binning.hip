// splatco_amd/csrc/binning.hip -- per-tile bucketing and depth sort (gfx950).
//
// The operator family sorts all (tile | depth) 64-bit keys with a global multi-pass radix sort.
// Here the same ordered lists are produced MI355X-style in two HBM passes:
//   1. bucket:  per-tile instance counts (integer atomics fused into the preprocess kernel) ->
//               exclusive scan -> every instance is dropped into its tile's segment as one 64-bit
//               key (depth_bits << 32 | gaussian id << 4 | quadrant mask).  Slot order inside a
//               segment is arbitrary.
//   2. sort:    one wave per tile sorts up to 1024 keys in registers (bitonic network), a four-wave
//               workgroup up to 8192 through LDS; bigger tiles add merge-path passes.  Sorting by (depth, id)
//               reproduces the stable (tile, depth) order of the reference semantics, so
//               point_list / ranges are bit-identical to the oracle's.  The final write also derives
//               each instance's Gaussian-major index (where the backward pass puts its gradient
//               record) from the Gaussian's tile rect, so no payload travels through the sort.
// Both passes move 8-12 B per instance once instead of 6+ radix passes over 12 B.
#include "common.h"

namespace scr {

// ------------------------------------------------------------------ block-wide exclusive scan
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_waves /*[THREADS/64+1]*/,
                                                         uint32_t& block_total) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t n = __shfl_up(inc, d, WAVE);
        if (lane >= d) inc += n;
    }
    if (lane == WAVE - 1) lds_waves[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < THREADS / WAVE; ++k) {
        uint32_t s = lds_waves[k];
        if (k < w) base += s;
        tot += s;
    }
    block_total = tot;
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------ plan scans (2 workgroups)
// block 0: block_sums[nblk] -> exclusive prefix in place, total -> *total
// block 1: tile_count[tiles] -> ranges[tiles] = (start, end), cursor[tiles] = 0
__global__ void __launch_bounds__(1024) plan_scan_kernel(uint32_t nblk, uint32_t* __restrict__ block_sums,
                                                         unsigned long long* __restrict__ total,
                                                         uint32_t tiles, const uint32_t* __restrict__ tile_count,
                                                         uint32_t* __restrict__ ranges,
                                                         uint32_t* __restrict__ cursor,
                                                         volatile unsigned long long* mailbox,
                                                         unsigned long long seq) {
    // mailbox (optional): four words of pinned host memory the GPU can write -- (I, seq, max tile count | plan flags << 32, seq).
    // The host spins on the two stamps instead of sleeping in a stream synchronisation.
    __shared__ uint32_t lds[1024 / WAVE + 1];
    __shared__ unsigned long long wide[1024 / WAVE];
    if (blockIdx.x == 0) {
        // the 32-bit prefixes are what the scatter kernel uses; the TOTAL is formed in 64 bits from the (saturated)
        // workgroup sums, so that num_rendered >= 2^32 shows up on the host instead of wrapping
        unsigned long long carry = 0, mine = 0;
        for (uint32_t base = 0; base < nblk; base += 1024) {
            uint32_t i = base + threadIdx.x;
            uint32_t v = i < nblk ? block_sums[i] : 0u, tot;
            uint32_t ex = block_exclusive_scan<1024>(v, lds, tot);
            if (i < nblk) block_sums[i] = (uint32_t)carry + ex;
            carry += tot;
            mine += v;
        }
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) mine += (unsigned long long)__shfl_down((long long)mine, d, WAVE);
        __syncthreads();
        if ((threadIdx.x & (WAVE - 1)) == 0) wide[threadIdx.x / WAVE] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            carry = 0;
            for (int w = 0; w < 1024 / WAVE; ++w) carry += wide[w];
            total[0] = carry;
            if (mailbox) {
                mailbox[0] = carry;
                __threadfence_system();
                mailbox[1] = seq;
            }
        }
    } else {
        uint32_t carry = 0, mx = 0;
        for (uint32_t base = 0; base < tiles; base += 1024) {
            uint32_t i = base + threadIdx.x;
            uint32_t v = i < tiles ? tile_count[i] : 0u, tot;
            uint32_t ex = block_exclusive_scan<1024>(v, lds, tot);
            if (i < tiles) {
                ranges[2 * i] = carry + ex;
                ranges[2 * i + 1] = carry + ex + v;
                cursor[i] = 0;
            }
            carry += tot;
            mx = max(mx, v);
        }
        // largest per-tile instance count (sizes the sort's chunk grid / merge passes on the host)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, WAVE));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = 0;
            for (int w = 0; w < 1024 / WAVE; ++w) m = max(m, lds[w]);
            total[1] = m;
            total[2] = 0;      // no tile order yet (tile_order_kernel sets it)
            if (mailbox) {
                mailbox[2] = (unsigned long long)m | (total[3] << 32);      // total[3]: the plan flags preprocess_kernel raised
                __threadfence_system();
                mailbox[3] = seq;
            }
        }
    }
}

// ------------------------------------------------------------------ scatter into tile segments
// Same workgroup shape as the preprocess kernel (1024 threads x 4 Gaussians, in index order).
// LDS_HIST: the workgroup counts its instances per tile in LDS, reserves a contiguous run of
// each tile's segment with ONE returning global atomic per touched tile, then hands out the
// slots of the run with returning LDS atomics.  Slot order inside a segment is arbitrary either
// way; the tile sort restores the canonical (depth, id) order.
template <bool LDS_HIST>
__global__ void __launch_bounds__(BIN_THREADS)
scatter_kernel(int64_t P, int gx, int tiles, const float4* __restrict__ rec, uint2* __restrict__ gm_base,
               uint32_t* __restrict__ live_bits, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_prefix,
               uint32_t* __restrict__ point_offsets, const uint32_t* __restrict__ ranges,
               uint32_t* __restrict__ cursor, unsigned long long* __restrict__ keys,
               const unsigned long long* __restrict__ total, unsigned long long cap_instances) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];  // [tiles] when LDS_HIST
    __shared__ uint32_t lds[BIN_THREADS / WAVE + 1];
    // Launched ahead of the host's look at the instance count (scr_forward_plan_run: the count is still on its way
    // through the mailbox) into a buffer sized from the previous forward: if this call's count does not fit, do nothing
    // at all -- the host sees the same number, re-zeroes nothing (no cursor was touched) and takes the two-call path.
    if (total[0] > cap_instances) return;
    if (LDS_HIST) {
        for (int t = threadIdx.x; t < tiles; t += BIN_THREADS) hist[t] = 0;
        __syncthreads();
    }
    uint32_t off[BIN_ROUNDS], tts[BIN_ROUNDS], rlo[BIN_ROUNDS], rhi[BIN_ROUNDS], dbits[BIN_ROUNDS];
    uint32_t carry = block_prefix[blockIdx.x];
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; ++r) {
        const int64_t i = (int64_t)blockIdx.x * BIN_GPW + r * BIN_THREADS + threadIdx.x;
        const uint32_t tt = i < P ? tiles_touched[i] : 0u;
        uint32_t tot;
        off[r] = carry + block_exclusive_scan<BIN_THREADS>(tt, lds, tot);
        carry += tot;
        tts[r] = tt;
        rlo[r] = rhi[r] = dbits[r] = 0;
        if (i < P) point_offsets[i] = off[r] + tt;  // inclusive, as in the reference semantics
        if (tt) {
            const float4 r2 = rec[3 * i + 2];
            dbits[r] = __float_as_uint(r2.y);
            rlo[r] = __float_as_uint(r2.z);
            rhi[r] = __float_as_uint(r2.w);
            // what the sort's final write needs to place an instance in Gaussian-major order: the
            // index of the instance in tile (tx, ty) is off + (ty - y0) * rw + (tx - x0)
            const uint32_t x0 = rlo[r] & 0xffff, y0 = rlo[r] >> 16, rw = (rhi[r] & 0xffff) - x0;
            gm_base[i] = make_uint2(off[r] - y0 * rw - x0, rw);
            if (LDS_HIST)
                for (int ty = rlo[r] >> 16; ty < (int)(rhi[r] >> 16); ++ty)
                    for (int tx = rlo[r] & 0xffff; tx < (int)(rhi[r] & 0xffff); ++tx) atomicAdd(&hist[ty * gx + tx], 1u);
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int t = threadIdx.x; t < tiles; t += BIN_THREADS) {
            const uint32_t c = hist[t];
            if (c) hist[t] = ranges[2 * t] + atomicAdd(&cursor[t], c);  // first slot of this workgroup's run
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < BIN_ROUNDS; ++r) {
        if (!tts[r]) continue;
        const int64_t i = (int64_t)blockIdx.x * BIN_GPW + r * BIN_THREADS + threadIdx.x;
        const uint32_t klo = (uint32_t)i << 4;
        const float4 r0 = rec[3 * i], r1 = rec[3 * i + 1];
        const unsigned long long khi = (unsigned long long)dbits[r] << 32;
        uint32_t live = 0, kbit = 1;     // bit k: tile k of this walk has a reachable quadrant (k < 32; later tiles: re-tested by the reader)
        for (int ty = rlo[r] >> 16; ty < (int)(rhi[r] >> 16); ++ty)
            for (int tx = rlo[r] & 0xffff; tx < (int)(rhi[r] & 0xffff); ++tx) {
                const uint32_t t = (uint32_t)(ty * gx + tx);
                const uint32_t slot = LDS_HIST ? atomicAdd(&hist[t], 1u) : ranges[2 * t] + atomicAdd(&cursor[t], 1u);
                const uint32_t qm = quadrant_mask(r0, r1, tx * TILE, ty * TILE);
                keys[slot] = khi | klo | qm;
                live |= qm ? kbit : 0u;
                kbit <<= 1;               // 0 from the 33rd tile on
            }
        live_bits[i] = live;
    }
}

// ------------------------------------------------------------------ per-tile sort
// One wave sorts up to 1024 64-bit keys entirely in registers: lane l holds
// E = m/64 elements (index i = l*E + e), m = 64..1024.  The bitonic network's compare-exchanges at
// distance j < E are register-to-register; at distance j >= E the partner sits in lane l ^ (j/E)
// and is fetched with DPP (quad_perm for lane distance 1, 2; bank-masked row_shl/row_shr for 4, 8)
// or v_permlane16_swap / v_permlane32_swap (16, 32).  No LDS, no barriers, no s_waitcnt inside the
// network.  Padding elements are +inf keys and sort to the end.
//   tiles with <= 1024 instances : one wave sorts the tile and writes the final lists;
//   up to 8192                   : an eight-wave workgroup -- each wave sorts 1024 keys this way, the runs are
//                                  merged in LDS, 16 outputs per thread (tile_sort_wg_kernel);
//   larger tiles                 : every 8192-chunk is sorted like that (in place), then log2(chunks)
//                                  merge-path passes double the sorted run length, ping-ponging between
//                                  two buffers; a tile's last pass writes its final lists.
template <int D>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {  // value of lane (l ^ D)
    if constexpr (D == 1) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
    } else if constexpr (D == 2) {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
    } else if constexpr (D == 4) {
        int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xf, 0x5, false);  // row_shl:4 -> banks 0,2
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xA, false);  // row_shr:4 -> banks 1,3
    } else if constexpr (D == 8) {
        int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xf, 0x3, false);  // row_shl:8 -> banks 0,1
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x118, 0xf, 0xC, false);  // row_shr:8 -> banks 2,3
    } else if constexpr (D == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = (x0,x0,x2,x2), r[1] = (x1,x1,x3,x3)
        return sel(0xFFFF0000FFFF0000ull, r[0], r[1]);
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r[0] = (lo,lo), r[1] = (hi,hi)
        return sel(0xFFFFFFFF00000000ull, r[0], r[1]);
    }
}

// lanes l (of 64) with (l & bit) == 0; bit >= 64 -> all lanes
constexpr unsigned long long lanes_with_bit_clear(int bit) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if ((l & bit) == 0) m |= 1ull << l;
    return m;
}

// The direction of every compare-exchange depends only on lane bits, so the "which lanes swap the other
// way" masks are compile-time constants: swap = lanes(a > b) ^ constant.  (Equal keys only occur between
// two padding elements, where either outcome is the same.)
template <int E, int K, int J>
__device__ __forceinline__ void bitonic_step(uint32_t (&klo)[E], uint32_t (&khi)[E], int lane) {
    if constexpr (J < E) {  // partner in the same lane
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if ((e & J) != 0) continue;
            const int f = e | J;
            // ascending where the K bit of the element index is clear (K == 64*E: everywhere)
            const unsigned long long desc = K < E ? ((e & K) == 0 ? 0ull : ~0ull) : ~lanes_with_bit_clear(K / E);
            const unsigned long long a = ((unsigned long long)khi[e] << 32) | klo[e];
            const unsigned long long b = ((unsigned long long)khi[f] << 32) | klo[f];
            const unsigned long long swm = lanes(a > b) ^ desc;
            const uint32_t t0 = klo[e], t1 = khi[e];
            klo[e] = sel(swm, klo[f], t0); khi[e] = sel(swm, khi[f], t1);
            klo[f] = sel(swm, t0, klo[f]); khi[f] = sel(swm, t1, khi[f]);
        }
    } else {  // partner in lane ^ (J / E)
        constexpr int D = J / E;
        // keep the smaller key where (lower lane of the pair) == (ascending block)
        constexpr unsigned long long keep_max = lanes_with_bit_clear(D) ^ lanes_with_bit_clear(K / E);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t olo = lane_xor<D>(klo[e]), ohi = lane_xor<D>(khi[e]);
            const unsigned long long mine = ((unsigned long long)khi[e] << 32) | klo[e];
            const unsigned long long theirs = ((unsigned long long)ohi << 32) | olo;
            const unsigned long long take = lanes(theirs < mine) ^ keep_max;
            klo[e] = sel(take, olo, klo[e]);
            khi[e] = sel(take, ohi, khi[e]);
        }
    }
}

template <int E, int K, int J>
__device__ __forceinline__ void bitonic_stage(uint32_t (&klo)[E], uint32_t (&khi)[E], int lane) {
    bitonic_step<E, K, J>(klo, khi, lane);
    if constexpr (J > 1) bitonic_stage<E, K, J / 2>(klo, khi, lane);
}
template <int E, int K>
__device__ __forceinline__ void bitonic_network(uint32_t (&klo)[E], uint32_t (&khi)[E], int lane) {
    bitonic_stage<E, K, K / 2>(klo, khi, lane);
    if constexpr (K < 64 * E) bitonic_network<E, K * 2>(klo, khi, lane);
}

constexpr int WAVE_SORT_MAX = 1024;

// What the blend kernels read per sorted instance: the Gaussian id, the quadrant mask, and the
// instance's Gaussian-major index = (first index of the Gaussian) + (position of this tile in the
// Gaussian's tile rect, row-major as the scatter kernel walks it), from the 8-byte gm_base entry.
struct FinalLists {
    const uint2* gm_base;
    uint32_t* point_list;
    uint32_t* gm_index;     // nullptr: deep lists (common.h deep_lists) -- the blend backward takes the index from gm_base itself
    uint8_t* qmask;
    int tx, ty;  // this tile
    __device__ __forceinline__ void write(uint32_t i, uint32_t klo) const {
        const uint32_t id = klo >> 4;
        point_list[i] = id;
        qmask[i] = (uint8_t)(klo & 15u);
        if (gm_index) {               // kernel-uniform
            const uint2 b = gm_base[id];  // 8 B per Gaussian: an XCD's band of tiles keeps its share in L2
            gm_index[i] = b.x + (uint32_t)ty * b.y + (uint32_t)tx;
        }
    }
    // U entries per thread -- list positions e0, e0 + stride, ... below cnt, their sorted low words from getk(e) -- with all U
    // gm_base gathers in flight before the first store.  As a loop of write() the stores of one entry and the gather of
    // the next may alias as far as the compiler knows: it kept them in program order, one memory round trip per entry
    // (tools/isa_waits.py: `S S L W0 S` sixteen times at the end of every wave-sorted tile).
    template <int U, typename GetK>
    __device__ __forceinline__ void write_batch(uint32_t lo, uint32_t e0, uint32_t stride, uint32_t cnt, GetK getk) const {
        uint32_t k[U];
        uint2 b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = e0 + (uint32_t)u * stride;
            k[u] = e < cnt ? getk(e, u) : 0u;           // padding reads Gaussian 0's entry: a valid address, never stored
        }
        if (gm_index) {
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = gm_base[k[u] >> 4];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = e0 + (uint32_t)u * stride;
            if (e < cnt) {
                point_list[lo + e] = k[u] >> 4;
                qmask[lo + e] = (uint8_t)(k[u] & 15u);
                if (gm_index) gm_index[lo + e] = b[u].x + (uint32_t)ty * b[u].y + (uint32_t)tx;
            }
        }
    }
};

// Sorts the n (<= 64*E) keys at keys[0..n).  FINAL: write the tile's final lists; otherwise write the
// sorted chunk back in place.
template <int E>
__device__ __forceinline__ void wave_sort_regs(uint32_t n, const unsigned long long* __restrict__ keys,
                                               uint32_t (&klo)[E], uint32_t (&khi)[E]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < E; ++e) {  // coalesced load; the network does not care where an element starts
        const uint32_t i = (uint32_t)e * 64 + lane;
        const unsigned long long v = i < n ? keys[i] : ~0ull;
        klo[e] = (uint32_t)v;
        khi[e] = (uint32_t)(v >> 32);
    }
    bitonic_network<E, 2>(klo, khi, lane);  // sorted position of (lane, e) is lane * E + e
}

template <int E, bool FINAL>
__device__ __forceinline__ void wave_sort(uint32_t n, unsigned long long* __restrict__ keys, uint32_t lo,
                                          const FinalLists& fl, uint32_t* __restrict__ tr) {
    const int lane = threadIdx.x & 63;
    uint32_t klo[E], khi[E];
    wave_sort_regs<E>(n, keys, klo, khi);
    // Sorted position of (lane, e) is lane * E + e: storing from here would put the 64 lanes of one store
    // instruction E words apart (one partial HBM sector each; measured 7.6x write amplification).  The
    // wave transposes through its 4 KB of LDS instead, so that store s covers positions s*64 .. s*64+63.
    // (position p lives at word p + p / E: a lane's E words start E + 1 apart, so neither side conflicts)
    auto transpose = [&](uint32_t (&w)[E]) {
        __syncthreads();  // one-wave workgroup: previous readers are done
#pragma unroll
        for (int e = 0; e < E; ++e) tr[lane * (E + 1) + e] = w[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int p = e * 64 + lane;
            w[e] = tr[p + p / E];
        }
    };
    transpose(klo);
    if (!FINAL) transpose(khi);
    if (FINAL) {
        fl.template write_batch<E>(lo, (uint32_t)lane, 64u, n, [&](uint32_t, int u) { return klo[u]; });   // sorted position u * 64 + lane
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = (uint32_t)e * 64 + lane;  // sorted position
            if (i < n) keys[i] = ((unsigned long long)khi[e] << 32) | klo[e];
        }
    }
}

// A network sorts a power of two: a tile of 538 instances pays for 1024 (880 compare-exchanges per lane at E = 16
// against 360 at E = 8).  Tiles just above a power of two are therefore sorted as TWO runs -- the first 64 EA keys and
// the remaining n - 64 EA <= 64 EB, each by its own network -- and the runs are merged through the wave's LDS: every lane
// forms EA + EB consecutive outputs with one bisection along its diagonal and EA + EB compare-and-advance steps (as the
// workgroup sort's merges do), then the outputs are transposed for coalesced stores like wave_sort's.
// mb: 8-byte slots, one pad slot per 8 keys (a lane's EA keys start EA + 1 slots apart at EA = 8).
__device__ __forceinline__ int split_slot(int p) { return p + (p >> 3); }
constexpr int SPLIT_SLOTS = WAVE_SORT_MAX + WAVE_SORT_MAX / 8;
template <int EA, int EB>
__device__ __forceinline__ void wave_sort_split(uint32_t n, const unsigned long long* __restrict__ keys, uint32_t lo,
                                                const FinalLists& fl, unsigned long long* __restrict__ mb) {
    constexpr int NA = 64 * EA, NB = 64 * EB, PER = EA + EB;
    const int lane = threadIdx.x & 63;
    {
        uint32_t klo[EA], khi[EA];
        wave_sort_regs<EA>(NA, keys, klo, khi);
#pragma unroll
        for (int e = 0; e < EA; ++e) mb[split_slot(lane * EA + e)] = ((unsigned long long)khi[e] << 32) | klo[e];
    }
    {
        uint32_t klo[EB], khi[EB];
        wave_sort_regs<EB>(n - NA, keys + NA, klo, khi);   // missing keys are +inf padding, they stay at the end
#pragma unroll
        for (int e = 0; e < EB; ++e) mb[split_slot(NA + lane * EB + e)] = ((unsigned long long)khi[e] << 32) | klo[e];
    }
    __syncthreads();  // one-wave workgroup
    uint32_t outk[PER];
    {
        auto keyA = [&](uint32_t i) { return mb[split_slot((int)i)]; };
        auto keyB = [&](uint32_t j) { return mb[split_slot(NA + (int)j)]; };
        const uint32_t d0 = (uint32_t)lane * PER;
        uint32_t a = d0 > (uint32_t)NB ? d0 - NB : 0u, b = min(d0, (uint32_t)NA);
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if (keyA(mid) < keyB(d0 - mid - 1)) a = mid + 1;
            else b = mid;
        }
        uint32_t ia = a, ib = d0 - a;
        unsigned long long x = ia < (uint32_t)NA ? keyA(ia) : ~0ull, y = ib < (uint32_t)NB ? keyB(ib) : ~0ull;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const bool takeA = x <= y;                // real keys are unique; +inf only pads run B
            outk[k] = (uint32_t)(takeA ? x : y);      // the final lists need the low word only (id | quadrant mask)
            if (takeA) {
                ++ia;
                x = ia < (uint32_t)NA ? keyA(ia) : ~0ull;
            } else {
                ++ib;
                y = ib < (uint32_t)NB ? keyB(ib) : ~0ull;
            }
        }
    }
    __syncthreads();  // everybody has read the runs
    uint32_t* tr = (uint32_t*)mb;
#pragma unroll
    for (int k = 0; k < PER; ++k) tr[lane * (PER + 1) + k] = outk[k];
    __syncthreads();
    fl.template write_batch<PER>(lo, (uint32_t)lane, 64u, n, [&](uint32_t i, int) { return tr[i + i / PER]; });   // sorted position i
}

template <bool FINAL>
__device__ __forceinline__ void wave_sort_any(uint32_t n, unsigned long long* keys, uint32_t lo, const FinalLists& fl,
                                              unsigned long long* mb) {
    uint32_t* tr = (uint32_t*)mb;
    if (n <= 64) wave_sort<1, FINAL>(n, keys, lo, fl, tr);
    else if (n <= 128) wave_sort<2, FINAL>(n, keys, lo, fl, tr);
    else if (n <= 256) wave_sort<4, FINAL>(n, keys, lo, fl, tr);
    else if (FINAL && n > 256 && n <= 320) wave_sort_split<4, 1>(n, keys, lo, fl, mb);
    else if (FINAL && n > 256 && n <= 384) wave_sort_split<4, 2>(n, keys, lo, fl, mb);
    else if (n <= 512) wave_sort<8, FINAL>(n, keys, lo, fl, tr);
    else if (FINAL && n <= 576) wave_sort_split<8, 1>(n, keys, lo, fl, mb);
    else if (FINAL && n <= 640) wave_sort_split<8, 2>(n, keys, lo, fl, mb);
    else if (FINAL && n <= 768) wave_sort_split<8, 4>(n, keys, lo, fl, mb);
    else if (FINAL) wave_sort_split<8, 8>(n, keys, lo, fl, mb);
    else wave_sort<16, FINAL>(n, keys, lo, fl, tr);
}

// Tiles with <= 1024 instances: one wave sorts the tile and writes its final lists.
__global__ void __launch_bounds__(64, 1)
tile_sort_wave_kernel(int tiles, int gx, const uint32_t* __restrict__ ranges, unsigned long long* __restrict__ keys,
                      const uint2* __restrict__ gm_base, uint32_t* __restrict__ point_list,
                      uint32_t* __restrict__ gm_index, uint8_t* __restrict__ qmask) {
    int t = xcd_tile(blockIdx.x, tiles);
    if (t < 0) return;
    const uint32_t lo = ranges[2 * t], n = ranges[2 * t + 1] - lo;
    if (n == 0 || n > (uint32_t)WAVE_SORT_MAX) return;
    const FinalLists fl{gm_base, point_list, gm_index, qmask, t % gx, t / gx};
    __shared__ unsigned long long mb[SPLIT_SLOTS];  // the wave's merge / transposition buffer (wave_sort, wave_sort_split)
    wave_sort_any<true>(n, keys + lo, lo, fl, mb);
}

// Tiles with more than 1024 instances: an eight-wave workgroup sorts a chunk of up to 8192 keys on chip --
// every wave sorts 1024 keys in registers, the eight sorted runs meet in LDS and three merge passes (a thread forms 16
// consecutive outputs: one bisection along its diagonal, 16 compare-and-advance steps) make one run of them.  A tile
// of up to 8192 instances is finished here (final lists written); a larger tile gets its 8192-chunks
// sorted in place and goes on to the global merge passes.
constexpr int WG_SORT_THREADS = WG_SORT_MAX / 16;   // 16 keys per lane
__device__ __forceinline__ int wg_slot(int p) { return p + (p >> 4); }  // 16 keys of a lane start 17 slots apart
__global__ void __launch_bounds__(WG_SORT_THREADS)
tile_sort_wg_kernel(int tiles, int gx, const uint32_t* __restrict__ ranges, unsigned long long* __restrict__ keys,
                    const uint2* __restrict__ gm_base, uint32_t* __restrict__ point_list,
                    uint32_t* __restrict__ gm_index, uint8_t* __restrict__ qmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long buf[];  // [WG_SORT_MAX * 17 / 16]: 68 KB, two workgroups per CU
    int t = xcd_tile(blockIdx.x, tiles);
    if (t < 0) return;
    const uint32_t lo = ranges[2 * t], n = ranges[2 * t + 1] - lo;
    if (n <= (uint32_t)WAVE_SORT_MAX) return;
    const uint32_t c0 = blockIdx.y * (uint32_t)WG_SORT_MAX;
    if (c0 >= n) return;
    const uint32_t cnt = min((uint32_t)WG_SORT_MAX, n - c0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long* chunk = keys + lo + c0;
    {   // ---- every wave sorts its 1024 keys (missing keys are +inf padding, they stay at the end)
        const uint32_t w0 = (uint32_t)wave * WAVE_SORT_MAX;
        const uint32_t wn = w0 < cnt ? min((uint32_t)WAVE_SORT_MAX, cnt - w0) : 0u;
        uint32_t klo[16], khi[16];
        wave_sort_regs<16>(wn, chunk + w0, klo, khi);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            buf[wg_slot((int)w0 + lane * 16 + e)] = ((unsigned long long)khi[e] << 32) | klo[e];
    }
    __syncthreads();
    // ---- merges in LDS: runs of L -> 2L.  One buffer: every thread forms its 16 outputs in registers while everybody
    // still sees the old runs, and the writes happen after a barrier.
    for (uint32_t L = WAVE_SORT_MAX; L < (uint32_t)WG_SORT_MAX && L < cnt; L <<= 1) {
        // every thread merges ITS 16 consecutive outputs of its run pair: one bisection along its diagonal, then 16
        // compare-and-advance steps (padding keys are +inf and come out last), instead of one bisection per key
        constexpr int PER = WG_SORT_MAX / WG_SORT_THREADS;
        unsigned long long outk[PER];
        const uint32_t o0 = threadIdx.x * PER, pairbase = o0 / (2u * L) * (2u * L), d0 = o0 - pairbase;
        {
            auto keyA = [&](uint32_t i) { return buf[wg_slot((int)(pairbase + i))]; };
            auto keyB = [&](uint32_t j) { return buf[wg_slot((int)(pairbase + L + j))]; };
            uint32_t a = d0 > L ? d0 - L : 0u, b = min(d0, L);
            while (a < b) {
                const uint32_t mid = (a + b) >> 1;
                if (keyA(mid) < keyB(d0 - mid - 1)) a = mid + 1;
                else b = mid;
            }
            uint32_t ia = a, ib = d0 - a;
            unsigned long long x = ia < L ? keyA(ia) : ~0ull, y = ib < L ? keyB(ib) : ~0ull;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const bool takeA = x <= y;                // real keys are unique; ties only between +inf paddings
                outk[k] = takeA ? x : y;
                if (takeA) {
                    ++ia;
                    x = ia < L ? keyA(ia) : ~0ull;
                } else {
                    ++ib;
                    y = ib < L ? keyB(ib) : ~0ull;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) buf[wg_slot((int)(o0 + k))] = outk[k];
        __syncthreads();
    }
    const unsigned long long* src = buf;
    // ---- out: final lists (the whole tile was this chunk) or the sorted chunk back in place
    if (n <= (uint32_t)WG_SORT_MAX) {
        const FinalLists fl{gm_base, point_list, gm_index, qmask, t % gx, t / gx};
        for (uint32_t e = threadIdx.x; e < cnt; e += 4 * WG_SORT_THREADS)
            fl.template write_batch<4>(lo, e, (uint32_t)WG_SORT_THREADS, cnt, [&](uint32_t i, int) { return (uint32_t)src[wg_slot((int)i)]; });
    } else {
        for (uint32_t e = threadIdx.x; e < cnt; e += WG_SORT_THREADS) chunk[e] = src[wg_slot((int)e)];
    }
}

constexpr int MP_SEG = 4096;        // keys of the merged output one merge-path workgroup produces
// Merge pass over sorted runs of length L = WG_SORT_MAX << pass -> runs of length 2L, for tiles above WG_SORT_MAX
// instances (data of pass p lives in buffer (p & 1); tiles that are already fully merged are skipped).
// Merge path: every workgroup produces one 4096-key segment of the merged output.  Two lanes find where
// the segment's first and last diagonal cut the two runs (one binary search each over global memory --
// per workgroup, not per key), the at most 4096 input keys between the cuts are staged in LDS as two
// short runs, merged there (a thread forms 16 consecutive outputs: one bisection, 16 compare-and-advance steps)
// and streamed out coalesced.
__device__ __forceinline__ uint32_t merge_passes_needed(uint32_t n) {  // ceil(log2(ceil(n / WG_SORT_MAX)))
    uint32_t chunks = (n + WG_SORT_MAX - 1) / WG_SORT_MAX, p = 0;
    while ((1u << p) < chunks) ++p;
    return p;
}

// number of A-keys among the first s keys of merge(A[0..la), B[0..lb)): the smallest i in [lo, hi] for which
// A[i] < B[s - i - 1] is false.  A WAVE searches: every step probes 64 evenly spaced positions at once and the ballot of
// the (monotone) predicate narrows the range 65-fold -- three dependent memory round trips for a run pair of 2^17 keys
// where a scalar bisection takes seventeen.
__device__ __forceinline__ uint32_t merge_path_cut(const unsigned long long* __restrict__ A, uint32_t la,
                                                   const unsigned long long* __restrict__ B, uint32_t lb, uint32_t s) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t lo = s > lb ? s - lb : 0u, hi = min(s, la);
    while (lo < hi) {
        const uint32_t span = hi - lo;                                  // candidates lo .. hi-1 for "first false", else hi
        // probe p_l = lo + floor(span * (l + 1) / 65), l = 0..63 (strictly inside [lo, hi) when span >= 65; for small
        // spans the probes are lo + l, clamped)
        const uint32_t p = span >= 65u ? lo + (uint32_t)(((unsigned long long)span * (lane + 1)) / 65u) : min(lo + lane, hi - 1);
        const bool below = A[p] < B[s - p - 1];                         // true: the cut lies right of p
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(below);
        const int c = __builtin_popcountll(bal);                        // monotone: the true lanes are the first c
        // new range: right of the last true probe, up to the first false probe
        const uint32_t plast = __shfl(p, c > 0 ? c - 1 : 0, 64), pfirst = __shfl(p, c < 64 ? c : 63, 64);
        const uint32_t nlo = c > 0 ? plast + 1 : lo, nhi = c < 64 ? pfirst : hi;
        if (span < 65u) {                                               // every candidate was probed
            return c < (int)min(span, 64u) ? lo + (uint32_t)c : hi;
        }
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
tile_merge_path_kernel(int tiles, int gx, uint32_t pass, const uint32_t* __restrict__ ranges,
                       const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                       const uint2* __restrict__ gm_base, uint32_t* __restrict__ point_list,
                       uint32_t* __restrict__ gm_index, uint8_t* __restrict__ qmask) {
    // run A at slots [0, na), run B at [na, na + nb), every index p stored at p + (p >> 4) (17 slots per 16 keys): the
    // threads walk the runs 16 keys apart, which would put them all on the same LDS banks
    __shared__ unsigned long long buf[MP_SEG + MP_SEG / 16];
    __shared__ uint32_t cut[2];
    int t = xcd_tile(blockIdx.x, tiles);
    if (t < 0) return;
    const uint32_t lo = ranges[2 * t], n = ranges[2 * t + 1] - lo;
    if (n <= (uint32_t)WG_SORT_MAX || pass >= merge_passes_needed(n)) return;
    const uint32_t L = (uint32_t)WG_SORT_MAX << pass;
    const uint32_t e0 = blockIdx.y * (uint32_t)MP_SEG;
    if (e0 >= n) return;
    const uint32_t pairbase = e0 / (2u * L) * (2u * L);
    const uint32_t lenA = min(L, n - pairbase);
    const uint32_t lenB = n - pairbase > L ? min(L, n - pairbase - L) : 0u;
    const uint32_t s0 = e0 - pairbase, s1 = min(s0 + (uint32_t)MP_SEG, lenA + lenB);
    const unsigned long long* A = src + lo + pairbase;
    const unsigned long long* B = A + L;
    if (threadIdx.x < 128) {  // waves 0 and 1: the two cuts, concurrently
        const int w = threadIdx.x >> 6;
        const uint32_t cw = merge_path_cut(A, lenA, B, lenB, w ? s1 : s0);
        if ((threadIdx.x & 63) == 0) cut[w] = cw;
    }
    __syncthreads();
    const uint32_t i0 = cut[0], na = cut[1] - i0, j0 = s0 - i0, nb = (s1 - s0) - na;
    const uint32_t tot = na + nb;
    {   // the segment's keys (na from run A, then nb from run B): all 16 loads of a thread in flight before the first LDS
        // write, from clamped addresses and without a branch around them -- as `load -> LDS write` loops every iteration
        // waited for its own load (sixteen memory round trips per workgroup: the pass ran at 28 % of the copy rate)
        unsigned long long v[MP_SEG / 256];
#pragma unroll
        for (int k = 0; k < MP_SEG / 256; ++k) {
            const uint32_t e = min(threadIdx.x + 256u * k, tot - 1);          // tot >= 1: the segment exists
            v[k] = e < na ? A[i0 + e] : B[j0 + (e - na)];
        }
#pragma unroll
        for (int k = 0; k < MP_SEG / 256; ++k) {
            const uint32_t e = threadIdx.x + 256u * k;
            if (e < tot) buf[wg_slot((int)e)] = v[k];
        }
    }
    __syncthreads();
    // ---- every thread merges ITS 16 consecutive outputs: one bisection along its diagonal (how many keys of run A lie
    // among the first 16 t outputs), then 16 compare-and-advance steps -- instead of one bisection per key
    constexpr int PER = MP_SEG / 256;
    unsigned long long outk[PER];
    const uint32_t d0 = min((uint32_t)threadIdx.x * PER, tot);
    {
        auto keyA = [&](uint32_t i) { return buf[wg_slot((int)i)]; };
        auto keyB = [&](uint32_t j) { return buf[wg_slot((int)(na + j))]; };
        uint32_t a = d0 > nb ? d0 - nb : 0u, b = min(d0, na);
        while (a < b) {
            const uint32_t mid = (a + b) >> 1;
            if (keyA(mid) < keyB(d0 - mid - 1)) a = mid + 1;
            else b = mid;
        }
        uint32_t ia = a, ib = d0 - a;
        unsigned long long x = ia < na ? keyA(ia) : ~0ull, y = ib < nb ? keyB(ib) : ~0ull;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const bool takeA = x < y;                     // keys are unique; +inf only when a run is exhausted
            outk[k] = takeA ? x : y;
            if (takeA) {
                ++ia;
                x = ia < na ? keyA(ia) : ~0ull;
            } else {
                ++ib;
                y = ib < nb ? keyB(ib) : ~0ull;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (d0 + k < tot) buf[wg_slot((int)(d0 + k))] = outk[k];
    __syncthreads();
    if (pass + 1 == merge_passes_needed(n)) {  // the tile's last pass writes the final lists directly
        const FinalLists fl{gm_base, point_list, gm_index, qmask, t % gx, t / gx};
        for (uint32_t e = threadIdx.x; e < tot; e += 4 * 256)
            fl.template write_batch<4>(lo + pairbase + s0, e, 256u, tot, [&](uint32_t i, int) { return (uint32_t)buf[wg_slot((int)i)]; });
    } else {
        unsigned long long* out = dst + lo + pairbase + s0;
        for (uint32_t e = threadIdx.x; e < tot; e += 256) out[e] = buf[wg_slot((int)e)];
    }
}

// ------------------------------------------------------------------ launchers
void launch_plan_scans(int64_t P, const KSettings& ks, const GeomView& gv, unsigned long long* mailbox,
                       unsigned long long seq, hipStream_t st) {
    Grid g(ks.H, ks.W);
    uint32_t nb = (uint32_t)((P + BIN_GPW - 1) / BIN_GPW);
    plan_scan_kernel<<<2, 1024, 0, st>>>(nb, gv.block_sums, gv.total, (uint32_t)g.tiles, gv.tile_count,
                                         gv.ranges, gv.cursor, mailbox, seq);
}

void launch_scatter(int64_t P, const KSettings& ks, const GeomView& gv, const BinView& bv, unsigned long long cap_instances,
                    hipStream_t st) {
    if (P <= 0) return;
    Grid g(ks.H, ks.W);
    const unsigned nb = (unsigned)((P + BIN_GPW - 1) / BIN_GPW);
    // histograms beyond the default 64 KB dynamic-LDS limit (gfx950 has 160 KB per CU): the attribute is per device
    static bool big_lds[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !big_lds[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)scatter_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 LDS_HIST_MAX_TILES * 4);
        if (e == hipSuccess && dev >= 0 && dev < 64) big_lds[dev] = true;
        (void)hipGetLastError();      // a refused attribute shows up as a launch error below
    }
    if (g.tiles <= LDS_HIST_MAX_TILES)
        scatter_kernel<true><<<nb, BIN_THREADS, (size_t)g.tiles * 4, st>>>(
            P, g.gx, g.tiles, gv.rec, gv.gm_base, gv.live_bits, gv.tiles_touched, gv.block_sums, gv.point_offsets, gv.ranges,
            gv.cursor, bv.keys, gv.total, cap_instances);
    else
        scatter_kernel<false><<<nb, BIN_THREADS, 0, st>>>(
            P, g.gx, g.tiles, gv.rec, gv.gm_base, gv.live_bits, gv.tiles_touched, gv.block_sums, gv.point_offsets, gv.ranges,
            gv.cursor, bv.keys, gv.total, cap_instances);
}

void launch_tile_sort(const KSettings& ks, const GeomView& gv, const BinView& bv, int64_t max_tile_instances,
                      bool with_gm_index, hipStream_t st) {
    uint32_t* const gm_index = with_gm_index ? bv.gm_index : nullptr;
    Grid g(ks.H, ks.W);
    const unsigned gt = (unsigned)xcd_grid(g.tiles);
    if (max_tile_instances <= 0) return;
    tile_sort_wave_kernel<<<gt, 64, 0, st>>>(g.tiles, g.gx, gv.ranges, bv.keys, gv.gm_base, bv.point_list, gm_index,
                                             bv.qmask);
    if (max_tile_instances <= WAVE_SORT_MAX) return;
    const unsigned chunks = (unsigned)((max_tile_instances + WG_SORT_MAX - 1) / WG_SORT_MAX);
    const size_t lds = (size_t)(WG_SORT_MAX + WG_SORT_MAX / 16) * 8;
    static bool big_lds[64] = {};      // more than the default 64 KB of dynamic LDS: the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !big_lds[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)tile_sort_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && dev >= 0 && dev < 64) big_lds[dev] = true;
        (void)hipGetLastError();
    }
    tile_sort_wg_kernel<<<dim3(gt, chunks), WG_SORT_THREADS, lds, st>>>(g.tiles, g.gx, gv.ranges, bv.keys, gv.gm_base, bv.point_list,
                                                                        gm_index, bv.qmask);
    if (chunks <= 1) return;
    unsigned passes = 0;
    while ((1u << passes) < chunks) ++passes;
    const unsigned segs = (unsigned)((max_tile_instances + MP_SEG - 1) / MP_SEG);
    for (unsigned p = 0; p < passes; ++p) {
        const unsigned long long* src = (p & 1u) ? bv.keys2 : bv.keys;  // data of pass p lives in buffer (p & 1)
        unsigned long long* dst = (p & 1u) ? bv.keys : bv.keys2;
        tile_merge_path_kernel<<<dim3(gt, segs), 256, 0, st>>>(g.tiles, g.gx, p, gv.ranges, src, dst, gv.gm_base,
                                                               bv.point_list, gm_index, bv.qmask);
    }
}

}  // namespace scr
