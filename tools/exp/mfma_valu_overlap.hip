// Developer probe (round 6): does v_mfma_f32_16x16x4_f32 run BESIDE fp32 vector work on one SIMD of gfx950, or does it
// take the vector pipe's slots?  The blend backward's moment reduction was moved to the matrix pipe on the assumption that
// it does; this probe measures it in isolation.
//   mode 0: V plain v_fma_f32 per iteration (eight independent chains)
//   mode 1: M v_mfma_f32_16x16x4_f32 per iteration (two accumulators, alternating)
//   mode 2: both, phase after phase (V fmas, then M MFMAs): overlap can only come from OTHER waves of the SIMD
//   mode 3: both, interleaved inside the wave (V / M fmas behind every MFMA)
//   mode 4: like 2 for even waves of a SIMD, MFMA phase first for odd ones (phases staggered by construction)
// Every wave of a 256-thread workgroup runs the same mode; WGS workgroups per CU put that many waves on each SIMD.
// Output: wall time of the launch (the figure to compare; the per-wave cycle count is block 0's own and depends on how the
// dispatcher filled its CU), so "mode 2 = mode 0 + mode 1" means no overlap and "mode 2 = max" means full overlap.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_valu_overlap.hip -o tools/exp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// KIND 0: v_mfma_f32_16x16x4_f32 (fp32 in); KIND 1: v_mfma_f32_16x16x32_bf16 (bf16 in, eight values per lane and operand)
template <int KIND>
__device__ __forceinline__ f4 mm(float a, float b, bf8 ha, bf8 hb, f4 c) {
    if (KIND == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c, 0, 0, 0);
}

#define FMA8                                                                                                   \
    asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\t"        \
                 "v_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\t"        \
                 "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"                                        \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                 : "v"(a), "v"(b))

template <int MODE, int V8, int M, int KIND>      // V8 = blocks of eight fmas per iteration, M = MFMAs per iteration
__global__ void __launch_bounds__(256) probe(float* out, int iters, unsigned long long* clk) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i);
    const float a = 0.999f, b = 0.001f;
    f4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    float ma = (float)(threadIdx.x & 15) * 1e-3f, mb = (float)(threadIdx.x >> 4) * 1e-3f;
    bf8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(ma + i); hb[i] = (__bf16)(mb - i); }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    const bool odd = ((blockIdx.x / 256) & 1) != 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || (MODE == 4 && !odd)) {
#pragma unroll
            for (int v = 0; v < V8; ++v) FMA8;
        }
        if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int m = 0; m < M; m += 2) {
                d0 = mm<KIND>(ma, mb, ha, hb, d0);
                d1 = mm<KIND>(mb, ma, hb, ha, d1);
            }
        }
        if (MODE == 4 && odd) {
#pragma unroll
            for (int v = 0; v < V8; ++v) FMA8;
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < M; m += 2) {
                d0 = mm<KIND>(ma, mb, ha, hb, d0);
#pragma unroll
                for (int v = 0; v < V8 / (M / 2) / 2; ++v) FMA8;
                d1 = mm<KIND>(mb, ma, hb, ha, d1);
#pragma unroll
                for (int v = 0; v < V8 / (M / 2) - V8 / (M / 2) / 2; ++v) FMA8;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    s += d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int V8, int M, int KIND>
static void run(const char* name, int wgs, float* out, unsigned long long* clk) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, V8, M, KIND><<<256 * wgs, 256>>>(out, 200, clk);
    hipEventRecord(e0);
    probe<MODE, V8, M, KIND><<<256 * wgs, 256>>>(out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);      // s_memrealtime ticks at 100 MHz
    const double cyc_iter_simd = (double)h[0] / iters;            // wave cycles per iteration: all waves of the SIMD run side by side
    printf("%s %-34s waves/SIMD %d  V=%3d fma  M=%2d mfma : %8.1f cycles per iteration of one wave (%6.1f per wave-iteration and SIMD), clock %.2f GHz, %.3f ms\n",
           KIND ? "bf16 16x16x32" : "f32  16x16x4 ", name, wgs, V8 * 8, M, cyc_iter_simd, cyc_iter_simd / wgs, ghz, ms);
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 1024); hipMalloc(&clk, 16);
    for (int wgs : {1, 2, 4}) {
        printf("---- %d waves per SIMD\n", wgs);
#define ALL(V8, M, KIND)                                                     \
        run<0, V8, M, KIND>("0 fma only", wgs, out, clk);                    \
        run<1, V8, M, KIND>("1 mfma only", wgs, out, clk);                   \
        run<2, V8, M, KIND>("2 fma phase, then mfma phase", wgs, out, clk);  \
        run<4, V8, M, KIND>("4 phases staggered between waves", wgs, out, clk); \
        run<3, V8, M, KIND>("3 interleaved inside the wave", wgs, out, clk);
        ALL(32, 16, 0)     // 256 fmas : 16 MFMAs  (the blend backward's ratio per eight splats is about 250 : 16)
        ALL(16, 16, 0)
        ALL(32, 16, 1)
        ALL(16, 16, 1)
    }
    return 0;
}
