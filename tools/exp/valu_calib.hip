// VALU calibration probe for gfx950 (developer tool; bench.py's roofline.valu block is normalised with its output,
// profiles/r03_valu_calibration.*).  For every instruction form the blend kernels lean on, one kernel of iters x 64
// back-to-back wave64 instructions (8 independent chains) is run at a given number of waves per SIMD; it reports
//   * the wall time (HIP events) and the SHADER clock during the run (s_memtime ticks per s_memrealtime tick, 100 MHz),
//   * hence cycles per instruction per SIMD at the measured clock;
// run under `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU ...` the same launches give the counter increments per
// instruction (tools/valu_calibration.py joins the two).  Kernel names carry the form: calib<OP, WPS>.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/valu_calib.hip -o tools/exp/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define V8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum { FMA, MUL, CMP_SGPR, CNDMASK_SGPR, DPP_ROW_SHR, DPP_QUAD, PERMLANE32, PERMLANE16, EXP, RCP, MED3, BLEND_MIX, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_cmp_lt_f32 -> sgpr pair", "v_cndmask_b32_e64 sgpr mask",
                                    "v_add_f32 dpp row_shr", "v_add_f32 dpp quad_perm", "v_permlane32_swap",
                                    "v_permlane16_swap", "v_exp_f32", "v_rcp_f32", "v_med3_f32",
                                    "blend-backward mix (39 plain : 4 swap : 4 dpp : 3 cmp : 3 cndmask : 2 exp/rcp per 55)"};

template <int OP, int WPS>
__global__ void __launch_bounds__(256) calib(float* out, unsigned long long* clk, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == FMA) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == MUL) {
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == CMP_SGPR) {
            REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[22:23], %3, %8\n"
                              "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[22:23], %7, %8\n"
                              : V8 : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");)
        } else if (OP == CNDMASK_SGPR) {
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n"
                              "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n"
                              "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
                              : V8 : "v"(b), "v"(c) : "s20", "s21");)
        } else if (OP == DPP_ROW_SHR) {
            REP8(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              "v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == DPP_QUAD) {
            REP8(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == PERMLANE32) {
            REP8(asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == PERMLANE16) {
            REP8(asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == EXP) {
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" : V8 : "v"(b), "v"(c));)
        } else if (OP == RCP) {
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n" : V8 : "v"(b), "v"(c));)
        } else if (OP == MED3) {
            REP8(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                              "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                              : V8 : "v"(b), "v"(c));)
        } else if (OP == BLEND_MIX) {
            // 55 instructions in the proportions of blend_backward_kernel's inner loop (profiles/r02t_isa_mix.txt: per four
            // splats 155 plain VALU, 18 permlane swaps, 17 DPP adds, 13 compares, 11 selects, 8 transcendentals), 64 / 55 per "8"
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_mul_f32 %2, %2, %8\n v_fma_f32 %3, %3, %8, %9\n v_add_f32 %4, %4, %8\n"
                "v_cmp_lt_f32 s[20:21], %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_mul_f32 %7, %7, %8\n v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n"
                "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_mul_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n"
                "s_nop 1\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                "v_permlane32_swap_b32 %2, %3\n v_fma_f32 %4, %4, %8, %9\n v_mul_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_cmp_lt_f32 s[22:23], %7, %8\n"
                "v_fma_f32 %0, %0, %8, %9\n v_add_f32 %1, %1, %8\n v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n"
                "s_nop 1\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fma_f32 %6, %6, %8, %9\n v_mul_f32 %7, %7, %8\n"
                "v_permlane16_swap_b32 %0, %1\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_rcp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n"
                "v_cmp_lt_f32 s[20:21], %6, %8\n v_fma_f32 %7, %7, %8, %9\n v_mul_f32 %0, %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n"
                "s_nop 1\n v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                "v_permlane32_swap_b32 %6, %7\n v_fma_f32 %0, %0, %8, %9\n v_add_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_mul_f32 %3, %3, %8\n"
                "s_nop 1\n v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fma_f32 %5, %5, %8, %9\n v_permlane16_swap_b32 %6, %7\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_mul_f32 %2, %2, %8\n"
                : V8 : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static float* g_out;
static unsigned long long* g_clk;

template <int OP, int WPS>
void run() {
    const int cus = 256, blocks = cus * WPS;            // 256 threads = 4 waves per block, one per SIMD: WPS waves per SIMD
    const int per_iter = OP == BLEND_MIX ? 55 : 64;
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    calib<OP, WPS><<<blocks, 256>>>(g_out, g_clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    calib<OP, WPS><<<blocks, 256>>>(g_out, g_clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { ticks += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    const double mhz = 100.0 * ticks / real;                              // s_memrealtime: 100 MHz
    const double wave_insts = (double)iters * per_iter;
    const double cyc_per_inst_simd = (ticks / blocks) / (wave_insts * WPS);   // a SIMD issues for its WPS waves in turn
    printf("{\"kernel\": \"calib<%d, %d>\", \"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_wave\": %.0f, "
           "\"valu_insts_per_launch\": %.0f, \"shader_clock_MHz\": %.1f, \"cycles_per_inst_per_simd\": %.3f, "
           "\"cycles_per_inst_per_simd_from_wall_time\": %.3f}\n",
           OP, WPS, kNames[OP], WPS, ms, wave_insts, wave_insts * blocks * 4, mhz, cyc_per_inst_simd,
           ms * 1e-3 * mhz * 1e6 / (wave_insts * WPS));
    fflush(stdout);
}

int main() {
    hipMalloc(&g_out, 256 * 8 * 256 * 4);
    hipMalloc(&g_clk, 256 * 8 * 16);
    run<FMA, 1>(); run<FMA, 2>(); run<FMA, 4>(); run<FMA, 5>(); run<FMA, 8>();
    run<MUL, 5>(); run<CMP_SGPR, 5>(); run<CNDMASK_SGPR, 5>(); run<DPP_ROW_SHR, 5>(); run<DPP_QUAD, 5>();
    run<PERMLANE32, 5>(); run<PERMLANE16, 5>(); run<EXP, 5>(); run<RCP, 5>(); run<MED3, 5>();
    run<BLEND_MIX, 1>(); run<BLEND_MIX, 4>(); run<BLEND_MIX, 5>(); run<BLEND_MIX, 8>();
    return 0;
}
