// Developer probe 2: per-instruction issue cost on gfx950 for more instruction forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define OPS8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define KERNEL(NAME, BODY, ...)                                                                              \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                      \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,  \
              a7 = a0 + 7, b = 1.0001f, c = 0.5f;                                                             \
        for (int i = 0; i < iters; ++i) {                                                                     \
            REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),     \
                              "+v"(a7) : "v"(b), "v"(c) : __VA_ARGS__);)                                             \
        }                                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                          \
    }
#define F_CND_VCC(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define F_CND_VCC_D(n) "v_cndmask_b32 %" #n ", %9, %8, vcc\n"
#define F_CND_S(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define F_CND_S0(n) "v_cndmask_b32_e64 %" #n ", 0, %8, s[20:21]\n"
#define F_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define F_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define F_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define F_MULLIT(n) "v_mul_f32 %" #n ", 0x3fb8aa3b, %" #n "\n"
#define F_MINLIT(n) "v_min_f32 %" #n ", 0x3f7d70a4, %" #n "\n"
#define F_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define F_CMPVCC(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define F_PL16(n) "v_permlane16_swap_b32 %" #n ", %8\n"
#define F_QP(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_MOVDPP(n) "v_mov_b32_dpp %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_RFL(n) "v_readfirstlane_b32 s20, %" #n "\n"
#define F_SUBS(n) "v_sub_f32 %" #n ", s20, %" #n "\n"
#define F_FMA_S(n) "v_fma_f32 %" #n ", %" #n ", s20, %8\n"
#define F_MADU64(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8\n"
#define F_SNOP(n) "s_nop 0\n"
#define F_SALU(n) "s_add_u32 s20, s20, 1\n"
#define F_SAND(n) "s_and_b64 s[20:21], s[20:21], s[22:23]\n"
#define F_MIXFS(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n s_and_b64 s[20:21], s[20:21], s[22:23]\n"
KERNEL(k_cnd_vcc, OPS8(F_CND_VCC), "vcc")
KERNEL(k_cnd_vcc_d, OPS8(F_CND_VCC_D), "vcc")
KERNEL(k_cnd_s, OPS8(F_CND_S), "s20", "s21")
KERNEL(k_cnd_s0, OPS8(F_CND_S0), "s20", "s21")
KERNEL(k_mov, OPS8(F_MOV), "s20")
KERNEL(k_addu, OPS8(F_ADDU), "s20")
KERNEL(k_mul, OPS8(F_MUL), "s20")
KERNEL(k_mullit, OPS8(F_MULLIT), "s20")
KERNEL(k_minlit, OPS8(F_MINLIT), "s20")
KERNEL(k_fmac, OPS8(F_FMAC), "s20")
KERNEL(k_cmpvcc, OPS8(F_CMPVCC), "vcc")
KERNEL(k_pl16, "s_nop 1\n" OPS8(F_PL16), "s20")
KERNEL(k_qp, "s_nop 1\n" OPS8(F_QP), "s20")
KERNEL(k_movdpp, "s_nop 1\n" OPS8(F_MOVDPP), "s20")
KERNEL(k_rfl, OPS8(F_RFL), "s20")
KERNEL(k_subs, OPS8(F_SUBS), "s20")
KERNEL(k_fma_s, OPS8(F_FMA_S), "s20")
KERNEL(k_lshladd, OPS8(F_MADU64), "s20")
KERNEL(k_snop, OPS8(F_SNOP), "s20")
KERNEL(k_salu, OPS8(F_SALU), "s20", "scc")
KERNEL(k_sand, OPS8(F_SAND), "s20", "s21", "s22", "s23", "scc")
KERNEL(k_mix, OPS8(F_MIXFS), "s20", "s21", "s22", "s23", "scc")

typedef void (*kern_t)(float*, int);
void run(const char* name, kern_t k, float* out, int wpsimd, double per_iter = 64) {
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<cus * wpsimd, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<cus * wpsimd, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (iters * per_iter * wpsimd);
    printf("%-26s waves/SIMD=%d  %.3f ns per instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wpsimd, ns, ns * 2.4); fflush(stdout);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 4}) {
#define R(k) run(#k, k, out, w);
        R(k_cnd_vcc) R(k_cnd_vcc_d) R(k_cnd_s) R(k_cnd_s0) R(k_mov) R(k_addu) R(k_mul) R(k_mullit) R(k_minlit) R(k_fmac)
        R(k_cmpvcc) R(k_pl16) R(k_qp) R(k_movdpp) R(k_rfl) R(k_subs) R(k_fma_s) R(k_lshladd) R(k_snop) R(k_salu) R(k_sand)
        run("k_mix (fma+s_and pairs)", k_mix, out, w, 64);
    }
    return 0;
}
