// Developer probe (round 6): layout of global_load_lds_dwordx4 / _dwordx3 on gfx950 -- does lane l's data land at
// dst + l * size?  Build: hipcc --offload-arch=gfx950 -O2 tools/exp/lds_direct_probe.hip -o tools/exp/lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k(const float* __restrict__ x, float* __restrict__ y4, float* __restrict__ y3) {
    __shared__ __attribute__((aligned(16))) float b4[2][64 * 4];
    __shared__ __attribute__((aligned(16))) float b3[2][64 * 3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // lane l reads 4 (3) floats from a lane-dependent, NON-linear place: x + 4 * perm(l)
    const int perm = (lane * 7 + 3) & 63;
    __builtin_amdgcn_global_load_lds(GP(x + 4 * perm + 1024 * w), LP(b4[w]), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GP(x + 3 * perm + 1024 * w), LP(b3[w]), 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = 0; e < 4; ++e) y4[(w * 64 + lane) * 4 + e] = b4[w][lane * 4 + e];
    for (int e = 0; e < 3; ++e) y3[(w * 64 + lane) * 3 + e] = b3[w][lane * 3 + e];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *x, *y4, *y3;
    hipMalloc(&x, 4096 * 4); hipMalloc(&y4, 128 * 4 * 4); hipMalloc(&y3, 128 * 3 * 4);
    hipMemcpy(x, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    k<<<1, 128>>>(x, y4, y3);
    std::vector<float> r4(512), r3(384);
    hipMemcpy(r4.data(), y4, 512 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r3.data(), y3, 384 * 4, hipMemcpyDeviceToHost);
    int bad4 = 0, bad3 = 0;
    for (int w = 0; w < 2; ++w)
        for (int l = 0; l < 64; ++l) {
            const int perm = (l * 7 + 3) & 63;
            for (int e = 0; e < 4; ++e) bad4 += r4[(w * 64 + l) * 4 + e] != (float)(4 * perm + 1024 * w + e);
            for (int e = 0; e < 3; ++e) bad3 += r3[(w * 64 + l) * 3 + e] != (float)(3 * perm + 1024 * w + e);
        }
    printf("dwordx4: %d mismatches; dwordx3: %d mismatches\n", bad4, bad3);
    printf("x4 lane 0..3: %g %g %g %g | %g %g %g %g\n", r4[0], r4[1], r4[2], r4[3], r4[4], r4[5], r4[6], r4[7]);
    printf("x3 lane 0..2: %g %g %g | %g %g %g\n", r3[0], r3[1], r3[2], r3[3], r3[4], r3[5]);
    return 0;
}
