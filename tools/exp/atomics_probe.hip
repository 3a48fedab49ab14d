// Developer probe: cost of global integer atomics on MI355X by scope / return / XCD privatisation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint32_t hashu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int MODE>
__global__ void k(uint32_t* cnt, int tiles, int per, uint32_t* sink) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    uint32_t* base = cnt;
    if (MODE >= 2) base = cnt + (size_t)xcc_id() * tiles;
    for (int r = 0; r < per; ++r) {
        uint32_t t = hashu(i * 16 + r) % tiles;
        if (MODE == 0) __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) acc += __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) acc += __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) __hip_atomic_fetch_add(&base[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // agent scope on XCD-private copy
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

template <int MODE>
int run(const char* name, uint32_t* cnt, uint32_t* sink, int tiles, int n, int per) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    std::vector<uint32_t> h((size_t)tiles * 16);
    unsigned long long total = 0;
    for (int it = 0; it < 5; ++it) {
        CK(hipMemset(cnt, 0, (size_t)tiles * 16 * 4));
        CK(hipEventRecord(a));
        k<MODE><<<n / 256, 256>>>(cnt, tiles, per, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(h.data(), cnt, (size_t)tiles * 16 * 4, hipMemcpyDeviceToHost));
    for (auto v : h) total += v;
    printf("%-34s %8.1f us  %6.2f G atomics/s   sum=%llu (want %llu) %s\n", name, best * 1e3, (double)n * per / best / 1e6, total,
           (unsigned long long)n * per, total == (unsigned long long)n * per ? "OK" : "MISMATCH");
    return 0;
}

int main() {
    int tiles = 8160, n = 1 << 20, per = 4;
    uint32_t *cnt, *sink;
    CK(hipMalloc(&cnt, (size_t)tiles * 16 * 4)); CK(hipMalloc(&sink, 64));
    run<0>("agent, no return", cnt, sink, tiles, n, per);
    run<1>("agent, returning", cnt, sink, tiles, n, per);
    run<2>("workgroup scope, XCD copy, no ret", cnt, sink, tiles, n, per);
    run<3>("workgroup scope, XCD copy, ret", cnt, sink, tiles, n, per);
    run<4>("agent scope, XCD copy, no ret", cnt, sink, tiles, n, per);
    return 0;
}
