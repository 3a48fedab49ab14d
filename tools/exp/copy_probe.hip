// Developer probe: which float4 copy shape reaches the highest HBM bandwidth on MI355X (bench.py's peak_measured).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_stride(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ void __launch_bounds__(256) copy_flat(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int U>
__global__ void __launch_bounds__(256) copy_unroll(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) d[base + u * 256] = v[u];
}
template <int U>
__global__ void __launch_bounds__(256) copy_nt(const float4* __restrict__ s4, float4* __restrict__ d4, size_t n) {
    const f4* s = (const f4*)s4; f4* d = (f4*)d4;
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = __builtin_nontemporal_load(&s[base + u * 256]);
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) __builtin_nontemporal_store(v[u], &d[base + u * 256]);
}
int main() {
    size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        printf("%-28s %.1f GB/s (read + write)\n", name, 2.0 * bytes / (best * 1e-3) / 1e9);
    };
    for (int g : {4096, 8192, 16384, 65536}) {
        char nm[64]; snprintf(nm, 64, "stride grid %d", g);
        run(nm, [&] { copy_stride<<<g, 256>>>(a, b, n); });
    }
    run("flat", [&] { copy_flat<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); });
    run("unroll 2", [&] { copy_unroll<2><<<(unsigned)((n + 511) / 512), 256>>>(a, b, n); });
    run("unroll 4", [&] { copy_unroll<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); });
    run("unroll 8", [&] { copy_unroll<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); });
    run("nontemporal unroll 4", [&] { copy_nt<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); });
    run("nontemporal unroll 8", [&] { copy_nt<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); });
    return 0;
}
