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
// read-only (sum kept alive through a never-taken store), write-only, and R reads : Wn writes per lane: the write-heavy
// kernels of the anchor path (heads forward 1 : 2, expand backward 1 : 1.4) against what a mix can reach
template <int U>
__global__ void __launch_bounds__(256) read_only(const float4* __restrict__ s4, float4* __restrict__ d4, size_t n) {
    const f4* s = (const f4*)s4;
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) acc += __builtin_nontemporal_load(&s[base + u * 256]);
    if (acc[0] == 123.456f) ((f4*)d4)[base] = acc;
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) write_only(float4* __restrict__ d4, size_t n, float x) {
    f4* d = (f4*)d4;
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const f4 v = {x, x, x, x};
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { if (NT) __builtin_nontemporal_store(v, &d[base + u * 256]); else d[base + u * 256] = v; }
}
template <int R, int Wn>
__global__ void __launch_bounds__(256) mix(const float4* __restrict__ s4, float4* __restrict__ d4, size_t nr, size_t nw) {
    const f4* s = (const f4*)s4; f4* d = (f4*)d4;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, blocks = gridDim.x;
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < R; ++u) { const size_t i = (u * blocks) * 256 + t; if (i < nr) acc += __builtin_nontemporal_load(&s[i]); }
#pragma unroll
    for (int u = 0; u < Wn; ++u) { const size_t i = (u * blocks) * 256 + t; if (i < nw) d[i] = acc; }
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
    auto run1 = [&](const char* name, double moved, auto launch) {
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        printf("%-28s %.1f GB/s\n", name, moved / (best * 1e-3) / 1e9);
    };
    run1("read only unroll 4", (double)bytes, [&] { read_only<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); });
    run1("read only unroll 8", (double)bytes, [&] { read_only<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); });
    run1("write only unroll 4", (double)bytes, [&] { write_only<4, false><<<(unsigned)((n + 1023) / 1024), 256>>>(b, n, 1.0f); });
    run1("write only unroll 4 nt", (double)bytes, [&] { write_only<4, true><<<(unsigned)((n + 1023) / 1024), 256>>>(b, n, 1.0f); });
    run1("write only unroll 8 nt", (double)bytes, [&] { write_only<8, true><<<(unsigned)((n + 2047) / 2048), 256>>>(b, n, 1.0f); });
    {   // 1 read : 2 writes (heads forward), 2 : 1, 1 : 1 with plain stores
        const unsigned blocks = (unsigned)(n / 2 / 256);
        run1("mix 1 read : 2 writes", 1.5 * bytes, [&] { mix<1, 2><<<blocks, 256>>>(a, b, n / 2, n); });
        run1("mix 2 reads : 1 write", 1.5 * bytes, [&] { mix<2, 1><<<blocks, 256>>>(a, b, n, n / 2); });
        run1("mix 2 reads : 2 writes", 2.0 * bytes, [&] { mix<2, 2><<<blocks, 256>>>(a, b, n, n); });
        run1("mix 4 reads : 4 writes", 2.0 * bytes, [&] { mix<4, 4><<<blocks / 2, 256>>>(a, b, n, n); });
    }
    return 0;
}
