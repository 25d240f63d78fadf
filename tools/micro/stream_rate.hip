// Streaming-pass rate of an MI355X for the shapes the library's element-wise kernels use: y = 2 x over N bytes (read N + write N),
// 16 bytes per lane and access, variants of grid size, unroll (chunks in flight per lane) and cache policy.
//   hipcc --offload-arch=gfx950 -O3 -o stream_rate stream_rate.hip && ./stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int U, bool NT>
__global__ __launch_bounds__(256) void scale_kernel(const u4* __restrict__ x, u4* __restrict__ y, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u4 o = v[u];
            o[0] ^= 0x80008000u;      // sign flip of the bf16 pairs: a real ALU op per dword
            o[1] ^= 0x80008000u;
            o[2] ^= 0x80008000u;
            o[3] ^= 0x80008000u;
            if (NT) __builtin_nontemporal_store(o, y + i + u * stride);
            else y[i + u * stride] = o;
        }
    }
    for (; i < n; i += stride) y[i] = x[i];
}

// one contiguous slab per workgroup (the workgroup walks its own range: DRAM pages / channels see long runs)
template <int U, bool NT>
__global__ __launch_bounds__(256) void slab_kernel(const u4* __restrict__ x, u4* __restrict__ y, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u4 o = v[u];
            o[0] ^= 0x80008000u; o[1] ^= 0x80008000u; o[2] ^= 0x80008000u; o[3] ^= 0x80008000u;
            if (NT) __builtin_nontemporal_store(o, y + i + u * 256);
            else y[i + u * 256] = o;
        }
    }
    for (; i < hi; i += 256) y[i] = x[i];
}

template <typename K>
static float run(K kern, int grid, const u4* x, u4* y, size_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, x, y, n);
    std::vector<float> t;
    for (int r = 0; r < 9; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, x, y, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    for (size_t mb : {154ul, 512ul, 1024ul}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        u4 *x, *y;
        if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) return 1;
        hipMemset(x, 1, bytes);
        printf("== %zu MiB in, %zu MiB out\n", mb, mb);
        for (int grid : {1024, 2048, 4096, 8192, 16384}) {
#define ROW(NAME, K) { float ms = run(K, grid, x, y, n); printf("%-22s grid %5d  %7.3f ms  %7.1f GB/s\n", NAME, grid, ms, 2.0 * bytes / ms / 1e6); }
            ROW("stride U1", (scale_kernel<1, false>));
            ROW("stride U4", (scale_kernel<4, false>));
            ROW("stride U4 nt", (scale_kernel<4, true>));
            ROW("stride U8 nt", (scale_kernel<8, true>));
            ROW("slab U4", (slab_kernel<4, false>));
            ROW("slab U4 nt", (slab_kernel<4, true>));
            ROW("slab U8 nt", (slab_kernel<8, true>));
        }
        hipFree(x); hipFree(y);
    }
    return 0;
}
