// VALU issue-rate probe for gfx950: v_fma_f32, v_pk_fma_f32, v_dot2_f32_bf16, bf16 unpack (shift / and), with 1-8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));
typedef __bf16 v2b __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int MODE> __global__ void probe(float* out, float a, float b) {
    v2f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v2f{(float)threadIdx.x, (float)i};
    v2f va = v2f{a, a * 1.0001f}, vb = v2f{b, b};
    unsigned u = __float_as_uint(a) + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MODE == 0) {          // 2 x v_fma_f32
                acc[i].x = __builtin_fmaf(acc[i].x, a, b);
                acc[i].y = __builtin_fmaf(acc[i].y, a, b);
            } else if constexpr (MODE == 1) {   // 1 x v_pk_fma_f32
                acc[i] = __builtin_elementwise_fma(acc[i], va, vb);
            } else if constexpr (MODE == 2) {   // 2 x v_dot2_f32_bf16
                unsigned p = u + i;
                acc[i].x = __builtin_amdgcn_fdot2_f32_bf16(*(v2b*)&p, *(v2b*)&u, acc[i].x, false);
                acc[i].y = __builtin_amdgcn_fdot2_f32_bf16(*(v2b*)&p, *(v2b*)&u, acc[i].y, false);
            } else if constexpr (MODE == 3) {   // unpack (2 ops) + 1 pk_fma
                unsigned p = u + i + it;
                v2f x = v2f{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
                acc[i] = __builtin_elementwise_fma(x, va, acc[i]);
            } else if constexpr (MODE == 4) {   // pk_fma with distinct weight regs (8 weights), x operand varies
                v2f x = v2f{acc[(i + 1) & 7].y, acc[(i + 3) & 7].x};
                acc[i] = __builtin_elementwise_fma(x, va, acc[i]);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int wps, float flop_per_inner) {
    float* out;
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wps, threads = 256;   // 4 waves per block -> one per SIMD, wps blocks per CU
    probe<MODE><<<blocks, threads>>>(out, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<MODE><<<blocks, threads>>>(out, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double inner = (double)blocks * threads * ITER * 8;
    printf("%-34s waves/SIMD %d  %8.3f ms  %8.1f TFLOP/s  %6.2f cycles per wave-iteration-slot (2.4 GHz)\n", name, wps, ms,
           inner * flop_per_inner / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * wps));
    hipFree(out);
}

int main() {
    for (int wps : {1, 2, 4}) {
        run<0>("2 x v_fma_f32", wps, 4);
        run<1>("1 x v_pk_fma_f32", wps, 4);
        run<2>("2 x v_dot2_f32_bf16", wps, 8);
        run<3>("unpack(2) + 1 x v_pk_fma_f32", wps, 4);
        run<4>("1 x v_pk_fma_f32 (mixed regs)", wps, 4);
    }
    return 0;
}
