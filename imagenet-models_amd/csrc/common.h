// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libgaext.
// wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/gaext.h"

typedef unsigned short bf16_t;  // storage type for bfloat16
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// ----------------------------------------------------------------------------------------------
// error plumbing (include/gaext.h: ga_last_error)
// ----------------------------------------------------------------------------------------------
void ga_set_error(const char* fmt, ...);
int ga_check_launch(const char* what);

// tuning knobs (runtime.hip): GAEXT_<NAME> from the environment, read ONCE when the knob is first looked up, or set through
// ga_set_knob(); dispatch code keeps the slot pointer in a function-local static, so no launch ever calls getenv
#include <atomic>
const std::atomic<int>* ga_knob_slot(const char* name, int dflt);
#define GA_KNOB(NAME, DFLT)                                                       \
    ([]() -> int {                                                                \
        static const std::atomic<int>* slot_ = ga_knob_slot(NAME, DFLT);          \
        return slot_->load(std::memory_order_relaxed);                            \
    }())

#define GA_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ga_set_error(__VA_ARGS__);        \
            return GA_ERR_BAD_ARG;            \
        }                                     \
    } while (0)

// ----------------------------------------------------------------------------------------------
// scalar conversions
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> struct elt;
template <> struct elt<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }   // value as stored
};
template <> struct elt<bf16_t> {
    static constexpr int EPC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32). Pointers must be 16-B aligned.
__device__ __forceinline__ void load8(const float* p, float v[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
    uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack8(const uint4& a, float v[8]) {   // 8 bf16 held in a uint4
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    // ONE v_cvt_pk_bf16_f32 for the pair (two scalar converts + shift/or otherwise)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t b = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return *reinterpret_cast<const unsigned*>(&b);
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
    uint4 a;
    a.x = pack2bf(v[0], v[1]); a.y = pack2bf(v[2], v[3]); a.z = pack2bf(v[4], v[5]); a.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = a;
}
// streaming (non-temporal) forms: tensors that a kernel writes once / reads once should not displace the re-used operand
// panels from the 4 MiB L2 of the XCD (measured: GEMM output tiles with nt stores, -5...20 % per launch)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint4 load16_nt(const void* p) {
    const u32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ void store16_nt(void* p, const uint4& v) {
    u32x4_t a;
    a.x = v.x; a.y = v.y; a.z = v.z; a.w = v.w;
    __builtin_nontemporal_store(a, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ void load8_nt(const bf16_t* p, float v[8]) { unpack8(load16_nt(p), v); }
__device__ __forceinline__ void load8_nt(const float* p, float v[8]) {
    const uint4 a = load16_nt(p), b = load16_nt(p + 4);
    v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
    v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y); v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
}
__device__ __forceinline__ void store8_nt(bf16_t* p, const float v[8]) {
    uint4 a;
    a.x = pack2bf(v[0], v[1]); a.y = pack2bf(v[2], v[3]); a.z = pack2bf(v[4], v[5]); a.w = pack2bf(v[6], v[7]);
    store16_nt(p, a);
}
__device__ __forceinline__ void store8_nt(float* p, const float v[8]) {
    store16_nt(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    store16_nt(p + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
}
// 4 consecutive elements
__device__ __forceinline__ void load4(const float* p, float v[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
    uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
    uint2 a;
    a.x = pack2bf(v[0], v[1]); a.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = a;
}

// ----------------------------------------------------------------------------------------------
// activations
// ----------------------------------------------------------------------------------------------
// exact-erf GELU (nn.GELU default, ga_convnext.py:94) through Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7):
// one v_rcp + one v_exp + ~12 FMAs instead of the libm erff call, since GELU / GELU' are RE-COMPUTED inside the
// GEMM operand loader and epilogue instead of being stored.  E = exp(-x^2/2) is shared by GELU'.
__device__ __forceinline__ float gelu_cdf_f(float x, float* e_out) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn is a full IEEE divide
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-z * z);           // exp(-x^2/2)
    const float half_erfc = 0.5f * p * t * e; // 0.5 * erfc(|x|/sqrt2)
    if (e_out) *e_out = e;
    return x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_cdf_f(x, nullptr); }
// gelu(x) and gelu'(x) from ONE erf/exp evaluation (the fc1 epilogue stores both)
__device__ __forceinline__ void gelu_both_f(float x, float& act, float& grad) {
    float e;
    const float cdf = gelu_cdf_f(x, &e);
    act = x * cdf;
    grad = fmaf(x * 0.3989422804014327f, e, cdf);
}
// bf16 throughput mode only: GELU through the logistic form Phi(x) ~ sigmoid(2*sqrt(2/pi)*(x + 0.044715 x^3)) (the
// tanh approximation, |gelu error| <= 5e-4 absolute -- below the bf16 rounding of the stored value for |gelu| >= 0.13)
// and its exact derivative: 14 VALU slots for both outputs instead of 25 for the erf form.  The fp32 parity mode and
// every standalone GELU keep the erf form above.
__device__ __forceinline__ void gelu_both_fast(float x, float& act, float& grad) {
    const float x2 = x * x;
    // -z*log2(e) with z = 1.5957691216 x (1 + 0.044715 x^2)
    const float e = __builtin_amdgcn_exp2f(x * fmaf(x2, -0.10294324f, -2.3022082f));
    const float s = __builtin_amdgcn_rcpf(1.0f + e);            // sigmoid(z)
    act = x * s;
    const float zp = fmaf(x2, 0.21406186f, 1.5957691f);          // dz/dx = 1.5957691216 (1 + 3*0.044715 x^2)
    grad = fmaf(x * (s - s * s), zp, s);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float e;
    const float cdf = gelu_cdf_f(x, &e);
    return fmaf(x * 0.3989422804014327f, e, cdf);
}

// ----------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over a sub-group of G consecutive lanes (G power of two <= 64)
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// conv3.hip: the direct 3 x 3 / 64-channel form of ga_gemm's GA_A_CONV3 product; returns 1 if it took the launch
int ga_conv3_c64_try(const ga_gemm_desc* d, hipStream_t s);
// ... and of the GA_A_CONV3S2 product of the 3 -> 64-channel first convolution on the NHWC8 image
int ga_conv0_c8_try(const ga_gemm_desc* d, hipStream_t s);
size_t ga_conv3_c64_wgrad_workspace(const ga_wgrad_desc* d);
int ga_conv3_c64_wgrad_try(const ga_wgrad_desc* d, hipStream_t s);
int ga_conv3s2_c64_wgrad_try(const ga_wgrad_desc* d, hipStream_t s);
int ga_conv0_c8_wgrad_try(const ga_wgrad_desc* d, hipStream_t s);
