// Alignment-free helpers for the odd-width variants (ga_convnext_{tiny,small}_688, base_976: ga_convnext.py:572-613).
//
// 688 / 8 = 86 and 688 / 4 = 172 channels per group put the per-group operands of the heads' grouped 1x1 convolutions
// (gram_embedding, GroupConvMlp: ga_convnext.py:190-222,418-420) off the 16-byte grid every MFMA kernel of this library
// loads on.  Those layers act on ONE token per image (M = batch rows), a few hundred MFLOP: they run here as LDS-tiled
// fp32 products with element-wise (any alignment) loads, straight from the fp32 master weights -- no weight preparation,
// no padding of the model.  The 172-wide stage-4 Bottleneck (50,176 rows) does use the MFMA kernels: its parameters are
// copied into zero-padded 176-wide buffers by ga_pad_copy_f32 (and their gradients copied back).
#include <algorithm>
#include "common.h"

namespace {

// C[i][j] = sum_l a(i, l) * b(l, j) over a TM x 64 tile per workgroup (256 threads: column j0 + (t & 63), TM / 4 rows each).
// AL / BL: the operand's memory-contiguous index is l (else i resp. j) -- consecutive threads then walk that index.
template <int TM, bool AL, bool BL, typename FA, typename FB, typename FS>
__device__ __forceinline__ void tiled_mm(int M, int N, int K, FA a, FB b, FS store) {
    constexpr int RPT = TM / 4;
    // 64-deep k steps: these launches are latency-bound (element-wise loads, some through a permutation index) -- a step keeps
    // 16 + TM / 4 independent loads per thread in flight and costs two barriers (head MLP layer of tiny_688: 0.073 ms with 16-deep steps, 0.029 now)
    constexpr int KS = 64;
    __shared__ float As[KS][TM + 1], Bs[KS][65];
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    const int i0 = blockIdx.y * TM, j0 = blockIdx.x * 64;
    float acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KS) {
        float av[KS * TM / 256], bv_[KS * 64 / 256];
#pragma unroll
        for (int u = 0; u < KS * TM / 256; ++u) {
            const int e = t + 256 * u;
            const int kk = AL ? e % KS : e / TM, ii = AL ? e / KS : e % TM;
            av[u] = (i0 + ii < M && k0 + kk < K) ? a(i0 + ii, k0 + kk) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KS * 64 / 256; ++u) {
            const int e = t + 256 * u;
            const int kk = BL ? e % KS : e >> 6, jj = BL ? e / KS : e & 63;
            bv_[u] = (j0 + jj < N && k0 + kk < K) ? b(k0 + kk, j0 + jj) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KS * TM / 256; ++u) {
            const int e = t + 256 * u;
            As[AL ? e % KS : e / TM][AL ? e / KS : e % TM] = av[u];
        }
#pragma unroll
        for (int u = 0; u < KS * 64 / 256; ++u) {
            const int e = t + 256 * u;
            Bs[BL ? e % KS : e >> 6][BL ? e / KS : e & 63] = bv_[u];
        }
        __syncthreads();
        const int kn = K - k0 < KS ? K - k0 : KS;
        for (int kk = 0; kk < kn; ++kk) {
            const float bv = Bs[kk][tx];
#pragma unroll
            for (int r = 0; r < RPT; ++r) acc[r] = fmaf(As[kk][ty * RPT + r], bv, acc[r]);
        }
        __syncthreads();
    }
    if (j0 + tx < N) {
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (i0 + ty * RPT + r < M) store(i0 + ty * RPT + r, j0 + tx, acc[r]);
    }
}

template <typename T> __device__ __forceinline__ float ldel(const void* p, long i) { return elt<T>::ld(reinterpret_cast<const T*>(p) + i); }
template <typename T> __device__ __forceinline__ void stel(void* p, long i, float v) { elt<T>::st(reinterpret_cast<T*>(p) + i, v); }

// Y[r][g*Ng + n] = R + rowscale * col_scale[.] * (sum_k A[r][acol(g*a_gstride + k)] W[g*Ng + n][k] + bias[.])
template <typename T, int TM>
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const ga_small_linear_desc d) {
    const int g = blockIdx.z;
    const long abase = (long)g * d.a_gstride;
    auto a = [&](int i, int l) {
        const long c = abase + l;
        return ldel<T>(d.A, (long)i * d.lda + (d.a_perm ? d.a_perm[c] : c));
    };
    auto b = [&](int l, int j) { return d.W[((long)g * d.Ng + j) * d.Kg + l]; };
    auto st = [&](int i, int j, float v) {
        const long n = (long)g * d.Ng + j;
        if (d.bias) v += d.bias[n];
        if (d.Yraw) stel<T>(d.Yraw, (long)i * d.ldy + n, v);
        if (d.col_scale) v *= d.col_scale[n];
        if (d.rowscale) v *= d.rowscale[i / d.rows_per_scale];
        if (d.R) v += ldel<T>(d.R, (long)i * d.ldr + n);
        stel<T>(d.Y, (long)i * d.ldy + n, v);
    };
    tiled_mm<TM, true, true>(d.rows, d.Ng, d.Kg, a, b, st);
}

// the gradient reaching the product: dY * rowscale * col_scale
template <typename T> __device__ __forceinline__ float dy_eff(const ga_small_linear_desc& d, const void* dY, int r, long n) {
    float v = ldel<T>(dY, (long)r * d.ldy + n);
    if (d.col_scale) v *= d.col_scale[n];
    if (d.rowscale) v *= d.rowscale[r / d.rows_per_scale];
    return v;
}

// dA[r][acol(g*a_gstride + k)] (+)= sum_n dYeff[r][g*Ng + n] W[g*Ng + n][k]
template <typename T, int TM>
__global__ __launch_bounds__(256) void small_linear_dgrad_kernel(const ga_small_linear_desc d, const void* dY, void* dA, int accumulate) {
    const int g = blockIdx.z;
    const long abase = (long)g * d.a_gstride;
    auto a = [&](int i, int l) { return dy_eff<T>(d, dY, i, (long)g * d.Ng + l); };
    auto b = [&](int l, int j) { return d.W[((long)g * d.Ng + l) * d.Kg + j]; };
    auto st = [&](int i, int j, float v) {
        const long c = abase + j;
        const long o = (long)i * d.lda + (d.a_perm ? d.a_perm[c] : c);
        stel<T>(dA, o, accumulate ? v + ldel<T>(dA, o) : v);
    };
    tiled_mm<TM, true, false>(d.rows, d.Kg, d.Ng, a, b, st);
}

// dW[g*Ng + n][k] += sum_r dYeff[r][g*Ng + n] A[r][acol(g*a_gstride + k)]
template <typename T, int TM>
__global__ __launch_bounds__(256) void small_linear_wgrad_kernel(const ga_small_linear_desc d, const void* dY, float* dW) {
    const int g = blockIdx.z;
    const long abase = (long)g * d.a_gstride;
    auto a = [&](int i, int l) { return dy_eff<T>(d, dY, l, (long)g * d.Ng + i); };
    auto b = [&](int l, int j) {
        const long c = abase + j;
        return ldel<T>(d.A, (long)l * d.lda + (d.a_perm ? d.a_perm[c] : c));
    };
    auto st = [&](int i, int j, float v) { dW[((long)g * d.Ng + i) * d.Kg + j] += v; };
    tiled_mm<TM, false, false>(d.Ng, d.Kg, d.rows, a, b, st);
}

// per output column n: dbias[n] += sum_r dYeff[r][n];  dcol_scale[n] += sum_r dY[r][n] * rowscale * Yraw[r][n]
// workgroup = 32 columns x 8 interleaved row ranges (a single thread per column walking every row is latency-bound)
template <typename T>
__global__ __launch_bounds__(256) void small_linear_colgrad_kernel(const ga_small_linear_desc d, const void* dY, float* dbias,
                                                                   float* dcs) {
    __shared__ float red[2][8][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long n = (long)blockIdx.x * 32 + cl;
    const bool ok = n < (long)d.groups * d.Ng;
    float sb = 0.f, sc = 0.f;
    if (ok) {
        for (int r = rg; r < d.rows; r += 8) {
            float v = ldel<T>(dY, (long)r * d.ldy + n);
            if (d.rowscale) v *= d.rowscale[r / d.rows_per_scale];
            if (dcs) sc = fmaf(v, ldel<T>(d.Yraw, (long)r * d.ldy + n), sc);
            sb += d.col_scale ? v * d.col_scale[n] : v;
        }
    }
    red[0][rg][cl] = sb;
    red[1][rg][cl] = sc;
    __syncthreads();
    if (rg == 0 && ok) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a += red[0][q][cl];
            c += red[1][q][cl];
        }
        if (dbias) dbias[n] += a;
        if (dcs) dcs[n] += c;
    }
}

// column sums / sums of squares of x [rows][C] (row stride ld): the BatchNorm batch statistics the MFMA GEMM's epilogue
// otherwise delivers
template <typename T>
__global__ __launch_bounds__(256) void colstats_kernel(const void* x, long ld, int rows, int C, float* s, float* q) {
    __shared__ float red[2][8][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
        for (int r = rg; r < rows; r += 8) {
            const float v = ldel<T>(x, (long)r * ld + c);
            a += v;
            b = fmaf(v, v, b);
        }
    }
    red[0][rg][cl] = a;
    red[1][rg][cl] = b;
    __syncthreads();
    if (rg == 0 && c < C) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sa += red[0][k][cl];
            sb += red[1][k][cl];
        }
        s[c] += sa;
        if (q) q[c] += sb;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pad_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, long cols,
                                                       long lds, long ldd, int accumulate) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cols, c = i - r * cols;
        const float v = elt<T>::ld(src + r * lds + c);
        elt<T>::st(dst + r * ldd + c, accumulate ? v + elt<T>::ld(dst + r * ldd + c) : v);
    }
}

// two-level group padding of an fp32 matrix: rows in groups of RG -> RGp, columns in groups of CG -> CGp (zero-initialised padded side)
__global__ __launch_bounds__(256) void pad_groups_kernel(const float* __restrict__ src, float* __restrict__ dst, long R, long C, int RG,
                                                         int RGp, int CG, int CGp, int unpad, int accumulate) {
    const long n = R * C, Cp = C / CG * CGp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / C, c = i - r * C;
        const long ip = (r / RG * RGp + r % RG) * Cp + c / CG * CGp + c % CG;
        if (unpad) dst[i] = accumulate ? dst[i] + src[ip] : src[ip];
        else dst[ip] = accumulate ? dst[ip] + src[i] : src[i];
    }
}

// compact [R][cg] <-> block-diagonal [R][ld] with the rows in groups of rg, ng groups per diagonal (R may stack several matrices):
// row r's cg values sit at columns ((r / rg) % ng) * cg ...
__global__ __launch_bounds__(256) void blockdiag_kernel(const float* __restrict__ src, float* __restrict__ dst, long R, int rg, int ng, int cg,
                                                        long ld, int to_diag, int accumulate) {
    const long n = R * cg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cg, c = i - r * cg;
        const long j = r * ld + ((r / rg) % ng) * cg + c;
        if (to_diag) dst[j] = accumulate ? dst[j] + src[i] : src[i];
        else dst[i] = accumulate ? dst[i] + src[j] : src[j];
    }
}

int check_desc(const ga_small_linear_desc* d, const char* what) {
    GA_REQUIRE(d && d->A && d->W && d->rows > 0 && d->groups > 0 && d->Ng > 0 && d->Kg > 0, "%s: null / empty descriptor", what);
    GA_REQUIRE(d->dtype == GA_F32 || d->dtype == GA_BF16, "%s: bad dtype", what);
    GA_REQUIRE(!d->rowscale || d->rows_per_scale > 0, "%s: rows_per_scale", what);
    return GA_OK;
}

}  // namespace

extern "C" int ga_small_linear_fwd(const ga_small_linear_desc* d, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_small_linear_fwd")) return rc;
    GA_REQUIRE(d->Y, "ga_small_linear_fwd: no output");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ct = (d->Ng + 63) / 64;
    if ((long)ct * ((d->rows + 31) / 32) * d->groups >= 512) {
        const dim3 grid(ct, (d->rows + 31) / 32, d->groups);
        if (d->dtype == GA_BF16) hipLaunchKernelGGL((small_linear_fwd_kernel<bf16_t, 32>), grid, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((small_linear_fwd_kernel<float, 32>), grid, dim3(256), 0, s, *d);
    } else {        // few output tiles (86-column groups): 8-row tiles put 4x the workgroups on the chip
        const dim3 grid(ct, (d->rows + 7) / 8, d->groups);
        if (d->dtype == GA_BF16) hipLaunchKernelGGL((small_linear_fwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((small_linear_fwd_kernel<float, 8>), grid, dim3(256), 0, s, *d);
    }
    return ga_check_launch("ga_small_linear_fwd");
}

extern "C" int ga_small_linear_bwd(const ga_small_linear_desc* d, const void* dY, void* dA, int accumulate_dA, float* dW, float* dbias,
                                   float* dcol_scale, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_small_linear_bwd")) return rc;
    GA_REQUIRE(dY, "ga_small_linear_bwd: no output gradient");
    GA_REQUIRE(!dcol_scale || d->Yraw, "ga_small_linear_bwd: the column-scale gradient needs the stored pre-scale output (Yraw)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool bf = d->dtype == GA_BF16;
    // few output tiles (these layers act on one token per image): 8-row tiles put 4x the workgroups on the chip
    if (dA) {
        const int ct = (d->Kg + 63) / 64;
        if ((long)ct * ((d->rows + 31) / 32) * d->groups >= 512) {
            const dim3 grid(ct, (d->rows + 31) / 32, d->groups);
            if (bf) hipLaunchKernelGGL((small_linear_dgrad_kernel<bf16_t, 32>), grid, dim3(256), 0, s, *d, dY, dA, accumulate_dA);
            else hipLaunchKernelGGL((small_linear_dgrad_kernel<float, 32>), grid, dim3(256), 0, s, *d, dY, dA, accumulate_dA);
        } else {
            const dim3 grid(ct, (d->rows + 7) / 8, d->groups);
            if (bf) hipLaunchKernelGGL((small_linear_dgrad_kernel<bf16_t, 8>), grid, dim3(256), 0, s, *d, dY, dA, accumulate_dA);
            else hipLaunchKernelGGL((small_linear_dgrad_kernel<float, 8>), grid, dim3(256), 0, s, *d, dY, dA, accumulate_dA);
        }
    }
    if (dW) {
        const int ct = (d->Kg + 63) / 64;
        if ((long)ct * ((d->Ng + 31) / 32) * d->groups >= 512) {
            const dim3 grid(ct, (d->Ng + 31) / 32, d->groups);
            if (bf) hipLaunchKernelGGL((small_linear_wgrad_kernel<bf16_t, 32>), grid, dim3(256), 0, s, *d, dY, dW);
            else hipLaunchKernelGGL((small_linear_wgrad_kernel<float, 32>), grid, dim3(256), 0, s, *d, dY, dW);
        } else {
            const dim3 grid(ct, (d->Ng + 7) / 8, d->groups);
            if (bf) hipLaunchKernelGGL((small_linear_wgrad_kernel<bf16_t, 8>), grid, dim3(256), 0, s, *d, dY, dW);
            else hipLaunchKernelGGL((small_linear_wgrad_kernel<float, 8>), grid, dim3(256), 0, s, *d, dY, dW);
        }
    }
    if (dbias || dcol_scale) {
        const int n = d->groups * d->Ng;
        if (bf) hipLaunchKernelGGL(small_linear_colgrad_kernel<bf16_t>, dim3((n + 31) / 32), dim3(256), 0, s, *d, dY, dbias, dcol_scale);
        else hipLaunchKernelGGL(small_linear_colgrad_kernel<float>, dim3((n + 31) / 32), dim3(256), 0, s, *d, dY, dbias, dcol_scale);
    }
    return ga_check_launch("ga_small_linear_bwd");
}

extern "C" int ga_colstats(const void* x, int64_t ld, int rows, int C, float* sum, float* sumsq, int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && sum && rows > 0 && C > 0 && ld >= C, "ga_colstats: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(colstats_kernel<bf16_t>, dim3((C + 31) / 32), dim3(256), 0, s, x, (long)ld, rows, C, sum, sumsq);
    else hipLaunchKernelGGL(colstats_kernel<float>, dim3((C + 31) / 32), dim3(256), 0, s, x, (long)ld, rows, C, sum, sumsq);
    return ga_check_launch("ga_colstats");
}

extern "C" int ga_pad_copy(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd, int accumulate, int dtype,
                           ga_stream_t stream) {
    GA_REQUIRE(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols && (dtype == GA_F32 || dtype == GA_BF16),
               "ga_pad_copy: bad args");
    const long n = rows * cols;
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(pad_copy_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(src),
                           reinterpret_cast<bf16_t*>(dst), (long)rows, (long)cols, (long)lds, (long)ldd, accumulate);
    else
        hipLaunchKernelGGL(pad_copy_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(src),
                           reinterpret_cast<float*>(dst), (long)rows, (long)cols, (long)lds, (long)ldd, accumulate);
    return ga_check_launch("ga_pad_copy");
}

extern "C" int ga_blockdiag_f32(const float* src, float* dst, int64_t R, int rg, int ng, int cg, int64_t ld, int to_diag, int accumulate,
                                ga_stream_t stream) {
    GA_REQUIRE(src && dst && R > 0 && rg > 0 && ng > 0 && cg > 0 && R % ((long)rg * ng) == 0 && ld >= (long)ng * cg, "ga_blockdiag_f32: bad args");
    const long n = R * cg;
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + 255) / 256));
    hipLaunchKernelGGL(blockdiag_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, (long)R, rg, ng, cg,
                       (long)ld, to_diag, accumulate);
    return ga_check_launch("ga_blockdiag_f32");
}

extern "C" int ga_pad_groups_f32(const float* src, float* dst, int64_t R, int64_t C, int RG, int RGp, int CG, int CGp, int unpad,
                                 int accumulate, ga_stream_t stream) {
    GA_REQUIRE(src && dst && R > 0 && C > 0 && RG > 0 && CG > 0 && RGp >= RG && CGp >= CG && R % RG == 0 && C % CG == 0,
               "ga_pad_groups_f32: bad args (R=%ld C=%ld RG=%d->%d CG=%d->%d)", (long)R, (long)C, RG, RGp, CG, CGp);
    const long n = R * C;
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n + 255) / 256));
    hipLaunchKernelGGL(pad_groups_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, (long)R, (long)C,
                       RG, RGp, CG, CGp, unpad, accumulate);
    return ga_check_launch("ga_pad_groups_f32");
}

extern "C" int ga_pad_copy_f32(const float* src, float* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd, int accumulate,
                               ga_stream_t stream) {
    return ga_pad_copy(src, dst, rows, cols, lds, ldd, accumulate, GA_F32, stream);
}
