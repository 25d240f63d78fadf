// Weight preparation / gradient un-folding and small fp32 utility kernels (gfx950).
// Per optimizer step the fp32 master weights are re-laid-out once into the "effective" operand copies the
// GEMMs read (dtype = bf16 or fp32): k-order (ky,kx,ci), LayerNorm scale / LayerScale gamma folded in,
// transposed copy for the data-gradient product.  The inverse maps the effective-weight gradients back.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ long w_index(int n, int kcol, int Ci, int KH, int KW, int stem, int* ci_out) {
    const int KK = Ci * KH * KW;
    if (stem || (KH == 1 && KW == 1)) {
        *ci_out = stem ? kcol / (KH * KW) : kcol;
        return (long)n * KK + kcol;
    }
    const int tap = kcol / Ci, ci = kcol - tap * Ci;
    *ci_out = ci;
    return ((long)n * Ci + ci) * (KH * KW) + tap;
}

template <typename T>
__global__ __launch_bounds__(256) void wprep_out_kernel(const float* __restrict__ w, const float* __restrict__ rs,
                                                        const float* __restrict__ cs, const int* __restrict__ perm,
                                                        T* __restrict__ out, long ldo, int rows, int Ci, int KH, int KW,
                                                        int stem) {
    const int KK = Ci * KH * KW;
    const long total = (long)rows * ldo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int no = (int)(i / ldo), kcol = (int)(i - (long)no * ldo);
        const int n = perm ? perm[no] : no;   // source row
        float v = 0.f;
        if (kcol < KK) {
            int ci;
            v = w[w_index(n, kcol, Ci, KH, KW, stem, &ci)];
            if (rs) v *= rs[n];
            if (cs) v *= cs[ci];
        }
        elt<T>::st(out + i, v);
    }
}

// outT: non-flip [G][KK][ldt] (co contiguous); flip [G][Ci][ldt] with column ((KH-1-ky)*KW + KW-1-kx)*Co + co
template <typename T>
__global__ __launch_bounds__(256) void wprep_outT_kernel(const float* __restrict__ w, const float* __restrict__ rs,
                                                         const float* __restrict__ cs, const int* __restrict__ perm,
                                                         T* __restrict__ outT, long ldt, int G, int Co, int Ci, int KH,
                                                         int KW, int stem, int flip) {
    const int KK = Ci * KH * KW;
    const int trows = flip ? Ci : KK;
    const long total = (long)G * trows * ldt;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % ldt);
        const long t = i / ldt;
        const int r = (int)(t % trows), g = (int)(t / trows);
        float v = 0.f;
        if (!flip) {
            if (col < Co) {
                const int n = perm ? perm[g * Co + col] : g * Co + col;
                int ci;
                v = w[w_index(n, r, Ci, KH, KW, stem, &ci)];
                if (rs) v *= rs[n];
                if (cs) v *= cs[ci];
            }
        } else {
            if (col < KH * KW * Co) {
                const int tapf = col / Co, co = col - tapf * Co;
                const int tap = KH * KW - 1 - tapf;  // (KH-1-ky, KW-1-kx) <-> flipped linear tap index
                const int n = perm ? perm[g * Co + co] : g * Co + co;
                v = w[((long)n * Ci + r) * (KH * KW) + tap];
                if (rs) v *= rs[n];
                if (cs) v *= cs[r];
            }
        }
        elt<T>::st(outT + i, v);
    }
}

// be[n] = rs[n] * (b[n] + sum_c W[n][c] v[c]); one wave per n
__global__ __launch_bounds__(256) void bias_fold_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                        const float* __restrict__ rs, const float* __restrict__ v,
                                                        const int* __restrict__ perm, float* __restrict__ be, int N,
                                                        int C) {
    const int no = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (no >= N) return;
    const int n = perm ? perm[no] : no;
    float s = 0.f;
    if (v)
        for (int c = lane; c < C; c += 64) s += W[(long)n * C + c] * v[c];
    s = wave_sum(s);
    if (lane == 0) be[no] = (rs ? rs[n] : 1.f) * ((b ? b[n] : 0.f) + s);
}

// dW[orig layout] += rs[n] * G[n][kcol] * cs[ci]
__global__ __launch_bounds__(256) void unfold_dw_kernel(const float* __restrict__ G, long ldg,
                                                        const float* __restrict__ rs, const float* __restrict__ cs,
                                                        const float* __restrict__ gb, const float* __restrict__ v,
                                                        const int* __restrict__ perm, float* __restrict__ dW, int N,
                                                        int Ci, int KH, int KW, int stem) {
    const int KK = Ci * KH * KW;
    const long total = (long)N * KK;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int no = (int)(i / KK), kcol = (int)(i - (long)no * KK);
        const int n = perm ? perm[no] : no;   // master-weight row of effective row `no`
        int ci;
        const long wi = w_index(n, kcol, Ci, KH, KW, stem, &ci);
        float val = G[(long)no * ldg + kcol];
        if (cs) val *= cs[ci];
        if (gb && v) val += gb[no] * v[ci];   // bias-fold path: be[n] = rs[n] * (b[n] + sum_c W[n][c] v[c])
        if (rs) val *= rs[n];
        dW[wi] += val;
    }
}

// per-row: d_rs[n] += sum_k G*W*cs + gb[n]*b[n];  db[n] += rs[n]*gb[n]      (one wave per n)
__global__ __launch_bounds__(256) void unfold_rows_kernel(const float* __restrict__ G, long ldg,
                                                          const float* __restrict__ gb, const float* __restrict__ W,
                                                          const float* __restrict__ b, const float* __restrict__ rs,
                                                          const float* __restrict__ cs, const float* __restrict__ v,
                                                          const int* __restrict__ perm, float* __restrict__ d_rs,
                                                          float* __restrict__ db, int N, int Ci, int KH, int KW,
                                                          int stem) {
    const int no = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (no >= N) return;
    const int n = perm ? perm[no] : no;
    const int KK = Ci * KH * KW;
    if (d_rs) {
        float s = 0.f;
        for (int k = lane; k < KK; k += 64) {
            int ci;
            const long wi = w_index(n, k, Ci, KH, KW, stem, &ci);
            s += G[(long)no * ldg + k] * W[wi] * (cs ? cs[ci] : 1.f);
            if (gb && v) s += gb[no] * W[wi] * v[ci];
        }
        s = wave_sum(s);
        if (lane == 0) d_rs[n] += s + ((gb && b) ? gb[no] * b[n] : 0.f);
    }
    if (db && gb && lane == 0) db[n] += (rs ? rs[n] : 1.f) * gb[no];
}

// per-column (KH=KW=1): d_cs[c] += sum_n rs[n] G[n][c] W[n][c];  d_v[c] += sum_n gb[n] rs[n] W[n][c]
// grid (C/64, N/64): 64 columns x 4 row lanes per workgroup, 16 rows per thread, fp32 atomics
__global__ __launch_bounds__(256) void unfold_cols_kernel(const float* __restrict__ G, long ldg,
                                                          const float* __restrict__ gb, const float* __restrict__ W,
                                                          const float* __restrict__ rs, float* __restrict__ d_cs,
                                                          float* __restrict__ d_v, int N, int C) {
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float a = 0.f, bsum = 0.f;
    if (c < C) {
        for (int i = 0; i < 16; ++i) {
            const int n = blockIdx.y * 64 + rl * 16 + i;
            if (n < N) {
                const float wv = W[(long)n * C + c] * (rs ? rs[n] : 1.f);
                a += G[(long)n * ldg + c] * wv;
                if (gb) bsum += gb[n] * wv;
            }
        }
    }
    red[0][rl][threadIdx.x & 63] = a;
    red[1][rl][threadIdx.x & 63] = bsum;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        const float s0 = red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
        const float s1 = red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
        if (d_cs) atomicAdd(d_cs + c, s0);
        if (d_v) atomicAdd(d_v + c, s1);
    }
}

// out[c][r] (+)= in[r][c]   (fp32; dwconv weight [C][49] <-> [49][C])
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int R, int C, int accumulate) {
    const long total = (long)R * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / R), r = (int)(i - (long)c * R);  // out index i = c*R + r
        const float v = in[(long)r * C + c];
        out[i] = accumulate ? out[i] + v : v;
    }
}

// y += w * (x - y)   (model EMA: ema = decay * ema + (1 - decay) * param, w = 1 - decay)
__global__ __launch_bounds__(256) void lerp_f32_kernel(float* __restrict__ y, const float* __restrict__ x, float w, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = fmaf(w, x[i] - y[i], y[i]);
}

__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ y, const float* __restrict__ x, float a,
                                                       long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}

// y[row][:] = x[row][:] * s[row / rows_per_scale]   (DropPath mask on a gradient), 8 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void rowscale_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                       T* __restrict__ y, long n8, long elems_per_scale) {
    // a pure streaming pass (every byte read once, written once): U chunks of 16 bytes in flight per lane, non-temporal both
    // ways, one scale lookup per chunk through a reciprocal (the 32-bit division cost ~20 VALU slots per chunk)
    constexpr int U = 4;
    constexpr int E = 16 / (int)sizeof(T);                    // elements per 16-byte chunk
    const unsigned epsc = (unsigned)(elems_per_scale / E);    // chunks per scale
    const float inv = 1.0f / (float)epsc;
    const unsigned nc = (unsigned)(n8 * 8 / E);
    const unsigned stride = gridDim.x * 256u;
    auto scale_of = [&](unsigned i) {
        unsigned q = (unsigned)((float)i * inv);              // exact up to 2^24 chunks per scale table entry boundary: fix up
        const long r = (long)i - (long)q * epsc;
        if (r < 0) --q;
        else if (r >= (long)epsc) ++q;
        return s[q];
    };
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    for (; (unsigned long long)i + (U - 1) * (unsigned long long)stride < nc; i += U * stride) {
        uint4 v[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load16_nt(x + (long)(i + u * stride) * E);
#pragma unroll
        for (int u = 0; u < U; ++u) sc[u] = scale_of(i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (sizeof(T) == 2) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= sc[u];
                store8_nt(y + (long)(i + u * stride) * E, f);
            } else {
                const uint4 o = make_uint4(__float_as_uint(__uint_as_float(v[u].x) * sc[u]), __float_as_uint(__uint_as_float(v[u].y) * sc[u]),
                                           __float_as_uint(__uint_as_float(v[u].z) * sc[u]), __float_as_uint(__uint_as_float(v[u].w) * sc[u]));
                store16_nt(y + (long)(i + u * stride) * E, o);
            }
        }
    }
    for (; i < nc; i += stride) {
        const float sc = scale_of(i);
        const uint4 v = load16_nt(x + (long)i * E);
        if constexpr (sizeof(T) == 2) {
            float f[8];
            unpack8(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] *= sc;
            store8_nt(y + (long)i * E, f);
        } else {
            store16_nt(y + (long)i * E, make_uint4(__float_as_uint(__uint_as_float(v.x) * sc), __float_as_uint(__uint_as_float(v.y) * sc),
                                                   __float_as_uint(__uint_as_float(v.z) * sc), __float_as_uint(__uint_as_float(v.w) * sc)));
        }
    }
}

// dst (T) = cast(src fp32), contiguous
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) elt<T>::st(dst + i, src[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void uncast_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = elt<T>::ld(src + i);
}


// ------------------------------------------------------------------------------------------------
// Batched variants: ONE launch walks a device-resident array of job descriptors (blockIdx.y = job).  A training
// step has ~120 weight-prep, ~90 un-fold and ~60 tiny axpy/transpose jobs of a few KB..MB each; as separate
// launches they cost ~3 ms of pure launch latency per step.
// ------------------------------------------------------------------------------------------------
// 1x1 weights (the bulk of the parameters): 64 x 64 tiles, the master rows are read ONCE with 16-byte loads, the
// row-major copy is written straight from registers and the transposed copy through a padded LDS tile, both with
// 4-element stores; padding columns (ldo > K, ldt > Co) are written as zeros.  The scalar path below reads the
// transposed copy's source with a stride of K floats and pays a 64-bit division per element.
template <typename T>
__device__ __forceinline__ void store4z(T* p, const float v[4]) { store4(p, v); }

template <typename T>
__device__ void wprep_job_1x1(const ga_wprep_desc& d, int nblk, int blk, float (*tile)[65]) {
    const int KK = d.Ci, Co = d.Co;
    T* out = reinterpret_cast<T*>(d.out);
    T* outT = reinterpret_cast<T*>(d.outT);
    const int tcols = d.t_cols > 0 ? d.t_cols : (int)d.ldt;     // columns of an outT row this job owns
    const int n_ext = outT ? max(Co, tcols) : Co, k_ext = out ? max(KK, (int)d.ldo) : KK;
    const int tn = (n_ext + 63) >> 6, tk = (k_ext + 63) >> 6;
    const int ntiles = d.G * tn * tk;
    const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
    for (int t = blk; t < ntiles; t += nblk) {
        const int g = t / (tn * tk), r = t - g * tn * tk;
        const int n0 = (r / tk) << 6, k0 = (r % tk) << 6;
        const int kcol = k0 + tc * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int no = n0 + tr + 16 * i;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (no < Co && kcol < KK) {
                const int n = d.row_perm ? d.row_perm[g * Co + no] : g * Co + no;
                load4(d.w + (long)n * KK + kcol, v);
                const float rs = d.rs ? d.rs[n] : 1.f;
                float cs[4] = {1.f, 1.f, 1.f, 1.f};
                if (d.cs) load4(d.cs + kcol, cs);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= rs * cs[j];
            }
            if (out && no < Co && kcol < d.ldo) store4(out + ((long)g * Co + no) * d.ldo + kcol, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[tc * 4 + j][tr + 16 * i] = v[j];
        }
        __syncthreads();
        if (outT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + tr + 16 * i, nc = n0 + tc * 4;
                if (k < KK && nc < tcols) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = tile[tr + 16 * i][tc * 4 + j];
                    store4(outT + ((long)g * KK + k) * d.ldt + nc, v);
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ bool wprep_is_1x1(const ga_wprep_desc& d) {
    return d.KH == 1 && d.KW == 1 && !d.stem && !d.flip && d.Ci % 4 == 0 && (reinterpret_cast<uintptr_t>(d.w) & 15) == 0 &&
           (!d.cs || (reinterpret_cast<uintptr_t>(d.cs) & 15) == 0) && (!d.out || d.ldo % 4 == 0) &&
           (!d.outT || (d.ldt % 4 == 0 && d.t_cols % 4 == 0 && (reinterpret_cast<uintptr_t>(d.outT) & 7) == 0));
}

template <typename T>
__device__ void wprep_job(const ga_wprep_desc& d, int nblk, int blk) {
    const int KK = d.Ci * d.KH * d.KW;
    if (d.out) {
        T* out = reinterpret_cast<T*>(d.out);
        const long total = (long)d.G * d.Co * d.ldo;
        for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
            const int no = (int)(i / d.ldo), kcol = (int)(i - (long)no * d.ldo);
            const int n = d.row_perm ? d.row_perm[no] : no;
            float v = 0.f;
            if (kcol < KK) {
                int ci;
                v = d.w[w_index(n, kcol, d.Ci, d.KH, d.KW, d.stem, &ci)];
                if (d.rs) v *= d.rs[n];
                if (d.cs) v *= d.cs[ci];
            }
            elt<T>::st(out + i, v);
        }
    }
    if (d.outT) {
        T* outT = reinterpret_cast<T*>(d.outT);
        const int trows = d.flip ? d.Ci : KK;
        const long tcols = d.t_cols > 0 ? d.t_cols : d.ldt;
        const long total = (long)d.G * trows * tcols;
        for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
            const int col = (int)(i % tcols);
            const long t = i / tcols;
            const int r = (int)(t % trows), g = (int)(t / trows);
            float v = 0.f;
            if (!d.flip) {
                if (col < d.Co) {
                    const int n = d.row_perm ? d.row_perm[g * d.Co + col] : g * d.Co + col;
                    int ci;
                    v = d.w[w_index(n, r, d.Ci, d.KH, d.KW, d.stem, &ci)];
                    if (d.rs) v *= d.rs[n];
                    if (d.cs) v *= d.cs[ci];
                }
            } else if (col < d.KH * d.KW * d.Co) {
                const int tapf = col / d.Co, co = col - tapf * d.Co;
                const int tap = d.KH * d.KW - 1 - tapf;
                const int n = d.row_perm ? d.row_perm[g * d.Co + co] : g * d.Co + co;
                v = d.w[((long)n * d.Ci + r) * (d.KH * d.KW) + tap];
                if (d.rs) v *= d.rs[n];
                if (d.cs) v *= d.cs[r];
            }
            elt<T>::st(outT + t * d.ldt + col, v);
        }
    }
}

__global__ __launch_bounds__(256) void wprep_batch_kernel(const ga_wprep_desc* __restrict__ jobs) {
    __shared__ float tile[64][65];
    const ga_wprep_desc d = jobs[blockIdx.y];
    if (wprep_is_1x1(d)) {
        if (d.dtype == GA_BF16) wprep_job_1x1<bf16_t>(d, gridDim.x, blockIdx.x, tile);
        else wprep_job_1x1<float>(d, gridDim.x, blockIdx.x, tile);
        return;
    }
    if (d.dtype == GA_BF16) wprep_job<bf16_t>(d, gridDim.x, blockIdx.x);
    else wprep_job<float>(d, gridDim.x, blockIdx.x);
}

// small fp32 jobs: kind 0 axpy (y += a*x), 1 transpose (out[c][r] (+)= in[r][c]), 2 bias fold
__global__ __launch_bounds__(256) void small_batch_kernel(const ga_small_desc* __restrict__ jobs) {
    const ga_small_desc d = jobs[blockIdx.y];
    if (d.kind == 0) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long)gridDim.x * 256) d.y[i] += d.a * d.x[i];
    } else if (d.kind == 1) {
        const long total = (long)d.R * d.C;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)(i / d.R), r = (int)(i - (long)c * d.R);
            const float v = d.x[(long)r * d.C + c];
            d.y[i] = d.accumulate ? d.y[i] + v : v;
        }
    } else {   // bias fold: y[n] = rs[m]*(b[m] + sum_c W[m][c] v[c]), one wave per row
        const int lane = threadIdx.x & 63;
        for (int no = blockIdx.x * 4 + (threadIdx.x >> 6); no < d.R; no += gridDim.x * 4) {
            const int n = d.row_perm ? d.row_perm[no] : no;
            float s = 0.f;
            if (d.v) {
                if (d.C % 4 == 0 && ((reinterpret_cast<uintptr_t>(d.x) | reinterpret_cast<uintptr_t>(d.v)) & 15) == 0) {
                    for (int c = lane * 4; c < d.C; c += 256) {
                        float w[4], vv[4];
                        load4(d.x + (long)n * d.C + c, w);
                        load4(d.v + c, vv);
                        s += w[0] * vv[0] + w[1] * vv[1] + w[2] * vv[2] + w[3] * vv[3];
                    }
                } else {
                    for (int c = lane; c < d.C; c += 64) s += d.x[(long)n * d.C + c] * d.v[c];
                }
            }
            s = wave_sum(s);
            if (lane == 0) d.y[no] = (d.rs ? d.rs[n] : 1.f) * ((d.b ? d.b[n] : 0.f) + s);
        }
    }
}

__global__ __launch_bounds__(256) void unfold_batch_kernel(const ga_wunfold_desc* __restrict__ jobs) {
    const ga_wunfold_desc d = jobs[blockIdx.y];
    const int KK = d.Ci * d.KH * d.KW;
    const bool vec4 = d.KH == 1 && d.KW == 1 && !d.stem && KK % 4 == 0 && d.ldg % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(d.G) | reinterpret_cast<uintptr_t>(d.dW) |
                        reinterpret_cast<uintptr_t>(d.cs) | reinterpret_cast<uintptr_t>(d.v)) & 15) == 0;
    if (d.dW && vec4) {   // 1x1 weights: 4 columns per thread, one 32-bit division per 16 bytes
        const unsigned P = (unsigned)KK >> 2, total = (unsigned)d.N * P;
        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
            const unsigned no = i / P, k4 = (i - no * P) << 2;
            const int n = d.row_perm ? d.row_perm[no] : (int)no;
            float g[4], w[4];
            load4(d.G + (long)no * d.ldg + k4, g);
            if (d.cs) {
                float cs[4];
                load4(d.cs + k4, cs);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] *= cs[j];
            }
            if (d.gb && d.v) {
                float vv[4];
                load4(d.v + k4, vv);
                const float gbn = d.gb[no];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = fmaf(gbn, vv[j], g[j]);
            }
            const float rs = d.rs ? d.rs[n] : 1.f;
            float* dst = d.dW + (long)n * KK + k4;
            load4(dst, w);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = fmaf(g[j], rs, w[j]);
            store4(dst, w);
        }
    } else if (d.dW) {
        const long total = (long)d.N * KK;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int no = (int)(i / KK), kcol = (int)(i - (long)no * KK);
            const int n = d.row_perm ? d.row_perm[no] : no;
            int ci;
            const long wi = w_index(n, kcol, d.Ci, d.KH, d.KW, d.stem, &ci);
            float val = d.G[(long)no * d.ldg + kcol];
            if (d.cs) val *= d.cs[ci];
            if (d.gb && d.v) val += d.gb[no] * d.v[ci];
            if (d.rs) val *= d.rs[n];
            d.dW[wi] += val;
        }
    }
    if (d.d_rs || (d.db && d.gb)) {   // per-row part, one wave per row
        const int lane = threadIdx.x & 63;
        for (int no = blockIdx.x * 4 + (threadIdx.x >> 6); no < d.N; no += gridDim.x * 4) {
            const int n = d.row_perm ? d.row_perm[no] : no;
            if (d.d_rs) {
                float s = 0.f;
                for (int k = lane; k < KK; k += 64) {
                    int ci;
                    const long wi = w_index(n, k, d.Ci, d.KH, d.KW, d.stem, &ci);
                    s += d.G[(long)no * d.ldg + k] * d.W[wi] * (d.cs ? d.cs[ci] : 1.f);
                    if (d.gb && d.v) s += d.gb[no] * d.W[wi] * d.v[ci];
                }
                s = wave_sum(s);
                if (lane == 0) d.d_rs[n] += s + ((d.gb && d.b) ? d.gb[no] * d.b[n] : 0.f);
            }
            if (d.db && d.gb && lane == 0) d.db[n] += (d.rs ? d.rs[n] : 1.f) * d.gb[no];
        }
    }
    if (d.d_cs || d.d_v) {            // per-column part (1x1 weights): thread = column, block = slice of rows
        const int rows_per_blk = (d.N + gridDim.x - 1) / gridDim.x;
        const int r0 = blockIdx.x * rows_per_blk, r1 = min(d.N, r0 + rows_per_blk);
        for (int c = threadIdx.x; c < d.Ci; c += 256) {
            float a = 0.f, bs = 0.f;
            for (int n = r0; n < r1; ++n) {
                const float wv = d.W[(long)n * d.Ci + c] * (d.rs ? d.rs[n] : 1.f);
                a += d.G[(long)n * d.ldg + c] * wv;
                if (d.gb) bs += d.gb[n] * wv;
            }
            if (r1 > r0) {
                if (d.d_cs) atomicAdd(d.d_cs + c, a);
                if (d.d_v) atomicAdd(d.d_v + c, bs);
            }
        }
    }
}

int nblocks(long n, int cap = 4096) { return (int)std::max<long>(1, std::min<long>(cap, (n + 255) / 256)); }

}  // namespace

extern "C" int ga_weight_prep(const ga_wprep_desc* d, ga_stream_t stream) {
    GA_REQUIRE(d && d->w && d->G >= 1 && d->Co >= 1 && d->Ci >= 1 && d->KH >= 1 && d->KW >= 1, "ga_weight_prep: bad args");
    GA_REQUIRE(!(d->cs && d->G > 1), "ga_weight_prep: column scale with groups unsupported");
    GA_REQUIRE(!(d->flip && d->stem), "ga_weight_prep: flip+stem unsupported");
    GA_REQUIRE(d->t_cols == 0 || d->t_cols == d->ldt, "ga_weight_prep: t_cols != ldt needs ga_weight_prep_batch");
    const int KK = d->Ci * d->KH * d->KW;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->out) {
        GA_REQUIRE(d->ldo >= KK, "ga_weight_prep: ldo < K");
        const long total = (long)d->G * d->Co * d->ldo;
        if (d->dtype == GA_BF16)
            hipLaunchKernelGGL(wprep_out_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, s, d->w, d->rs, d->cs,
                               d->row_perm, (bf16_t*)d->out, (long)d->ldo, d->G * d->Co, d->Ci, d->KH, d->KW, d->stem);
        else
            hipLaunchKernelGGL(wprep_out_kernel<float>, dim3(nblocks(total)), dim3(256), 0, s, d->w, d->rs, d->cs,
                               d->row_perm, (float*)d->out, (long)d->ldo, d->G * d->Co, d->Ci, d->KH, d->KW, d->stem);
    }
    if (d->outT) {
        GA_REQUIRE(d->ldt >= (d->flip ? d->KH * d->KW * d->Co : d->Co), "ga_weight_prep: ldt too small");
        const long total = (long)d->G * (d->flip ? d->Ci : KK) * d->ldt;
        if (d->dtype == GA_BF16)
            hipLaunchKernelGGL(wprep_outT_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, s, d->w, d->rs, d->cs,
                               d->row_perm, (bf16_t*)d->outT, (long)d->ldt, d->G, d->Co, d->Ci, d->KH, d->KW, d->stem, d->flip);
        else
            hipLaunchKernelGGL(wprep_outT_kernel<float>, dim3(nblocks(total)), dim3(256), 0, s, d->w, d->rs, d->cs,
                               d->row_perm, (float*)d->outT, (long)d->ldt, d->G, d->Co, d->Ci, d->KH, d->KW, d->stem, d->flip);
    }
    return ga_check_launch("ga_weight_prep");
}

extern "C" int ga_bias_fold(const float* W, const float* b, const float* rs, const float* v, const int* row_perm,
                            float* be, int N, int C, ga_stream_t stream) {
    GA_REQUIRE(be && N > 0 && (!v || W), "ga_bias_fold: bad args");
    hipLaunchKernelGGL(bias_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, b, rs,
                       v, row_perm, be, N, C);
    return ga_check_launch("ga_bias_fold");
}

extern "C" int ga_weight_unfold(const ga_wunfold_desc* d, ga_stream_t stream) {
    GA_REQUIRE(d && d->G && d->N > 0 && d->Ci > 0, "ga_weight_unfold: bad args");
    GA_REQUIRE(!((d->d_cs || d->d_v || d->v) && (d->KH != 1 || d->KW != 1)),
               "ga_weight_unfold: column grads / bias-fold vector need a 1x1 kernel");
    GA_REQUIRE(!((d->d_rs || d->d_cs || d->d_v) && !d->W), "ga_weight_unfold: W required");
    GA_REQUIRE(!(d->row_perm && (d->d_cs || d->d_v)), "ga_weight_unfold: row_perm with column grads unsupported");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int KK = d->Ci * d->KH * d->KW;
    if (d->dW)
        hipLaunchKernelGGL(unfold_dw_kernel, dim3(nblocks((long)d->N * KK)), dim3(256), 0, s, d->G, (long)d->ldg, d->rs,
                           d->cs, d->gb, d->v, d->row_perm, d->dW, d->N, d->Ci, d->KH, d->KW, d->stem);
    if (d->d_rs || (d->db && d->gb))
        hipLaunchKernelGGL(unfold_rows_kernel, dim3(cdiv(d->N, 4)), dim3(256), 0, s, d->G, (long)d->ldg, d->gb, d->W,
                           d->b, d->rs, d->cs, d->v, d->row_perm, d->d_rs, d->db, d->N, d->Ci, d->KH, d->KW, d->stem);
    if (d->d_cs || d->d_v)
        hipLaunchKernelGGL(unfold_cols_kernel, dim3(cdiv(d->Ci, 64), cdiv(d->N, 64)), dim3(256), 0, s, d->G,
                           (long)d->ldg, d->gb, d->W, d->rs, d->d_cs, d->d_v, d->N, d->Ci);
    return ga_check_launch("ga_weight_unfold");
}

extern "C" int ga_transpose_f32(const float* in, float* out, int R, int C, int accumulate, ga_stream_t stream) {
    GA_REQUIRE(in && out && R > 0 && C > 0, "ga_transpose_f32: bad args");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3(nblocks((long)R * C)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in, out, R, C, accumulate);
    return ga_check_launch("ga_transpose_f32");
}

extern "C" int ga_axpy_f32(float* y, const float* x, float a, int64_t n, ga_stream_t stream) {
    GA_REQUIRE(y && x && n > 0, "ga_axpy_f32: bad args");
    hipLaunchKernelGGL(axpy_f32_kernel, dim3(nblocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, x, a,
                       (long)n);
    return ga_check_launch("ga_axpy_f32");
}

extern "C" int ga_lerp_f32(float* y, const float* x, float w, int64_t n, ga_stream_t stream) {
    GA_REQUIRE(y && x && n > 0, "ga_lerp_f32: bad args");
    hipLaunchKernelGGL(lerp_f32_kernel, dim3(nblocks(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, x, w,
                       (long)n);
    return ga_check_launch("ga_lerp_f32");
}

extern "C" int ga_rowscale(const void* x, const float* s, void* y, int64_t n, int64_t elems_per_scale, int dtype,
                           ga_stream_t stream) {
    GA_REQUIRE(x && s && y && n > 0 && n % 8 == 0 && elems_per_scale % 8 == 0, "ga_rowscale: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(rowscale_kernel<bf16_t>, dim3(nblocks(n / 8 / 4, 2048)), dim3(256), 0, st, (const bf16_t*)x, s,
                           (bf16_t*)y, (long)(n / 8), (long)elems_per_scale);
    else
        hipLaunchKernelGGL(rowscale_kernel<float>, dim3(nblocks(n / 8 / 2, 2048)), dim3(256), 0, st, (const float*)x, s,
                           (float*)y, (long)(n / 8), (long)elems_per_scale);
    return ga_check_launch("ga_rowscale");
}

extern "C" int ga_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, ga_stream_t stream) {
    GA_REQUIRE(src && dst && n > 0, "ga_cast_from_f32: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, src, (bf16_t*)dst, (long)n);
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, src, (float*)dst, (long)n);
    return ga_check_launch("ga_cast_from_f32");
}

extern "C" int ga_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, ga_stream_t stream) {
    GA_REQUIRE(src && dst && n > 0, "ga_cast_to_f32: bad args");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(uncast_kernel<bf16_t>, dim3(nblocks(n)), dim3(256), 0, st, (const bf16_t*)src, dst, (long)n);
    else
        hipLaunchKernelGGL(uncast_kernel<float>, dim3(nblocks(n)), dim3(256), 0, st, (const float*)src, dst, (long)n);
    return ga_check_launch("ga_cast_to_f32");
}

extern "C" int ga_weight_prep_batch(const ga_wprep_desc* jobs_dev, int n, ga_stream_t stream) {
    GA_REQUIRE(jobs_dev && n > 0, "ga_weight_prep_batch: bad args");
    // grid.x workgroups walk one job with a grid stride: the few big jobs of a batch (fc1 / fc2 of stage 3: 2.4 M elements)
    // set its duration, so give every job enough workgroups to fill the chip on its own
    const int gx = std::max(1, GA_KNOB("BATCH_GX", 256));
    hipLaunchKernelGGL(wprep_batch_kernel, dim3(gx, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev);
    return ga_check_launch("ga_weight_prep_batch");
}

extern "C" int ga_small_batch(const ga_small_desc* jobs_dev, int n, ga_stream_t stream) {
    GA_REQUIRE(jobs_dev && n > 0, "ga_small_batch: bad args");
    hipLaunchKernelGGL(small_batch_kernel, dim3(48, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev);
    return ga_check_launch("ga_small_batch");
}

extern "C" int ga_weight_unfold_batch(const ga_wunfold_desc* jobs_dev, int n, ga_stream_t stream) {
    GA_REQUIRE(jobs_dev && n > 0, "ga_weight_unfold_batch: bad args");
    const int gx = std::max(1, GA_KNOB("BATCH_GX", 256));
    hipLaunchKernelGGL(unfold_batch_kernel, dim3(gx, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs_dev);
    return ga_check_launch("ga_weight_unfold_batch");
}
