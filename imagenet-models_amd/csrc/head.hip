// GA head kernels for gfx950 (all small / HBM- or latency-bound, no MFMA here; the GEMM-shaped parts of the
// head go through ga_gemm / ga_wgrad):
//   multi-scale aggregate (avg-pool / copy / bilinear x2 into the channel-concat buffer) + backward,
//   squeeze-excite (pool, per-sample MLP, gate) + backward,
//   Gram triu-pack + L2 normalise + backward (produces the symmetric matrix for the dX product),
//   single-query class attention + backward, token concat / split.
#include <algorithm>
#include "common.h"

namespace {

int nblk(long n, int per = 256, int cap = 8192) { return (int)std::max<long>(1, std::min<long>(cap, (n + per - 1) / per)); }

// ------------------------------------------------------------------------------------------------
// aggregate: dst[b, oy, ox, c_off + c] = pool(src)[b, oy, ox, c]      (4 channels per thread)
//   mode 0: f x f average (f = Hin / Hout; f == 1 -> copy);  mode 1: bilinear x2 (align_corners = False)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_coef(int o, int n_in, int& i0, int& i1, float& l) {
    float s = 0.5f * (o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l = s - (float)i0;
}

// E channels per thread (8 = 16-byte accesses for bf16 when C, ldd and c_off allow, else 4); 32-bit index math
template <int E> __device__ __forceinline__ void ldE(const float* p, float v[E]) { if constexpr (E == 8) load8(p, v); else load4(p, v); }
template <int E> __device__ __forceinline__ void ldE(const bf16_t* p, float v[E]) { if constexpr (E == 8) load8(p, v); else load4(p, v); }
template <int E> __device__ __forceinline__ void stE(float* p, const float v[E]) { if constexpr (E == 8) store8(p, v); else store4(p, v); }
template <int E> __device__ __forceinline__ void stE(bf16_t* p, const float v[E]) { if constexpr (E == 8) store8(p, v); else store4(p, v); }

template <typename T, int E>
__global__ __launch_bounds__(256) void pool_concat_kernel(const T* __restrict__ src, T* __restrict__ dst, int B,
                                                          int Hin, int Win, int C, int Hout, int Wout, int ldd,
                                                          int c_off, int mode) {
    const unsigned cgs = (unsigned)C / E;
    const unsigned total = (unsigned)B * Hout * Wout * cgs;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cg = i % cgs;
        unsigned p = i / cgs;
        const int ox = (int)(p % (unsigned)Wout); p /= (unsigned)Wout;
        const int oy = (int)(p % (unsigned)Hout);
        const long b = p / (unsigned)Hout;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        const T* sb = src + b * Hin * Win * C + cg * E;
        if (mode == 0) {
            const int f = Hin / Hout;
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) {
                    float v[E];
                    ldE<E>(sb + ((long)(oy * f + dy) * Win + ox * f + dx) * C, v);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += v[e];
                }
            const float inv = 1.f / (float)(f * f);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] *= inv;
        } else if (mode == 2) {          // bilinear reduction by an even factor f: the mean of the central 2 x 2 of each f x f block
            const int f = Hin / Hout, o = f / 2 - 1;
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    float v[E];
                    ldE<E>(sb + ((long)(oy * f + o + dy) * Win + ox * f + o + dx) * C, v);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += 0.25f * v[e];
                }
        } else if (mode == 3) {          // adaptive_avg_pool2d to twice the size: every input pixel replicated 2 x 2
            ldE<E>(sb + ((long)(oy >> 1) * Win + (ox >> 1)) * C, acc);
        } else {
            int y0, y1, x0, x1;
            float ly, lx;
            bil_coef(oy, Hin, y0, y1, ly);
            bil_coef(ox, Win, x0, x1, lx);
            float a[E], bb[E], c[E], d[E];
            ldE<E>(sb + ((long)y0 * Win + x0) * C, a);
            ldE<E>(sb + ((long)y0 * Win + x1) * C, bb);
            ldE<E>(sb + ((long)y1 * Win + x0) * C, c);
            ldE<E>(sb + ((long)y1 * Win + x1) * C, d);
#pragma unroll
            for (int e = 0; e < E; ++e)
                acc[e] = (1.f - ly) * ((1.f - lx) * a[e] + lx * bb[e]) + ly * ((1.f - lx) * c[e] + lx * d[e]);
        }
        stE<E>(dst + ((b * Hout + oy) * Wout + ox) * (long)ldd + c_off + cg * E, acc);
    }
}

// backward: dsrc[b, iy, ix, c] = (dres) + sum over the outputs that read this input pixel
template <typename T, int E>
__global__ __launch_bounds__(256) void pool_concat_bwd_kernel(const T* __restrict__ dcat, const T* __restrict__ dres,
                                                              T* __restrict__ dsrc, int B, int Hin, int Win, int C,
                                                              int Hout, int Wout, int ldd, int c_off, int mode) {
    const unsigned cgs = (unsigned)C / E;
    const unsigned total = (unsigned)B * Hin * Win * cgs;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cg = i % cgs;
        unsigned p = i / cgs;
        const int ix = (int)(p % (unsigned)Win); p /= (unsigned)Win;
        const int iy = (int)(p % (unsigned)Hin);
        const long b = p / (unsigned)Hin;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        const T* db = dcat + b * Hout * Wout * ldd + c_off + cg * E;
        if (mode == 0) {
            const int f = Hin / Hout;
            float v[E];
            ldE<E>(db + ((long)(iy / f) * Wout + ix / f) * ldd, v);
            const float inv = 1.f / (float)(f * f);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = v[e] * inv;
        } else if (mode == 2) {
            const int f = Hin / Hout, o = f / 2 - 1;
            const int ry = iy % f - o, rx = ix % f - o;
            if ((unsigned)ry < 2u && (unsigned)rx < 2u) {
                float v[E];
                ldE<E>(db + ((long)(iy / f) * Wout + ix / f) * ldd, v);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = 0.25f * v[e];
            }
        } else if (mode == 3) {
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    float v[E];
                    ldE<E>(db + ((long)(2 * iy + dy) * Wout + 2 * ix + dx) * ldd, v);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += v[e];
                }
        } else {
            for (int oy = max(0, 2 * iy - 2); oy <= min(Hout - 1, 2 * iy + 3); ++oy) {
                int y0, y1;
                float ly;
                bil_coef(oy, Hin, y0, y1, ly);
                const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
                if (wy == 0.f) continue;
                for (int ox = max(0, 2 * ix - 2); ox <= min(Wout - 1, 2 * ix + 3); ++ox) {
                    int x0, x1;
                    float lx;
                    bil_coef(ox, Win, x0, x1, lx);
                    const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                    if (wx == 0.f) continue;
                    float v[E];
                    ldE<E>(db + ((long)oy * Wout + ox) * ldd, v);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += wy * wx * v[e];
                }
            }
        }
        const long off = ((b * Hin + iy) * Win + ix) * (long)C + cg * E;
        if (dres) {
            float r[E];
            ldE<E>(dres + off, r);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += r[e];
        }
        stE<E>(dsrc + off, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// per-(b,c) spatial reductions:  out[b][c] = scale * sum_hw a[b,hw,c] (* b2[b,hw,c])
// one workgroup per (b, 64-channel slice); thread = (4 channels, 16 row slots)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void spatial_sum_kernel(const T* __restrict__ a, const T* __restrict__ b2,
                                                          float* __restrict__ out, int HW, int C, float scale) {
    __shared__ float red[16][64];
    const long b = blockIdx.x;
    const int c0 = blockIdx.y * 64;
    const int cgs = min(64, C - c0) >> 2;
    const int cg = threadIdx.x & 15, slot = threadIdx.x >> 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (cg < cgs) {
        for (int p = slot; p < HW; p += 16) {
            const long off = (b * HW + p) * C + c0 + cg * 4;
            float v[4];
            load4(a + off, v);
            if (b2) {
                float w[4];
                load4(b2 + off, w);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= w[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[slot][cg * 4 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < cgs * 4) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
        out[b * C + c0 + threadIdx.x] = s * scale;
    }
}

// SE MLP: gate = sigmoid(W2 relu(W1 s + b1) + b2); one workgroup per sample.  hid saved for backward.
__global__ __launch_bounds__(256) void se_mlp_kernel(const float* __restrict__ s, const float* __restrict__ W1,
                                                     const float* __restrict__ b1, const float* __restrict__ W2,
                                                     const float* __restrict__ b2, float* __restrict__ hid,
                                                     float* __restrict__ gate, int C, int R) {
    extern __shared__ float sm[];  // s[C], h[R]
    float* ss = sm;
    float* hh = sm + C;
    const long b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) ss[c] = s[b * C + c];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < R; r += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += W1[(long)r * C + c] * ss[c];
        a = wave_sum(a);
        if (lane == 0) {
            a = fmaxf(a + b1[r], 0.f);
            hh[r] = a;
            hid[b * R + r] = a;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = b2[c];
        for (int r = 0; r < R; ++r) a += W2[(long)c * R + r] * hh[r];
        gate[b * C + c] = 1.f / (1.f + __expf(-a));
    }
}

// SE MLP backward: dgate[b][c] -> ds[b][c] (grad of the pooled mean) + parameter gradients (atomics)
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                         const float* __restrict__ hid, const float* __restrict__ s,
                                                         const float* __restrict__ W1, const float* __restrict__ W2,
                                                         float* __restrict__ ds, float ds_scale, float* __restrict__ dW1,
                                                         float* __restrict__ db1, float* __restrict__ dW2,
                                                         float* __restrict__ db2, int C, int R) {
    extern __shared__ float sm[];  // da[C], hh[R], dh[R], ss[C]
    float* da = sm;
    float* hh = sm + C;
    float* dh = hh + R;
    float* ss = dh + R;
    const long b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gate[b * C + c];
        da[c] = dgate[b * C + c] * g * (1.f - g);
        ss[c] = s[b * C + c];
    }
    for (int r = threadIdx.x; r < R; r += 256) hh[r] = hid[b * R + r];
    __syncthreads();
    // dW2[c][r] += da[c] * h[r]; db2[c] += da[c]
    for (int i = threadIdx.x; i < C * R; i += 256) {
        const int c = i / R, r = i - c * R;
        atomicAdd(dW2 + i, da[c] * hh[r]);
    }
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(db2 + c, da[c]);
    // dh[r] = (h[r] > 0) * sum_c W2[c][r] da[c]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < R; r += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += W2[(long)c * R + r] * da[c];
        a = wave_sum(a);
        if (lane == 0) dh[r] = hh[r] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * R; i += 256) {
        const int r = i / C, c = i - r * C;
        atomicAdd(dW1 + i, dh[r] * ss[c]);
    }
    for (int r = threadIdx.x; r < R; r += 256) atomicAdd(db1 + r, dh[r]);
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int r = 0; r < R; ++r) a += W1[(long)r * C + c] * dh[r];
        ds[b * C + c] = a * ds_scale;
    }
}

// y[b,hw,c] = x[b,hw,c] * g[b][c] + add[b][c]        (g / add may be NULL), 8 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void chan_scale_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ add, T* __restrict__ y, long n8,
                                                         int HW, int C) {
    const unsigned P = (unsigned)C >> 3, PB = P * (unsigned)HW;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)n8; i += gridDim.x * 256u) {
        const long e = (long)i * 8;
        const int c = (int)(i % P) << 3;
        const long b = i / PB;
        float v[8], gv[8], av[8];
        load8(x + e, v);
        // (two 32-byte loads, not 16 conditional dword loads in series: the per-element `if (g)` form compiled to one exec-masked
        //  branch + load per element)
        if (g) load8(g + b * C + c, gv);
        if (add) load8(add + b * C + c, av);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (g) v[j] *= gv[j];
            if (add) v[j] += av[j];
        }
        store8(y + e, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Gram pack: G fp32 [B][C][C] -> upper-tri (row-major, i<=j) vector, L2-normalised, written in the grouped/padded
// layout [B][groups][Kp] that the grouped embedding GEMM reads.  One workgroup per sample.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_ij(int t, int C, int& i, int& j) {
    // row-major upper-triangular index t -> (i, j>=i); rows before i hold i*C - i*(i-1)/2 entries
    float fi = ((2.f * C + 1.f) - sqrtf((2.f * C + 1.f) * (2.f * C + 1.f) - 8.f * (float)t)) * 0.5f;
    i = (int)fi;
    while (i > 0 && (long)i * C - (long)i * (i - 1) / 2 > t) --i;
    while ((long)(i + 1) * C - (long)(i + 1) * i / 2 <= t) ++i;
    j = i + t - (int)((long)i * C - (long)i * (i - 1) / 2);
}

// One workgroup of 1024 threads per sample, a wave per matrix row (coalesced reads of the upper-triangular part, no
// index inversion: the packed position of (i, j) is rowstart(i) + j - i and the lane steps through it).
template <typename T>
__global__ __launch_bounds__(1024) void gram_pack_kernel(const float* __restrict__ G, T* __restrict__ out,
                                                         float* __restrict__ inv_norm, int C, int groups, int Kg,
                                                         int Kp, int ntok, int per_tok) {
    __shared__ float red[16];
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* Gb = G + b * C * C;
    float ss = 0.f;
    for (int i = wave; i < C; i += 16)
        for (int j = i + lane; j < C; j += 64) {
            const float v = Gb[i * C + j];
            ss = fmaf(v, v, ss);
        }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float inv = 1.f / fmaxf(sqrtf(tot), 1e-12f);
    if (threadIdx.x == 0) inv_norm[b] = inv;
    T* ob = out + b * groups * Kp;
    for (int i = wave; i < C; i += 16) {
        const unsigned t0 = (unsigned)(i * C - i * (i - 1) / 2);        // packed index of (i, i)
        unsigned t = t0 + lane;
        unsigned g = t / (unsigned)Kg, k = t - g * (unsigned)Kg;
        if (ntok > 1) {     // GramToken's token interleave (map.py:225-227): packed entry t goes to (t % T) * (ntri / T) + t / T
            for (int j = i + lane; j < C; j += 64, t += 64) {
                const unsigned d = (t % (unsigned)ntok) * (unsigned)per_tok + t / (unsigned)ntok;
                const unsigned gd = d / (unsigned)Kg;
                elt<T>::st(ob + gd * Kp + (d - gd * (unsigned)Kg), Gb[i * C + j] * inv);
            }
            continue;
        }
        for (int j = i + lane; j < C; j += 64) {
            if (g < (unsigned)groups) elt<T>::st(ob + g * Kp + k, Gb[i * C + j] * inv);
            k += 64;
            while (k >= (unsigned)Kg) { k -= Kg; ++g; }
        }
    }
    const int pad = Kp - Kg;                                             // zero padding behind every group
    for (int t = threadIdx.x; t < groups * pad; t += 1024) {
        const int g = t / pad;
        elt<T>::st(ob + g * Kp + Kg + (t - g * pad), 0.f);
    }
}

// The same pack with the sample's packed vector staged in LDS (bf16, C (C + 1) / 2 entries <= 75 K): the form above pays three
// runtime integer divisions per entry for the token interleave and the group layout and scatters 2-byte stores; here the scaled
// upper triangle goes to LDS at its (interleaved) packed position, then [groups][Kp] is written group by group, coalesced.
template <int NTOK>
__global__ __launch_bounds__(1024) void gram_pack_lds_kernel(const float* __restrict__ G, bf16_t* __restrict__ out,
                                                             float* __restrict__ inv_norm, int C, int groups, int Kg, int Kp,
                                                             int ntok_rt, int per_tok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    bf16_t* packed = reinterpret_cast<bf16_t*>(smem_g);
    __shared__ float red[16];
    const int ntok = NTOK > 0 ? NTOK : ntok_rt;
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* Gb = G + b * C * C;
    float ss = 0.f;
    for (int i = wave; i < C; i += 16)
        for (int j = i + lane; j < C; j += 64) {
            const float v = Gb[i * C + j];
            ss = fmaf(v, v, ss);
        }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float inv = 1.f / fmaxf(sqrtf(tot), 1e-12f);
    if (threadIdx.x == 0) inv_norm[b] = inv;
    for (int i = wave; i < C; i += 16) {
        const unsigned t0 = (unsigned)(i * C - i * (i - 1) / 2);        // packed index of (i, i)
        for (int j = i + lane; j < C; j += 64) {
            unsigned t = t0 + (unsigned)(j - i);
            if (ntok > 1) t = (t % (unsigned)ntok) * (unsigned)per_tok + t / (unsigned)ntok;
            packed[t] = f2bf(Gb[i * C + j] * inv);
        }
    }
    __syncthreads();
    bf16_t* ob = out + b * groups * Kp;
    for (int g = 0; g < groups; ++g)
        for (int k = threadIdx.x; k < Kp; k += 1024) ob[g * Kp + k] = k < Kg ? packed[g * Kg + k] : (bf16_t)0;
}

// backward of normalise + pack: S[b][i][j] (T, symmetric; diagonal doubled) = d(raw gram entry)
//   draw = inv * (dvec - vhat * <vhat, dvec>)
template <typename T>
__global__ __launch_bounds__(1024) void gram_pack_bwd_kernel(const T* __restrict__ dvec, const T* __restrict__ vhat,
                                                             const float* __restrict__ inv_norm, T* __restrict__ S,
                                                             int C, int groups, int Kg, int Kp, int ntok, int per_tok) {
    __shared__ float red[16];
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* db = dvec + b * groups * Kp;
    const T* vb = vhat + b * groups * Kp;
    float dot = 0.f;
    for (int g = 0; g < groups; ++g)
        for (int k = threadIdx.x; k < Kg; k += 1024) dot = fmaf(elt<T>::ld(db + g * Kp + k), elt<T>::ld(vb + g * Kp + k), dot);
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) dot += red[w];
    const float inv = inv_norm[b];
    T* Sb = S + b * C * C;
    for (int i = wave; i < C; i += 16) {                 // output row i: entries (i, j) read packed (min, max)
        for (int j = lane; j < C; j += 64) {
            const int lo = min(i, j), hi = max(i, j);
            unsigned t = (unsigned)(lo * C - lo * (lo - 1) / 2 + (hi - lo));
            if (ntok > 1) t = (t % (unsigned)ntok) * (unsigned)per_tok + t / (unsigned)ntok;
            const unsigned g = t / (unsigned)Kg, k = t - g * (unsigned)Kg;
            const float draw = g < (unsigned)groups ? inv * (elt<T>::ld(db + g * Kp + k) - elt<T>::ld(vb + g * Kp + k) * dot) : 0.f;
            elt<T>::st(Sb + i * C + j, i == j ? 2.f * draw : draw);
        }
    }
}

// The same backward with the sample's packed gradient staged in LDS (bf16, C (C + 1) / 2 entries <= 75 K): the gather of the
// form above reads dvec and vhat element-wise from memory for every one of the C x C outputs -- the lower triangle column-wise,
// the token interleave strided -- and pays three runtime integer divisions per output (0.22-0.34 ms per launch at C = 384, 13x
// its bytes).  Here pass 2 computes draw once per packed entry with coalesced reads into LDS [t], and pass 3 fills the C x C
// matrix from LDS with coalesced stores (bf16(2 x) = 2 bf16(x): the same values up to the summation order of the dot product).
template <int NTOK>
__global__ __launch_bounds__(1024) void gram_pack_bwd_lds_kernel(const bf16_t* __restrict__ dvec, const bf16_t* __restrict__ vhat,
                                                                 const float* __restrict__ inv_norm, bf16_t* __restrict__ S, int C,
                                                                 int groups, int Kg, int Kp, int ntok_rt, int per_tok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    bf16_t* packed = reinterpret_cast<bf16_t*>(smem_g);
    __shared__ float red[16];
    const int ntok = NTOK > 0 ? NTOK : ntok_rt;
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bf16_t* db = dvec + b * groups * Kp;
    const bf16_t* vb = vhat + b * groups * Kp;
    // both passes over the sample's two vectors: 8-byte (4-element) accesses, 2 x 4 independent loads per iteration -- with one
    // workgroup of 16 waves per CU (the LDS image) the 2-byte, one-load-per-iteration form was latency-bound (0.10 ms per launch at
    // 73,920 entries: 72 dependent round trips)
    const bool vec4 = (Kg & 3) == 0;               // (group bases are 16-byte aligned: Kp % 8 == 0)
    float dot = 0.f;
    if (vec4) {
        const int K4 = Kg >> 2, n4 = groups * K4;
        for (int q = threadIdx.x; q < n4; q += 4 * 1024) {
            uint2 dv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q + u * 1024;
                const int g = qq < n4 ? qq / K4 : 0, k = qq < n4 ? (qq - g * K4) * 4 : 0;
                dv[u] = qq < n4 ? *reinterpret_cast<const uint2*>(db + g * Kp + k) : make_uint2(0u, 0u);
                vv[u] = qq < n4 ? *reinterpret_cast<const uint2*>(vb + g * Kp + k) : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dot = fmaf(__uint_as_float(dv[u].x << 16), __uint_as_float(vv[u].x << 16), dot);
                dot = fmaf(__uint_as_float(dv[u].x & 0xffff0000u), __uint_as_float(vv[u].x & 0xffff0000u), dot);
                dot = fmaf(__uint_as_float(dv[u].y << 16), __uint_as_float(vv[u].y << 16), dot);
                dot = fmaf(__uint_as_float(dv[u].y & 0xffff0000u), __uint_as_float(vv[u].y & 0xffff0000u), dot);
            }
        }
    } else {
        for (int g = 0; g < groups; ++g)
            for (int k = threadIdx.x; k < Kg; k += 1024) dot = fmaf(bf2f(db[g * Kp + k]), bf2f(vb[g * Kp + k]), dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) dot += red[w];
    const float inv = inv_norm[b];
    if (vec4) {
        const int K4 = Kg >> 2, n4 = groups * K4;
        for (int q = threadIdx.x; q < n4; q += 4 * 1024) {
            uint2 dv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q + u * 1024;
                const int g = qq < n4 ? qq / K4 : 0, k = qq < n4 ? (qq - g * K4) * 4 : 0;
                dv[u] = qq < n4 ? *reinterpret_cast<const uint2*>(db + g * Kp + k) : make_uint2(0u, 0u);
                vv[u] = qq < n4 ? *reinterpret_cast<const uint2*>(vb + g * Kp + k) : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q + u * 1024;
                if (qq < n4) {
                    const float o0 = inv * (__uint_as_float(dv[u].x << 16) - __uint_as_float(vv[u].x << 16) * dot);
                    const float o1 = inv * (__uint_as_float(dv[u].x & 0xffff0000u) - __uint_as_float(vv[u].x & 0xffff0000u) * dot);
                    const float o2 = inv * (__uint_as_float(dv[u].y << 16) - __uint_as_float(vv[u].y << 16) * dot);
                    const float o3 = inv * (__uint_as_float(dv[u].y & 0xffff0000u) - __uint_as_float(vv[u].y & 0xffff0000u) * dot);
                    *reinterpret_cast<uint2*>(packed + qq * 4) = make_uint2(pack2bf(o0, o1), pack2bf(o2, o3));   // g * Kg + k = 4 qq
                }
            }
        }
    } else {
        for (int g = 0; g < groups; ++g)
            for (int k = threadIdx.x; k < Kg; k += 1024)
                packed[g * Kg + k] = f2bf(inv * (bf2f(db[g * Kp + k]) - bf2f(vb[g * Kp + k]) * dot));
    }
    __syncthreads();
    bf16_t* Sb = S + b * C * C;
    for (int i = wave; i < C; i += 16) {                 // output row i: entries (i, j) read packed (min, max)
        for (int j = lane; j < C; j += 64) {
            const int lo = min(i, j), hi = max(i, j);
            unsigned t = (unsigned)(lo * C - lo * (lo - 1) / 2 + (hi - lo));
            if (ntok > 1) t = (t % (unsigned)ntok) * (unsigned)per_tok + t / (unsigned)ntok;
            const bf16_t v = packed[t];
            Sb[i * C + j] = i == j ? f2bf(2.f * bf2f(v)) : v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// token concat / split for the class-attention input u = cat(x_cls, tokens)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void token_cat_kernel(const T* __restrict__ cls, const T* __restrict__ tok,
                                                        T* __restrict__ u, long n8, int N, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long e = i * 8;
        const int c = (int)(e % C);
        const long r = e / C;
        const int n = (int)(r % (N + 1));
        const long b = r / (N + 1);
        const T* src = n == 0 ? cls + b * C + c : tok + (b * N + n - 1) * C + c;
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(u) + e * sizeof(T)) =
            *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src));
        if constexpr (sizeof(T) == 4)
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(u) + e * 4 + 16) =
                *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(src) + 16);
    }
}
// dcls[b] (+)= du[b,0];  dtok[b,n] (+)= du[b,1+n]
template <typename T>
__global__ __launch_bounds__(256) void token_split_kernel(const T* __restrict__ du, T* __restrict__ dcls,
                                                          T* __restrict__ dtok, long n8, int N, int C, int acc_cls,
                                                          int acc_tok) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long e = i * 8;
        const int c = (int)(e % C);
        const long r = e / C;
        const int n = (int)(r % (N + 1));
        const long b = r / (N + 1);
        float v[8];
        load8(du + e, v);
        T* dst = n == 0 ? dcls + b * C + c : dtok + (b * N + n - 1) * C + c;
        if (n == 0 ? acc_cls : acc_tok) {
            float o[8];
            load8(dst, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        store8(dst, v);
    }
}

// ------------------------------------------------------------------------------------------------
// class attention, one query per (sample, head).   q [B][E] (T), kv [B*N][2E] (k | v), E = heads*hd
//   p = softmax_n(scale * q.k_n);  out[b][h*hd+d] = sum_n p_n v[n][h*hd+d];   p saved fp32 [B][heads][N]
// one workgroup (256 threads = 4 waves) per sample; wave loops over heads
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void class_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                             T* __restrict__ out, float* __restrict__ P, int N,
                                                             int heads, int hd, float scale) {
    extern __shared__ float sm[];  // per wave: p[N], q[64]
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int E = heads * hd;
    float* pw = sm + wave * (N + 64);
    float* qv = pw + N;
    for (int h = wave; h < heads; h += 4) {
        if (lane < hd) qv[lane] = elt<T>::ld(q + b * E + h * hd + lane) * scale;
        float mx = -3.0e38f;
        for (int n = lane; n < N; n += 64) {
            const T* kp = kv + (b * N + n) * (long)(2 * E) + h * hd;
            float s = 0.f;
            for (int d = 0; d < hd; ++d) s += qv[d] * elt<T>::ld(kp + d);
            pw[n] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int n = lane; n < N; n += 64) {
            const float e = __expf(pw[n] - mx);
            pw[n] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int n = lane; n < N; n += 64) {
            const float p = pw[n] * inv;
            pw[n] = p;
            P[(b * heads + h) * N + n] = p;
        }
        // out[d] = sum_n p[n] v[n][d]: lane = d (hd <= 64); a wave's LDS accesses complete in issue order
        if (lane < hd) {
            float a = 0.f;
            for (int n = 0; n < N; ++n) a += pw[n] * elt<T>::ld(kv + (b * N + n) * (long)(2 * E) + E + h * hd + lane);
            elt<T>::st(out + b * E + h * hd + lane, a);
        }
    }
}

// backward: dout [B][E] -> dq [B][E], dkv [B*N][2E]
template <typename T>
__global__ __launch_bounds__(256) void class_attn_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                             const T* __restrict__ kv, const float* __restrict__ P,
                                                             T* __restrict__ dq, T* __restrict__ dkv, int N, int heads,
                                                             int hd, float scale) {
    extern __shared__ float sm[];  // per wave: ds[N], q[64], dout[64]
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int E = heads * hd;
    float* dsw = sm + wave * (N + 128);
    float* qv = dsw + N;
    float* dov = qv + 64;
    for (int h = wave; h < heads; h += 4) {
        if (lane < hd) {
            dov[lane] = elt<T>::ld(dout + b * E + h * hd + lane);
            qv[lane] = elt<T>::ld(q + b * E + h * hd + lane);
        }
        const float* pp = P + (b * heads + h) * N;
        float dot = 0.f;
        for (int n = lane; n < N; n += 64) {
            const T* vp = kv + (b * N + n) * (long)(2 * E) + E + h * hd;
            float dp = 0.f;
            for (int d = 0; d < hd; ++d) dp += dov[d] * elt<T>::ld(vp + d);
            dsw[n] = dp;
            dot += dp * pp[n];
        }
        dot = wave_sum(dot);
        for (int n = lane; n < N; n += 64) {
            const float p = pp[n];
            const float ds = p * (dsw[n] - dot);  // grad wrt the scaled score
            dsw[n] = ds;
            T* dk = dkv + (b * N + n) * (long)(2 * E) + h * hd;
            T* dv = dk + E;
            for (int d = 0; d < hd; ++d) {
                elt<T>::st(dk + d, ds * scale * qv[d]);
                elt<T>::st(dv + d, p * dov[d]);
            }
        }
        if (lane < hd) {
            float a = 0.f;
            for (int n = 0; n < N; ++n) a += dsw[n] * elt<T>::ld(kv + (b * N + n) * (long)(2 * E) + h * hd + lane);
            elt<T>::st(dq + b * E + h * hd + lane, a * scale);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// class attention, coalesced form (hd % 8 == 0, E <= 512): one workgroup of 16 waves per sample, a WAVE per token
// row -- lane c holds the 8-channel chunk c of the row, so the k | v rows are read and the dk | dv rows written as
// whole 16-byte-per-lane lines (the per-head form above walks tokens across lanes: 2-byte accesses 2E apart).
//   A: per token, per chunk partial dot products -> LDS part[n][chunk]
//   B: a wave per head finishes the dot products over the head's chunks and does the softmax (fwd) / its gradient (bwd)
//   C: per token, p-weighted sum of v (fwd); dk, dv rows and the ds-weighted sum of k for dq (bwd); the 16 waves' partial
//      sums meet in LDS
// ------------------------------------------------------------------------------------------------
constexpr int kCaWaves = 16, kCaThreads = 64 * kCaWaves;

template <typename T>
__global__ __launch_bounds__(kCaThreads) void class_attn_fwd_rows_kernel(const T* __restrict__ q, const T* __restrict__ kv_cls,
                                                                         long cls_stride, const T* __restrict__ kv_tok,
                                                                         long tok_stride, long tok_ld, T* __restrict__ out,
                                                                         float* __restrict__ P, int N, int heads, int hd,
                                                                         float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = heads * hd, NCH = E >> 3, cph = hd >> 3;   // chunks per row, per head
    float* part = sm;                  // [N][NCH]
    float* pl = part + N * NCH;        // [heads][N]
    float* red = pl + heads * N;       // [16][E]
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool live = lane < NCH;
    // row n of the sample: n == 0 the class token (kv_cls), n >= 1 token n-1 (kv_tok); the two may be one array
    // ([B][N][2E]: cls_stride = tok_stride = N*2E, kv_tok = kv_cls + 2E) or separate ([B][2E] and [B][N-1][2E])
    const T* kvc = kv_cls + b * cls_stride;
    const T* kvt = kv_tok + b * tok_stride - tok_ld;     // so that row n >= 1 is kvt + n * tok_ld
    float qv[8];
    if (live) {
        load8(q + b * E + lane * 8, qv);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] *= scale;
    }
    for (int n = wave; n < N; n += kCaWaves) {
        if (live) {
            float k[8];
            load8((n == 0 ? kvc : kvt + (long)n * tok_ld) + lane * 8, k);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[e], k[e], s);
            part[n * NCH + lane] = s;
        }
    }
    __syncthreads();
    for (int h = wave; h < heads; h += kCaWaves) {
        float mx = -3.0e38f;
        for (int n = lane; n < N; n += 64) {
            float s = 0.f;
            for (int i = 0; i < cph; ++i) s += part[n * NCH + h * cph + i];
            pl[h * N + n] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int n = lane; n < N; n += 64) {
            const float e = __expf(pl[h * N + n] - mx);
            pl[h * N + n] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int n = lane; n < N; n += 64) {
            const float p = pl[h * N + n] * inv;
            pl[h * N + n] = p;
            P[(b * heads + h) * N + n] = p;
        }
    }
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int hl = live ? lane / cph : 0;
    for (int n = wave; n < N; n += kCaWaves) {
        if (live) {
            float v[8];
            load8((n == 0 ? kvc : kvt + (long)n * tok_ld) + E + lane * 8, v);
            const float p = pl[hl * N + n];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, v[e], acc[e]);
        }
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * E + lane * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += kCaThreads) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < kCaWaves; ++w) a += red[w * E + c];
        elt<T>::st(out + b * E + c, a);
    }
}

template <typename T>
__global__ __launch_bounds__(kCaThreads) void class_attn_bwd_rows_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                                         const T* __restrict__ kv_cls, long cls_stride,
                                                                         const T* __restrict__ kv_tok, long tok_stride, long tok_ld,
                                                                         const float* __restrict__ P, T* __restrict__ dq,
                                                                         T* __restrict__ dkv_cls, T* __restrict__ dkv_tok,
                                                                         int N, int heads, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = heads * hd, NCH = E >> 3, cph = hd >> 3;
    float* part = sm;                  // [N][NCH]
    float* pl = part + N * NCH;        // [heads][N]  softmax probabilities
    float* dsl = pl + heads * N;       // [heads][N]  gradient wrt the scaled scores
    float* red = dsl + heads * N;      // [16][E]
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool live = lane < NCH;
    const T* kvc = kv_cls + b * cls_stride;
    const T* kvt = kv_tok + b * tok_stride - tok_ld;
    T* dkc = dkv_cls + b * cls_stride;
    T* dkt = dkv_tok + b * tok_stride - tok_ld;
    float dov[8], qv[8];
    if (live) {
        load8(dout + b * E + lane * 8, dov);
        load8(q + b * E + lane * 8, qv);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] *= scale;
    }
    for (int n = wave; n < N; n += kCaWaves) {
        if (live) {
            float v[8];
            load8((n == 0 ? kvc : kvt + (long)n * tok_ld) + E + lane * 8, v);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(dov[e], v[e], s);
            part[n * NCH + lane] = s;
        }
    }
    __syncthreads();
    for (int h = wave; h < heads; h += kCaWaves) {
        const float* pp = P + (b * heads + h) * N;
        float dot = 0.f;
        for (int n = lane; n < N; n += 64) {
            float dp = 0.f;
            for (int i = 0; i < cph; ++i) dp += part[n * NCH + h * cph + i];
            const float p = pp[n];
            pl[h * N + n] = p;
            dsl[h * N + n] = dp;
            dot = fmaf(dp, p, dot);
        }
        dot = wave_sum(dot);
        for (int n = lane; n < N; n += 64) dsl[h * N + n] = pl[h * N + n] * (dsl[h * N + n] - dot);
    }
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int hl = live ? lane / cph : 0;
    for (int n = wave; n < N; n += kCaWaves) {
        if (live) {
            float k[8], dk[8], dv[8];
            load8((n == 0 ? kvc : kvt + (long)n * tok_ld) + lane * 8, k);
            const float ds = dsl[hl * N + n], p = pl[hl * N + n];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] = fmaf(ds, k[e], acc[e]);
                dk[e] = ds * qv[e];
                dv[e] = p * dov[e];
            }
            T* drow = n == 0 ? dkc : dkt + (long)n * tok_ld;
            store8(drow + lane * 8, dk);
            store8(drow + E + lane * 8, dv);
        }
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * E + lane * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += kCaThreads) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < kCaWaves; ++w) a += red[w * E + c];
        elt<T>::st(dq + b * E + c, a * scale);
    }
}

// LDS bytes of the row form (0 = not applicable: use the per-head kernels)
static size_t class_attn_rows_lds(int N, int heads, int hd, bool bwd) {
    const int E = heads * hd;
    if (hd % 8 != 0 || E > 512) return 0;
    const size_t fl = (size_t)N * (E / 8) + (size_t)(bwd ? 2 : 1) * heads * N + (size_t)kCaWaves * E;
    return fl * sizeof(float) <= 150 * 1024 ? fl * sizeof(float) : 0;
}

#define DISPATCH_T(dtype, KERNEL, grid, block, lds, s, ...)                                              \
    do {                                                                                                 \
        if ((dtype) == GA_BF16) { using T = bf16_t; hipLaunchKernelGGL(KERNEL<T>, grid, block, lds, s, __VA_ARGS__); } \
        else { using T = float; hipLaunchKernelGGL(KERNEL<T>, grid, block, lds, s, __VA_ARGS__); }          \
    } while (0)

extern "C" int ga_pool_concat_fwd(const void* src, void* dst, int B, int Hin, int Win, int C, int Hout, int Wout,
                                  int ldd, int c_off, int mode, int dtype, ga_stream_t stream) {
    GA_REQUIRE(src && dst && C % 4 == 0 && c_off % 4 == 0 && ldd % 4 == 0, "ga_pool_concat_fwd: alignment");
    GA_REQUIRE(mode >= 0 && mode <= 3 &&
                   ((mode == 0 || mode == 2) ? (Hin % Hout == 0 && Win % Wout == 0 && Hin / Hout == Win / Wout &&
                                                (mode == 0 || (Hin / Hout) % 2 == 0))
                                             : (Hout == 2 * Hin && Wout == 2 * Win)),
               "ga_pool_concat_fwd: unsupported geometry %dx%d -> %dx%d mode %d", Hin, Win, Hout, Wout, mode);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool w8 = C % 8 == 0 && c_off % 8 == 0 && ldd % 8 == 0;
    const long total = (long)B * Hout * Wout * (C / (w8 ? 8 : 4));
    GA_REQUIRE(total < (1L << 32), "ga_pool_concat_fwd: tensor too large for 32-bit indexing");
#define GA_PC(E_)                                                                                                     \
    do {                                                                                                              \
        if (dtype == GA_BF16)                                                                                         \
            hipLaunchKernelGGL((pool_concat_kernel<bf16_t, E_>), dim3(nblk(total)), dim3(256), 0, s, (const bf16_t*)src, \
                               (bf16_t*)dst, B, Hin, Win, C, Hout, Wout, ldd, c_off, mode);                            \
        else                                                                                                          \
            hipLaunchKernelGGL((pool_concat_kernel<float, E_>), dim3(nblk(total)), dim3(256), 0, s, (const float*)src,  \
                               (float*)dst, B, Hin, Win, C, Hout, Wout, ldd, c_off, mode);                             \
    } while (0)
    if (w8) GA_PC(8);
    else GA_PC(4);
#undef GA_PC
    return ga_check_launch("ga_pool_concat_fwd");
}

extern "C" int ga_pool_concat_bwd(const void* dcat, const void* dres, void* dsrc, int B, int Hin, int Win, int C,
                                  int Hout, int Wout, int ldd, int c_off, int mode, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dcat && dsrc && C % 4 == 0 && c_off % 4 == 0 && ldd % 4 == 0, "ga_pool_concat_bwd: alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool w8 = C % 8 == 0 && c_off % 8 == 0 && ldd % 8 == 0;
    const long total = (long)B * Hin * Win * (C / (w8 ? 8 : 4));
    GA_REQUIRE(total < (1L << 32), "ga_pool_concat_bwd: tensor too large for 32-bit indexing");
#define GA_PC(E_)                                                                                                      \
    do {                                                                                                               \
        if (dtype == GA_BF16)                                                                                          \
            hipLaunchKernelGGL((pool_concat_bwd_kernel<bf16_t, E_>), dim3(nblk(total)), dim3(256), 0, s,                \
                               (const bf16_t*)dcat, (const bf16_t*)dres, (bf16_t*)dsrc, B, Hin, Win, C, Hout, Wout, ldd, \
                               c_off, mode);                                                                           \
        else                                                                                                           \
            hipLaunchKernelGGL((pool_concat_bwd_kernel<float, E_>), dim3(nblk(total)), dim3(256), 0, s,                 \
                               (const float*)dcat, (const float*)dres, (float*)dsrc, B, Hin, Win, C, Hout, Wout, ldd,   \
                               c_off, mode);                                                                           \
    } while (0)
    if (w8) GA_PC(8);
    else GA_PC(4);
#undef GA_PC
    return ga_check_launch("ga_pool_concat_bwd");
}

extern "C" int ga_spatial_sum(const void* a, const void* b2, float* out, int B, int HW, int C, float scale, int dtype,
                              ga_stream_t stream) {
    GA_REQUIRE(a && out && C % 4 == 0, "ga_spatial_sum: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, spatial_sum_kernel, dim3(B, cdiv(C, 64)), dim3(256), 0, s, (const T*)a, (const T*)b2, out, HW, C,
               scale);
    return ga_check_launch("ga_spatial_sum");
}

extern "C" int ga_se_mlp_fwd(const float* s, const float* W1, const float* b1, const float* W2, const float* b2,
                             float* hid, float* gate, int B, int C, int R, ga_stream_t stream) {
    GA_REQUIRE(s && W1 && b1 && W2 && b2 && hid && gate, "ga_se_mlp_fwd: null");
    hipLaunchKernelGGL(se_mlp_kernel, dim3(B), dim3(256), (C + R) * sizeof(float), reinterpret_cast<hipStream_t>(stream),
                       s, W1, b1, W2, b2, hid, gate, C, R);
    return ga_check_launch("ga_se_mlp_fwd");
}

extern "C" int ga_se_mlp_bwd(const float* dgate, const float* gate, const float* hid, const float* s, const float* W1,
                             const float* W2, float* ds, float ds_scale, float* dW1, float* db1, float* dW2, float* db2,
                             int B, int C, int R, ga_stream_t stream) {
    GA_REQUIRE(dgate && gate && hid && s && W1 && W2 && ds && dW1 && db1 && dW2 && db2, "ga_se_mlp_bwd: null");
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(B), dim3(256), (2 * C + 2 * R) * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), dgate, gate, hid, s, W1, W2, ds, ds_scale, dW1, db1, dW2, db2, C,
                       R);
    return ga_check_launch("ga_se_mlp_bwd");
}

extern "C" int ga_chan_scale(const void* x, const float* g, const float* add, void* y, int B, int HW, int C, int dtype,
                             ga_stream_t stream) {
    GA_REQUIRE(x && y && C % 8 == 0, "ga_chan_scale: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n8 = (long)B * HW * C / 8;
    DISPATCH_T(dtype, chan_scale_kernel, dim3(nblk(n8)), dim3(256), 0, s, (const T*)x, g, add, (T*)y, n8, HW, C);
    return ga_check_launch("ga_chan_scale");
}

extern "C" int ga_gram_pack_fwd2(const float* G, void* out, float* inv_norm, int B, int C, int groups, int Kp, int ntok,
                                 int dtype, ga_stream_t stream) {
    const int ntri = C * (C + 1) / 2;
    GA_REQUIRE(G && out && inv_norm && ntri % groups == 0 && Kp >= ntri / groups && ntok >= 1 && ntri % ntok == 0,
               "ga_gram_pack_fwd: bad args (C=%d groups=%d Kp=%d ntok=%d)", C, groups, Kp, ntok);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)ntri * 2;
    if (dtype == GA_BF16 && lds <= 150 * 1024 && GA_KNOB("GRAM_LDS", 1)) {
#define GA_GPF(NT)                                                                                                              \
    do {                                                                                                                        \
        auto k = gram_pack_lds_kernel<NT>;                                                                                      \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   150 * 1024) == hipSuccess;                                                   \
        GA_REQUIRE(ok, "ga_gram_pack_fwd: cannot reserve LDS");                                                                 \
        hipLaunchKernelGGL(k, dim3(B), dim3(1024), lds, s, G, (bf16_t*)out, inv_norm, C, groups, ntri / groups, Kp, ntok,       \
                           ntri / ntok);                                                                                        \
    } while (0)
        if (ntok == 1) GA_GPF(1);
        else if (ntok == 2) GA_GPF(2);
        else if (ntok == 3) GA_GPF(3);
        else GA_GPF(0);
#undef GA_GPF
        return ga_check_launch("ga_gram_pack_fwd");
    }
    DISPATCH_T(dtype, gram_pack_kernel, dim3(B), dim3(1024), 0, s, G, (T*)out, inv_norm, C, groups, ntri / groups, Kp, ntok,
               ntri / ntok);
    return ga_check_launch("ga_gram_pack_fwd");
}

extern "C" int ga_gram_pack_bwd2(const void* dvec, const void* vhat, const float* inv_norm, void* S, int B, int C,
                                 int groups, int Kp, int ntok, int dtype, ga_stream_t stream) {
    const int ntri = C * (C + 1) / 2;
    GA_REQUIRE(dvec && vhat && inv_norm && S && ntri % groups == 0 && ntok >= 1 && ntri % ntok == 0, "ga_gram_pack_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)ntri * 2;
    if (dtype == GA_BF16 && lds <= 150 * 1024 && GA_KNOB("GRAM_LDS", 1)) {      // every packed entry of a sample fits in LDS
#define GA_GPB(NT)                                                                                                              \
    do {                                                                                                                        \
        auto k = gram_pack_bwd_lds_kernel<NT>;                                                                                  \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   150 * 1024) == hipSuccess;                                                   \
        GA_REQUIRE(ok, "ga_gram_pack_bwd: cannot reserve LDS");                                                                 \
        hipLaunchKernelGGL(k, dim3(B), dim3(1024), lds, s, (const bf16_t*)dvec, (const bf16_t*)vhat, inv_norm, (bf16_t*)S, C,   \
                           groups, ntri / groups, Kp, ntok, ntri / ntok);                                                       \
    } while (0)
        if (ntok == 1) GA_GPB(1);
        else if (ntok == 2) GA_GPB(2);
        else if (ntok == 3) GA_GPB(3);
        else GA_GPB(0);
#undef GA_GPB
        return ga_check_launch("ga_gram_pack_bwd");
    }
    DISPATCH_T(dtype, gram_pack_bwd_kernel, dim3(B), dim3(1024), 0, s, (const T*)dvec, (const T*)vhat, inv_norm, (T*)S, C,
               groups, ntri / groups, Kp, ntok, ntri / ntok);
    return ga_check_launch("ga_gram_pack_bwd");
}

extern "C" int ga_gram_pack_fwd(const float* G, void* out, float* inv_norm, int B, int C, int groups, int Kp, int dtype,
                                ga_stream_t stream) {
    return ga_gram_pack_fwd2(G, out, inv_norm, B, C, groups, Kp, 1, dtype, stream);
}

extern "C" int ga_gram_pack_bwd(const void* dvec, const void* vhat, const float* inv_norm, void* S, int B, int C,
                                int groups, int Kp, int dtype, ga_stream_t stream) {
    return ga_gram_pack_bwd2(dvec, vhat, inv_norm, S, B, C, groups, Kp, 1, dtype, stream);
}

extern "C" int ga_token_cat(const void* cls, const void* tok, void* u, int B, int N, int C, int dtype,
                            ga_stream_t stream) {
    GA_REQUIRE(cls && tok && u && C % 8 == 0, "ga_token_cat: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n8 = (long)B * (N + 1) * C / 8;
    DISPATCH_T(dtype, token_cat_kernel, dim3(nblk(n8)), dim3(256), 0, s, (const T*)cls, (const T*)tok, (T*)u, n8, N, C);
    return ga_check_launch("ga_token_cat");
}

extern "C" int ga_token_split(const void* du, void* dcls, void* dtok, int B, int N, int C, int acc_cls, int acc_tok,
                              int dtype, ga_stream_t stream) {
    GA_REQUIRE(du && dcls && dtok && C % 8 == 0, "ga_token_split: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n8 = (long)B * (N + 1) * C / 8;
    DISPATCH_T(dtype, token_split_kernel, dim3(nblk(n8)), dim3(256), 0, s, (const T*)du, (T*)dcls, (T*)dtok, n8, N, C,
               acc_cls, acc_tok);
    return ga_check_launch("ga_token_split");
}

// rows-form launchers; kv_cls / kv_tok may be two views of one [B][N][2E] array or two separate arrays
template <typename T>
static int class_attn_fwd_rows(const void* q, const void* kv_cls, long cls_stride, const void* kv_tok, long tok_stride,
                               long tok_ld, void* out, float* P, int B, int N, int heads, int hd, float scale, size_t lds,
                               hipStream_t s) {
    auto k = class_attn_fwd_rows_kernel<T>;
    GA_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) ==
                   hipSuccess, "ga_class_attn_fwd: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k, dim3(B), dim3(kCaThreads), lds, s, (const T*)q, (const T*)kv_cls, cls_stride, (const T*)kv_tok,
                       tok_stride, tok_ld, (T*)out, P, N, heads, hd, scale);
    return ga_check_launch("ga_class_attn_fwd");
}
template <typename T>
static int class_attn_bwd_rows(const void* dout, const void* q, const void* kv_cls, long cls_stride, const void* kv_tok,
                               long tok_stride, long tok_ld, const float* P, void* dq, void* dkv_cls, void* dkv_tok, int B, int N,
                               int heads, int hd, float scale, size_t lds, hipStream_t s) {
    auto k = class_attn_bwd_rows_kernel<T>;
    GA_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) ==
                   hipSuccess, "ga_class_attn_bwd: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(k, dim3(B), dim3(kCaThreads), lds, s, (const T*)dout, (const T*)q, (const T*)kv_cls, cls_stride,
                       (const T*)kv_tok, tok_stride, tok_ld, P, (T*)dq, (T*)dkv_cls, (T*)dkv_tok, N, heads, hd, scale);
    return ga_check_launch("ga_class_attn_bwd");
}

extern "C" int ga_class_attn_fwd(const void* q, const void* kv, void* out, float* P, int B, int N, int heads, int hd,
                                 float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(q && kv && out && P && hd >= 1 && hd <= 64 && N >= 1, "ga_class_attn_fwd: bad args (hd<=64)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long E2 = 2L * heads * hd, esz = dtype == GA_BF16 ? 2 : 4;
    if (const size_t lds = class_attn_rows_lds(N, heads, hd, false)) {
        const void* tok = static_cast<const char*>(kv) + E2 * esz;
        return dtype == GA_BF16 ? class_attn_fwd_rows<bf16_t>(q, kv, N * E2, tok, N * E2, E2, out, P, B, N, heads, hd, scale, lds, s)
                                : class_attn_fwd_rows<float>(q, kv, N * E2, tok, N * E2, E2, out, P, B, N, heads, hd, scale, lds, s);
    }
    DISPATCH_T(dtype, class_attn_fwd_kernel, dim3(B), dim3(256), 4 * (N + 64) * sizeof(float), s, (const T*)q,
               (const T*)kv, (T*)out, P, N, heads, hd, scale);
    return ga_check_launch("ga_class_attn_fwd");
}

extern "C" int ga_class_attn_bwd(const void* dout, const void* q, const void* kv, const float* P, void* dq, void* dkv,
                                 int B, int N, int heads, int hd, float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dout && q && kv && P && dq && dkv && hd >= 1 && hd <= 64, "ga_class_attn_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long E2 = 2L * heads * hd, esz = dtype == GA_BF16 ? 2 : 4;
    if (const size_t lds = class_attn_rows_lds(N, heads, hd, true)) {
        const void* tok = static_cast<const char*>(kv) + E2 * esz;
        void* dtok = static_cast<char*>(dkv) + E2 * esz;
        return dtype == GA_BF16
                   ? class_attn_bwd_rows<bf16_t>(dout, q, kv, N * E2, tok, N * E2, E2, P, dq, dkv, dtok, B, N, heads, hd, scale, lds, s)
                   : class_attn_bwd_rows<float>(dout, q, kv, N * E2, tok, N * E2, E2, P, dq, dkv, dtok, B, N, heads, hd, scale, lds, s);
    }
    DISPATCH_T(dtype, class_attn_bwd_kernel, dim3(B), dim3(256), 4 * (N + 128) * sizeof(float), s, (const T*)dout,
               (const T*)q, (const T*)kv, P, (T*)dq, (T*)dkv, N, heads, hd, scale);
    return ga_check_launch("ga_class_attn_bwd");
}

// split form: the class-token row and the N-1 image-token rows of k | v live in separate arrays (kv_cls [B][2E],
// kv_tok [B][N-1][2E]) -- the image tokens' LayerNorm is shared by all heads, so their k | v rows come from one GEMM over
// the shared normalised tokens and are never concatenated with the per-head class token.  hd % 8 == 0, E <= 512.
extern "C" int ga_class_attn_fwd2(const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, void* out, float* P,
                                  int B, int N, int heads, int hd, float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(q && kv_cls && kv_tok && out && P && N >= 2, "ga_class_attn_fwd2: bad args");
    const size_t lds = class_attn_rows_lds(N, heads, hd, false);
    GA_REQUIRE(lds > 0, "ga_class_attn_fwd2: needs hd %% 8 == 0 and heads*hd <= 512 (hd=%d heads=%d N=%d)", hd, heads, N);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long E2 = 2L * heads * hd, ld = tok_ld > 0 ? tok_ld : E2;
    GA_REQUIRE(ld >= E2 && ld % 8 == 0, "ga_class_attn_fwd2: tok_ld must be a multiple of 8 and >= 2E");
    return dtype == GA_BF16 ? class_attn_fwd_rows<bf16_t>(q, kv_cls, E2, kv_tok, (N - 1) * ld, ld, out, P, B, N, heads, hd, scale, lds, s)
                            : class_attn_fwd_rows<float>(q, kv_cls, E2, kv_tok, (N - 1) * ld, ld, out, P, B, N, heads, hd, scale, lds, s);
}

extern "C" int ga_class_attn_bwd2(const void* dout, const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld,
                                  const float* P, void* dq, void* dkv_cls, void* dkv_tok, int B, int N, int heads, int hd,
                                  float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dout && q && kv_cls && kv_tok && P && dq && dkv_cls && dkv_tok && N >= 2, "ga_class_attn_bwd2: bad args");
    const size_t lds = class_attn_rows_lds(N, heads, hd, true);
    GA_REQUIRE(lds > 0, "ga_class_attn_bwd2: needs hd %% 8 == 0 and heads*hd <= 512 (hd=%d heads=%d N=%d)", hd, heads, N);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long E2 = 2L * heads * hd, ld = tok_ld > 0 ? tok_ld : E2;
    GA_REQUIRE(ld >= E2 && ld % 8 == 0, "ga_class_attn_bwd2: tok_ld must be a multiple of 8 and >= 2E");
    return dtype == GA_BF16 ? class_attn_bwd_rows<bf16_t>(dout, q, kv_cls, E2, kv_tok, (N - 1) * ld, ld, P, dq, dkv_cls, dkv_tok, B,
                                                          N, heads, hd, scale, lds, s)
                            : class_attn_bwd_rows<float>(dout, q, kv_cls, E2, kv_tok, (N - 1) * ld, ld, P, dq, dkv_cls, dkv_tok, B,
                                                         N, heads, hd, scale, lds, s);
}
