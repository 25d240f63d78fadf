// MAP head kernels for gfx950 (/root/reference/MAP/models/map.py): all small / latency- or HBM-bound, no MFMA here
// (the GEMM-shaped parts -- ch_reduction, bp_reduction, q / k / v / proj, the grouped MLP, the classifiers -- go through
// ga_gemm / ga_wgrad):
//   * Gram tokens -> class tokens: de-interleave of GramToken's (out_dim, T) channel layout + the self-distillation
//     mean token (map.py:231-232, 273-275) and its backward;
//   * ClassAttention with T query tokens against T class rows + N image rows of k | v (map.py:118-144), with the optional
//     attention-dropout mask, forward and backward;
//   * elementwise helpers: GELU fwd / bwd (MultiScale's ConvNormAct with non_linearity = GELU), ReLU derivative x dropout
//     mask (GroupConvMlp with act = ReLU), mask multiply, strided 2-D copy.
#include <algorithm>
#include "common.h"

namespace {

int nblk(long n, int per = 256, int cap = 4096) { return (int)std::max<long>(1, std::min<long>(cap, (n + per - 1) / per)); }

// e [B][C*T] with channel c*T + t  ->  tok [B][T (+1)][C]; the extra row = mean over t (self-distillation token)
template <typename T>
__global__ __launch_bounds__(256) void map_tokens_fwd_kernel(const T* __restrict__ e, T* __restrict__ tok, long B, int C, int Tn,
                                                             int add_mean) {
    const int rows = Tn + (add_mean ? 1 : 0);
    const long n = B * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / C;
        const int c = (int)(i - b * C);
        float s = 0.f;
        for (int t = 0; t < Tn; ++t) {
            const float v = elt<T>::ld(e + b * C * Tn + (long)c * Tn + t);
            s += elt<T>::round(v);
            elt<T>::st(tok + (b * rows + t) * C + c, v);
        }
        if (add_mean) elt<T>::st(tok + (b * rows + Tn) * C + c, s / (float)Tn);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void map_tokens_bwd_kernel(const T* __restrict__ dtok, T* __restrict__ de, long B, int C, int Tn,
                                                             int add_mean) {
    const int rows = Tn + (add_mean ? 1 : 0);
    const long n = B * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / C;
        const int c = (int)(i - b * C);
        const float dm = add_mean ? elt<T>::ld(dtok + (b * rows + Tn) * C + c) / (float)Tn : 0.f;
        for (int t = 0; t < Tn; ++t) elt<T>::st(de + b * C * Tn + (long)c * Tn + t, elt<T>::ld(dtok + (b * rows + t) * C + c) + dm);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// multi-token class attention.  q [B][T][E]; kv_cls [B][T][2E] (k | v of the class rows); kv_tok rows of k | v of the
// N - T image tokens (row stride tok_ld, sample stride (N - T) * tok_ld); E = heads * hd, hd % 8 == 0, E <= 512.
//   a_t = softmax_n(scale * q_t . k_n);  out_t = sum_n (a_t[n] * m_t[n]) v_n     (m = attention dropout mask or NULL)
//   P [B][T][heads][N] fp32 = a (saved for backward)
// One workgroup of 16 waves per sample; a WAVE per k | v row, lane c = the row's 8-channel chunk c (whole 16-byte lines).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kW = 16, kThr = 64 * kW, kMaxT = 8;   // kernels are built for MT = 4 (T <= 4, the ConvNeXt / ViT heads), 6 (map_pit_s: T = 5) and 8

template <typename T, int MT>
__global__ __launch_bounds__(kThr) void mt_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv_cls,
                                                           const T* __restrict__ kv_tok, long tok_ld, T* __restrict__ out,
                                                           float* __restrict__ P, const float* __restrict__ mask, int Tn, int N,
                                                           int heads, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = heads * hd, NCH = E >> 3, cph = hd >> 3;
    float* part = sm;                       // [N][NCH]
    float* pl = part + N * NCH;             // [T][heads][N]  (a * m)
    float* red = pl + Tn * heads * N;       // [kW][E]
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool live = lane < NCH;
    const T* kvc = kv_cls + b * Tn * 2 * E;
    const T* kvt = kv_tok + b * (N - Tn) * tok_ld - (long)Tn * tok_ld;    // row n >= T is kvt + n * tok_ld
    auto row = [&](int n) { return n < Tn ? kvc + (long)n * 2 * E : kvt + (long)n * tok_ld; };
    for (int t = 0; t < Tn; ++t) {
        float qv[8];
        if (live) {
            load8(q + (b * Tn + t) * E + lane * 8, qv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] *= scale;
        }
        __syncthreads();                     // `part` of the previous query is consumed
        for (int n = wave; n < N; n += kW) {
            if (live) {
                float k[8];
                load8(row(n) + lane * 8, k);
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qv[e], k[e], s);
                part[n * NCH + lane] = s;
            }
        }
        __syncthreads();
        for (int h = wave; h < heads; h += kW) {
            float* pr = pl + (t * heads + h) * N;
            float mx = -3.0e38f;
            for (int n = lane; n < N; n += 64) {
                float s = 0.f;
                for (int i = 0; i < cph; ++i) s += part[n * NCH + h * cph + i];
                pr[n] = s;
                mx = fmaxf(mx, s);
            }
            mx = wave_max(mx);
            float sum = 0.f;
            for (int n = lane; n < N; n += 64) {
                const float e = __expf(pr[n] - mx);
                pr[n] = e;
                sum += e;
            }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            const long pofs = ((b * Tn + t) * heads + h) * N;
            for (int n = lane; n < N; n += 64) {
                const float p = pr[n] * inv;
                P[pofs + n] = p;
                pr[n] = mask ? p * mask[pofs + n] : p;
            }
        }
    }
    __syncthreads();
    float acc[MT][8];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const int hl = live ? lane / cph : 0;
    for (int n = wave; n < N; n += kW) {
        if (live) {
            float v[8];
            load8(row(n) + E + lane * 8, v);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (t < Tn) {
                    const float p = pl[(t * heads + hl) * N + n];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(p, v[e], acc[t][e]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (t < Tn) {
            __syncthreads();
            if (live) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[wave * E + lane * 8 + e] = acc[t][e];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < E; c += kThr) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < kW; ++w) a += red[w * E + c];
                elt<T>::st(out + (b * Tn + t) * E + c, a);
            }
        }
    }
}

// backward: dout [B][T][E] -> dq [B][T][E], dkv_cls [B][T][2E], dkv_tok rows (overwritten)
template <typename T, int MT>
__global__ __launch_bounds__(512) void mt_attn_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                           const T* __restrict__ kv_cls, const T* __restrict__ kv_tok, long tok_ld,
                                                           const float* __restrict__ P, const float* __restrict__ mask,
                                                           T* __restrict__ dq, T* __restrict__ dkv_cls, T* __restrict__ dkv_tok,
                                                           long dtok_ld, int Tn, int N, int heads, int hd, float scale) {
    constexpr int kW = 8, kThr = 64 * kW;      // 8 waves (not the 16 of the forward): the per-token registers fit without spills
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = heads * hd, NCH = E >> 3, cph = hd >> 3;
    float* part = sm;                       // [N][NCH]
    float* pl = part + N * NCH;             // [T][heads][N]  a * m   (weights of dv)
    float* dsl = pl + Tn * heads * N;       // [T][heads][N]  gradient wrt the scaled scores
    float* red = dsl + Tn * heads * N;      // [kW][E]
    const long b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool live = lane < NCH;
    const T* kvc = kv_cls + b * Tn * 2 * E;
    const T* kvt = kv_tok + b * (N - Tn) * tok_ld - (long)Tn * tok_ld;
    T* dkc = dkv_cls + b * Tn * 2 * E;
    T* dkt = dkv_tok + b * (N - Tn) * dtok_ld - (long)Tn * dtok_ld;
    auto row = [&](int n) { return n < Tn ? kvc + (long)n * 2 * E : kvt + (long)n * tok_ld; };
    float dov[MT][8], qv[MT][8];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (t < Tn && live) {
            load8(dout + (b * Tn + t) * E + lane * 8, dov[t]);
            load8(q + (b * Tn + t) * E + lane * 8, qv[t]);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[t][e] *= scale;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dov[t][e] = qv[t][e] = 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (t >= Tn) break;
        __syncthreads();
        for (int n = wave; n < N; n += kW) {
            if (live) {
                float v[8];
                load8(row(n) + E + lane * 8, v);
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(dov[t][e], v[e], s);
                part[n * NCH + lane] = s;
            }
        }
        __syncthreads();
        for (int h = wave; h < heads; h += kW) {
            const long pofs = ((b * Tn + t) * heads + h) * N;
            float* pr = pl + (t * heads + h) * N;
            float* dr = dsl + (t * heads + h) * N;
            float dot = 0.f;
            for (int n = lane; n < N; n += 64) {
                float dp = 0.f;
                for (int i = 0; i < cph; ++i) dp += part[n * NCH + h * cph + i];
                const float a = P[pofs + n], m = mask ? mask[pofs + n] : 1.f;
                dp *= m;                                   // d(a) = d(a*m) * m
                pr[n] = a * m;
                dr[n] = dp;
                dot = fmaf(dp, a, dot);
            }
            dot = wave_sum(dot);
            for (int n = lane; n < N; n += 64) dr[n] = P[pofs + n] * (dr[n] - dot);
        }
    }
    __syncthreads();
    float acc[MT][8];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const int hl = live ? lane / cph : 0;
    for (int n = wave; n < N; n += kW) {
        if (live) {
            float k[8], dk[8], dv[8];
            load8(row(n) + lane * 8, k);
#pragma unroll
            for (int e = 0; e < 8; ++e) dk[e] = dv[e] = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (t < Tn) {
                    const float ds = dsl[(t * heads + hl) * N + n], p = pl[(t * heads + hl) * N + n];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        acc[t][e] = fmaf(ds, k[e], acc[t][e]);
                        dk[e] = fmaf(ds, qv[t][e], dk[e]);
                        dv[e] = fmaf(p, dov[t][e], dv[e]);
                    }
                }
            }
            T* drow = n < Tn ? dkc + (long)n * 2 * E : dkt + (long)n * dtok_ld;
            store8(drow + lane * 8, dk);
            store8(drow + E + lane * 8, dv);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (t < Tn) {
            __syncthreads();
            if (live) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[wave * E + lane * 8 + e] = acc[t][e];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < E; c += kThr) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < kW; ++w) a += red[w * E + c];
                elt<T>::st(dq + (b * Tn + t) * E + c, a * scale);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// `interactive` class attention (map.py:96-98,130-136): two linears over the HEAD axis around the softmax,
//     S = scale q k^T;  U = S + W1 S + b1;  A = softmax_n(U);  Pm = A + W2 A + b2;  D = Pm * mask;  out = D v
// (W1, W2 [heads][heads], the "S" / "A" above indexed [head][key]).  Used by the MAP heads of map_resnet50 / map_mobilenet_v1 /
// map_faster_vit_3_224; none of the registered in-scope models turns it on, so these kernels are written for clarity, not
// speed: one workgroup per sample, fp32 in LDS, a thread per (head, key) for the head mixing, a wave per head for the softmax.
// Saved for backward: P = A (the softmax output), as in the plain kernels.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kIaThr = 256;

template <typename T>
__global__ __launch_bounds__(kIaThr) void mt_attn_ia_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv_cls, const T* __restrict__ kv_tok,
                                                               long tok_ld, T* __restrict__ out, float* __restrict__ P,
                                                               const float* __restrict__ mask, const float* __restrict__ W1,
                                                               const float* __restrict__ b1, const float* __restrict__ W2,
                                                               const float* __restrict__ b2, int Tn, int N, int heads, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = heads * hd, HN = heads * N;
    float* S = sm;              // [heads][N]
    float* A = S + HN;          // [heads][N]
    const long b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const T* kvc = kv_cls + b * Tn * 2 * E;
    const T* kvt = kv_tok + b * (N - Tn) * tok_ld - (long)Tn * tok_ld;
    auto row = [&](int n) { return n < Tn ? kvc + (long)n * 2 * E : kvt + (long)n * tok_ld; };
    for (int t = 0; t < Tn; ++t) {
        const T* qt = q + (b * Tn + t) * E;
        for (int i = tid; i < HN; i += kIaThr) {                 // raw scores
            const int h = i / N, n = i - h * N;
            const T* kr = row(n) + h * hd;
            float s = 0.f;
            for (int e = 0; e < hd; ++e) s = fmaf(elt<T>::ld(qt + h * hd + e), elt<T>::ld(kr + e), s);
            S[i] = s * scale;
        }
        __syncthreads();
        for (int i = tid; i < HN; i += kIaThr) {                 // U = S + W1 S + b1
            const int h = i / N, n = i - h * N;
            float u = S[i] + b1[h];
            for (int g = 0; g < heads; ++g) u = fmaf(W1[h * heads + g], S[g * N + n], u);
            A[i] = u;
        }
        __syncthreads();
        for (int h = wave; h < heads; h += kIaThr / 64) {        // softmax over the keys, a wave per head
            float* ar = A + h * N;
            float mx = -3.0e38f;
            for (int n = lane; n < N; n += 64) mx = fmaxf(mx, ar[n]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int n = lane; n < N; n += 64) {
                const float e = __expf(ar[n] - mx);
                ar[n] = e;
                sum += e;
            }
            const float inv = 1.f / wave_sum(sum);
            const long pofs = ((b * Tn + t) * heads + h) * N;
            for (int n = lane; n < N; n += 64) {
                const float a = ar[n] * inv;
                ar[n] = a;
                P[pofs + n] = a;
            }
        }
        __syncthreads();
        for (int i = tid; i < HN; i += kIaThr) {                 // D = (A + W2 A + b2) * mask   (into S)
            const int h = i / N, n = i - h * N;
            float pm = A[i] + b2[h];
            for (int g = 0; g < heads; ++g) pm = fmaf(W2[h * heads + g], A[g * N + n], pm);
            S[i] = mask ? pm * mask[((b * Tn + t) * heads + h) * N + n] : pm;
        }
        __syncthreads();
        for (int c = tid; c < E; c += kIaThr) {                  // out = D v
            const int h = c / hd;
            float o = 0.f;
            for (int n = 0; n < N; ++n) o = fmaf(S[h * N + n], elt<T>::ld(row(n) + E + c), o);
            elt<T>::st(out + (b * Tn + t) * E + c, o);
        }
        __syncthreads();
    }
}

// backward: dq, dkv_cls, dkv_tok rows (overwritten); dW1, db1, dW2, db2 accumulated (fp32 atomics over the samples)
template <typename T>
__global__ __launch_bounds__(kIaThr) void mt_attn_ia_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ q, const T* __restrict__ kv_cls,
                                                               const T* __restrict__ kv_tok, long tok_ld, const float* __restrict__ P,
                                                               const float* __restrict__ mask, const float* __restrict__ W1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                                               T* __restrict__ dq, T* __restrict__ dkv_cls, T* __restrict__ dkv_tok,
                                                               long dtok_ld, float* __restrict__ dW1, float* __restrict__ db1,
                                                               float* __restrict__ dW2, float* __restrict__ db2, int Tn, int N, int heads,
                                                               int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int E = heads * hd, HN = heads * N;
    float* Dall = sm;                   // [T][heads][N]   D = Pm * mask          (weights of dv)
    float* dSall = Dall + Tn * HN;      // [T][heads][N]   gradient wrt the scaled raw scores
    float* St = dSall + Tn * HN;        // [heads][N]      raw scores of token t
    float* At = St + HN;                // [heads][N]      softmax output
    float* Gt = At + HN;                // [heads][N]      dPm, then dU
    float* Xt = Gt + HN;                // [heads][N]      dA
    const long b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NW = kIaThr / 64;
    const T* kvc = kv_cls + b * Tn * 2 * E;
    const T* kvt = kv_tok + b * (N - Tn) * tok_ld - (long)Tn * tok_ld;
    T* dkc = dkv_cls + b * Tn * 2 * E;
    T* dkt = dkv_tok + b * (N - Tn) * dtok_ld - (long)Tn * dtok_ld;
    auto row = [&](int n) { return n < Tn ? kvc + (long)n * 2 * E : kvt + (long)n * tok_ld; };
    for (int t = 0; t < Tn; ++t) {
        const T* qt = q + (b * Tn + t) * E;
        const T* dot_ = dout + (b * Tn + t) * E;
        const long pbase = (b * Tn + t) * (long)HN;
        for (int i = tid; i < HN; i += kIaThr) {
            const int h = i / N, n = i - h * N;
            const T* kr = row(n) + h * hd;
            float s = 0.f, g = 0.f;
            for (int e = 0; e < hd; ++e) {
                s = fmaf(elt<T>::ld(qt + h * hd + e), elt<T>::ld(kr + e), s);
                g = fmaf(elt<T>::ld(dot_ + h * hd + e), elt<T>::ld(kr + E + e), g);     // dD = dout . v
            }
            St[i] = s * scale;
            At[i] = P[pbase + i];
            Gt[i] = mask ? g * mask[pbase + i] : g;                                       // dPm = dD * mask
        }
        __syncthreads();
        for (int i = tid; i < HN; i += kIaThr) {                 // D (for dv) and dA = dPm + W2^T dPm
            const int h = i / N, n = i - h * N;
            float pm = At[i] + b2[h], da = Gt[i];
            for (int g = 0; g < heads; ++g) {
                pm = fmaf(W2[h * heads + g], At[g * N + n], pm);
                da = fmaf(W2[g * heads + h], Gt[g * N + n], da);
            }
            Dall[t * HN + i] = mask ? pm * mask[pbase + i] : pm;
            Xt[i] = da;
        }
        for (int pr = wave; pr < heads * heads; pr += NW) {      // dW2[h][g] += sum_n dPm[h][n] A[g][n];  db2[h] += sum_n dPm[h][n]
            const int h = pr / heads, g = pr - h * heads;
            float a = 0.f, bsum = 0.f;
            for (int n = lane; n < N; n += 64) {
                a = fmaf(Gt[h * N + n], At[g * N + n], a);
                bsum += Gt[h * N + n];
            }
            a = wave_sum(a);
            bsum = wave_sum(bsum);
            if (lane == 0) {
                atomicAdd(dW2 + pr, a);
                if (g == 0) atomicAdd(db2 + h, bsum);
            }
        }
        __syncthreads();
        for (int h = wave; h < heads; h += NW) {                 // softmax backward: dU = A (dA - <dA, A>)   (into Gt)
            float dot = 0.f;
            for (int n = lane; n < N; n += 64) dot = fmaf(Xt[h * N + n], At[h * N + n], dot);
            dot = wave_sum(dot);
            for (int n = lane; n < N; n += 64) Gt[h * N + n] = At[h * N + n] * (Xt[h * N + n] - dot);
        }
        __syncthreads();
        for (int i = tid; i < HN; i += kIaThr) {                 // dS = dU + W1^T dU
            const int h = i / N, n = i - h * N;
            float ds = Gt[i];
            for (int g = 0; g < heads; ++g) ds = fmaf(W1[g * heads + h], Gt[g * N + n], ds);
            dSall[t * HN + i] = ds;
        }
        for (int pr = wave; pr < heads * heads; pr += NW) {      // dW1[h][g] += sum_n dU[h][n] S[g][n];  db1[h] += sum_n dU[h][n]
            const int h = pr / heads, g = pr - h * heads;
            float a = 0.f, bsum = 0.f;
            for (int n = lane; n < N; n += 64) {
                a = fmaf(Gt[h * N + n], St[g * N + n], a);
                bsum += Gt[h * N + n];
            }
            a = wave_sum(a);
            bsum = wave_sum(bsum);
            if (lane == 0) {
                atomicAdd(dW1 + pr, a);
                if (g == 0) atomicAdd(db1 + h, bsum);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < Tn * E; i += kIaThr) {                 // dq[t][c] = scale sum_n dS[t][h][n] k[n][c]
        const int t = i / E, c = i - t * E, h = c / hd;
        float a = 0.f;
        for (int n = 0; n < N; ++n) a = fmaf(dSall[(t * heads + h) * N + n], elt<T>::ld(row(n) + c), a);
        elt<T>::st(dq + (b * Tn + t) * E + c, a * scale);
    }
    for (int i = tid; i < N * E; i += kIaThr) {                  // dk[n][c] = scale sum_t dS q;  dv[n][c] = sum_t D dout
        const int n = i / E, c = i - n * E, h = c / hd;
        float dk = 0.f, dv = 0.f;
        for (int t = 0; t < Tn; ++t) {
            dk = fmaf(dSall[(t * heads + h) * N + n], elt<T>::ld(q + (b * Tn + t) * E + c), dk);
            dv = fmaf(Dall[(t * heads + h) * N + n], elt<T>::ld(dout + (b * Tn + t) * E + c), dv);
        }
        T* drow = n < Tn ? dkc + (long)n * 2 * E : dkt + (long)n * dtok_ld;
        elt<T>::st(drow + c, dk * scale);
        elt<T>::st(drow + E + c, dv);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// elementwise helpers (8 elements per thread)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8];
        load8(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        store8(y + i * 8, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8], g[8];
        load8(x + i * 8, v);
        load8(dy + i * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= gelu_grad_f(v[e]);
        store8(dx + i * 8, g);
    }
}

// a (ReLU output) -> a * m, deriv = (a > 0) * m   (m fp32 mask or NULL = 1)
template <typename T>
__global__ __launch_bounds__(256) void relu_drop_kernel(const T* __restrict__ a, const float* __restrict__ m, T* __restrict__ out,
                                                        T* __restrict__ deriv, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8], mm[8], d[8];
        load8(a + i * 8, v);
        if (m) load8(m + i * 8, mm);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float k = m ? mm[e] : 1.f;
            d[e] = v[e] > 0.f ? k : 0.f;
            v[e] *= k;
        }
        store8(out + i * 8, v);
        if (deriv) store8(deriv + i * 8, d);
    }
}

// y = x * m (+ res)
template <typename T>
__global__ __launch_bounds__(256) void mask_mul_kernel(const T* __restrict__ x, const float* __restrict__ m, const T* __restrict__ res,
                                                       T* __restrict__ y, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float v[8], mm[8];
        load8(x + i * 8, v);
        load8(m + i * 8, mm);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mm[e];
        if (res) {
            float r[8];
            load8(res + i * 8, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        store8(y + i * 8, v);
    }
}

// dst[r][0..cols) (row stride ldd) (+)= src[r][0..cols) (row stride lds); cols % 8 == 0
template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(const T* __restrict__ src, long lds, T* __restrict__ dst, long ldd, long rows,
                                                     int cols, int accumulate) {
    const int c8 = cols / 8;
    const long n = rows * c8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        float v[8];
        load8(src + r * lds + c, v);
        if (accumulate) {
            float o[8];
            load8(dst + r * ldd + c, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += o[e];
        }
        store8(dst + r * ldd + c, v);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

size_t mt_lds(int Tn, int N, int heads, int hd, bool bwd) {
    const int E = heads * hd;
    return ((size_t)N * (E / 8) + (size_t)(bwd ? 2 : 1) * Tn * heads * N + (size_t)kW * E) * sizeof(float);
}

}  // namespace

#define MAP_DISPATCH(dtype, KERNEL, grid, block, lds, s, ...)                                                       \
    do {                                                                                                            \
        if ((dtype) == GA_BF16) { using T = bf16_t; hipLaunchKernelGGL(KERNEL<T>, grid, block, lds, s, __VA_ARGS__); } \
        else { using T = float; hipLaunchKernelGGL(KERNEL<T>, grid, block, lds, s, __VA_ARGS__); }                  \
    } while (0)

extern "C" int ga_map_tokens_fwd(const void* e, void* tok, int B, int C, int T_, int add_mean, int dtype, ga_stream_t stream) {
    GA_REQUIRE(e && tok && B > 0 && C > 0 && T_ >= 1, "ga_map_tokens_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, map_tokens_fwd_kernel, dim3(nblk((long)B * C)), dim3(256), 0, s, (const T*)e, (T*)tok, (long)B, C, T_, add_mean);
    return ga_check_launch("ga_map_tokens_fwd");
}

extern "C" int ga_map_tokens_bwd(const void* dtok, void* de, int B, int C, int T_, int add_mean, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dtok && de && B > 0 && C > 0 && T_ >= 1, "ga_map_tokens_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, map_tokens_bwd_kernel, dim3(nblk((long)B * C)), dim3(256), 0, s, (const T*)dtok, (T*)de, (long)B, C, T_, add_mean);
    return ga_check_launch("ga_map_tokens_bwd");
}

extern "C" int ga_class_attn_mt_fwd(const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, void* out, float* P,
                                    const float* mask, int B, int T_, int N, int heads, int hd, float scale, int dtype,
                                    ga_stream_t stream) {
    GA_REQUIRE(q && kv_cls && kv_tok && out && P && B > 0 && T_ >= 1 && T_ <= kMaxT && N > T_, "ga_class_attn_mt_fwd: bad args (T <= %d)", kMaxT);
    const int E = heads * hd;
    GA_REQUIRE(hd % 8 == 0 && E <= 512 && tok_ld >= 2 * E && tok_ld % 8 == 0 && al16(q) && al16(kv_cls) && al16(kv_tok),
               "ga_class_attn_mt_fwd: needs head_dim %% 8 == 0, heads*head_dim <= 512, 16-byte aligned rows");
    const size_t lds = mt_lds(T_, N, heads, hd, false);
    GA_REQUIRE(lds <= 160 * 1024, "ga_class_attn_mt_fwd: %zu B of LDS needed", lds);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto reserve = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    static const bool ok = reserve(reinterpret_cast<const void*>(mt_attn_fwd_kernel<bf16_t, 4>)) && reserve(reinterpret_cast<const void*>(mt_attn_fwd_kernel<float, 4>)) &&
                           reserve(reinterpret_cast<const void*>(mt_attn_fwd_kernel<bf16_t, 6>)) && reserve(reinterpret_cast<const void*>(mt_attn_fwd_kernel<float, 6>)) &&
                           reserve(reinterpret_cast<const void*>(mt_attn_fwd_kernel<bf16_t, 8>)) && reserve(reinterpret_cast<const void*>(mt_attn_fwd_kernel<float, 8>));
    GA_REQUIRE(ok, "ga_class_attn_mt_fwd: cannot reserve LDS");
#define MT_LAUNCH(TT, MTT) hipLaunchKernelGGL((mt_attn_fwd_kernel<TT, MTT>), dim3(B), dim3(kThr), lds, s, (const T*)q, (const T*)kv_cls, (const T*)kv_tok, (long)tok_ld, (T*)out, P, mask, T_, N, heads, hd, scale)
    if (dtype == GA_BF16) { using T = bf16_t; if (T_ <= 4) MT_LAUNCH(T, 4); else if (T_ <= 6) MT_LAUNCH(T, 6); else MT_LAUNCH(T, 8); }
    else { using T = float; if (T_ <= 4) MT_LAUNCH(T, 4); else if (T_ <= 6) MT_LAUNCH(T, 6); else MT_LAUNCH(T, 8); }
#undef MT_LAUNCH
    return ga_check_launch("ga_class_attn_mt_fwd");
}

extern "C" int ga_class_attn_mt_bwd(const void* dout, const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld,
                                    const float* P, const float* mask, void* dq, void* dkv_cls, void* dkv_tok, int64_t dtok_ld,
                                    int B, int T_, int N, int heads, int hd, float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dout && q && kv_cls && kv_tok && P && dq && dkv_cls && dkv_tok && B > 0 && T_ >= 1 && T_ <= kMaxT && N > T_,
               "ga_class_attn_mt_bwd: bad args (T <= %d)", kMaxT);
    const int E = heads * hd;
    GA_REQUIRE(hd % 8 == 0 && E <= 512 && tok_ld >= 2 * E && dtok_ld >= 2 * E && tok_ld % 8 == 0 && dtok_ld % 8 == 0,
               "ga_class_attn_mt_bwd: needs head_dim %% 8 == 0, heads*head_dim <= 512");
    const size_t lds = mt_lds(T_, N, heads, hd, true);
    GA_REQUIRE(lds <= 160 * 1024, "ga_class_attn_mt_bwd: %zu B of LDS needed", lds);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto reserve = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    static const bool ok = reserve(reinterpret_cast<const void*>(mt_attn_bwd_kernel<bf16_t, 4>)) && reserve(reinterpret_cast<const void*>(mt_attn_bwd_kernel<float, 4>)) &&
                           reserve(reinterpret_cast<const void*>(mt_attn_bwd_kernel<bf16_t, 6>)) && reserve(reinterpret_cast<const void*>(mt_attn_bwd_kernel<float, 6>)) &&
                           reserve(reinterpret_cast<const void*>(mt_attn_bwd_kernel<bf16_t, 8>)) && reserve(reinterpret_cast<const void*>(mt_attn_bwd_kernel<float, 8>));
    GA_REQUIRE(ok, "ga_class_attn_mt_bwd: cannot reserve LDS");
#define MT_LAUNCH(TT, MTT) hipLaunchKernelGGL((mt_attn_bwd_kernel<TT, MTT>), dim3(B), dim3(512), lds, s, (const T*)dout, (const T*)q, (const T*)kv_cls, (const T*)kv_tok, (long)tok_ld, P, mask, (T*)dq, (T*)dkv_cls, (T*)dkv_tok, (long)dtok_ld, T_, N, heads, hd, scale)
    if (dtype == GA_BF16) { using T = bf16_t; if (T_ <= 4) MT_LAUNCH(T, 4); else if (T_ <= 6) MT_LAUNCH(T, 6); else MT_LAUNCH(T, 8); }
    else { using T = float; if (T_ <= 4) MT_LAUNCH(T, 4); else if (T_ <= 6) MT_LAUNCH(T, 6); else MT_LAUNCH(T, 8); }
#undef MT_LAUNCH
    return ga_check_launch("ga_class_attn_mt_bwd");
}

extern "C" int ga_class_attn_mt_ia_fwd(const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, void* out, float* P,
                                       const float* mask, const float* W1, const float* b1, const float* W2, const float* b2, int B, int T_,
                                       int N, int heads, int hd, float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(q && kv_cls && kv_tok && out && P && W1 && b1 && W2 && b2 && B > 0 && T_ >= 1 && T_ <= kMaxT && N > T_ && heads >= 1 && hd >= 1,
               "ga_class_attn_mt_ia_fwd: bad args (T <= %d)", kMaxT);
    GA_REQUIRE(tok_ld >= 2 * heads * hd, "ga_class_attn_mt_ia_fwd: tok_ld");
    const size_t lds = (size_t)2 * heads * N * sizeof(float);
    GA_REQUIRE(lds <= 160 * 1024, "ga_class_attn_mt_ia_fwd: %zu B of LDS needed", lds);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto reserve = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    static const bool ok = reserve(reinterpret_cast<const void*>(mt_attn_ia_fwd_kernel<bf16_t>)) && reserve(reinterpret_cast<const void*>(mt_attn_ia_fwd_kernel<float>));
    GA_REQUIRE(ok, "ga_class_attn_mt_ia_fwd: cannot reserve LDS");
    MAP_DISPATCH(dtype, mt_attn_ia_fwd_kernel, dim3(B), dim3(kIaThr), lds, s, (const T*)q, (const T*)kv_cls, (const T*)kv_tok, (long)tok_ld, (T*)out, P,
                 mask, W1, b1, W2, b2, T_, N, heads, hd, scale);
    return ga_check_launch("ga_class_attn_mt_ia_fwd");
}

extern "C" int ga_class_attn_mt_ia_bwd(const void* dout, const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, const float* P,
                                       const float* mask, const float* W1, const float* W2, const float* b2, void* dq, void* dkv_cls,
                                       void* dkv_tok, int64_t dtok_ld, float* dW1, float* db1, float* dW2, float* db2, int B, int T_, int N,
                                       int heads, int hd, float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dout && q && kv_cls && kv_tok && P && W1 && W2 && b2 && dq && dkv_cls && dkv_tok && dW1 && db1 && dW2 && db2 && B > 0 && T_ >= 1 &&
                   T_ <= kMaxT && N > T_, "ga_class_attn_mt_ia_bwd: bad args (T <= %d)", kMaxT);
    GA_REQUIRE(tok_ld >= 2 * heads * hd && dtok_ld >= 2 * heads * hd, "ga_class_attn_mt_ia_bwd: row strides");
    const size_t lds = (size_t)(2 * T_ + 4) * heads * N * sizeof(float);
    GA_REQUIRE(lds <= 160 * 1024, "ga_class_attn_mt_ia_bwd: %zu B of LDS needed", lds);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto reserve = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    static const bool ok = reserve(reinterpret_cast<const void*>(mt_attn_ia_bwd_kernel<bf16_t>)) && reserve(reinterpret_cast<const void*>(mt_attn_ia_bwd_kernel<float>));
    GA_REQUIRE(ok, "ga_class_attn_mt_ia_bwd: cannot reserve LDS");
    MAP_DISPATCH(dtype, mt_attn_ia_bwd_kernel, dim3(B), dim3(kIaThr), lds, s, (const T*)dout, (const T*)q, (const T*)kv_cls, (const T*)kv_tok,
                 (long)tok_ld, P, mask, W1, W2, b2, (T*)dq, (T*)dkv_cls, (T*)dkv_tok, (long)dtok_ld, dW1, db1, dW2, db2, T_, N, heads, hd, scale);
    return ga_check_launch("ga_class_attn_mt_ia_bwd");
}

extern "C" int ga_gelu_fwd(const void* x, void* y, int64_t n, int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && y && n > 0 && n % 8 == 0 && al16(x) && al16(y), "ga_gelu_fwd: n must be a multiple of 8, 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, gelu_fwd_kernel, dim3(nblk(n / 8)), dim3(256), 0, s, (const T*)x, (T*)y, (long)(n / 8));
    return ga_check_launch("ga_gelu_fwd");
}

extern "C" int ga_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dy && x && dx && n > 0 && n % 8 == 0, "ga_gelu_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, gelu_bwd_kernel, dim3(nblk(n / 8)), dim3(256), 0, s, (const T*)dy, (const T*)x, (T*)dx, (long)(n / 8));
    return ga_check_launch("ga_gelu_bwd");
}

extern "C" int ga_relu_drop(const void* a, const float* mask, void* out, void* deriv, int64_t n, int dtype, ga_stream_t stream) {
    GA_REQUIRE(a && out && n > 0 && n % 8 == 0, "ga_relu_drop: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, relu_drop_kernel, dim3(nblk(n / 8)), dim3(256), 0, s, (const T*)a, mask, (T*)out, (T*)deriv, (long)(n / 8));
    return ga_check_launch("ga_relu_drop");
}

extern "C" int ga_mask_mul(const void* x, const float* mask, const void* res, void* y, int64_t n, int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && mask && y && n > 0 && n % 8 == 0, "ga_mask_mul: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, mask_mul_kernel, dim3(nblk(n / 8)), dim3(256), 0, s, (const T*)x, mask, (const T*)res, (T*)y, (long)(n / 8));
    return ga_check_launch("ga_mask_mul");
}

extern "C" int ga_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int cols, int accumulate, int dtype,
                         ga_stream_t stream) {
    GA_REQUIRE(src && dst && rows > 0 && cols > 0 && cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && al16(src) && al16(dst),
               "ga_copy2d: cols / leading dimensions must be multiples of 8, 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MAP_DISPATCH(dtype, copy2d_kernel, dim3(nblk(rows * (cols / 8))), dim3(256), 0, s, (const T*)src, (long)lds, (T*)dst, (long)ldd,
                 (long)rows, cols, accumulate);
    return ga_check_launch("ga_copy2d");
}
