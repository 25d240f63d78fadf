// LayerNorm (rows x C) and BatchNorm (train-mode batch statistics) kernels for gfx950.  All HBM-bound:
// a row of C channels is spread over G = 4..64 lanes (4 channels per lane per step, 8/16-byte accesses), row
// reductions are wave shuffles; parameter-gradient column sums are kept in registers by persistent
// workgroups and reach memory as one fp32 atomic per column per workgroup.
#include <algorithm>
#include "common.h"

namespace {

constexpr int kMaxChunks = 4;  // 16-byte chunks (8 bf16 / 4 fp32 channels) per lane, at most

int pick_group(int C, int epc) {
    const int nch = C / epc;
    if (nch == 12) {
        // 192-byte rows (C = 96 bf16, the stage-0 / stem norms): with 4 lanes per row one load instruction touches 64 B
        // of each row, i.e. HALF of every 128-byte line, and the streaming (non-temporal) loads do not keep the line
        // for the instruction that wants the other half: FETCH_SIZE showed 1.73x the algorithmic bytes.  16 lanes per
        // row (12 active) make every instruction cover whole rows: 0.053 vs 0.072 ms forward, 0.132 vs 0.145 backward
        return GA_KNOB("LN_G12", 16);
    }
    for (int g = 4; g <= 64; g <<= 1)
        if (kMaxChunks * g >= nch) return g;
    if (nch <= 8 * 64) return 64;      // very wide rows (fp32 NormHead of map_pit_s: 4 x 384 channels): 8 chunks per lane
    return 0;
}

// one 16-byte chunk <-> floats
__device__ __forceinline__ void ld_chunk(const float* p, float v[4]) { load4(p, v); }
__device__ __forceinline__ void ld_chunk(const bf16_t* p, float v[8]) { load8(p, v); }
// read-once input (dead after this kernel): streaming load, keeps the L2 for what the next GEMM re-reads
__device__ __forceinline__ void ld_chunk_nt(const float* p, float v[4]) { load4(p, v); }
__device__ __forceinline__ void ld_chunk_nt(const bf16_t* p, float v[8]) { load8_nt(p, v); }
__device__ __forceinline__ void st_chunk(float* p, const float v[4]) { store4(p, v); }
__device__ __forceinline__ void st_chunk(bf16_t* p, const float v[8]) { store8(p, v); }

int grid_blocks(long work_items, int per_block, int max_blocks = 2048) {
    return (int)std::max<long>(1, std::min<long>(max_blocks, (work_items + per_block - 1) / per_block));
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward
// ------------------------------------------------------------------------------------------------
template <typename T, int G, bool AFF, int kMaxCh>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                     int C, float eps) {
    constexpr int E = elt<T>::EPC;            // channels per 16-byte chunk
    const int lg = threadIdx.x % G, rib = threadIdx.x / G, rpb = 256 / G;
    const int nch = C / E;
    const float invC = 1.f / (float)C;
    // the lane's chunks are the same for every row: affine parameters are read once
    float wv[AFF ? kMaxCh : 1][E], bv[AFF ? kMaxCh : 1][E];
    if constexpr (AFF) {
#pragma unroll
        for (int j = 0; j < kMaxCh; ++j) {
            const int ci = lg + G * j;
            // UNCONDITIONAL loads from a clamped chunk + selects: as `ci < nch ? w[...] : 1` hipcc emitted one exec-masked branch
            // with a single dword load per element -- 2 E dependent round trips per workgroup before its first row (44 us for ANY
            // tensor of the step against 8-29 us for the form without parameters: tools/r03/ln_bench.py)
            const int cc = ci < nch ? ci : 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float wl = w[cc * E + e], bl = b[cc * E + e];
                wv[j][e] = ci < nch ? wl : 1.f;
                bv[j][e] = ci < nch ? bl : 0.f;
            }
        }
    }
    // U rows per thread and iteration (their loads issued back to back).  Measured round 3, same box, alternating builds
    // (tools/ew_bench.py): U = 2 changes nothing forward (0.053 ms = 5.8 TB/s at C = 96, M = 802,816 either way) -- the
    // 2048-thread-per-CU occupancy already keeps enough loads in flight -- so the simple form stays
#ifdef GA_LN_UNROLL
    constexpr int U = GA_LN_UNROLL;       // (A/B builds)
#else
    constexpr int U = 1;
#endif
    const long rstep = (long)gridDim.x * rpb;
    for (long row0 = (long)blockIdx.x * rpb + rib; row0 < rows; row0 += U * rstep) {
        float v[U][kMaxCh][E];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * rstep;
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j) {
                const int ci = lg + G * j;
                if (ci < nch && row < rows) {
                    ld_chunk_nt(x + row * C + ci * E, v[u][j]);
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) v[u][j][e] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * rstep;
            if (row >= rows) break;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j)
#pragma unroll
                for (int e = 0; e < E; ++e) s += v[u][j][e];
            const float mu = group_sum<G>(s) * invC;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j) {
                const int ci = lg + G * j;
                if (ci < nch) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const float dlt = v[u][j][e] - mu;
                        q += dlt * dlt;
                    }
                }
            }
            const float rs = rsqrtf(group_sum<G>(q) * invC + eps);
            if (lg == 0) {
                if (mean) mean[row] = mu;
                if (rstd) rstd[row] = rs;
            }
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j) {
                const int ci = lg + G * j;
                if (ci < nch) {
                    float o[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        o[e] = (v[u][j][e] - mu) * rs;
                        if constexpr (AFF) o[e] = fmaf(o[e], wv[j][e], bv[j][e]);
                    }
                    st_chunk(y + row * C + ci * E, o);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  xhat = xnorm ? x : (x - mean) * rstd
//   dx = rstd * (gw - mean_c(gw) - xhat * mean_c(gw * xhat)) + dres,   gw = g * w
//   dw[c] += sum_rows g * xhat ;  db[c] += sum_rows g
// ------------------------------------------------------------------------------------------------
template <typename T, int G, bool AFF, int kMaxCh>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ w, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dw,
                                                     float* __restrict__ db, long rows, int C, int xnorm,
                                                     T* __restrict__ dx2, const float* __restrict__ scale2, long rows_per_scale) {
    constexpr int E = elt<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [2][rpb][C] when dw != null
    const int lg = threadIdx.x % G, rib = threadIdx.x / G, rpb = 256 / G;
    const int nch = C / E;
    const float invC = 1.f / (float)C;
    // AFF: affine weight and / or parameter gradients present (the block LayerNorms have neither: their affine part is
    // folded into fc1) -- 96 fewer VGPRs without
    float aw[AFF ? kMaxCh : 1][E], ab[AFF ? kMaxCh : 1][E], wv[AFF ? kMaxCh : 1][E];
    if constexpr (AFF) {
#pragma unroll
        for (int j = 0; j < kMaxCh; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                aw[j][e] = ab[j][e] = 0.f;
                const int ci = lg + G * j;
                // (unconditional load from a clamped chunk + select: see ln_fwd_kernel)
                const float* wsafe = w ? w : rstd;                 // any readable fp32 address when there is no weight
                const float wl = wsafe[(w && ci < nch) ? ci * E + e : 0];
                wv[j][e] = (w && ci < nch) ? wl : 1.f;
            }
    }
    // U rows per thread and iteration: measured SLOWER backward with U = 2 / 4 (block norm of C = 96: 0.097 vs 0.079 ms, affine
    // form 0.149 vs 0.115; same box, alternating builds): the extra registers cost occupancy, which is what keeps loads in flight
#ifdef GA_LN_UNROLL
    constexpr int U = GA_LN_UNROLL;
#else
    constexpr int U = 1;
#endif
    const long rstep = (long)gridDim.x * rpb;
    for (long row0 = (long)blockIdx.x * rpb + rib; row0 < rows; row0 += U * rstep) {
        float gv[U][kMaxCh][E], xh[U][kMaxCh][E], rsv[U], muv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * rstep;
            const bool live = row < rows;
            rsv[u] = live ? rstd[row] : 0.f;
            muv[u] = (live && !xnorm) ? mean[row] : 0.f;
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j) {
                const int ci = lg + G * j;
                if (ci < nch && live) {
                    ld_chunk_nt(g + row * C + ci * E, gv[u][j]);
                    ld_chunk(x + row * C + ci * E, xh[u][j]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = row0 + u * rstep;
            if (row >= rows) break;
            const float rs = rsv[u];
            const float mu = muv[u];
            const float sc = xnorm ? 1.f : rs;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j) {
                const int ci = lg + G * j;
                if (ci < nch) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        xh[u][j][e] = (xh[u][j][e] - mu) * sc;
                        if constexpr (AFF) {
                            aw[j][e] = fmaf(gv[u][j][e], xh[u][j][e], aw[j][e]);
                            ab[j][e] += gv[u][j][e];
                            gv[u][j][e] *= wv[j][e];
                        }
                        s1 += gv[u][j][e];
                        s2 = fmaf(gv[u][j][e], xh[u][j][e], s2);
                    }
                }
            }
            s1 = group_sum<G>(s1) * invC;
            s2 = group_sum<G>(s2) * invC;
#pragma unroll
            for (int j = 0; j < kMaxCh; ++j) {
                const int ci = lg + G * j;
                if (ci < nch) {
                    float o[E];
                    if (dres) {
                        ld_chunk(dres + row * C + ci * E, o);
                    } else {
#pragma unroll
                        for (int e = 0; e < E; ++e) o[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] += rs * (gv[u][j][e] - s1 - xh[u][j][e] * s2);
                    st_chunk(dx + row * C + ci * E, o);
                    if (dx2) {      // second output: the stored value times a per-sample scale (the consumer block's DropPath factor)
                        const float sc2 = scale2[row / rows_per_scale];
#pragma unroll
                        for (int e = 0; e < E; ++e) o[e] = elt<T>::round(o[e]) * sc2;
                        st_chunk(dx2 + row * C + ci * E, o);
                    }
                }
            }
        }
    }
    if (AFF && dw) {  // reduce the per-thread column partials over the rpb row slots of this workgroup
        float* rw = red;
        float* rb = red + rpb * C;
#pragma unroll
        for (int j = 0; j < kMaxCh; ++j) {
            const int ci = lg + G * j;
            if (ci < nch) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    rw[rib * C + ci * E + e] = aw[j][e];
                    rb[rib * C + ci * E + e] = ab[j][e];
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f, bb = 0.f;
            for (int r = 0; r < rpb; ++r) {
                a += rw[r * C + c];
                bb += rb[r * C + c];
            }
            atomicAdd(dw + c, a);
            if (db) atomicAdd(db + c, bb);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm: finalize (C threads), affine apply, backward reduce/apply
// ------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* sum, const float* sumsq, float n, const float* w, const float* b,
                                   float eps, float momentum, float* rmean, float* rvar, float* mean_out,
                                   float* rstd_out, float* scale, float* shift, int C, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mu, var;
    if (training) {
        mu = sum[c] / n;
        var = fmaxf(sumsq[c] / n - mu * mu, 0.f);
        if (rmean) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
        }
    } else {
        mu = rmean[c];
        var = rvar[c];
    }
    const float rs = rsqrtf(var + eps);
    if (mean_out) mean_out[c] = mu;
    if (rstd_out) rstd_out[c] = rs;
    const float a = (w ? w[c] : 1.f) * rs;
    scale[c] = a;
    shift[c] = (b ? b[c] : 0.f) - mu * a;
}

// walk over the 8-element chunks i = i0, i0 + S, ... of a [rows][C] tensor keeping (row, chunk column) without a
// division per chunk (one at the start; 32-bit integer division costs ~20 VALU instructions)
struct ChunkWalk {
    unsigned i, row, col, P, drow, dcol, step;
    __device__ ChunkWalk(unsigned i0, unsigned stride, unsigned P_) : i(i0), P(P_), step(stride) {
        row = i0 / P_;
        col = i0 - row * P_;
        drow = stride / P_;
        dcol = stride - drow * P_;
    }
    __device__ void next() {
        i += step;
        row += drow;
        col += dcol;
        if (col >= P) { col -= P; ++row; }
    }
};
// row / rps for row < 2^24 (per-sample DropPath scale lookup)
__device__ __forceinline__ unsigned div_small(unsigned row, unsigned rps, float inv_rps) {
    unsigned q = (unsigned)((float)row * inv_rps);
    const int r = (int)(row - q * rps);
    if (r < 0) --q;
    else if (r >= (int)rps) ++q;
    return q;
}

// y = x * scale[c] + shift[c] (+ res) (relu); 8 channels per thread; per-channel coefficients staged in LDS once
template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res,
                                                         const float* __restrict__ rowscale, long elems_per_scale,
                                                         T* __restrict__ y, long n8, int C, int relu, long ldx) {
    extern __shared__ __attribute__((aligned(16))) float coef[];      // [2][C]: scale, shift
    if (scale) {
        for (int c = threadIdx.x; c < C; c += 256) {
            coef[c] = scale[c];
            coef[C + c] = shift[c];
        }
        __syncthreads();
    }
    const unsigned rps = (unsigned)(elems_per_scale / C);
    const float inv_rps = 1.0f / (float)rps;
    for (ChunkWalk w(blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u, (unsigned)C >> 3); w.i < (unsigned)n8; w.next()) {
        const long e = (long)w.i * 8;
        const int c = (int)w.col << 3;
        const float rsc = rowscale ? rowscale[div_small(w.row, rps, inv_rps)] : 1.f;
        float v[8], r[8], sc[8], sh[8];
        load8(x + (long)w.row * ldx + c, v);          // x may be a column slice of a wider matrix (row stride ldx)
        if (res) load8(res + e, r);
        if (scale) {
            load8(coef + c, sc);
            load8(coef + C + c, sh);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j];
            if (scale) t = fmaf(t, sc[j], sh[j]);
            t *= rsc;
            if (res) t += r[j];
            if (relu) t = fmaxf(t, 0.f);
            v[j] = t;
        }
        store8(y + e, v);
    }
}

// column sums s1[c] += sum g, s2[c] += sum g*xhat with g = dy * (yrelu > 0); persistent workgroups,
// thread = (4-channel group, row slot); blockIdx.y = 64-channel slice
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ yrelu,
                                                            const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ rowscale,
                                                            long rows_per_scale, float* __restrict__ s1,
                                                            float* __restrict__ s2, long rows, int C, long ldx) {
    __shared__ float red[2][16][64];
    const int c0 = blockIdx.y * 64;
    const int cgs = min(64, C - c0) >> 2;
    const int cg = threadIdx.x & 15, slot = threadIdx.x >> 4;
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    if (cg < cgs) {
        float mu[4], rs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mu[e] = mean[c0 + cg * 4 + e];
            rs[e] = rstd[c0 + cg * 4 + e];
        }
        for (long row = (long)blockIdx.x * 16 + slot; row < rows; row += (long)gridDim.x * 16) {
            const long off = row * C + c0 + cg * 4;
            float g[4], xv[4];
            load4(dy + off, g);
            load4(x + row * ldx + c0 + cg * 4, xv);
            if (yrelu) {
                float yv[4];
                load4(yrelu + off, yv);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
            }
            if (rowscale) {
                const float rsc = rowscale[row / rows_per_scale];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] *= rsc;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a1[e] += g[e];
                a2[e] += g[e] * (xv[e] - mu[e]) * rs[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][slot][cg * 4 + e] = a1[e];
        red[1][slot][cg * 4 + e] = a2[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        if (c < cgs * 4) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[which][r][c];
            atomicAdd((which ? s2 : s1) + c0 + c, s);
        }
    }
}

// dx = w*rstd * (g - s1/n - xhat * s2/n), g = dy * (yrelu > 0)  ==  A[c]*g + B[c]*x + D[c] with the per-channel
// A = w*rstd, B = -A*rstd*s2/n, D = -A*s1/n (and the mean: x - mean is formed first, no cancellation against D)
// staged in LDS once per workgroup:  dx = A*g + B*(x - mean) + D
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ yrelu,
                                                           const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ w,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           const float* __restrict__ rowscale, long elems_per_scale,
                                                           float inv_n, T* __restrict__ dx, long n8, int C, long ldx,
                                                           long lddx) {
    extern __shared__ __attribute__((aligned(16))) float coef[];      // [4][C]
    for (int c = threadIdx.x; c < C; c += 256) {
        const float rs = rstd[c];
        const float A = (w ? w[c] : 1.f) * rs;
        const float Bc = -A * rs * s2[c] * inv_n;
        coef[c] = A;
        coef[C + c] = Bc;
        coef[2 * C + c] = -A * s1[c] * inv_n;
        coef[3 * C + c] = mean[c];
    }
    __syncthreads();
    const unsigned rps = (unsigned)(elems_per_scale / C);
    const float inv_rps = 1.0f / (float)rps;
    for (ChunkWalk wk(blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u, (unsigned)C >> 3); wk.i < (unsigned)n8; wk.next()) {
        const long e = (long)wk.i * 8;
        const int c = (int)wk.col << 3;
        const float rsc = rowscale ? rowscale[div_small(wk.row, rps, inv_rps)] : 1.f;
        float g[8], xv[8], yv[8], A[8], Bc[8], D[8], mu[8];
        load8(dy + e, g);
        load8(x + (long)wk.row * ldx + c, xv);        // x / dx may be column slices of wider matrices
        if (yrelu) load8(yrelu + e, yv);
        load8(coef + c, A);
        load8(coef + C + c, Bc);
        load8(coef + 2 * C + c, D);
        load8(coef + 3 * C + c, mu);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gg = g[j] * rsc;
            if (yrelu && !(yv[j] > 0.f)) gg = 0.f;
            g[j] = fmaf(A[j], gg, fmaf(Bc[j], xv[j] - mu[j], D[j]));
        }
        store8(dx + (long)wk.row * lddx + c, g);
    }
}

}  // namespace

// chunks per lane: 3 for the C = 96 * 2^k of this model family (12 / 24 / 48 / 96 chunks over 4 / 8 / 16 / 32 lanes), else 4
#define LN_DISPATCH__(G, AFF, NCH, KERNEL, ...)                                                  \
    switch (G) {                                                                                 \
        case 4: hipLaunchKernelGGL((KERNEL<T, 4, AFF, NCH>), __VA_ARGS__); break;                \
        case 8: hipLaunchKernelGGL((KERNEL<T, 8, AFF, NCH>), __VA_ARGS__); break;                \
        case 16: hipLaunchKernelGGL((KERNEL<T, 16, AFF, NCH>), __VA_ARGS__); break;              \
        case 32: hipLaunchKernelGGL((KERNEL<T, 32, AFF, NCH>), __VA_ARGS__); break;              \
        default: hipLaunchKernelGGL((KERNEL<T, 64, AFF, NCH>), __VA_ARGS__); break;              \
    }
#define LN_DISPATCH_(G, AFF, nch3, KERNEL, ...)                                                                  \
    do {                                                                                                         \
        if ((nch3) == 1 && (G) == 16) { hipLaunchKernelGGL((KERNEL<T, 16, AFF, 1>), __VA_ARGS__); }             \
        else if ((nch3) == 8) { hipLaunchKernelGGL((KERNEL<T, 64, AFF, 8>), __VA_ARGS__); }                      \
        else if (nch3) { LN_DISPATCH__(G, AFF, 3, KERNEL, __VA_ARGS__) }                                          \
        else { LN_DISPATCH__(G, AFF, 4, KERNEL, __VA_ARGS__) }                                                   \
    } while (0)
#define LN_DISPATCH(G, aff, nch3, KERNEL, ...)                            \
    do {                                                                  \
        if (aff) LN_DISPATCH_(G, true, nch3, KERNEL, __VA_ARGS__);        \
        else LN_DISPATCH_(G, false, nch3, KERNEL, __VA_ARGS__);           \
    } while (0)

template <typename T>
static int ln_fwd_t(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t rows,
                    int C, float eps, hipStream_t s) {
    const int G = pick_group(C, elt<T>::EPC);
    const int rpb = 256 / G;
    dim3 grid(grid_blocks(rows, rpb, 4096)), block(256);
    // 1: one chunk per lane (the 12 chunks of C = 96 on 16 lanes: a third of the registers of the 3-chunk form, which
    // matters for the affine variants -- stem / downsample norms), 2: up to 3 chunks, 0: up to 4
    const int nch3 = C / elt<T>::EPC <= G ? 1 : (C / elt<T>::EPC <= 3 * G ? 2 : (C / elt<T>::EPC <= 4 * G ? 0 : 8));
    LN_DISPATCH(G, w != nullptr, nch3, ln_fwd_kernel, grid, block, 0, s, (const T*)x, w, b, (T*)y, mean, rstd, (long)rows, C, eps);
    return ga_check_launch("ga_layernorm_fwd");
}

template <typename T>
static int ln_bwd_t(const void* g, const void* x, const float* mean, const float* rstd, const float* w,
                    const void* dres, void* dx, float* dw, float* db, int64_t rows, int C, int xnorm, hipStream_t s,
                    void* dx2 = nullptr, const float* scale2 = nullptr, int64_t rows_per_scale = 1) {
    const int G = pick_group(C, elt<T>::EPC);
    const int rpb = 256 / G;
    // persistent when parameter gradients are reduced: every workgroup ends with 2 C atomics onto the same 2 C addresses, so the
    // workgroup count trades bytes in flight against contended atomics (tools/r03/r03_lnb.sh, B = 256 stage shapes, ms at
    // 1024 / 512 / 256 workgroups: C = 96 0.114 / 0.145 / 0.241; C = 192 0.065 / 0.052 / 0.071; C = 384 0.046 / 0.037 / 0.042;
    // C = 768 0.040 / 0.027 / 0.028)
    const int wgs_knob = GA_KNOB("LN_BWD_WGS", 0);
    const int wgs = wgs_knob > 0 ? std::max(64, wgs_knob) : (C <= 128 ? 1024 : 512);
    dim3 grid(grid_blocks(rows, rpb, dw ? wgs : 4096)), block(256);
    const size_t lds = dw ? (size_t)2 * rpb * C * sizeof(float) : 0;
    const int nch3 = C / elt<T>::EPC <= G ? 1 : (C / elt<T>::EPC <= 3 * G ? 2 : (C / elt<T>::EPC <= 4 * G ? 0 : 8));
    LN_DISPATCH(G, (w != nullptr || dw != nullptr), nch3, ln_bwd_kernel, grid, block, lds, s, (const T*)g, (const T*)x, mean, rstd, w, (const T*)dres, (T*)dx,
                dw, db, (long)rows, C, xnorm, (T*)dx2, scale2, (long)rows_per_scale);
    return ga_check_launch("ga_layernorm_bwd");
}

extern "C" int ga_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                int64_t rows, int C, float eps, int dtype, ga_stream_t stream) {
    const int epc = dtype == GA_BF16 ? 8 : 4;   // whole 16-byte chunks
    GA_REQUIRE(x && y && rows > 0 && C > 0 && C % epc == 0 && pick_group(C, epc) > 0, "ga_layernorm_fwd: C=%d unsupported", C);
    GA_REQUIRE((w == nullptr) == (b == nullptr), "ga_layernorm_fwd: w and b must both be given or both NULL");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? ln_fwd_t<bf16_t>(x, w, b, y, mean, rstd, rows, C, eps, s)
                            : ln_fwd_t<float>(x, w, b, y, mean, rstd, rows, C, eps, s);
}

extern "C" int ga_layernorm_bwd(const void* g, const void* x, const float* mean, const float* rstd, const float* w,
                                const void* dres, void* dx, float* dw, float* db, int64_t rows, int C,
                                int x_is_normalized, int dtype, ga_stream_t stream) {
    const int epc = dtype == GA_BF16 ? 8 : 4;
    GA_REQUIRE(g && x && rstd && dx && rows > 0 && C % epc == 0 && pick_group(C, epc) > 0, "ga_layernorm_bwd: bad args");
    GA_REQUIRE(x_is_normalized || mean, "ga_layernorm_bwd: mean required");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? ln_bwd_t<bf16_t>(g, x, mean, rstd, w, dres, dx, dw, db, rows, C, x_is_normalized, s)
                            : ln_bwd_t<float>(g, x, mean, rstd, w, dres, dx, dw, db, rows, C, x_is_normalized, s);
}

extern "C" int ga_layernorm_bwd_dp(const void* g, const void* x, const float* mean, const float* rstd, const float* w,
                                   const void* dres, void* dx, float* dw, float* db, int64_t rows, int C,
                                   int x_is_normalized, void* dx2, const float* scale2, int64_t rows_per_scale, int dtype,
                                   ga_stream_t stream) {
    const int epc = dtype == GA_BF16 ? 8 : 4;
    GA_REQUIRE(g && x && rstd && dx && dx2 && scale2 && rows_per_scale > 0 && rows > 0 && C % epc == 0 && pick_group(C, epc) > 0,
               "ga_layernorm_bwd_dp: bad args");
    GA_REQUIRE(x_is_normalized || mean, "ga_layernorm_bwd_dp: mean required");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? ln_bwd_t<bf16_t>(g, x, mean, rstd, w, dres, dx, dw, db, rows, C, x_is_normalized, s, dx2, scale2, rows_per_scale)
                            : ln_bwd_t<float>(g, x, mean, rstd, w, dres, dx, dw, db, rows, C, x_is_normalized, s, dx2, scale2, rows_per_scale);
}

extern "C" int ga_bn_finalize(const float* sum, const float* sumsq, int64_t n, const float* w, const float* b,
                              float eps, float momentum, float* running_mean, float* running_var, float* mean_out,
                              float* rstd_out, float* scale, float* shift, int C, int training, ga_stream_t stream) {
    GA_REQUIRE(scale && shift && C > 0 && (training ? (sum && sumsq && n > 0) : (running_mean && running_var)),
               "ga_bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), sum,
                       sumsq, (float)n, w, b, eps, momentum, running_mean, running_var, mean_out, rstd_out, scale, shift,
                       C, training);
    return ga_check_launch("ga_bn_finalize");
}

extern "C" int ga_affine_act(const void* x, const float* scale, const float* shift, const void* res,
                             const float* rowscale, int64_t rows_per_scale, void* y, int64_t rows, int C, int relu,
                             int dtype, int64_t ldx, ga_stream_t stream) {
    if (ldx == 0) ldx = C;
    GA_REQUIRE(ldx >= C && ldx % 8 == 0, "ga_affine_act: ldx must be a multiple of 8 and >= C");
    GA_REQUIRE(x && y && rows > 0 && C % 8 == 0 && ((scale == nullptr) == (shift == nullptr)) &&
                   rows * C / 8 < (1L << 31), "ga_affine_act: bad args");
    GA_REQUIRE(C <= 8192, "ga_affine_act: C=%d too large for the LDS coefficient table", C);
    const long n8 = rows * C / 8;
    dim3 grid(grid_blocks(n8, 256, 2048)), block(256);
    const size_t lds = (size_t)2 * C * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(affine_act_kernel<bf16_t>, grid, block, lds, s, (const bf16_t*)x, scale, shift,
                           (const bf16_t*)res, rowscale, (long)rows_per_scale * C, (bf16_t*)y, n8, C, relu, (long)ldx);
    else
        hipLaunchKernelGGL(affine_act_kernel<float>, grid, block, lds, s, (const float*)x, scale, shift, (const float*)res,
                           rowscale, (long)rows_per_scale * C, (float*)y, n8, C, relu, (long)ldx);
    return ga_check_launch("ga_affine_act");
}

extern "C" int ga_bn_bwd_reduce(const void* dy, const void* y_relu, const void* x, const float* mean, const float* rstd,
                                const float* rowscale, int64_t rows_per_scale, float* s1, float* s2, int64_t rows, int C,
                                int dtype, int64_t ldx, ga_stream_t stream) {
    if (ldx == 0) ldx = C;
    GA_REQUIRE(ldx >= C && ldx % 4 == 0, "ga_bn_bwd_reduce: ldx must be a multiple of 4 and >= C");
    GA_REQUIRE(dy && x && mean && rstd && s1 && s2 && rows > 0 && C % 4 == 0, "ga_bn_bwd_reduce: bad args");
    const int slices = cdiv(C, 64);
    dim3 grid(std::max(1, std::min<int>((int)((rows + 15) / 16), std::max(1, 1024 / slices))), slices), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)dy, (const bf16_t*)y_relu,
                           (const bf16_t*)x, mean, rstd, rowscale, (long)rows_per_scale, s1, s2, (long)rows, C, (long)ldx);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, block, 0, s, (const float*)dy, (const float*)y_relu,
                           (const float*)x, mean, rstd, rowscale, (long)rows_per_scale, s1, s2, (long)rows, C, (long)ldx);
    return ga_check_launch("ga_bn_bwd_reduce");
}

extern "C" int ga_bn_bwd_apply(const void* dy, const void* y_relu, const void* x, const float* mean, const float* rstd,
                               const float* w, const float* s1, const float* s2, const float* rowscale,
                               int64_t rows_per_scale, int64_t n, void* dx, int64_t rows, int C, int dtype,
                               int64_t ldx, int64_t lddx, ga_stream_t stream) {
    if (ldx == 0) ldx = C;
    if (lddx == 0) lddx = C;
    GA_REQUIRE(ldx >= C && lddx >= C && ldx % 8 == 0 && lddx % 8 == 0, "ga_bn_bwd_apply: ldx / lddx must be multiples of 8 and >= C");
    GA_REQUIRE(dy && x && mean && rstd && s1 && s2 && dx && rows > 0 && C % 8 == 0 && n > 0 && rows * C / 8 < (1L << 31),
               "ga_bn_bwd_apply: bad args");
    GA_REQUIRE(C <= 4096, "ga_bn_bwd_apply: C=%d too large for the LDS coefficient table", C);
    const long n8 = rows * C / 8;
    dim3 grid(grid_blocks(n8, 256, 2048)), block(256);
    const size_t lds = (size_t)4 * C * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, block, lds, s, (const bf16_t*)dy, (const bf16_t*)y_relu,
                           (const bf16_t*)x, mean, rstd, w, s1, s2, rowscale, (long)rows_per_scale * C, 1.f / (float)n,
                           (bf16_t*)dx, n8, C, (long)ldx, (long)lddx);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, block, lds, s, (const float*)dy, (const float*)y_relu,
                           (const float*)x, mean, rstd, w, s1, s2, rowscale, (long)rows_per_scale * C, 1.f / (float)n,
                           (float*)dx, n8, C, (long)ldx, (long)lddx);
    return ga_check_launch("ga_bn_bwd_apply");
}
