// Fused MLP bodies for the narrow stages (bf16, C = 96 / 192, hidden H = 4C): the 128 x H hidden tile never goes to HBM.
//
//   ga_mlp_fwd:  Y = R + rowscale * (gelu(X W1^T + b1) W2^T + b2)                  (Block.forward, ga_convnext.py:86-101:
//                pwconv1 -> GELU -> pwconv2 -> gamma -> DropPath -> residual; LayerNorm affine and gamma are folded into
//                the effective weights by ga_weight_prep)
//   ga_mlp_bwd:  re-computes Hd = X W1^T + b1, then A = gelu(Hd), DH = (DY W2) * gelu'(Hd), DX = DH W1;
//                writes A and DH once (the weight-gradient GEMMs read them) and DX.
//
// Why: at C = 96 the unfused fc1 / fc2 / dgrad2 / dgrad1 GEMMs have 40-60 FLOP per HBM byte and run at the HBM rate
// (4.0-4.4 TB/s, 170-390 TFLOP/s); the hidden activation and its GELU' (8C per token written, 8C read back) are 3/4 of
// the forward traffic of a block.  Here a workgroup owns 128 token rows: X (and DY) stay in LDS, the hidden dimension is
// walked in chunks of HC columns, each chunk's weights come from L2 through registers (prefetched one chunk ahead),
// and the chunk of the hidden tile lives in accumulators / one LDS tile between the two chained MFMA products.
//
// LDS operand images are "k slabs": [rows][32 bf16] = 64-byte rows, the 16-byte unit c of row r at c ^ ((r >> 2) & 3),
// so the ds_read_b128 of a 16x16x32 fragment (16 rows x one unit per 16-lane group) covers all 16 slots of the 256-byte
// bank row.  The weight fragment is the MFMA's first operand: a lane then holds 4 CONSECUTIVE hidden / output columns
// of ONE token row, i.e. an 8-byte piece of the next product's row-major operand tile.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace {

constexpr int BM = 128;

// workgroup barrier that orders LDS traffic only: the weight pieces of the NEXT chunk stay in flight across it
// (__syncthreads() would wait for vmcnt(0) as well and expose their L2 latency once per chunk)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned slab_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ bf16x8_t frag(const unsigned char* slab, int r0, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(slab + slab_off(r0 + (lane & 15), lane >> 4));
}

// global rows [row0, row0 + ROWS) x (32 * KS) bf16 -> KS slabs of [ROWS][64 B]; rows >= limit are zero.  All the loads of a
// thread are issued before the first LDS write (one exposed latency, not one per piece).
template <int ROWS, int KS, int NTHR>
__device__ __forceinline__ void stage_rows(unsigned char* dst, const bf16_t* src, long ld, long row0, long limit) {
    constexpr int CPR = KS * 4, NP = ROWS * CPR / NTHR;
    static_assert(ROWS * CPR % NTHR == 0, "tile must split evenly over the threads");
    u32x4_t v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = threadIdx.x + NTHR * i;
        const int r = idx / CPR, c = idx - r * CPR;
        v[i] = u32x4_t{0, 0, 0, 0};
        if (row0 + r < limit) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src + (row0 + r) * ld + c * 8));
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int idx = threadIdx.x + NTHR * i;
        const int r = idx / CPR, c = idx - r * CPR;
        *reinterpret_cast<u32x4_t*>(dst + (c >> 2) * ROWS * 64 + slab_off(r, c & 3)) = v[i];
    }
}

// the residual pieces a thread adds in store_rows, requested ahead (during the last chunk) so that their HBM latency is not
// paid row by row in the store loop
template <int C, int NTHR> struct RowPre {
    static constexpr int P8 = C / 8, RG = NTHR / P8, NR = (64 + RG - 1) / RG;
};
template <int C, int NTHR> using PreArr = u32x4_t[BM / 64][RowPre<C, NTHR>::NR];
template <int C, int NTHR>
__device__ __forceinline__ void rows_prefetch(PreArr<C, NTHR>& v, const bf16_t* R, long ldr, long m0, long M) {
    using P = RowPre<C, NTHR>;
    const int c8 = threadIdx.x % P::P8, rg = threadIdx.x / P::P8;
#pragma unroll
    for (int piece = 0; piece < BM / 64; ++piece)
#pragma unroll
        for (int it = 0; it < P::NR; ++it) {
            const int row = rg + it * P::RG;
            const long m = m0 + piece * 64 + row;
            v[piece][it] = u32x4_t{0, 0, 0, 0};
            if (rg < P::RG && row < 64 && m < M) v[piece][it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(R + m * ldr + c8 * 8));
        }
}

// fp32 accumulator tiles -> rows of Y through an LDS staging piece of 64 rows x (C + 4) floats, epilogue fused:
//   y = R + rowscale * (acc + bias)
// acc[tn][tm]: token row 16 tm + (lane & 15) of the wave's WROWS rows, columns wn * (C/2) + 16 tn + 4 (lane >> 4) .. +3
template <int C, int NTHR, int NWM, int TM>
__device__ __forceinline__ void store_rows(float* Cs, const f32x4_t (&acc)[C / 32][TM], int wm, int wn, int lane, long m0, long M,
                                           const float* bias, const float* rowscale, int rps, bool has_r,
                                           const PreArr<C, NTHR>& pre, bf16_t* Y, long ldy) {
    constexpr int LDC = C + 4, P8 = C / 8, RG = NTHR / P8, WROWS = 16 * TM, WPH = 64 / WROWS;   // waves (in m) per 64-row piece
    constexpr int NR = (64 + RG - 1) / RG;
    const int c8 = threadIdx.x % P8, rg = threadIdx.x / P8;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[c8 * 8 + j] : 0.f;
#pragma unroll
    for (int piece = 0; piece < NWM / WPH; ++piece) {
        if (piece) __syncthreads();
        if (wm / WPH == piece) {
#pragma unroll
            for (int tn = 0; tn < C / 32; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row = (wm % WPH) * WROWS + 16 * tm + (lane & 15), col = wn * (C / 2) + 16 * tn + 4 * (lane >> 4);
                    *reinterpret_cast<f32x4_t*>(Cs + row * LDC + col) = acc[tn][tm];
                }
        }
        __syncthreads();
        if (rg < RG) {
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int row = rg + it * RG;
                const long m = m0 + piece * 64 + row;
                if (row >= 64 || m >= M) break;
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(Cs + row * LDC + c8 * 8);
                const f32x4_t b = *reinterpret_cast<const f32x4_t*>(Cs + row * LDC + c8 * 8 + 4);
                float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += bv[j];
                if (rowscale) {
                    const float s = rowscale[m / rps];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= s;
                }
                if (has_r) {
                    float r[8];
                    const u32x4_t q = pre[piece][it];
                    unpack8(make_uint4(q.x, q.y, q.z, q.w), r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += r[j];
                }
                store8_nt(Y + m * ldy + c8 * 8, v);
            }
        }
    }
}

// =================================================================================================================
// forward.  256 threads = 4 waves as 2 (tokens) x 2 (columns); chunk of HC hidden columns per round:
//   P1: h[128 x HC] = Xs . W1s^T (K = C)   -> + b1, GELU, bf16 -> As      (wave: 64 rows x HC/2)
//   P2: y[128 x C] += As . W2s^T (K = HC)                                   (wave: 64 rows x C/2)
// two barriers per round; the next chunk's weights travel global -> registers during P1 / P2
// =================================================================================================================
template <int C, int HC>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(const ga_mlp_desc d) {
    constexpr int KS1 = C / 32, KS2 = HC / 32, TN1 = HC / 32, TN2 = C / 32;
    constexpr int XS = KS1 * BM * 64, W1S = KS1 * HC * 64, AS = KS2 * BM * 64;
    constexpr int NI = HC * C / 8 / 256;                  // 16-byte weight pieces per thread and matrix
    static_assert(HC * C / 8 % 256 == 0, "weight chunk must split evenly");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* W1s = Xs + XS;
    unsigned char* As = W1s + W1S;
    unsigned char* W2s = As + AS;
    float* B1s = reinterpret_cast<float*>(W2s + KS2 * C * 64);    // [4C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, g = lane >> 4;
    const long m0 = (long)blockIdx.x * BM;
    const bf16_t* W1 = reinterpret_cast<const bf16_t*>(d.W1);
    const bf16_t* W2 = reinterpret_cast<const bf16_t*>(d.W2);

    u32x4_t w1r[NI], w2r[NI];
    auto w_load = [&](int j0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + 256 * i;
            const int r1 = idx / (KS1 * 4), c1 = idx - r1 * (KS1 * 4);
            w1r[i] = *reinterpret_cast<const u32x4_t*>(W1 + (long)(j0 + r1) * d.ldw1 + c1 * 8);
            const int r2 = idx / (HC / 8), c2 = idx - r2 * (HC / 8);
            w2r[i] = *reinterpret_cast<const u32x4_t*>(W2 + (long)r2 * d.ldw2 + j0 + c2 * 8);
        }
    };
    auto w1_store = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + 256 * i;
            const int r1 = idx / (KS1 * 4), c1 = idx - r1 * (KS1 * 4);
            *reinterpret_cast<u32x4_t*>(W1s + (c1 >> 2) * HC * 64 + slab_off(r1, c1 & 3)) = w1r[i];
        }
    };
    auto w2_store = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = tid + 256 * i;
            const int r2 = idx / (HC / 8), c2 = idx - r2 * (HC / 8);
            *reinterpret_cast<u32x4_t*>(W2s + (c2 >> 2) * C * 64 + slab_off(r2, c2 & 3)) = w2r[i];
        }
    };

    w_load(0);
    stage_rows<BM, KS1, 256>(Xs, reinterpret_cast<const bf16_t*>(d.X), d.ldx, m0, d.M);
    for (int i = tid; i < 4 * C; i += 256) B1s[i] = d.b1[i];
    w1_store();
    w2_store();
    __syncthreads();

    f32x4_t y[TN2][4];
#pragma unroll
    for (int i = 0; i < TN2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) y[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // one round = one chunk of HC hidden columns; the last one (more = false) runs after the residual prefetch
    auto round = [&](const int j0, const bool more) __attribute__((always_inline)) {
        if (more) w_load(j0 + HC);
        // ---- P1
        f32x4_t h[TN1][4];
#pragma unroll
        for (int i = 0; i < TN1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) h[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            bf16x8_t xf[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) xf[tm] = frag(Xs + ks * BM * 64, wm * 64 + 16 * tm, lane);
#pragma unroll
            for (int tn = 0; tn < TN1; ++tn) {
                const bf16x8_t wf = frag(W1s + ks * HC * 64, wn * (HC / 2) + 16 * tn, lane);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) h[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[tm], h[tn][tm], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN1; ++tn) {
            const int nl = wn * (HC / 2) + 16 * tn + 4 * g;              // first of this lane's 4 hidden columns in the chunk
            const float4 b = *reinterpret_cast<const float4*>(B1s + j0 + nl);
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                float a[4], unused;
#pragma unroll
                for (int r = 0; r < 4; ++r) gelu_both_fast(h[tn][tm][r] + bb[r], a[r], unused);
                const int row = wm * 64 + 16 * tm + (lane & 15);
                uint2 p;
                p.x = pack2bf(a[0], a[1]);
                p.y = pack2bf(a[2], a[3]);
                *reinterpret_cast<uint2*>(As + (nl >> 5) * BM * 64 + slab_off(row, (nl & 31) >> 3) + (nl & 7) * 2) = p;
            }
        }
        lds_barrier();                       // As complete; W1s is free
        // ---- P2
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            bf16x8_t af[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) af[tm] = frag(As + ks * BM * 64, wm * 64 + 16 * tm, lane);
#pragma unroll
            for (int tn = 0; tn < TN2; ++tn) {
                const bf16x8_t wf = frag(W2s + ks * C * 64, wn * (C / 2) + 16 * tn, lane);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) y[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[tm], y[tn][tm], 0, 0, 0);
            }
        }
        if (more) w1_store();                // its pieces had P1 + P2 to arrive from L2
        lds_barrier();                       // As / W2s are free, the new W1s is visible to everyone
        if (more) w2_store();                // read by P2 of the next round, after its first barrier
    };
    for (int j0 = 0; j0 + HC < d.H; j0 += HC) round(j0, true);
    u32x4_t rpre[BM / 64][RowPre<C, 256>::NR];
    if (d.R) rows_prefetch<C, 256>(rpre, reinterpret_cast<const bf16_t*>(d.R), d.ldr, m0, d.M);
    round(d.H - HC, false);
    // (the last barrier of the round has passed: every LDS operand is dead)
    store_rows<C, 256, 2, 4>(reinterpret_cast<float*>(smem), y, wm, wn, lane, m0, d.M, d.b2, d.rowscale, d.rows_per_scale,
                             d.R != nullptr, rpre, reinterpret_cast<bf16_t*>(d.Y), d.ldy);
}

template <int C, int HC> constexpr int fwd_lds() {
    constexpr int ab = (C / 32) * BM * 64 + (C / 32) * HC * 64 + (HC / 32) * BM * 64 + (HC / 32) * C * 64 + 16 * C;
    constexpr int st = 64 * (C + 4) * 4;
    return ab > st ? ab : st;
}

// =================================================================================================================
// backward.  512 threads = 8 waves as 4 (tokens) x 2 (columns), one workgroup per CU (X and DY tiles both resident);
// per chunk of HC hidden columns:
//   P1: h = Xs . W1s^T,  t = Ds . W2Ts^T   (K = C; wave: 32 rows x HC/2 each)
//       a = gelu(h + b1) -> A2s,  dh = t * gelu'(h + b1) -> DHs
//   P2: dx[128 x C] += DHs . W1Ts^T (K = HC; wave: 32 rows x C/2);  A2s / DHs rows -> global with 16-byte stores
// (measured at C = 96, M = 802816: 0.54 ms; a 4-wave / two-workgroup form writing a / dh as 8-byte pieces straight from the
//  accumulators: 1.06 ms -- the 32-byte runs cost more than the LDS staging)
// =================================================================================================================
template <int C, int HC>
__global__ __launch_bounds__(512, 1) void mlp_bwd_kernel(const ga_mlp_bwd_desc d) {
    constexpr int KS1 = C / 32, KS2 = HC / 32, TN1 = HC / 32, TN2 = C / 32;
    constexpr int XS = KS1 * BM * 64, WS = KS1 * HC * 64, TS = KS2 * BM * 64;
    constexpr int NPC = HC * C / 8, NI = (NPC + 511) / 512;     // 16-byte pieces per weight chunk / per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xs = smem;
    unsigned char* Ds = Xs + XS;
    unsigned char* W1s = Ds + XS;
    unsigned char* W2Ts = W1s + WS;
    unsigned char* W1Ts = W2Ts + WS;          // [C][HC]: KS2 slabs of C rows
    unsigned char* DHs = W1Ts + KS2 * C * 64;
    unsigned char* A2s = DHs + TS;
    float* B1s = reinterpret_cast<float*>(A2s + TS);               // [4C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, g = lane >> 4;
    const long m0 = (long)blockIdx.x * BM;
    const bf16_t* W1 = reinterpret_cast<const bf16_t*>(d.W1);
    const bf16_t* W2T = reinterpret_cast<const bf16_t*>(d.W2T);
    const bf16_t* W1T = reinterpret_cast<const bf16_t*>(d.W1T);
    bf16_t* Aout = reinterpret_cast<bf16_t*>(d.A);
    bf16_t* DHout = reinterpret_cast<bf16_t*>(d.DH);

    u32x4_t w1r[NI], w2r[NI], w3r[NI];
    auto w_load = [&](int j0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = NPC % 512 == 0 ? tid + 512 * i : min(tid + 512 * i, NPC - 1);   // clamped: a few pieces twice
            const int r1 = idx / (KS1 * 4), c1 = idx - r1 * (KS1 * 4);
            w1r[i] = *reinterpret_cast<const u32x4_t*>(W1 + (long)(j0 + r1) * d.ldw1 + c1 * 8);
            w2r[i] = *reinterpret_cast<const u32x4_t*>(W2T + (long)(j0 + r1) * d.ldw2t + c1 * 8);
            const int r3 = idx / (HC / 8), c3 = idx - r3 * (HC / 8);
            w3r[i] = *reinterpret_cast<const u32x4_t*>(W1T + (long)r3 * d.ldw1t + j0 + c3 * 8);
        }
    };
    auto w12_store = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = NPC % 512 == 0 ? tid + 512 * i : min(tid + 512 * i, NPC - 1);
            const int r1 = idx / (KS1 * 4), c1 = idx - r1 * (KS1 * 4);
            const unsigned o = (c1 >> 2) * HC * 64 + slab_off(r1, c1 & 3);
            *reinterpret_cast<u32x4_t*>(W1s + o) = w1r[i];
            *reinterpret_cast<u32x4_t*>(W2Ts + o) = w2r[i];
        }
    };
    auto w3_store = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = NPC % 512 == 0 ? tid + 512 * i : min(tid + 512 * i, NPC - 1);
            const int r3 = idx / (HC / 8), c3 = idx - r3 * (HC / 8);
            *reinterpret_cast<u32x4_t*>(W1Ts + (c3 >> 2) * C * 64 + slab_off(r3, c3 & 3)) = w3r[i];
        }
    };

    w_load(0);
    stage_rows<BM, KS1, 512>(Xs, reinterpret_cast<const bf16_t*>(d.X), d.ldx, m0, d.M);
    stage_rows<BM, KS1, 512>(Ds, reinterpret_cast<const bf16_t*>(d.DY), d.lddy, m0, d.M);
    for (int i = tid; i < 4 * C; i += 512) B1s[i] = d.b1[i];
    w12_store();
    w3_store();
    __syncthreads();

    f32x4_t dx[TN2][2];
#pragma unroll
    for (int i = 0; i < TN2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dx[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int j0 = 0; j0 < d.H; j0 += HC) {
        const bool more = j0 + HC < d.H;
        if (more) w_load(j0 + HC);
        // ---- P1
        f32x4_t h[TN1][2], t[TN1][2];
#pragma unroll
        for (int i = 0; i < TN1; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) h[i][j] = t[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            bf16x8_t xf[2], df[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                xf[tm] = frag(Xs + ks * BM * 64, wm * 32 + 16 * tm, lane);
                df[tm] = frag(Ds + ks * BM * 64, wm * 32 + 16 * tm, lane);
            }
#pragma unroll
            for (int tn = 0; tn < TN1; ++tn) {
                const bf16x8_t w1f = frag(W1s + ks * HC * 64, wn * (HC / 2) + 16 * tn, lane);
                const bf16x8_t w2f = frag(W2Ts + ks * HC * 64, wn * (HC / 2) + 16 * tn, lane);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    h[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f, xf[tm], h[tn][tm], 0, 0, 0);
                    t[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f, df[tm], t[tn][tm], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN1; ++tn) {
            const int nl = wn * (HC / 2) + 16 * tn + 4 * g;
            const float4 b = *reinterpret_cast<const float4*>(B1s + j0 + nl);
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                float a[4], dh[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gr;
                    gelu_both_fast(h[tn][tm][r] + bb[r], a[r], gr);
                    dh[r] = t[tn][tm][r] * gr;
                }
                const int row = wm * 32 + 16 * tm + (lane & 15);
                const unsigned o = (nl >> 5) * BM * 64 + slab_off(row, (nl & 31) >> 3) + (nl & 7) * 2;
                uint2 p;
                p.x = pack2bf(a[0], a[1]);
                p.y = pack2bf(a[2], a[3]);
                *reinterpret_cast<uint2*>(A2s + o) = p;
                p.x = pack2bf(dh[0], dh[1]);
                p.y = pack2bf(dh[2], dh[3]);
                *reinterpret_cast<uint2*>(DHs + o) = p;
            }
        }
        lds_barrier();                       // A2s / DHs complete; W1s / W2Ts are free
        // ---- P2
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            bf16x8_t af[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) af[tm] = frag(DHs + ks * BM * 64, wm * 32 + 16 * tm, lane);
#pragma unroll
            for (int tn = 0; tn < TN2; ++tn) {
                const bf16x8_t wf = frag(W1Ts + ks * C * 64, wn * (C / 2) + 16 * tn, lane);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) dx[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[tm], dx[tn][tm], 0, 0, 0);
            }
        }
        // rows of a / dh of this chunk -> global (HC bf16 = HC/8 16-byte pieces per row)
        {
            constexpr int PPR = HC / 8, NP = BM * PPR / 512;
            static_assert(BM * PPR % 512 == 0, "chunk rows must split evenly");
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int idx = tid + 512 * i;
                const int r = idx / PPR, c = idx - r * PPR;
                const long m = m0 + r;
                if (m < d.M) {
                    const unsigned o = (c >> 2) * BM * 64 + slab_off(r, c & 3);
                    store16_nt(Aout + m * d.lda + j0 + c * 8, *reinterpret_cast<const uint4*>(A2s + o));
                    store16_nt(DHout + m * d.lddh + j0 + c * 8, *reinterpret_cast<const uint4*>(DHs + o));
                }
            }
        }
        if (more) w12_store();
        lds_barrier();                       // A2s / DHs / W1Ts are free, the new W1s / W2Ts visible
        if (more) w3_store();
    }
    u32x4_t nopre[BM / 64][RowPre<C, 512>::NR];
#pragma unroll
    for (int i = 0; i < BM / 64; ++i)
#pragma unroll
        for (int j = 0; j < RowPre<C, 512>::NR; ++j) nopre[i][j] = u32x4_t{0, 0, 0, 0};
    store_rows<C, 512, 4, 2>(reinterpret_cast<float*>(smem), dx, wm, wn, lane, m0, d.M, nullptr, nullptr, 1, false, nopre,
                             reinterpret_cast<bf16_t*>(d.DX), d.lddx);
}

template <int C, int HC> constexpr int bwd_lds() {
    return 2 * (C / 32) * BM * 64 + 2 * (C / 32) * HC * 64 + (HC / 32) * C * 64 + 2 * (HC / 32) * BM * 64 + 16 * C;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename K> bool set_lds(K kern, size_t bytes) {
    return bytes <= 65536 ||
           hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace

extern "C" int ga_mlp_supported(int C, int H, int dtype) { return dtype == GA_BF16 && H == 4 * C && (C == 96 || C == 192); }

extern "C" int ga_mlp_fwd(const ga_mlp_desc* d, ga_stream_t stream) {
    GA_REQUIRE(d && d->X && d->W1 && d->b1 && d->W2 && d->Y && d->M > 0, "ga_mlp_fwd: null / empty descriptor");
    GA_REQUIRE(ga_mlp_supported(d->C, d->H, d->dtype), "ga_mlp_fwd: C=%d H=%d dtype=%d has no fused form (ga_mlp_supported)", d->C,
               d->H, d->dtype);
    GA_REQUIRE(aligned16(d->X) && aligned16(d->W1) && aligned16(d->W2) && aligned16(d->Y) && aligned16(d->b1) &&
                   (!d->R || aligned16(d->R)) && d->ldx % 8 == 0 && d->ldw1 % 8 == 0 && d->ldw2 % 8 == 0 && d->ldy % 8 == 0 &&
                   (!d->R || d->ldr % 8 == 0) && d->ldx >= d->C && d->ldw1 >= d->C && d->ldw2 >= d->H && d->ldy >= d->C,
               "ga_mlp_fwd: operands must be 16-byte aligned with leading dimensions in multiples of 8");
    GA_REQUIRE(!d->rowscale || d->rows_per_scale > 0, "ga_mlp_fwd: rows_per_scale");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((d->M + BM - 1) / BM);
    if (d->C == 96) {
        constexpr int lds = fwd_lds<96, 64>();
        GA_REQUIRE(set_lds(mlp_fwd_kernel<96, 64>, lds), "ga_mlp_fwd: LDS");
        hipLaunchKernelGGL((mlp_fwd_kernel<96, 64>), dim3(grid), dim3(256), lds, s, *d);
    } else {
        constexpr int lds = fwd_lds<192, 32>();
        GA_REQUIRE(set_lds(mlp_fwd_kernel<192, 32>, lds), "ga_mlp_fwd: LDS");
        hipLaunchKernelGGL((mlp_fwd_kernel<192, 32>), dim3(grid), dim3(256), lds, s, *d);
    }
    return ga_check_launch("ga_mlp_fwd");
}

extern "C" int ga_mlp_bwd(const ga_mlp_bwd_desc* d, ga_stream_t stream) {
    GA_REQUIRE(d && d->X && d->DY && d->W1 && d->b1 && d->W2T && d->W1T && d->A && d->DH && d->DX && d->M > 0,
               "ga_mlp_bwd: null / empty descriptor");
    GA_REQUIRE(ga_mlp_supported(d->C, d->H, d->dtype), "ga_mlp_bwd: C=%d H=%d dtype=%d has no fused form (ga_mlp_supported)", d->C,
               d->H, d->dtype);
    GA_REQUIRE(aligned16(d->X) && aligned16(d->DY) && aligned16(d->W1) && aligned16(d->W2T) && aligned16(d->W1T) &&
                   aligned16(d->A) && aligned16(d->DH) && aligned16(d->DX) && aligned16(d->b1) && d->ldx % 8 == 0 &&
                   d->lddy % 8 == 0 && d->ldw1 % 8 == 0 && d->ldw2t % 8 == 0 && d->ldw1t % 8 == 0 && d->lda % 8 == 0 &&
                   d->lddh % 8 == 0 && d->lddx % 8 == 0 && d->ldx >= d->C && d->lddy >= d->C && d->ldw1 >= d->C &&
                   d->ldw2t >= d->C && d->ldw1t >= d->H && d->lda >= d->H && d->lddh >= d->H && d->lddx >= d->C,
               "ga_mlp_bwd: operands must be 16-byte aligned with leading dimensions in multiples of 8");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((d->M + BM - 1) / BM);
    if (d->C == 96) {
        constexpr int lds = bwd_lds<96, 64>();
        GA_REQUIRE(set_lds(mlp_bwd_kernel<96, 64>, lds), "ga_mlp_bwd: LDS");
        hipLaunchKernelGGL((mlp_bwd_kernel<96, 64>), dim3(grid), dim3(512), lds, s, *d);
    } else {
        constexpr int lds = bwd_lds<192, 32>();
        GA_REQUIRE(set_lds(mlp_bwd_kernel<192, 32>, lds), "ga_mlp_bwd: LDS");
        hipLaunchKernelGGL((mlp_bwd_kernel<192, 32>), dim3(grid), dim3(512), lds, s, *d);
    }
    return ga_check_launch("ga_mlp_bwd");
}
