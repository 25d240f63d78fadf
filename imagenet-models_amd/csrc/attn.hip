// Global multi-head self-attention over token sequences (timm vision_transformer.Attention as the reference uses it through
// `Block`: MAP/models/map_pit.py:14,35-44):  out = softmax(q k^T * scale) v  per (image, head), N tokens, head_dim 64.
//
//   qkv [B*N][ldq]: q | k | v column blocks of width C = H * hd (head h owns columns h*hd .. of each block), out [B*N][ldo].
//
// bf16, head_dim 64: flash-style on the matrix cores.  A workgroup (4 waves) owns 64 query rows of one (image, head); keys /
// values stream through LDS in blocks of 64.  As in the stripe-attention kernels (cswin.hip) the scores are computed
// TRANSPOSED, S^T = K . Q^T with v_mfma_f32_16x16x32_bf16: a lane then holds 16 keys of ONE query, so the running maximum /
// sum of the online softmax are in-lane reductions plus two shuffles, and the P^T accumulators are directly the A operand
// of the next product (O += P . V; the V fragment is read with the transposing ds_read_b64_tr_b16 in the matching k order).
// Backward = three launches: delta[q] = dO . O, a dQ kernel (per query block, streaming keys) and a dK / dV kernel (per key
// block, streaming queries; S and dP recomputed in the [query][key] layout whose accumulators are the A operands of
// dV += P^T dO and dK += dS^T Q).  The row statistics lse[b][h][q] are kept by the forward pass.
//
// Everything else (fp32 parity mode, other head dims): a plain fp32 kernel, one thread per query (forward / dQ) or per key
// (dK / dV) streaming the other side from L2 -- meant for the parity tests at small batches, not for throughput.
#include <algorithm>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// LDS tile = [rows][32 bf16] slabs (64-byte rows, 16-byte unit c of row r at c ^ ((r >> 2) & 3)); a 64-wide head = 2 slabs
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned tile_off(int r, int chunk) { return r * 64 + ((chunk ^ ((r >> 2) & 3)) << 4); }

__device__ __forceinline__ bf16x8_t row_frag(const unsigned char* slab, int r0, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(slab + tile_off(r0 + (lane & 15), lane >> 4));
}

// B fragment taken column-wise with the transposing LDS read: element j of lane (g, i) is slab[k0 + 16 (j >> 2) + 4 g + (j & 3)][16 dt + i]
__device__ __forceinline__ bf16x8_t col_frag_acc(const unsigned char* slab, int k0, int dt, int lane) {
    const int gq = lane >> 4, i = lane & 15;
    s16x4_t h[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = k0 + 16 * q + 4 * gq + (i >> 2);
        const unsigned a = tile_off(r, 2 * dt + ((i & 3) >> 1)) + 8 * (i & 1);
        h[q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(slab + a));
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
    return *reinterpret_cast<const bf16x8_t*>(&v);
}

__device__ __forceinline__ bf16x8_t acc_pair_frag(const f32x4_t& a, const f32x4_t& b) {
    uint4 u;
    u.x = pack2bf(a[0], a[1]); u.y = pack2bf(a[2], a[3]); u.z = pack2bf(b[0], b[1]); u.w = pack2bf(b[2], b[3]);
    return *reinterpret_cast<const bf16x8_t*>(&u);
}

// max / sum over the four lane rows (lanes l, l^16, l^32, l^48) on the VALU: v_permlane16_swap(x, x) leaves the lane's own value in
// one result and its l^16 partner's in the other (which is which depends on the row, the combination does not); likewise 32
__device__ __forceinline__ float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

constexpr int BQ = 64;                  // rows of a block (queries or keys)
constexpr int SLAB = BQ * 64;           // bytes of one [64][32] slab
constexpr int TILE = 2 * SLAB;          // [64][64] bf16

// rows [r0, r0 + 64) x 64 columns of a [B*N][ld] column slice -> two slabs; rows >= nrows (end of the sequence) are zero
// (hc = head_dim / 8 column chunks are real, the rest of the 64-wide tile is zero: head_dim 16 / 32 / 48 ride the same code)
__device__ __forceinline__ void load_block(unsigned char* dst, const bf16_t* src, long ld, int r0, int nrows, int hc) {
    for (int i = threadIdx.x; i < BQ * 8; i += 256) {
        const int r = i >> 3, c = i & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < nrows && c < hc) v = *reinterpret_cast<const uint4*>(src + (long)(r0 + r) * ld + c * 8);
        *reinterpret_cast<uint4*>(dst + (c >> 2) * SLAB + tile_off(r, c & 3)) = v;
    }
}

// the same block in two steps: global -> registers (issued a whole block ahead of its use), registers -> LDS
struct BlockRegs { u32x4_t v[2]; };
__device__ __forceinline__ void fetch_block(BlockRegs& b, const bf16_t* src, long ld, int r0, int nrows, int hc) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = threadIdx.x + 256 * j, r = i >> 3, c = i & 7;
        b.v[j] = u32x4_t{0, 0, 0, 0};
        if (r0 + r < nrows && c < hc) b.v[j] = *reinterpret_cast<const u32x4_t*>(src + (long)(r0 + r) * ld + c * 8);
    }
}
__device__ __forceinline__ void put_block(unsigned char* dst, const BlockRegs& b) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = threadIdx.x + 256 * j, r = i >> 3, c = i & 7;
        *reinterpret_cast<u32x4_t*>(dst + (c >> 2) * SLAB + tile_off(r, c & 3)) = b.v[j];
    }
}

// fp32 accumulator tiles o[dt] (rows 4g + r, column 16 dt + (lane & 15)) of one wave's 16 rows -> bf16 rows through this wave's
// 2 KiB staging piece -> 16-byte global stores (8 lanes per row)
__device__ __forceinline__ void store_rows16(unsigned char* stage, const f32x4_t (&o)[4], bf16_t* dst, long ld, int r0, int nrows,
                                             int lane, int hc) {
    bf16_t* st = reinterpret_cast<bf16_t*>(stage);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * (lane >> 4) + r) * 64 + 16 * dt + (lane & 15)] = f2bf(o[dt][r]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // a wave's LDS accesses complete in order
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = 8 * p + (lane >> 3), c = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + r * 128 + c * 16);
        if (r0 + r < nrows && c < hc) *reinterpret_cast<uint4*>(dst + (long)(r0 + r) * ld + c * 8) = v;
    }
}

// the same for TRANSPOSED accumulator tiles o[dt] (row = channel 16 dt + 4 g + r, column = the wave's row lane & 15)
__device__ __forceinline__ void store_rows16_t(unsigned char* stage, const f32x4_t (&o)[4], bf16_t* dst, long ld, int r0, int nrows,
                                               int lane, int hc) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        uint2 u;
        u.x = pack2bf(o[dt][0], o[dt][1]);
        u.y = pack2bf(o[dt][2], o[dt][3]);
        *reinterpret_cast<uint2*>(stage + (lane & 15) * 128 + (16 * dt + 4 * (lane >> 4)) * 2) = u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // a wave's LDS accesses complete in order
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = 8 * p + (lane >> 3), c = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(stage + r * 128 + c * 16);
        if (r0 + r < nrows && c < hc) *reinterpret_cast<uint4*>(dst + (long)(r0 + r) * ld + c * 8) = v;
    }
}

__device__ __forceinline__ int xcd_walk(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct Item { int b, h, blk; long row0; };   // row0 = first row of image b in the [B*N] matrices

__device__ __forceinline__ Item item_of(const ga_attn_desc& d, int id, int nblk) {
    Item it;
    it.blk = id % nblk;
    const int bh = id / nblk;
    it.h = bh % d.H;
    it.b = bh / d.H;
    it.row0 = (long)it.b * d.N;
    return it;
}

// =================================================================================================================
// forward, bf16, head_dim 64
// =================================================================================================================
// QT = 16-query tiles per wave: a workgroup owns 64 * QT queries.  With QT = 2 every K / V fragment read from LDS feeds two MFMAs
// (one wave's 16 queries alone make the loop LDS-bound: 8 ds_read_b128 + 16 transposing reads per 16 MFMAs)
template <int QT>
__global__ __launch_bounds__(256) void attn_fwd_mfma(const ga_attn_desc d, const int nblk, const int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qs = smem;
    unsigned char* Ks = Qs + TILE;
    unsigned char* Vs = Ks + TILE;
    unsigned char* Stage = Vs + TILE;            // 4 x 2 KiB
    const Item it = item_of(d, xcd_walk(blockIdx.x, nwg), nblk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int hc = d.hd >> 3, C = d.H * d.hd, q0 = it.blk * (BQ * QT);
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(d.qkv) + it.row0 * d.ldq + it.h * d.hd;
    bf16x8_t qf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (t) __syncthreads();
        load_block(Qs, qkv, d.ldq, q0 + BQ * t, d.N, hc);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[t][s] = row_frag(Qs + s * SLAB, 16 * wave, lane);
    }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t o[QT][4];
    float m[QT], l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -3.0e38f;
        l[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = zero;
    }
    const float sc = d.scale * 1.44269504f;                    // exp2 domain
    BlockRegs kr, vr;                                          // the next key / value block travels in registers
    fetch_block(kr, qkv + C, d.ldq, 0, d.N, hc);
    fetch_block(vr, qkv + 2 * C, d.ldq, 0, d.N, hc);
    for (int k0 = 0; k0 < d.N; k0 += BQ) {
        __syncthreads();                                       // everyone is done with the previous key block
        put_block(Ks, kr);
        put_block(Vs, vr);
        __syncthreads();
        if (k0 + BQ < d.N) {
            fetch_block(kr, qkv + C, d.ldq, k0 + BQ, d.N, hc);
            fetch_block(vr, qkv + 2 * C, d.ldq, k0 + BQ, d.N, hc);
        }
        f32x4_t st[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int t = 0; t < QT; ++t) st[t][kt] = zero;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t kf = row_frag(Ks + s * SLAB, 16 * kt, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) st[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[t][s], st[t][kt], 0, 0, 0);
            }
        }
        bf16x8_t pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[t][kt][r] = (k0 + 16 * kt + 4 * g + r < d.N) ? st[t][kt][r] * sc : -3.0e38f;
                    mx = fmaxf(mx, st[t][kt][r]);
                }
            mx = rows_max(mx);
            const float mn = fmaxf(m[t], mx);
            const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(st[t][kt][r] - mn);      // masked keys: exp2(-huge) = 0
                    st[t][kt][r] = p;
                    sum += p;
                }
            sum = rows_sum(sum);
            l[t] = l[t] * alpha + sum;
            m[t] = mn;
            // O is accumulated TRANSPOSED (rows = channels, column = the query lane & 15): the statistics of a lane's query are its own
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) pf[t][kp] = acc_pair_frag(st[t][2 * kp], st[t][2 * kp + 1]);
        }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t vf = col_frag_acc(Vs + (dt >> 1) * SLAB, 32 * kp, dt & 1, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t][kp], o[t][dt], 0, 0, 0);   // (V^T P^T)
            }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float linv = 1.f / l[t];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][dt][r] *= linv;
        const int qw = q0 + BQ * t + 16 * wave;
        store_rows16_t(Stage + wave * 2048, o[t], reinterpret_cast<bf16_t*>(d.out) + it.row0 * d.ldo + it.h * d.hd, d.ldo, qw, d.N, lane, hc);
        if (lane < 16 && qw + lane < d.N) d.lse[((long)it.b * d.H + it.h) * d.N + qw + lane] = m[t] * 0.69314718f + __logf(l[t]);   // natural-log units
    }
}

// delta[b][h][q] = sum_d dO[q][d] * O[q][d]   (one thread per (b, h, q))
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const ga_attn_desc d, const void* dout_, float* delta) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)d.B * d.H * d.N;
    if (i >= n) return;
    const int q = (int)(i % d.N), h = (int)((i / d.N) % d.H), b = (int)(i / ((long)d.N * d.H));
    const T* o = reinterpret_cast<const T*>(d.out) + ((long)b * d.N + q) * d.ldo + h * d.hd;
    const T* go = reinterpret_cast<const T*>(dout_) + ((long)b * d.N + q) * d.ldo + h * d.hd;
    float s = 0.f;
    for (int c = 0; c < d.hd; ++c) s = fmaf(elt<T>::ld(o + c), elt<T>::ld(go + c), s);
    delta[i] = s;
}

// the same with whole-row coalesced reads (hd / 8 a power of two): thread = 8 consecutive channels of one token row, the hd / 8
// lanes of a head reduce by shuffles (the per-(b, h, q) form above reads 64 scattered 128-byte pieces per wave: 1.4 ms at
// ViT-B / 384 / B = 128, more than the three MFMA kernels of the backward together)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_rows_kernel(const ga_attn_desc d, const void* dout_, float* delta) {
    const int C8 = d.H * d.hd / 8, L = d.hd / 8;
    const long n = (long)d.B * d.N * C8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    long row = 0;
    int c8 = 0;
    if (i < n) {
        row = i / C8;
        c8 = (int)(i - row * C8);
        float a[8], g[8];
        load8(reinterpret_cast<const T*>(d.out) + row * d.ldo + c8 * 8, a);
        load8(reinterpret_cast<const T*>(dout_) + row * d.ldo + c8 * 8, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(a[j], g[j], s);
    }
    for (int o = 1; o < L; o <<= 1) s += __shfl_xor(s, o, 64);
    if (i < n && (c8 & (L - 1)) == 0) {
        const long b = row / d.N, q = row - b * d.N;
        delta[(b * d.H + c8 / L) * d.N + q] = s;
    }
}

// head dims whose chunk count is not a power of two (PiT: 48): thread = one (row, head), hd / 8 16-byte loads each; the lanes of a
// wave cover consecutive heads of a row, so the wave's reads are one contiguous span when ldo = H * hd
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_heads_kernel(const ga_attn_desc d, const void* dout_, float* delta) {
    const long n = (long)d.B * d.N * d.H;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = i / d.H;
    const int h = (int)(i - row * d.H);
    const T* o = reinterpret_cast<const T*>(d.out) + row * d.ldo + h * d.hd;
    const T* go = reinterpret_cast<const T*>(dout_) + row * d.ldo + h * d.hd;
    float s = 0.f;
    for (int c = 0; c < d.hd; c += 8) {
        float a[8], g[8];
        load8(o + c, a);
        load8(go + c, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(a[j], g[j], s);
    }
    const long b = row / d.N, q = row - b * d.N;
    delta[(b * d.H + h) * d.N + q] = s;
}

// =================================================================================================================
// backward dQ, bf16, head_dim 64: one workgroup per 64 queries, keys / values streamed
// =================================================================================================================
template <int QT>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma(const ga_attn_desc d, const void* dout_, void* dqkv_, const float* delta,
                                                        const int nblk, const int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qs = smem;
    unsigned char* Gs = Qs + TILE;
    unsigned char* Ks = Gs + TILE;
    unsigned char* Vs = Ks + TILE;
    unsigned char* Stage = Vs + TILE;
    const Item it = item_of(d, xcd_walk(blockIdx.x, nwg), nblk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int hc = d.hd >> 3, C = d.H * d.hd, q0 = it.blk * (BQ * QT);
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(d.qkv) + it.row0 * d.ldq + it.h * d.hd;
    bf16x8_t qf[QT][2], gf[QT][2];
    float lse[QT], dl[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        if (t) __syncthreads();
        load_block(Qs, qkv, d.ldq, q0 + BQ * t, d.N, hc);
        load_block(Gs, reinterpret_cast<const bf16_t*>(dout_) + it.row0 * d.ldo + it.h * d.hd, d.ldo, q0 + BQ * t, d.N, hc);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[t][s] = row_frag(Qs + s * SLAB, 16 * wave, lane);
            gf[t][s] = row_frag(Gs + s * SLAB, 16 * wave, lane);
        }
        const int qi = q0 + BQ * t + 16 * wave + (lane & 15);
        const long sidx = ((long)it.b * d.H + it.h) * d.N + qi;
        lse[t] = qi < d.N ? d.lse[sidx] : 0.f;
        dl[t] = qi < d.N ? delta[sidx] : 0.f;
    }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t o[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = zero;
    BlockRegs kr, vr;
    fetch_block(kr, qkv + C, d.ldq, 0, d.N, hc);
    fetch_block(vr, qkv + 2 * C, d.ldq, 0, d.N, hc);
    for (int k0 = 0; k0 < d.N; k0 += BQ) {
        __syncthreads();
        put_block(Ks, kr);
        put_block(Vs, vr);
        __syncthreads();
        if (k0 + BQ < d.N) {
            fetch_block(kr, qkv + C, d.ldq, k0 + BQ, d.N, hc);
            fetch_block(vr, qkv + 2 * C, d.ldq, k0 + BQ, d.N, hc);
        }
        f32x4_t ds[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4_t st[QT], dp[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) st[t] = dp[t] = zero;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t kfr = row_frag(Ks + s * SLAB, 16 * kt, lane), vfr = row_frag(Vs + s * SLAB, 16 * kt, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[t][s], st[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, gf[t][s], dp[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = k0 + 16 * kt + 4 * g + r < d.N;
                    const float p = ok ? __expf(st[t][r] * d.scale - lse[t]) : 0.f;
                    ds[t][kt][r] = p * (dp[t][r] - dl[t]);
                }
        }
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            bf16x8_t sf[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) sf[t] = acc_pair_frag(ds[t][2 * kp], ds[t][2 * kp + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t kc = col_frag_acc(Ks + (dt >> 1) * SLAB, 32 * kp, dt & 1, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf[t], kc, o[t][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][dt][r] *= d.scale;
        store_rows16(Stage + wave * 2048, o[t], reinterpret_cast<bf16_t*>(dqkv_) + it.row0 * d.ldq + it.h * d.hd, d.ldq, q0 + BQ * t + 16 * wave, d.N, lane, hc);
    }
}

// =================================================================================================================
// backward dK / dV, bf16, head_dim 64: one workgroup per 64 keys, queries / dO streamed
// =================================================================================================================
// KT = 16-key tiles per wave (a workgroup owns 64 * KT keys): with KT = 2 every Q / dO fragment read from LDS (16 row reads + 32
// transposing reads per query block) feeds two MFMAs
template <int KT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma(const ga_attn_desc d, const void* dout_, void* dqkv_, const float* delta,
                                                         const int nblk, const int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vs = Ks + TILE;
    unsigned char* Qs = Vs + TILE;
    unsigned char* Gs = Qs + TILE;
    unsigned char* Stage = Gs + TILE;                              // 4 x 2 KiB
    float* lse_s = reinterpret_cast<float*>(Stage + 8192);         // [64]
    float* dlt_s = lse_s + BQ;                                     // [64]
    const Item it = item_of(d, xcd_walk(blockIdx.x, nwg), nblk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int hc = d.hd >> 3, C = d.H * d.hd, k0 = it.blk * (BQ * KT);
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(d.qkv) + it.row0 * d.ldq + it.h * d.hd;
    const bf16_t* dout = reinterpret_cast<const bf16_t*>(dout_) + it.row0 * d.ldo + it.h * d.hd;
    bf16_t* dqkv = reinterpret_cast<bf16_t*>(dqkv_) + it.row0 * d.ldq + it.h * d.hd;
    bf16x8_t kf[KT][2], vf[KT][2];
    bool key_ok[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (t) __syncthreads();
        load_block(Ks, qkv + C, d.ldq, k0 + BQ * t, d.N, hc);
        load_block(Vs, qkv + 2 * C, d.ldq, k0 + BQ * t, d.N, hc);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[t][s] = row_frag(Ks + s * SLAB, 16 * wave, lane);
            vf[t][s] = row_frag(Vs + s * SLAB, 16 * wave, lane);
        }
        key_ok[t] = k0 + BQ * t + 16 * wave + (lane & 15) < d.N;
    }
    const long sbase = ((long)it.b * d.H + it.h) * d.N;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t dv[KT][4], dk[KT][4];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[t][dt] = dk[t][dt] = zero;
    BlockRegs qr, gr;
    float lr = 0.f, dr = 0.f;                                  // this thread's lse / delta entry of the next query block (threads < 64)
    auto fetch_q = [&](int q0) {
        fetch_block(qr, qkv, d.ldq, q0, d.N, hc);
        fetch_block(gr, dout, d.ldo, q0, d.N, hc);
        if (threadIdx.x < BQ) {
            const int q = q0 + threadIdx.x;
            lr = q < d.N ? d.lse[sbase + q] : 0.f;
            dr = q < d.N ? delta[sbase + q] : 0.f;
        }
    };
    fetch_q(0);
    for (int q0 = 0; q0 < d.N; q0 += BQ) {
        __syncthreads();
        put_block(Qs, qr);
        put_block(Gs, gr);
        if (threadIdx.x < BQ) {
            lse_s[threadIdx.x] = lr;
            dlt_s[threadIdx.x] = dr;
        }
        __syncthreads();
        if (q0 + BQ < d.N) fetch_q(q0 + BQ);
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            f32x4_t p[KT][2], ds[KT][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qt = 32 * qs + 16 * h;              // rows of this 16-query tile inside the block
#pragma unroll
                for (int t = 0; t < KT; ++t) p[t][h] = ds[t][h] = zero;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8_t qfr = row_frag(Qs + s * SLAB, qt, lane), gfr = row_frag(Gs + s * SLAB, qt, lane);
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        p[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[t][s], p[t][h], 0, 0, 0);      // [query 4g+r][key lane&15]
                        ds[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfr, vf[t][s], ds[t][h], 0, 0, 0);
                    }
                }
                const f32x4_t ls = *reinterpret_cast<const f32x4_t*>(lse_s + qt + 4 * g);
                const f32x4_t de = *reinterpret_cast<const f32x4_t*>(dlt_s + qt + 4 * g);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = key_ok[t] && q0 + qt + 4 * g + r < d.N;
                        const float pv = ok ? __expf(p[t][h][r] * d.scale - ls[r]) : 0.f;
                        p[t][h][r] = pv;
                        ds[t][h][r] = pv * (ds[t][h][r] - de[r]);
                    }
            }
            bf16x8_t pf[KT], sf[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                pf[t] = acc_pair_frag(p[t][0], p[t][1]);
                sf[t] = acc_pair_frag(ds[t][0], ds[t][1]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t gc = col_frag_acc(Gs + (dt >> 1) * SLAB, 32 * qs, dt & 1, lane), qc = col_frag_acc(Qs + (dt >> 1) * SLAB, 32 * qs, dt & 1, lane);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    dv[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[t], gc, dv[t][dt], 0, 0, 0);
                    dk[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf[t], qc, dk[t][dt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dk[t][dt][r] *= d.scale;
        store_rows16(Stage + wave * 2048, dk[t], dqkv + C, d.ldq, k0 + BQ * t + 16 * wave, d.N, lane, hc);
        store_rows16(Stage + wave * 2048, dv[t], dqkv + 2 * C, d.ldq, k0 + BQ * t + 16 * wave, d.N, lane, hc);
    }
}

// =================================================================================================================
// generic fp32 forms (any dtype / head_dim <= 128): one thread per query (forward, dQ) or per key (dK, dV)
// =================================================================================================================
template <typename T>
__global__ __launch_bounds__(128) void attn_fwd_simple(const ga_attn_desc d) {
    const long i = (long)blockIdx.x * 128 + threadIdx.x, n = (long)d.B * d.H * d.N;
    if (i >= n) return;
    const int q = (int)(i % d.N), h = (int)((i / d.N) % d.H), b = (int)(i / ((long)d.N * d.H));
    const int C = d.H * d.hd, hd = d.hd;
    const T* base = reinterpret_cast<const T*>(d.qkv) + (long)b * d.N * d.ldq + h * hd;
    float qv[128], acc[128];
    for (int c = 0; c < hd; ++c) {
        qv[c] = elt<T>::ld(base + (long)q * d.ldq + c) * d.scale;
        acc[c] = 0.f;
    }
    float m = -3.0e38f, l = 0.f;
    for (int j = 0; j < d.N; ++j) {
        const T* kr = base + (long)j * d.ldq + C;
        float s = 0.f;
        for (int c = 0; c < hd; ++c) s = fmaf(qv[c], elt<T>::ld(kr + c), s);
        const float mn = fmaxf(m, s), a = __expf(m - mn), p = __expf(s - mn);
        l = l * a + p;
        const T* vr = base + (long)j * d.ldq + 2 * C;
        for (int c = 0; c < hd; ++c) acc[c] = fmaf(p, elt<T>::ld(vr + c), acc[c] * a);
        m = mn;
    }
    T* o = reinterpret_cast<T*>(d.out) + ((long)b * d.N + q) * d.ldo + h * hd;
    const float inv = 1.f / l;
    for (int c = 0; c < hd; ++c) elt<T>::st(o + c, acc[c] * inv);
    d.lse[i] = m + __logf(l);
}

template <typename T>
__global__ __launch_bounds__(128) void attn_bwd_dq_simple(const ga_attn_desc d, const void* dout_, void* dqkv_, const float* delta) {
    const long i = (long)blockIdx.x * 128 + threadIdx.x, n = (long)d.B * d.H * d.N;
    if (i >= n) return;
    const int q = (int)(i % d.N), h = (int)((i / d.N) % d.H), b = (int)(i / ((long)d.N * d.H));
    const int C = d.H * d.hd, hd = d.hd;
    const T* base = reinterpret_cast<const T*>(d.qkv) + (long)b * d.N * d.ldq + h * hd;
    const T* go = reinterpret_cast<const T*>(dout_) + ((long)b * d.N + q) * d.ldo + h * hd;
    float qv[128], gv[128], acc[128];
    for (int c = 0; c < hd; ++c) {
        qv[c] = elt<T>::ld(base + (long)q * d.ldq + c);
        gv[c] = elt<T>::ld(go + c);
        acc[c] = 0.f;
    }
    const float lse = d.lse[i], dl = delta[i];
    for (int j = 0; j < d.N; ++j) {
        const T* kr = base + (long)j * d.ldq + C;
        const T* vr = base + (long)j * d.ldq + 2 * C;
        float s = 0.f, dp = 0.f;
        for (int c = 0; c < hd; ++c) {
            s = fmaf(qv[c], elt<T>::ld(kr + c), s);
            dp = fmaf(gv[c], elt<T>::ld(vr + c), dp);
        }
        const float ds = __expf(s * d.scale - lse) * (dp - dl);
        for (int c = 0; c < hd; ++c) acc[c] = fmaf(ds, elt<T>::ld(kr + c), acc[c]);
    }
    T* o = reinterpret_cast<T*>(dqkv_) + ((long)b * d.N + q) * d.ldq + h * hd;
    for (int c = 0; c < hd; ++c) elt<T>::st(o + c, acc[c] * d.scale);
}

template <typename T>
__global__ __launch_bounds__(128) void attn_bwd_dkv_simple(const ga_attn_desc d, const void* dout_, void* dqkv_, const float* delta) {
    const long i = (long)blockIdx.x * 128 + threadIdx.x, n = (long)d.B * d.H * d.N;
    if (i >= n) return;
    const int j = (int)(i % d.N), h = (int)((i / d.N) % d.H), b = (int)(i / ((long)d.N * d.H));
    const int C = d.H * d.hd, hd = d.hd;
    const T* base = reinterpret_cast<const T*>(d.qkv) + (long)b * d.N * d.ldq + h * hd;
    const T* gbase = reinterpret_cast<const T*>(dout_) + (long)b * d.N * d.ldo + h * hd;
    float kv[128], vv[128], ak[128], av[128];
    for (int c = 0; c < hd; ++c) {
        kv[c] = elt<T>::ld(base + (long)j * d.ldq + C + c);
        vv[c] = elt<T>::ld(base + (long)j * d.ldq + 2 * C + c);
        ak[c] = av[c] = 0.f;
    }
    const long sbase = ((long)b * d.H + h) * d.N;
    for (int q = 0; q < d.N; ++q) {
        const T* qr = base + (long)q * d.ldq;
        const T* gr = gbase + (long)q * d.ldo;
        float s = 0.f, dp = 0.f;
        for (int c = 0; c < hd; ++c) {
            s = fmaf(elt<T>::ld(qr + c), kv[c], s);
            dp = fmaf(elt<T>::ld(gr + c), vv[c], dp);
        }
        const float p = __expf(s * d.scale - d.lse[sbase + q]);
        const float ds = p * (dp - delta[sbase + q]);
        for (int c = 0; c < hd; ++c) {
            av[c] = fmaf(p, elt<T>::ld(gr + c), av[c]);
            ak[c] = fmaf(ds, elt<T>::ld(qr + c), ak[c]);
        }
    }
    T* o = reinterpret_cast<T*>(dqkv_) + ((long)b * d.N + j) * d.ldq + h * hd;
    for (int c = 0; c < hd; ++c) {
        elt<T>::st(o + C + c, ak[c] * d.scale);
        elt<T>::st(o + 2 * C + c, av[c]);
    }
}

// =================================================================================================================
// ViT stem helpers (timm PatchEmbed + cls_token / pos_embed): the P x P / stride P patch convolution is a GEMM over patches
//   ga_patchify:  fp32 NCHW [B,3,H,W] -> [B*(H/P)*(W/P)][3*P*P] in T, k = (c, ky, kx) = the flattened conv weight's order
//   ga_patchify_strided: the same for a patch convolution whose stride S differs from P (PiT conv_embedding: 16 / 8, overlapping)
//   ga_vit_embed_fwd:  x0[b][0] = cls + pos[0];  x0[b][1+p] = tok[b][p] + pos[1+p]
//   ga_vit_embed_bwd:  dtok = dx0[b][1+p];  dcls += sum_b dx0[b][0];  dpos[t] += sum_b dx0[b][t]
// =================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int CH, int H, int W, int P,
                                                       int S) {
    const int gw = (W - P) / S + 1, gh = (H - P) / S + 1, K = CH * P * P, K8 = K / 8;
    const long n = (long)B * gh * gw * K8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int k = (int)(i % K8) * 8;
        const long row = i / K8;
        const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long)gw * gh));
        const int c = k / (P * P), ky = (k / P) % P, kx = k % P;
        const float* src = x + (((long)b * CH + c) * H + py * S + ky) * W + px * S + kx;
        const float4 a = *reinterpret_cast<const float4*>(src), bq = *reinterpret_cast<const float4*>(src + 4);
        float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        store8(out + row * K + k, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_embed_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                                            T* __restrict__ x0, int B, int Np, int C) {
    const int C8 = C / 8;
    const long n = (long)B * (Np + 1) * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long row = i / C8;
        const int t = (int)(row % (Np + 1)), b = (int)(row / (Np + 1));
        float v[8], pv[8];
        load8(pos + (long)t * C + c, pv);
        if (t == 0) load8(cls + c, v);
        else load8(tok + ((long)b * Np + t - 1) * C + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += pv[j];
        store8(x0 + row * C + c, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_embed_bwd_kernel(const T* __restrict__ dx0, T* __restrict__ dtok, float* __restrict__ dcls,
                                                            float* __restrict__ dpos, int B, int Np, int C) {
    const int C8 = C / 8;
    const long n = (long)(Np + 1) * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8, t = (int)(i / C8);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            float v[8];
            load8(dx0 + ((long)b * (Np + 1) + t) * C + c, v);
            if (t > 0) store8(dtok + ((long)b * Np + t - 1) * C + c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dpos[(long)t * C + c + j] += acc[j];
            if (t == 0) dcls[c + j] += acc[j];
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_desc(const ga_attn_desc* d, const char* what) {
    GA_REQUIRE(d && d->qkv && d->out && d->lse && d->B > 0 && d->N > 0 && d->H > 0 && d->hd > 0 && d->hd <= 128, "%s: null / empty descriptor", what);
    GA_REQUIRE(d->dtype == GA_F32 || d->dtype == GA_BF16, "%s: bad dtype", what);
    GA_REQUIRE(d->ldq >= 3L * d->H * d->hd && d->ldo >= (long)d->H * d->hd, "%s: leading dimensions", what);
    return GA_OK;
}

bool use_mfma(const ga_attn_desc* d) {
    return GA_KNOB("ATTN_MFMA", 1) && d->dtype == GA_BF16 && d->hd % 16 == 0 && d->hd <= 64 && d->ldq % 8 == 0 && d->ldo % 8 == 0 && aligned16(d->qkv) && aligned16(d->out);
}

}  // namespace

extern "C" int ga_attn_fwd(const ga_attn_desc* d, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_attn_fwd")) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (use_mfma(d)) {
        const int qt_env = GA_KNOB("ATTN_QT", 0);
        const int qt = qt_env ? qt_env : (d->N >= 128 ? 2 : 1);
        const int nblk = (d->N + BQ * qt - 1) / (BQ * qt), nwg = nblk * d->B * d->H;
        if (qt == 2) hipLaunchKernelGGL(attn_fwd_mfma<2>, dim3(nwg), dim3(256), 3 * TILE + 8192, s, *d, nblk, nwg);
        else hipLaunchKernelGGL(attn_fwd_mfma<1>, dim3(nwg), dim3(256), 3 * TILE + 8192, s, *d, nblk, nwg);
        return ga_check_launch("ga_attn_fwd");
    }
    const long n = (long)d->B * d->H * d->N;
    if (d->dtype == GA_BF16) hipLaunchKernelGGL(attn_fwd_simple<bf16_t>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, *d);
    else hipLaunchKernelGGL(attn_fwd_simple<float>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, *d);
    return ga_check_launch("ga_attn_fwd");
}

extern "C" size_t ga_attn_bwd_workspace(const ga_attn_desc* d) { return d ? (size_t)d->B * d->H * d->N * sizeof(float) : 0; }

extern "C" int ga_attn_bwd(const ga_attn_desc* d, const void* dout, void* dqkv, void* workspace, size_t ws_bytes, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_attn_bwd")) return rc;
    GA_REQUIRE(dout && dqkv && workspace && ws_bytes >= ga_attn_bwd_workspace(d), "ga_attn_bwd: dout / dqkv / a workspace of %zu bytes",
               ga_attn_bwd_workspace(d));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* delta = reinterpret_cast<float*>(workspace);
    const long n = (long)d->B * d->H * d->N;
    const unsigned g256 = (unsigned)((n + 255) / 256), g128 = (unsigned)((n + 127) / 128);
    const int L8 = d->hd / 8;
    const bool rows_form = d->hd % 8 == 0 && (L8 & (L8 - 1)) == 0 && L8 <= 64 && (d->H * L8) % L8 == 0 && aligned16(d->out) && aligned16(dout) &&
                           d->ldo % (d->dtype == GA_BF16 ? 8 : 4) == 0 && (256 % L8) == 0;
    if (rows_form) {
        const long nt = (long)d->B * d->N * d->H * L8;
        const unsigned gr = (unsigned)((nt + 255) / 256);
        if (d->dtype == GA_BF16) hipLaunchKernelGGL(attn_delta_rows_kernel<bf16_t>, dim3(gr), dim3(256), 0, s, *d, dout, delta);
        else hipLaunchKernelGGL(attn_delta_rows_kernel<float>, dim3(gr), dim3(256), 0, s, *d, dout, delta);
    } else if (d->hd % 8 == 0 && aligned16(d->out) && aligned16(dout) && d->ldo % 8 == 0) {
        if (d->dtype == GA_BF16) hipLaunchKernelGGL(attn_delta_heads_kernel<bf16_t>, dim3(g256), dim3(256), 0, s, *d, dout, delta);
        else hipLaunchKernelGGL(attn_delta_heads_kernel<float>, dim3(g256), dim3(256), 0, s, *d, dout, delta);
    } else if (d->dtype == GA_BF16) {
        hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3(g256), dim3(256), 0, s, *d, dout, delta);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(g256), dim3(256), 0, s, *d, dout, delta);
    }
    if (use_mfma(d) && aligned16(dout) && aligned16(dqkv)) {
        const int qt_env = GA_KNOB("ATTN_QT", 0);
        const int qt = qt_env ? qt_env : (d->N >= 256 ? 2 : 1);      // (N = 197: 0.381 vs 0.393 ms per backward with 2)
        const int nblk1 = (d->N + BQ * qt - 1) / (BQ * qt), nwg1 = nblk1 * d->B * d->H;
        if (qt == 2) hipLaunchKernelGGL(attn_bwd_dq_mfma<2>, dim3(nwg1), dim3(256), 4 * TILE + 8192, s, *d, dout, dqkv, delta, nblk1, nwg1);
        else hipLaunchKernelGGL(attn_bwd_dq_mfma<1>, dim3(nwg1), dim3(256), 4 * TILE + 8192, s, *d, dout, dqkv, delta, nblk1, nwg1);
        const int kt_env = GA_KNOB("ATTN_KT", 0);
        const int kt = kt_env ? kt_env : (d->N >= 128 ? 2 : 1);
        const int nblk2 = (d->N + BQ * kt - 1) / (BQ * kt), nwg2 = nblk2 * d->B * d->H;
        if (kt == 2) hipLaunchKernelGGL(attn_bwd_dkv_mfma<2>, dim3(nwg2), dim3(256), 4 * TILE + 8192 + 512, s, *d, dout, dqkv, delta, nblk2, nwg2);
        else hipLaunchKernelGGL(attn_bwd_dkv_mfma<1>, dim3(nwg2), dim3(256), 4 * TILE + 8192 + 512, s, *d, dout, dqkv, delta, nblk2, nwg2);
        return ga_check_launch("ga_attn_bwd");
    }
    if (d->dtype == GA_BF16) {
        hipLaunchKernelGGL(attn_bwd_dq_simple<bf16_t>, dim3(g128), dim3(128), 0, s, *d, dout, dqkv, delta);
        hipLaunchKernelGGL(attn_bwd_dkv_simple<bf16_t>, dim3(g128), dim3(128), 0, s, *d, dout, dqkv, delta);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_simple<float>, dim3(g128), dim3(128), 0, s, *d, dout, dqkv, delta);
        hipLaunchKernelGGL(attn_bwd_dkv_simple<float>, dim3(g128), dim3(128), 0, s, *d, dout, dqkv, delta);
    }
    return ga_check_launch("ga_attn_bwd");
}

extern "C" int ga_patchify_strided(const float* x, void* out, int B, int CH, int H, int W, int P, int S, int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && out && B > 0 && CH > 0 && P > 0 && P % 8 == 0 && S > 0 && S % 4 == 0 && H >= P && W >= P && W % 4 == 0 && aligned16(x) &&
                   aligned16(out),
               "ga_patchify_strided: patch size %% 8, stride %% 4, a 16-byte aligned image at least one patch large");
    const long n = (long)B * ((H - P) / S + 1) * ((W - P) / S + 1) * (CH * P * P / 8);
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, x, (bf16_t*)out, B, CH, H, W, P, S);
    else hipLaunchKernelGGL(patchify_kernel<float>, dim3(blocks), dim3(256), 0, s, x, (float*)out, B, CH, H, W, P, S);
    return ga_check_launch("ga_patchify_strided");
}

extern "C" int ga_patchify(const float* x, void* out, int B, int CH, int H, int W, int P, int dtype, ga_stream_t stream) {
    GA_REQUIRE(P > 0 && H % P == 0 && W % P == 0, "ga_patchify: the patch size must divide the image");
    return ga_patchify_strided(x, out, B, CH, H, W, P, P, dtype, stream);
}

extern "C" int ga_vit_embed_fwd(const void* tok, const float* cls, const float* pos, void* x0, int B, int Np, int C, int dtype,
                                ga_stream_t stream) {
    GA_REQUIRE(tok && cls && pos && x0 && B > 0 && Np > 0 && C > 0 && C % 8 == 0 && aligned16(tok) && aligned16(x0) && aligned16(cls) && aligned16(pos),
               "ga_vit_embed_fwd: bad args (C %% 8, 16-byte alignment)");
    const long n = (long)B * (Np + 1) * (C / 8);
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(vit_embed_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)tok, cls, pos, (bf16_t*)x0, B, Np, C);
    else hipLaunchKernelGGL(vit_embed_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)tok, cls, pos, (float*)x0, B, Np, C);
    return ga_check_launch("ga_vit_embed_fwd");
}

extern "C" int ga_vit_embed_bwd(const void* dx0, void* dtok, float* dcls, float* dpos, int B, int Np, int C, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dx0 && dtok && dcls && dpos && B > 0 && Np > 0 && C > 0 && C % 8 == 0 && aligned16(dx0) && aligned16(dtok),
               "ga_vit_embed_bwd: bad args (C %% 8, 16-byte alignment)");
    const long n = (long)(Np + 1) * (C / 8);
    const int blocks = (int)std::max<long>(1, std::min<long>(4096, (n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(vit_embed_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dx0, (bf16_t*)dtok, dcls, dpos, B, Np, C);
    else hipLaunchKernelGGL(vit_embed_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)dx0, (float*)dtok, dcls, dpos, B, Np, C);
    return ga_check_launch("ga_vit_embed_bwd");
}
