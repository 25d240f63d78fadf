// GA-CSWin kernels for gfx950: cross-shaped (stripe) window attention with the LePE depthwise 3x3 of V
// (LePEAttention.forward, /root/reference/GA/ga_cswin.py:110-136; get_lepe :95-108; img2windows / windows2img
// :215-233), its backward, the LePE weight gradient, and the small layout helpers of the deep stem.
//
// Layout: q | k | v are the three C-wide column blocks of ONE [B*L][ldq] matrix (the qkv GEMM output, tokens in image
// raster order); branch i of a CSWinBlock owns channels [i*C/nb, (i+1)*C/nb) of each block and its own stripe shape
// Hs x Ws.  Window partition / merge (img2windows, windows2img) never materialise: a workgroup = one (image, window,
// head) and addresses its tokens in place.
//
// Two forms of each kernel:
//   * generic (any dtype, head_dim 8 / 16 / 32, N = Hs*Ws <= 128 tokens): one thread per query / key row, scores in
//     LDS, fp32 arithmetic -- the parity math mode and every narrow configuration;
//   * bf16, head_dim 32 (every stage of the GA-CSWin models): Q.K^T and P.V on v_mfma_f32_16x16x32_bf16, softmax by
//     wave shuffles in the S^T accumulator layout, V / K / dO / Q consumed column-wise with ds_read_b64_tr_b16.
#include <algorithm>
#include "common.h"

namespace {

struct WinGeom {
    int b, head, branch, hb;       // image, global head, branch, head inside the branch
    int Hs, Ws, N;                 // stripe shape, tokens per window
    int y0, x0;                    // window origin in the image
    int ch;                        // first channel of this head inside a C-wide block
    int chb;                       // first channel of this head inside its branch (LePE weight row)
};

__device__ __forceinline__ WinGeom win_geom(const ga_cswin_attn_desc& d, int item) {
    WinGeom g;
    const int hpb = d.heads / d.nbranch;
    const int hd = d.C / d.heads;
    g.head = item % d.heads;
    const int t = item / d.heads;
    g.branch = g.head / hpb;
    g.hb = g.head - g.branch * hpb;
    g.Hs = d.Hs[g.branch];
    g.Ws = d.Ws[g.branch];
    g.N = g.Hs * g.Ws;
    const int nwx = d.reso / g.Ws, nwin = (d.reso / g.Hs) * nwx;
    const int win = t % nwin;
    g.b = t / nwin;
    g.y0 = (win / nwx) * g.Hs;
    g.x0 = (win % nwx) * g.Ws;
    g.ch = g.head * hd;
    g.chb = g.hb * hd;
    return g;
}

// image row (token index in [0, B*L)) of window token t
__device__ __forceinline__ long tok_row(const ga_cswin_attn_desc& d, const WinGeom& g, int t) {
    const int ty = t / g.Ws, tx = t - ty * g.Ws;
    return ((long)g.b * d.reso + g.y0 + ty) * d.reso + g.x0 + tx;
}

__device__ __forceinline__ int xcd_walk(int bid, int nwg) {   // bijective XCD-aware remap (heads of a window share lines)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// =================================================================================================================
// generic forward: 128 threads, thread t = token t of the window
// LDS: q k v [N][HD+1] fp32, scores [N][N+1] fp32
// =================================================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(128) void cswin_attn_fwd_simple(const ga_cswin_attn_desc d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const WinGeom g = win_geom(d, blockIdx.x);
    const int N = g.N, LD = HD + 1;
    float* qs = sm;
    float* ks = qs + 128 * LD;
    float* vs = ks + 128 * LD;
    float* sc = vs + 128 * LD;           // [N][N+1]
    const int t = threadIdx.x;
    const T* qkv = reinterpret_cast<const T*>(d.qkv);
    long row = 0;
    if (t < N) {
        row = tok_row(d, g, t);
        const T* p = qkv + row * d.ldq + g.ch;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            qs[t * LD + c] = elt<T>::ld(p + c);
            ks[t * LD + c] = elt<T>::ld(p + d.C + c);
            vs[t * LD + c] = elt<T>::ld(p + 2 * d.C + c);
        }
    }
    __syncthreads();
    if (t >= N) return;
    float q[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] = qs[t * LD + c] * d.scale;   // q * scale, then q @ k^T (ga_cswin.py:125-126)
    float m = -3.0e38f;
    for (int j = 0; j < N; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s = fmaf(q[c], ks[j * LD + c], s);
        sc[t * (N + 1) + j] = s;
        m = fmaxf(m, s);
    }
    float o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
    float l = 0.f;
    for (int j = 0; j < N; ++j) {
        const float p = __expf(sc[t * (N + 1) + j] - m);
        l += p;
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = fmaf(p, vs[j * LD + c], o[c]);
    }
    const float inv = 1.f / l;
    // LePE: depthwise 3x3 of v inside the window, zero padded at the window border (ga_cswin.py:101-104)
    const float* w = d.lepe_w[g.branch] + (long)g.chb * 9;
    const float* bb = d.lepe_b[g.branch] + g.chb;
    const int ty = t / g.Ws, tx = t - ty * g.Ws;
    T* op = reinterpret_cast<T*>(d.out) + row * d.ldo + g.ch;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        float a = bb[c];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = ty + dy, xx = tx + dx;
                if ((unsigned)yy < (unsigned)g.Hs && (unsigned)xx < (unsigned)g.Ws)
                    a = fmaf(w[c * 9 + (dy + 1) * 3 + dx + 1], vs[(yy * g.Ws + xx) * LD + c], a);
            }
        elt<T>::st(op + c, fmaf(o[c], inv, a));
    }
}

// =================================================================================================================
// generic backward: dq, dk, dv (incl. the LePE transpose) of one (image, window, head)
// LDS: q k v do [128][HD+1] fp32; P, dS [N][N+1] fp32
// =================================================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(128) void cswin_attn_bwd_simple(const ga_cswin_attn_desc d, const void* dout_, void* dqkv_) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const WinGeom g = win_geom(d, blockIdx.x);
    const int N = g.N, LD = HD + 1, LS = N + 1;
    float* qs = sm;
    float* ks = qs + 128 * LD;
    float* vs = ks + 128 * LD;
    float* gs = vs + 128 * LD;           // dO
    float* P = gs + 128 * LD;
    float* dS = P + N * LS;
    const int t = threadIdx.x;
    const T* qkv = reinterpret_cast<const T*>(d.qkv);
    const T* dout = reinterpret_cast<const T*>(dout_);
    T* dqkv = reinterpret_cast<T*>(dqkv_);
    long row = 0;
    if (t < N) {
        row = tok_row(d, g, t);
        const T* p = qkv + row * d.ldq + g.ch;
        const T* pg = dout + row * d.ldo + g.ch;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            qs[t * LD + c] = elt<T>::ld(p + c);
            ks[t * LD + c] = elt<T>::ld(p + d.C + c);
            vs[t * LD + c] = elt<T>::ld(p + 2 * d.C + c);
            gs[t * LD + c] = elt<T>::ld(pg + c);
        }
    }
    __syncthreads();
    if (t < N) {                          // query row t
        float q[HD], go[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            q[c] = qs[t * LD + c] * d.scale;
            go[c] = gs[t * LD + c];
        }
        float m = -3.0e38f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < HD; ++c) {
                s = fmaf(q[c], ks[j * LD + c], s);
                dp = fmaf(go[c], vs[j * LD + c], dp);
            }
            P[t * LS + j] = s;
            dS[t * LS + j] = dp;
            m = fmaxf(m, s);
        }
        float l = 0.f;
        for (int j = 0; j < N; ++j) {
            const float p = __expf(P[t * LS + j] - m);
            P[t * LS + j] = p;
            l += p;
        }
        const float inv = 1.f / l;
        float delta = 0.f;
        for (int j = 0; j < N; ++j) {
            const float p = P[t * LS + j] * inv;
            P[t * LS + j] = p;
            delta = fmaf(p, dS[t * LS + j], delta);
        }
        float dq[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = 0.f;
        for (int j = 0; j < N; ++j) {
            const float ds = P[t * LS + j] * (dS[t * LS + j] - delta);
            dS[t * LS + j] = ds;
#pragma unroll
            for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, ks[j * LD + c], dq[c]);
        }
        T* o = dqkv + row * d.ldq + g.ch;
#pragma unroll
        for (int c = 0; c < HD; ++c) elt<T>::st(o + c, dq[c] * d.scale);
    }
    __syncthreads();
    if (t >= N) return;                   // key row t
    float dk[HD], dv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dk[c] = dv[c] = 0.f;
    for (int i = 0; i < N; ++i) {
        const float p = P[i * LS + t], ds = dS[i * LS + t];
#pragma unroll
        for (int c = 0; c < HD; ++c) {
            dv[c] = fmaf(p, gs[i * LD + c], dv[c]);
            dk[c] = fmaf(ds, qs[i * LD + c], dk[c]);
        }
    }
    // LePE transpose: dv[s] += sum_taps w[tap] * dO[s - tap]  (inside the window)
    const float* w = d.lepe_w[g.branch] + (long)g.chb * 9;
    const int ty = t / g.Ws, tx = t - ty * g.Ws;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = ty - dy, xx = tx - dx;
            if ((unsigned)yy < (unsigned)g.Hs && (unsigned)xx < (unsigned)g.Ws) {
#pragma unroll
                for (int c = 0; c < HD; ++c)
                    dv[c] = fmaf(w[c * 9 + (dy + 1) * 3 + dx + 1], gs[(yy * g.Ws + xx) * LD + c], dv[c]);
            }
        }
    T* o = dqkv + row * d.ldq + g.ch;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        elt<T>::st(o + d.C + c, dk[c] * d.scale);
        elt<T>::st(o + 2 * d.C + c, dv[c]);
    }
}

// =================================================================================================================
// bf16, head_dim 32, MFMA.  One workgroup (4 waves) per (image, window, head); NT = 16-token tiles (4: N <= 64,
// 7: N <= 112), KS = 32-key steps, KP = KS*32 rows per LDS tile (pad rows zero).
// LDS tiles are [KP][32] bf16 (64-byte rows); the 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3) so that the
// 16 rows x one chunk of a ds_read_b128 fragment read cover all 16 slots of the 256-byte bank row.
// =================================================================================================================
template <int NT> struct AT {
    static constexpr int KS = (NT + 1) / 2;
    static constexpr int KP = KS * 32;
    static constexpr int TILE = KP * 64;              // bytes
};

__device__ __forceinline__ unsigned tile_off(int r, int chunk) { return r * 64 + ((chunk ^ ((r >> 2) & 3)) << 4); }

// global [tokens][32 channels] bf16 column slice -> LDS tile (zero rows beyond N)
template <int NT>
__device__ __forceinline__ void load_tile(unsigned char* dst, const bf16_t* src, long ld, const ga_cswin_attn_desc& d,
                                          const WinGeom& g) {
    constexpr int KP = AT<NT>::KP;
    for (int i = threadIdx.x; i < KP * 4; i += 256) {
        const int r = i >> 2, c = i & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < g.N) v = *reinterpret_cast<const uint4*>(src + tok_row(d, g, r) * ld + c * 8);
        *reinterpret_cast<uint4*>(dst + tile_off(r, c)) = v;
    }
}

// A / B fragment of the 16x16x32 MFMA taken row-wise: row = r0 + (lane & 15), elements 8*(lane>>4) .. +7 of the 32 channels
__device__ __forceinline__ bf16x8_t row_frag(const unsigned char* tile, int r0, int lane) {
    const int r = r0 + (lane & 15);
    return *reinterpret_cast<const bf16x8_t*>(tile + tile_off(r, lane >> 4));
}

// B fragment taken column-wise (transposing LDS read): element j of lane (g = lane>>4, i = lane&15) is
// tile[row = k0 + 16*(j>>2) + 4g + (j&3)][col = 16*dt + i] -- the k order in which an S^T accumulator pair presents
// its rows when it is used as the A operand (cdna guide, "an accumulator tile as the next MFMA's operand")
__device__ __forceinline__ bf16x8_t col_frag_acc(const unsigned char* tile, int k0, int dt, int lane) {
    const int gq = lane >> 4, i = lane & 15;
    s16x4_t lo, hi;
    {
        const int r = k0 + 4 * gq + (i >> 2);
        const unsigned a = tile_off(r, 2 * dt + ((i & 3) >> 1)) + 8 * (i & 1);
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + a));
    }
    {
        const int r = k0 + 16 + 4 * gq + (i >> 2);
        const unsigned a = tile_off(r, 2 * dt + ((i & 3) >> 1)) + 8 * (i & 1);
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + a));
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return *reinterpret_cast<const bf16x8_t*>(&v);
}

// two accumulator tiles (rows 4g+r of tile a = k slots 0..3, of tile b = k slots 4..7) -> one A fragment
__device__ __forceinline__ bf16x8_t acc_pair_frag(const f32x4_t& a, const f32x4_t& b) {
    uint4 u;
    u.x = pack2bf(a[0], a[1]); u.y = pack2bf(a[2], a[3]); u.z = pack2bf(b[0], b[1]); u.w = pack2bf(b[2], b[3]);
    return *reinterpret_cast<const bf16x8_t*>(&u);
}

// S^T tiles of one 16-query tile: st[kt][r] = scale * sum_d K[16kt + 4g + r][d] * Q[q0 + (lane&15)][d]; keys >= N masked
template <int NT>
__device__ __forceinline__ void st_tiles(const unsigned char* Kt, const unsigned char* Qt, int q0, int N, float scale, int lane,
                                         f32x4_t (&st)[NT]) {
    const bf16x8_t qf = row_frag(Qt, q0, lane);
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Kt, 16 * kt, lane), qf, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * (lane >> 4) + r;
            st[kt][r] = key < N ? st[kt][r] * scale : -3.0e38f;
        }
    }
}

// softmax over the keys of each query column (in-lane over r / kt, then lanes l, l^16, l^32 hold the other key rows)
template <int NT>
__device__ __forceinline__ void softmax_cols(f32x4_t (&st)[NT], float& m_out, float& l_out) {
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, st[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __expf(st[kt][r] - m);    // masked keys: exp(-huge) = 0
            st[kt][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][r] *= inv;
    m_out = m;
    l_out = l;
}

// out tile [16 rows q0..][32 ch] fp32 accumulators (o[dt][r]: row 4g+r, col 16dt + (lane&15)) -> bf16 rows of `stage`
// (this wave's own 1 KiB) -> 16-byte global stores, one token row per 4 lanes
__device__ __forceinline__ void store_tile(unsigned char* stage, const f32x4_t (&o)[2], int q0, bf16_t* dst, long ld,
                                           const ga_cswin_attn_desc& d, const WinGeom& g, int lane) {
    bf16_t* st = reinterpret_cast<bf16_t*>(stage);
    asm volatile("" ::: "memory");                        // the previous use of this staging piece is complete
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * (lane >> 4) + r) * 32 + 16 * dt + (lane & 15)] = f2bf(o[dt][r]);
    // LDS accesses of one wave complete in order: the reads below see the writes above (the asm statement keeps the
    // compiler from re-ordering the differently typed accesses)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int r = lane >> 2, c = lane & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(stage + r * 64 + c * 16);
    const int t = q0 + r;
    if (t < g.N) *reinterpret_cast<uint4*>(dst + tok_row(d, g, t) * ld + c * 8) = v;
}


// ---------------------------------------------------------------------------------------------------------------
// LePE on the matrix cores.  out[t][c] += sum_tap w[c][tap] * X[nbr(t, tap)][c] is, per 16-channel half dt and per PAIR
// of taps, one 16x16x32 product  D[t][c] += sum_{k = (s, c')} A[t][k] * B[k][c]  with
//     A[t][(s, c')] = X[nbr(t, tap_{2p+s})][16 dt + c']   (0 outside the window)         -- one 16-byte LDS read per lane
//     B[(s, c')][c] = (c' == c) * w[16 dt + c][tap_{2p+s}]                                   -- 10 fragments held in registers
// (tap 9 does not exist: its B rows are zero).  5 MFMAs per half instead of ~36 VALU multiply-adds, bounds tests and 2-byte
// LDS reads per accumulator element; the transpose (backward, X = dO) only flips the neighbour offsets.
// ---------------------------------------------------------------------------------------------------------------
struct LepeW { bf16x8_t f[2][5]; };

__device__ __forceinline__ LepeW lepe_wfrags(const float* wp, int lane) {
    LepeW W;
    const int g = lane >> 4, c = lane & 15, e0 = c - 8 * (g & 1);          // this lane's non-zero element, if 0 <= e0 < 8
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const int tap = 2 * p + (g >> 1);
            const float w = tap < 9 ? wp[(16 * dt + c) * 9 + tap] : 0.f;
            const unsigned wb = (unsigned)f2bf(w);
            unsigned u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = (e0 == 2 * q ? wb : 0u) | (e0 == 2 * q + 1 ? wb << 16 : 0u);
            const uint4 v = make_uint4(u[0], u[1], u[2], u[3]);
            W.f[dt][p] = *reinterpret_cast<const bf16x8_t*>(&v);
        }
    return W;
}

// o[dt] (+)= LePE rows q0 .. q0+15 of `tile` (TR: the transpose); zoff = LDS byte offset (relative to tile) of 16 zero bytes
template <bool TR>
__device__ __forceinline__ void lepe_mfma(f32x4_t (&o)[2], const unsigned char* tile, int q0, const WinGeom& g, const LepeW& W,
                                          int lane, const unsigned char* zero16) {
    const int gq = lane >> 4, t = q0 + (lane & 15);
    const int ty = t / g.Ws, tx = t - ty * g.Ws;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int tap = 2 * p + (gq >> 1);
        const int dy = (tap * 11 >> 5) - 1, dx = tap - 3 * (tap * 11 >> 5) - 1;      // tap / 3 for tap < 12
        const int yy = TR ? ty - dy : ty + dy, xx = TR ? tx - dx : tx + dx;
        const bool ok = tap < 9 && t < g.N && (unsigned)yy < (unsigned)g.Hs && (unsigned)xx < (unsigned)g.Ws;
        const int row = yy * g.Ws + xx;
        const unsigned base = row * 64, sw = (row >> 2) & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const unsigned char* src = ok ? tile + base + (((2 * dt + (gq & 1)) ^ sw) << 4) : zero16;
            const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(src);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, W.f[dt][p], o[dt], 0, 0, 0);
        }
    }
}

// B fragments (both 16-channel halves) of the SHIFTED tile taken column-wise in the k order of col_frag_acc: element j of
// lane (g, i) is tile[nbr(row, tap)][16 dt + i] with row = k0 + 16 (j >> 2) + 4 g + (j & 3); rows whose neighbour is outside
// the window read zeros.  ty / tx: window coordinates of this lane's two rows (h = 0, 1), computed once per k step.
__device__ __forceinline__ void col_frag_shift2(bf16x8_t (&out)[2], const unsigned char* tile, int lane, int dy, int dx, const WinGeom& g,
                                                const int (&ty)[2], const int (&tx)[2], const bool (&rok)[2], const unsigned char* zero16) {
    const int i = lane & 15;
    s16x4_t half[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int yy = ty[h] + dy, xx = tx[h] + dx;
        const bool ok = rok[h] && (unsigned)yy < (unsigned)g.Hs && (unsigned)xx < (unsigned)g.Ws;
        const int row = yy * g.Ws + xx;
        const unsigned base = row * 64 + 8 * (i & 1), sw = (row >> 2) & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const unsigned char* src = ok ? tile + base + (((2 * dt + ((i & 3) >> 1)) ^ sw) << 4) : zero16 + 8 * (i & 1);
            half[dt][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)src);
        }
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const s16x8_t v = {half[dt][0][0], half[dt][0][1], half[dt][0][2], half[dt][0][3],
                           half[dt][1][0], half[dt][1][1], half[dt][1][2], half[dt][1][3]};
        out[dt] = *reinterpret_cast<const bf16x8_t*>(&v);
    }
}

// LePE weight-gradient partial of one (window, head) on the matrix cores: per tap, D[c][c''] = sum_t dO[t][c] * V[nbr(t, tap)][c'']
// over the window's tokens (k = token, both operands read column-wise); its diagonal is dw[c][tap], and with an all-ones
// operand db[c].  Wave w takes taps w, w+4 and (w = 0) 8 / (w = 1) the bias.  ws_item[32][10] gets plain stores.
template <int NT>
__device__ __forceinline__ void lepe_wgrad_mfma(const unsigned char* Gt, const unsigned char* Vt, const WinGeom& g, float* ws_item,
                                                int lane, int wave, const unsigned char* zero16) {
    constexpr int KS = AT<NT>::KS;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    bf16x8_t gf[2][KS];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gf[dt][ks] = col_frag_acc(Gt, 32 * ks, dt, lane);
    const bool diag = ((lane & 15) >> 2) == (lane >> 4);
    // window coordinates of this lane's rows of every k step (row = 32 ks + 16 h + 4 g + (i >> 2)): shared by the wave's taps
    int ty[KS][2], tx[KS][2];
    bool rok[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 32 * ks + 16 * h + 4 * (lane >> 4) + ((lane & 15) >> 2);
            ty[ks][h] = r / g.Ws;
            tx[ks][h] = r - ty[ks][h] * g.Ws;
            rok[ks][h] = r < g.N;
        }
#pragma unroll 1
    for (int n = 0; n < 3; ++n) {
        const int tap = wave + 4 * n;                       // 0..11; 9 = bias, > 9: nothing
        if (tap > 9) break;
        f32x4_t acc[2] = {zero, zero};
        if (tap < 9) {
            const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t vf[2];
                col_frag_shift2(vf, Vt, lane, dy, dx, g, ty[ks], tx[ks], rok[ks], zero16);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[dt][ks], vf[dt], acc[dt], 0, 0, 0);
            }
        } else {
            const uint4 o4 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);      // bf16 ones
            const bf16x8_t ones = *reinterpret_cast<const bf16x8_t*>(&o4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[dt][ks], ones, acc[dt], 0, 0, 0);
        }
        if (diag) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int r = lane & 3;
                const float v = r == 0 ? acc[dt][0] : r == 1 ? acc[dt][1] : r == 2 ? acc[dt][2] : acc[dt][3];
                ws_item[(16 * dt + (lane & 15)) * 10 + tap] = v;
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void cswin_attn_fwd_mfma(const ga_cswin_attn_desc d, const int nwg) {
    using A = AT<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qt = smem;
    unsigned char* Kt = Qt + A::TILE;
    unsigned char* Vt = Kt + A::TILE;
    unsigned char* Stage = Vt + A::TILE;              // 4 x 1 KiB
    unsigned char* Zero16 = Stage + 4096;              // 16 zero bytes: the LePE operand of neighbours outside the window
    const WinGeom g = win_geom(d, xcd_walk(blockIdx.x, nwg));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(d.qkv) + g.ch;
    load_tile<NT>(Qt, qkv, d.ldq, d, g);
    load_tile<NT>(Kt, qkv + d.C, d.ldq, d, g);
    load_tile<NT>(Vt, qkv + 2 * d.C, d.ldq, d, g);
    if (threadIdx.x < 4) reinterpret_cast<unsigned*>(Zero16)[threadIdx.x] = 0u;
    const LepeW LW = lepe_wfrags(d.lepe_w[g.branch] + (long)g.chb * 9, lane);
    float lbias[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) lbias[dt] = d.lepe_b[g.branch][g.chb + 16 * dt + (lane & 15)];
    __syncthreads();
    for (int qt = wave; qt < NT; qt += 4) {
        const int q0 = 16 * qt;
        if (q0 >= g.N) break;
        f32x4_t st[NT];
        st_tiles<NT>(Kt, Qt, q0, g.N, d.scale, lane, st);
        float m, l;
        softmax_cols<NT>(st, m, l);
        f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            const bf16x8_t pf = acc_pair_frag(st[2 * ks], 2 * ks + 1 < NT ? st[2 * ks + 1 < NT ? 2 * ks + 1 : 0] : zero);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, col_frag_acc(Vt, 32 * ks, dt, lane), o[dt], 0, 0, 0);
        }
        // + LePE of these 16 rows (5 + 5 MFMAs against the banded tap operand), + its bias
        lepe_mfma<false>(o, Vt, q0, g, LW, lane, Zero16);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] += lbias[dt];
        store_tile(Stage + wave * 1024, o, q0, reinterpret_cast<bf16_t*>(d.out) + g.ch, d.ldo, d, g, lane);
    }
}

// backward, bf16 hd 32.  Phase A (per 16-query tile): S^T, P^T, dP^T = V.dO^T, delta, dS^T -> dQ = scale * dS.K; the
// row statistics lse / delta go to LDS.  Phase B (per 16-key tile): S = Q.K^T and dP = dO.V^T recomputed in the
// [query][key] layout, whose accumulators are directly the A operands of dV += P^T.dO and dK += dS^T.Q.
template <int NT>
__global__ __launch_bounds__(256) void cswin_attn_bwd_mfma(const ga_cswin_attn_desc d, const void* dout_, void* dqkv_,
                                                           float* lepe_ws, const int nwg) {
    using A = AT<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Qt = smem;
    unsigned char* Kt = Qt + A::TILE;
    unsigned char* Vt = Kt + A::TILE;
    unsigned char* Gt = Vt + A::TILE;                 // dO
    unsigned char* Stage = Gt + A::TILE;              // 4 x 1 KiB
    float* lse = reinterpret_cast<float*>(Stage + 4096);   // [KP]
    float* dlt = lse + A::KP;                              // [KP]
    unsigned char* Zero16 = reinterpret_cast<unsigned char*>(dlt + A::KP);
    const int item = xcd_walk(blockIdx.x, nwg);
    const WinGeom g = win_geom(d, item);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(d.qkv) + g.ch;
    bf16_t* dqkv = reinterpret_cast<bf16_t*>(dqkv_) + g.ch;
    load_tile<NT>(Qt, qkv, d.ldq, d, g);
    load_tile<NT>(Kt, qkv + d.C, d.ldq, d, g);
    load_tile<NT>(Vt, qkv + 2 * d.C, d.ldq, d, g);
    load_tile<NT>(Gt, reinterpret_cast<const bf16_t*>(dout_) + g.ch, d.ldo, d, g);
    for (int i = threadIdx.x; i < 2 * A::KP + 4; i += 256) lse[i] = 0.f;     // (+ the 16 zero bytes behind dlt)
    __syncthreads();
    if (lepe_ws) lepe_wgrad_mfma<NT>(Gt, Vt, g, lepe_ws + (long)item * 320, lane, wave, Zero16);
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    // ---------------- phase A ----------------
    for (int qt = wave; qt < NT; qt += 4) {
        const int q0 = 16 * qt;
        if (q0 >= g.N) break;
        f32x4_t st[NT], dp[NT];
        st_tiles<NT>(Kt, Qt, q0, g.N, d.scale, lane, st);
        float m, l;
        softmax_cols<NT>(st, m, l);
        const bf16x8_t gf = row_frag(Gt, q0, lane);
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vt, 16 * kt, lane), gf, zero, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) delta = fmaf(st[kt][r], dp[kt][r], delta);
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[kt][r] = st[kt][r] * (dp[kt][r] - delta);     // dS^T
        if (lane < 16) {
            lse[q0 + lane] = m + __logf(l);
            dlt[q0 + lane] = delta;
        }
        f32x4_t o[2] = {zero, zero};
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            const bf16x8_t sf = acc_pair_frag(dp[2 * ks], 2 * ks + 1 < NT ? dp[2 * ks + 1 < NT ? 2 * ks + 1 : 0] : zero);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, col_frag_acc(Kt, 32 * ks, dt, lane), o[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= d.scale;
        store_tile(Stage + wave * 1024, o, q0, dqkv, d.ldq, d, g, lane);
    }
    __syncthreads();
    // ---------------- phase B ----------------
    const LepeW LW = lepe_wfrags(d.lepe_w[g.branch] + (long)g.chb * 9, lane);
    for (int kt = wave; kt < NT; kt += 4) {
        const int k0 = 16 * kt;
        if (k0 >= g.N) break;
        const bf16x8_t kf = row_frag(Kt, k0, lane), vf = row_frag(Vt, k0, lane);
        const bool key_ok = k0 + (lane & 15) < g.N;
        f32x4_t dv[2] = {zero, zero}, dk[2] = {zero, zero};
#pragma unroll
        for (int qs = 0; qs < A::KS; ++qs) {
            f32x4_t p[2], ds[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qt = 2 * qs + h;
                if (qt < NT) {
                    const int q0 = 16 * qt;
                    // [query 4g + r][key lane&15]
                    p[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qt, q0, lane), kf, zero, 0, 0, 0);
                    ds[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Gt, q0, lane), vf, zero, 0, 0, 0);
                    const f32x4_t ls = *reinterpret_cast<const f32x4_t*>(lse + q0 + 4 * (lane >> 4));
                    const f32x4_t de = *reinterpret_cast<const f32x4_t*>(dlt + q0 + 4 * (lane >> 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = key_ok ? __expf(p[h][r] * d.scale - ls[r]) : 0.f;
                        p[h][r] = pv;
                        ds[h][r] = pv * (ds[h][r] - de[r]);
                    }
                } else {
                    p[h] = zero;
                    ds[h] = zero;
                }
            }
            const bf16x8_t pf = acc_pair_frag(p[0], p[1]), sf = acc_pair_frag(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, col_frag_acc(Gt, 32 * qs, dt, lane), dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, col_frag_acc(Qt, 32 * qs, dt, lane), dk[dt], 0, 0, 0);
            }
        }
        // dv += LePE^T(dO); dk *= scale
        lepe_mfma<true>(dv, Gt, k0, g, LW, lane, Zero16);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dk[dt][r] *= d.scale;
        store_tile(Stage + wave * 1024, dk, k0, dqkv + d.C, d.ldq, d, g, lane);
        store_tile(Stage + wave * 1024, dv, k0, dqkv + 2 * d.C, d.ldq, d, g, lane);
    }
}

// =================================================================================================================
// LePE weight gradient: dw[ch][tap] += sum_tokens dO[t][ch] * v[t + tap][ch] (inside the window), db[ch] += sum dO
// thread = one 8-channel chunk x a strided set of tokens; workgroup partials through LDS, then fp32 atomics
// =================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void lepe_wgrad_kernel(const ga_cswin_attn_desc d, const void* dout_, float* dw0, float* db0,
                                                         float* dw1, float* db1) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rows per block][C8][80]
    const int C8 = d.C / 8;
    const int cc = threadIdx.x % C8, rib = threadIdx.x / C8, rpb = 256 / C8;
    const T* dout = reinterpret_cast<const T*>(dout_);
    const T* v = reinterpret_cast<const T*>(d.qkv) + 2 * d.C;
    const int cb = d.C / d.nbranch;
    const int branch = (cc * 8) / cb;
    const int Hs = d.Hs[branch], Ws = d.Ws[branch];
    const long rows = (long)d.B * d.reso * d.reso;
    float acc[10][8];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    if (rib < rpb) {
        for (long row = (long)blockIdx.x * rpb + rib; row < rows; row += (long)gridDim.x * rpb) {
            const int x = (int)(row % d.reso), y = (int)((row / d.reso) % d.reso);
            const int ty = y % Hs, tx = x % Ws;
            float g[8];
            load8(dout + row * d.ldo + cc * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[9][e] += g[e];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if ((unsigned)(ty + dy) < (unsigned)Hs && (unsigned)(tx + dx) < (unsigned)Ws) {
                        float vv[8];
                        load8(v + (row + (long)dy * d.reso + dx) * d.ldq + cc * 8, vv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[(dy + 1) * 3 + dx + 1][e] = fmaf(g[e], vv[e], acc[(dy + 1) * 3 + dx + 1][e]);
                    }
                }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(rib * C8 + cc) * 80 + k * 8 + e] = acc[k][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C8 * 80; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += red[r * C8 * 80 + i];
        const int c8 = i / 80, k = (i % 80) / 8, e = i % 8;
        const int ch = c8 * 8 + e, br = ch / cb, chb = ch - br * cb;
        float* dw = br ? dw1 : dw0;
        float* db = br ? db1 : db0;
        if (k < 9) atomicAdd(dw + (long)chb * 9 + k, s);
        else atomicAdd(db + chb, s);
    }
}

// sum of the per-(window, head) partials the fused backward wrote: ws[(t * heads + head)][32][10] over t = (image, window).
// grid (heads, 5, splits); thread = one of 64 outputs x 4 interleaved partial sums; fp32 atomics across the splits.
__global__ __launch_bounds__(256) void lepe_wgrad_reduce_kernel(const float* __restrict__ ws, int T, int heads, int hpb,
                                                                float* dw0, float* db0, float* dw1, float* db1) {
    __shared__ float red[256];
    const int head = blockIdx.x, i = blockIdx.y * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int per = (T + gridDim.z - 1) / gridDim.z;
    const int t0 = blockIdx.z * per, t1 = min(T, t0 + per);
    float s = 0.f;
    for (int t = t0 + part; t < t1; t += 4) s += ws[((long)t * heads + head) * 320 + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        s = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        const int br = head / hpb, chb = (head - br * hpb) * 32 + i / 10, k = i % 10;
        float* dw = br ? dw1 : dw0;
        float* db = br ? db1 : db0;
        if (k < 9) atomicAdd(dw + (long)chb * 9 + k, s);
        else atomicAdd(db + chb, s);
    }
}

// =================================================================================================================
// deep-stem helpers
// =================================================================================================================
// fp32 NCHW [B,3,H,W] -> NHWC [B,H,W,8] (channels 3..7 zero) in T: the first stem conv then is an ordinary NHWC gather
template <typename T>
__global__ __launch_bounds__(256) void nchw3_to_nhwc8_kernel(const float* __restrict__ x, T* __restrict__ y, long npix, long HW) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        const long b = i / HW, p = i - b * HW;
        const float* s = x + b * 3 * HW + p;
        float v[8] = {s[0], s[HW], s[2 * HW], 0.f, 0.f, 0.f, 0.f, 0.f};
        store8(y + i * 8, v);
    }
}

// conv weight fp32 [Co][Ci][T taps] -> effective [Co][ldo] in T with k = tap*Cp + ci (ci < Ci, zero padded to Cp)
template <typename T>
__global__ __launch_bounds__(256) void convw_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int Tp,
                                                         int Cp, long ldo) {
    const long n = (long)Co * ldo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int co = (int)(i / ldo), k = (int)(i - (long)co * ldo);
        const int tap = k / Cp, ci = k - tap * Cp;
        float v = 0.f;
        if (tap < Tp && ci < Ci) v = w[((long)co * Ci + ci) * Tp + tap];
        elt<T>::st(out + i, v);
    }
}

// dW[co][ci][tap] += G[co][tap*Cp + ci]
__global__ __launch_bounds__(256) void convw_unpack_grad_kernel(const float* __restrict__ G, float* __restrict__ dW, int Co, int Ci,
                                                                int Tp, int Cp, long ldg) {
    const long n = (long)Co * Ci * Tp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int tap = (int)(i % Tp);
        const long t = i / Tp;
        const int ci = (int)(t % Ci), co = (int)(t / Ci);
        dW[i] += G[(long)co * ldg + tap * Cp + ci];
    }
}

// data-gradient operand of a 3x3 / stride 2 / pad 1 convolution: Bt[(py,px,ci)][(ay,ax,co)] = w[co][ci][ky][kx] with
// (py,ay) -> ky: (0,0) -> 1, (1,0) -> 2, (1,1) -> 0, (0,1) -> none (same for x); zeros elsewhere.  The product
// GA_A_NEIGH2(dy) . Bt^T scattered with GA_C_UNPATCH2 is the transposed convolution.
template <typename T>
__global__ __launch_bounds__(256) void conv3s2_dgrad_prep_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci,
                                                                 long ldo) {
    const long n = (long)4 * Ci * ldo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int row = (int)(i / ldo), k = (int)(i - (long)row * ldo);
        const int pq = row / Ci, ci = row - pq * Ci;
        const int py = pq >> 1, px = pq & 1;
        float v = 0.f;
        if (k < 4 * Co) {
            const int a = k / Co, co = k - a * Co;
            const int ay = a >> 1, ax = a & 1;
            const int ky = py == 0 ? (ay == 0 ? 1 : -1) : (ay == 0 ? 2 : 0);
            const int kx = px == 0 ? (ax == 0 ? 1 : -1) : (ax == 0 ? 2 : 0);
            if (ky >= 0 && kx >= 0) v = w[(((long)co * Ci + ci) * 3 + ky) * 3 + kx];
        }
        elt<T>::st(out + i, v);
    }
}

// =================================================================================================================
// LayerNorm (affine, over C) followed by GELU, as the deep stem applies it between its convs (ga_cswin.py:466-473):
// a row of C = 8 * G channels on G lanes (G a power of two <= 64), one 8-channel chunk per lane.
//   fwd: y = gelu(xhat * w + b), mean / rstd saved;   bwd: gz = g * gelu'(xhat * w + b), then the LayerNorm backward
// =================================================================================================================
template <typename T, int G>
__global__ __launch_bounds__(256) void ln_gelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, T* __restrict__ y, float* __restrict__ mean,
                                                          float* __restrict__ rstd, long rows, float eps) {
    constexpr int C = 8 * G, RPB = 256 / G;
    const int lg = threadIdx.x % G, rib = threadIdx.x / G;
    float wv[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        wv[e] = w[lg * 8 + e];
        bv[e] = b[lg * 8 + e];
    }
    for (long row = (long)blockIdx.x * RPB + rib; row < rows; row += (long)gridDim.x * RPB) {
        float v[8];
        load8(x + row * C + lg * 8, v);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
        const float mu = group_sum<G>(s) * (1.f / C);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) q += (v[e] - mu) * (v[e] - mu);
        const float rs = rsqrtf(group_sum<G>(q) * (1.f / C) + eps);
        if (lg == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_f(fmaf((v[e] - mu) * rs, wv[e], bv[e]));
        store8(y + row * C + lg * 8, v);
    }
}

template <typename T, int G>
__global__ __launch_bounds__(256) void ln_gelu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                          long rows) {
    constexpr int C = 8 * G, RPB = 256 / G;
    __shared__ float red[2 * 256 * 8];
    const int lg = threadIdx.x % G, rib = threadIdx.x / G;
    float wv[8], bv[8], aw[8], ab[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        wv[e] = w[lg * 8 + e];
        bv[e] = b[lg * 8 + e];
        aw[e] = ab[e] = 0.f;
    }
    for (long row = (long)blockIdx.x * RPB + rib; row < rows; row += (long)gridDim.x * RPB) {
        float gv[8], xh[8];
        load8(g + row * C + lg * 8, gv);
        load8(x + row * C + lg * 8, xh);
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh[e] = (xh[e] - mu) * rs;
            const float gz = gv[e] * gelu_grad_f(fmaf(xh[e], wv[e], bv[e]));
            aw[e] = fmaf(gz, xh[e], aw[e]);
            ab[e] += gz;
            gv[e] = gz * wv[e];
            s1 += gv[e];
            s2 = fmaf(gv[e], xh[e], s2);
        }
        s1 = group_sum<G>(s1) * (1.f / C);
        s2 = group_sum<G>(s2) * (1.f / C);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = rs * (gv[e] - s1 - xh[e] * s2);
        store8(dx + row * C + lg * 8, gv);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[threadIdx.x * 8 + e] = aw[e];
        red[2048 + threadIdx.x * 8 + e] = ab[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, bb = 0.f;
        for (int r = 0; r < RPB; ++r) {
            a += red[(r * G + c / 8) * 8 + (c & 7)];
            bb += red[2048 + (r * G + c / 8) * 8 + (c & 7)];
        }
        atomicAdd(dw + c, a);
        atomicAdd(db + c, bb);
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_desc(const ga_cswin_attn_desc* d, const char* what) {
    GA_REQUIRE(d && d->qkv && d->B > 0 && d->reso > 0 && d->C > 0 && d->heads > 0, "%s: null / empty descriptor", what);
    GA_REQUIRE(d->nbranch == 1 || d->nbranch == 2, "%s: nbranch must be 1 or 2", what);
    GA_REQUIRE(d->heads % d->nbranch == 0 && d->C % d->heads == 0, "%s: C=%d heads=%d nbranch=%d do not divide", what, d->C,
               d->heads, d->nbranch);
    const int hd = d->C / d->heads;
    GA_REQUIRE(hd == 8 || hd == 16 || hd == 32, "%s: head_dim %d not in {8, 16, 32}", what, hd);
    for (int i = 0; i < d->nbranch; ++i) {
        GA_REQUIRE(d->Hs[i] > 0 && d->Ws[i] > 0 && d->reso % d->Hs[i] == 0 && d->reso % d->Ws[i] == 0 &&
                       d->Hs[i] * d->Ws[i] <= 128,
                   "%s: stripe %dx%d does not tile %d or exceeds 128 tokens", what, d->Hs[i], d->Ws[i], d->reso);
        GA_REQUIRE(d->lepe_w[i] && d->lepe_b[i], "%s: LePE weights missing", what);
    }
    if (d->nbranch == 2)
        GA_REQUIRE(d->reso / d->Hs[0] * (d->reso / d->Ws[0]) == d->reso / d->Hs[1] * (d->reso / d->Ws[1]),
                   "%s: both branches must have the same number of windows", what);
    GA_REQUIRE(d->dtype == GA_F32 || d->dtype == GA_BF16, "%s: bad dtype", what);
    const int epc = d->dtype == GA_BF16 ? 8 : 4;
    GA_REQUIRE(aligned16(d->qkv) && d->ldq % epc == 0 && d->ldq >= 3 * d->C && d->C % 8 == 0, "%s: qkv alignment / ldq", what);
    return GA_OK;
}

bool use_mfma(const ga_cswin_attn_desc* d) {
    const int force = GA_KNOB("CSWIN_MFMA", 1);            // 0: generic form everywhere (tests switch it through ga_set_knob)
    if (!force || d->dtype != GA_BF16 || d->C / d->heads != 32) return false;
    for (int i = 0; i < d->nbranch; ++i)
        if (d->Hs[i] * d->Ws[i] > 112) return false;
    return true;
}

template <typename K> bool set_lds(K kern, size_t bytes) {
    return bytes <= 65536 ||
           hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace

extern "C" int ga_cswin_attn_fwd(const ga_cswin_attn_desc* d, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_cswin_attn_fwd")) return rc;
    GA_REQUIRE(d->out && aligned16(d->out) && d->ldo >= d->C && d->ldo % (d->dtype == GA_BF16 ? 8 : 4) == 0,
               "ga_cswin_attn_fwd: out alignment / ldo");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nwin = (d->reso / d->Hs[0]) * (d->reso / d->Ws[0]);
    const int items = d->B * nwin * d->heads;
    const int hd = d->C / d->heads;
    int nmax = 0;
    for (int i = 0; i < d->nbranch; ++i) nmax = std::max(nmax, d->Hs[i] * d->Ws[i]);
    if (use_mfma(d)) {
        if (nmax <= 64) {
            const size_t lds = 3 * AT<4>::TILE + 4096 + 16;
            hipLaunchKernelGGL(cswin_attn_fwd_mfma<4>, dim3(items), dim3(256), lds, s, *d, items);
        } else {
            const size_t lds = 3 * AT<7>::TILE + 4096 + 16;
            hipLaunchKernelGGL(cswin_attn_fwd_mfma<7>, dim3(items), dim3(256), lds, s, *d, items);
        }
        return ga_check_launch("ga_cswin_attn_fwd");
    }
    const size_t lds = ((size_t)3 * 128 * (hd + 1) + (size_t)nmax * (nmax + 1)) * sizeof(float);
#define GA_FWD(T, HD)                                                                                              \
    do {                                                                                                           \
        GA_REQUIRE(set_lds(cswin_attn_fwd_simple<T, HD>, lds), "ga_cswin_attn_fwd: cannot reserve %zu B of LDS", lds); \
        hipLaunchKernelGGL((cswin_attn_fwd_simple<T, HD>), dim3(items), dim3(128), lds, s, *d);                    \
    } while (0)
    if (d->dtype == GA_BF16) {
        if (hd == 8) GA_FWD(bf16_t, 8); else if (hd == 16) GA_FWD(bf16_t, 16); else GA_FWD(bf16_t, 32);
    } else {
        if (hd == 8) GA_FWD(float, 8); else if (hd == 16) GA_FWD(float, 16); else GA_FWD(float, 32);
    }
#undef GA_FWD
    return ga_check_launch("ga_cswin_attn_fwd");
}

extern "C" size_t ga_cswin_attn_bwd_workspace(const ga_cswin_attn_desc* d) {
    if (!d || check_desc(d, "ga_cswin_attn_bwd_workspace") || !use_mfma(d)) return 0;
    const size_t nwin = (size_t)(d->reso / d->Hs[0]) * (d->reso / d->Ws[0]);
    return (size_t)d->B * nwin * d->heads * 320 * sizeof(float);
}

extern "C" int ga_cswin_attn_bwd(const ga_cswin_attn_desc* d, const void* dout, void* dqkv, void* lepe_ws, size_t ws_bytes,
                                 ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_cswin_attn_bwd")) return rc;
    GA_REQUIRE(!lepe_ws || (use_mfma(d) && ws_bytes >= ga_cswin_attn_bwd_workspace(d) && aligned16(lepe_ws)),
               "ga_cswin_attn_bwd: LePE workspace of %zu B given, ga_cswin_attn_bwd_workspace() asks for %zu B (0 = this shape does "
               "not take one: use ga_cswin_lepe_wgrad)", ws_bytes, ga_cswin_attn_bwd_workspace(d));
    GA_REQUIRE(dout && dqkv && aligned16(dout) && aligned16(dqkv) && d->ldo >= d->C && d->ldo % (d->dtype == GA_BF16 ? 8 : 4) == 0,
               "ga_cswin_attn_bwd: dout / dqkv alignment");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nwin = (d->reso / d->Hs[0]) * (d->reso / d->Ws[0]);
    const int items = d->B * nwin * d->heads;
    const int hd = d->C / d->heads;
    int nmax = 0;
    for (int i = 0; i < d->nbranch; ++i) nmax = std::max(nmax, d->Hs[i] * d->Ws[i]);
    if (use_mfma(d)) {
        if (nmax <= 64) {
            const size_t lds = 4 * AT<4>::TILE + 4096 + 2 * AT<4>::KP * sizeof(float) + 16;
            hipLaunchKernelGGL(cswin_attn_bwd_mfma<4>, dim3(items), dim3(256), lds, s, *d, dout, dqkv, (float*)lepe_ws, items);
        } else {
            const size_t lds = 4 * AT<7>::TILE + 4096 + 2 * AT<7>::KP * sizeof(float) + 16;
            hipLaunchKernelGGL(cswin_attn_bwd_mfma<7>, dim3(items), dim3(256), lds, s, *d, dout, dqkv, (float*)lepe_ws, items);
        }
        return ga_check_launch("ga_cswin_attn_bwd");
    }
    const size_t lds = ((size_t)4 * 128 * (hd + 1) + (size_t)2 * nmax * (nmax + 1)) * sizeof(float);
#define GA_BWD(T, HD)                                                                                              \
    do {                                                                                                           \
        GA_REQUIRE(set_lds(cswin_attn_bwd_simple<T, HD>, lds), "ga_cswin_attn_bwd: cannot reserve %zu B of LDS", lds); \
        hipLaunchKernelGGL((cswin_attn_bwd_simple<T, HD>), dim3(items), dim3(128), lds, s, *d, dout, dqkv);        \
    } while (0)
    if (d->dtype == GA_BF16) {
        if (hd == 8) GA_BWD(bf16_t, 8); else if (hd == 16) GA_BWD(bf16_t, 16); else GA_BWD(bf16_t, 32);
    } else {
        if (hd == 8) GA_BWD(float, 8); else if (hd == 16) GA_BWD(float, 16); else GA_BWD(float, 32);
    }
#undef GA_BWD
    return ga_check_launch("ga_cswin_attn_bwd");
}

extern "C" int ga_cswin_lepe_wgrad(const ga_cswin_attn_desc* d, const void* dout, float* dw0, float* db0, float* dw1,
                                   float* db1, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_cswin_lepe_wgrad")) return rc;
    GA_REQUIRE(dout && dw0 && db0 && (d->nbranch == 1 || (dw1 && db1)), "ga_cswin_lepe_wgrad: null gradient");
    GA_REQUIRE(d->C / 8 <= 256 && (d->C / d->nbranch) % 8 == 0, "ga_cswin_lepe_wgrad: C=%d unsupported", d->C);
    const int C8 = d->C / 8, rpb = 256 / C8;
    const long rows = (long)d->B * d->reso * d->reso;
    const int grid = (int)std::max<long>(1, std::min<long>(1024, (rows + 4L * rpb - 1) / (4L * rpb)));
    const size_t lds = (size_t)rpb * C8 * 80 * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == GA_BF16) {
        GA_REQUIRE(set_lds(lepe_wgrad_kernel<bf16_t>, lds), "ga_cswin_lepe_wgrad: LDS");
        hipLaunchKernelGGL(lepe_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), lds, s, *d, dout, dw0, db0, dw1, db1);
    } else {
        GA_REQUIRE(set_lds(lepe_wgrad_kernel<float>, lds), "ga_cswin_lepe_wgrad: LDS");
        hipLaunchKernelGGL(lepe_wgrad_kernel<float>, dim3(grid), dim3(256), lds, s, *d, dout, dw0, db0, dw1, db1);
    }
    return ga_check_launch("ga_cswin_lepe_wgrad");
}

extern "C" int ga_cswin_lepe_wgrad_reduce(const ga_cswin_attn_desc* d, const void* lepe_ws, float* dw0, float* db0, float* dw1,
                                          float* db1, ga_stream_t stream) {
    if (int rc = check_desc(d, "ga_cswin_lepe_wgrad_reduce")) return rc;
    GA_REQUIRE(lepe_ws && dw0 && db0 && (d->nbranch == 1 || (dw1 && db1)), "ga_cswin_lepe_wgrad_reduce: null pointer");
    GA_REQUIRE(use_mfma(d), "ga_cswin_lepe_wgrad_reduce: this shape has no fused partials (ga_cswin_attn_bwd_workspace() == 0)");
    const int T = d->B * (d->reso / d->Hs[0]) * (d->reso / d->Ws[0]);
    // a thread sums T / splits / 4 partials one load after the other (latency-bound): 16-32 loads per thread (T / 256 splits: 21 us
    // per launch for 5 MB of partials, 26 launches per step on the dgrad chain)
    const int splits = std::max(1, std::min(64, T / 32));
    hipLaunchKernelGGL(lepe_wgrad_reduce_kernel, dim3(d->heads, 5, splits), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float*>(lepe_ws), T, d->heads, d->heads / d->nbranch, dw0, db0, dw1, db1);
    return ga_check_launch("ga_cswin_lepe_wgrad_reduce");
}

extern "C" int ga_nchw3_to_nhwc8(const float* x, void* y, int B, int H, int W, int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && aligned16(y), "ga_nchw3_to_nhwc8: bad args");
    const long npix = (long)B * H * W;
    const int grid = (int)std::min<long>(4096, (npix + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(nchw3_to_nhwc8_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, (bf16_t*)y, npix, (long)H * W);
    else hipLaunchKernelGGL(nchw3_to_nhwc8_kernel<float>, dim3(grid), dim3(256), 0, s, x, (float*)y, npix, (long)H * W);
    return ga_check_launch("ga_nchw3_to_nhwc8");
}

extern "C" int ga_convw_pack(const float* w, void* out, int Co, int Ci, int taps, int Cp, int64_t ldo, int dtype,
                             ga_stream_t stream) {
    GA_REQUIRE(w && out && Co > 0 && Ci > 0 && taps > 0 && Cp >= Ci && ldo >= (int64_t)taps * Cp, "ga_convw_pack: bad args");
    const long n = (long)Co * ldo;
    const int grid = (int)std::min<long>(2048, (n + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(convw_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, taps, Cp, (long)ldo);
    else hipLaunchKernelGGL(convw_pack_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, Co, Ci, taps, Cp, (long)ldo);
    return ga_check_launch("ga_convw_pack");
}

extern "C" int ga_convw_unpack_grad(const float* G, float* dW, int Co, int Ci, int taps, int Cp, int64_t ldg,
                                    ga_stream_t stream) {
    GA_REQUIRE(G && dW && Co > 0 && Ci > 0 && taps > 0 && Cp >= Ci && ldg >= (int64_t)taps * Cp, "ga_convw_unpack_grad: bad args");
    const long n = (long)Co * Ci * taps;
    const int grid = (int)std::min<long>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(convw_unpack_grad_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), G, dW, Co, Ci,
                       taps, Cp, (long)ldg);
    return ga_check_launch("ga_convw_unpack_grad");
}

extern "C" int ga_conv3s2_dgrad_prep(const float* w, void* out, int Co, int Ci, int64_t ldo, int dtype, ga_stream_t stream) {
    GA_REQUIRE(w && out && Co > 0 && Ci > 0 && ldo >= 4L * Co, "ga_conv3s2_dgrad_prep: bad args");
    const long n = 4L * Ci * ldo;
    const int grid = (int)std::min<long>(2048, (n + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) hipLaunchKernelGGL(conv3s2_dgrad_prep_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, (long)ldo);
    else hipLaunchKernelGGL(conv3s2_dgrad_prep_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, Co, Ci, (long)ldo);
    return ga_check_launch("ga_conv3s2_dgrad_prep");
}

#define GA_LNG_DISPATCH(KERN, ...)                                                                     \
    switch (G) {                                                                                       \
        case 1: hipLaunchKernelGGL((KERN<T, 1>), __VA_ARGS__); break;                                  \
        case 2: hipLaunchKernelGGL((KERN<T, 2>), __VA_ARGS__); break;                                  \
        case 4: hipLaunchKernelGGL((KERN<T, 4>), __VA_ARGS__); break;                                  \
        case 8: hipLaunchKernelGGL((KERN<T, 8>), __VA_ARGS__); break;                                  \
        case 16: hipLaunchKernelGGL((KERN<T, 16>), __VA_ARGS__); break;                                \
        case 32: hipLaunchKernelGGL((KERN<T, 32>), __VA_ARGS__); break;                                \
        default: hipLaunchKernelGGL((KERN<T, 64>), __VA_ARGS__); break;                                \
    }

template <typename T>
static void ln_gelu_fwd_t(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t rows, int C,
                          float eps, hipStream_t s) {
    const int G = C / 8;
    const int grid = (int)std::max<long>(1, std::min<long>(4096, (rows + 256 / G - 1) / (256 / G)));
    GA_LNG_DISPATCH(ln_gelu_fwd_kernel, dim3(grid), dim3(256), 0, s, (const T*)x, w, b, (T*)y, mean, rstd, (long)rows, eps);
}

template <typename T>
static void ln_gelu_bwd_t(const void* g, const void* x, const float* mean, const float* rstd, const float* w, const float* b,
                          void* dx, float* dw, float* db, int64_t rows, int C, hipStream_t s) {
    const int G = C / 8;
    const int grid = (int)std::max<long>(1, std::min<long>(1024, (rows + 256 / G - 1) / (256 / G)));
    GA_LNG_DISPATCH(ln_gelu_bwd_kernel, dim3(grid), dim3(256), 0, s, (const T*)g, (const T*)x, mean, rstd, w, b, (T*)dx, dw, db,
                    (long)rows);
}

extern "C" int ga_layernorm_gelu_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                     int64_t rows, int C, float eps, int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && w && b && y && mean && rstd && rows > 0, "ga_layernorm_gelu_fwd: null operand");
    GA_REQUIRE(C >= 8 && C <= 512 && (C & (C - 1)) == 0, "ga_layernorm_gelu_fwd: C=%d must be a power of two in [8, 512]", C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) ln_gelu_fwd_t<bf16_t>(x, w, b, y, mean, rstd, rows, C, eps, s);
    else ln_gelu_fwd_t<float>(x, w, b, y, mean, rstd, rows, C, eps, s);
    return ga_check_launch("ga_layernorm_gelu_fwd");
}

extern "C" int ga_layernorm_gelu_bwd(const void* g, const void* x, const float* mean, const float* rstd, const float* w,
                                     const float* b, void* dx, float* dw, float* db, int64_t rows, int C, int dtype,
                                     ga_stream_t stream) {
    GA_REQUIRE(g && x && mean && rstd && w && b && dx && dw && db && rows > 0, "ga_layernorm_gelu_bwd: null operand");
    GA_REQUIRE(C >= 8 && C <= 512 && (C & (C - 1)) == 0, "ga_layernorm_gelu_bwd: C=%d must be a power of two in [8, 512]", C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == GA_BF16) ln_gelu_bwd_t<bf16_t>(g, x, mean, rstd, w, b, dx, dw, db, rows, C, s);
    else ln_gelu_bwd_t<float>(g, x, mean, rstd, w, b, dx, dw, db, rows, C, s);
    return ga_check_launch("ga_layernorm_gelu_bwd");
}
