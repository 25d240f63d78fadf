// Direct 3 x 3 convolution (pad 1, stride 1) for 64 -> 64 channels, NHWC bf16, on the matrix cores -- the form of ga_gemm's
// GA_A_CONV3 product that GA-CSWin's deep stem needs (ga_cswin.py:463-477: Conv2d(64, 64, 3, 1, 1) on the 112 x 112 map of every
// image: 237 GFLOP per pass at batch 256, forward and -- with the flipped / transposed weight image ga_weight_prep already writes --
// backward-data).  The implicit-GEMM gather fetches every input pixel nine times from L2 and ran these launches at 215-260 TFLOP/s
// (1.1 / 0.92 ms); their HBM time is 0.2 ms.
//
//   workgroup (4 waves, persistent over tiles) = 8 x 16 output pixels x 64 channels; the 10 x 18 input tile with its halo lives in
//   LDS ([pixel][128 B], the 16-byte chunk c of pixel P at slot c ^ ((P >> 1) & 7): the 16 pixels of a fragment read conflict-
//   free), double-buffered and filled by `buffer_load_dwordx4 ... lds` one tile ahead (pixels outside the image carry an
//   out-of-range offset and land as zeros); the whole weight matrix [64][576] stays in LDS (72 KiB, same swizzle by output
//   channel).  wave = 2 image rows: per k step (one tap, 32 input channels) 4 weight fragments x 2 pixel fragments -> 8
//   v_mfma_f32_16x16x32_bf16 with the weights as the A operand, so that a lane ends with 4 consecutive output channels of one
//   pixel; results go through a per-wave staging piece to 16-byte NHWC stores.
//   Bound: LDS reads (6 ds_read_b128 per 8 MFMAs, one workgroup of 133 KiB per CU).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned c3_u4;

constexpr int TH = 8, TW = 16, HH = TH + 2, HW_ = TW + 2, NPX = HH * HW_;      // 180 halo pixels
constexpr int W_BYTES = 64 * 1152;                                             // weights: 64 rows x 576 bf16
constexpr int DMA_PER_WAVE = 6;                                                // 24 x 1 KiB >= 22.5 KiB: the last pieces are out of range
constexpr int IN_BUF = DMA_PER_WAVE * 4 * 1024;                                // 24576 B per buffer (so every DMA piece has its own KiB)
constexpr int ST_BYTES = 4096;                                                 // per-wave staging: 32 pixels x 128 B
constexpr int LDS_TOTAL = W_BYTES + 2 * IN_BUF + 4 * ST_BYTES;                 // 73728 + 49152 + 16384 = 139264

__device__ __forceinline__ c3_u4 c3_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    c3_u4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}

// six 1-KiB pieces of this wave: lane l of piece i writes LDS [dst + 1024 i + 16 l]; v[i] = source offset - 1024 i
__device__ __forceinline__ void c3_dma6(c3_u4 rs, unsigned dst, const unsigned* v) {
    unsigned keep;
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[c0], %[rs], 0 offen lds\n\t"
                 "buffer_load_dwordx4 %[c1], %[rs], 0 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %[c2], %[rs], 0 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %[c3], %[rs], 0 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %[d2]\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[c4], %[rs], 0 offen lds\n\t"
                 "buffer_load_dwordx4 %[c5], %[rs], 0 offen offset:1024 lds\n\t"
                 "s_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep)
                 : [c0] "v"(v[0]), [c1] "v"(v[1]), [c2] "v"(v[2]), [c3] "v"(v[3]), [c4] "v"(v[4]), [c5] "v"(v[5]), [rs] "s"(rs), [d] "s"(dst),
                   [d2] "s"(dst + 4096u)
                 : "memory");
}
template <int N> __device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, 1) void conv3_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wmat, long ldb,
                                                           bf16_t* __restrict__ y, int nimg, int H, int W, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* inb = smem + W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stg = smem + W_BYTES + 2 * IN_BUF + wv * ST_BYTES;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- weights -> LDS: row n (output channel) of 1152 B, its 16-byte chunk q at (q & ~7) + ((q & 7) ^ ((n >> 1) & 7))
    for (int u = tid; u < 64 * 72; u += 256) {
        const int n = u / 72, q = u - n * 72;
        const c3_u4 v = *reinterpret_cast<const c3_u4*>(wmat + (long)n * ldb + q * 8);
        *reinterpret_cast<c3_u4*>(wl + n * 1152 + (q & ~7) * 16 + (((q & 7) ^ ((n >> 1) & 7)) << 4)) = v;
    }

    const int tiles_x = W / TW, tiles_y = H / TH;
    const int tpi = tiles_x * tiles_y;
    const long ntiles = (long)nimg * tpi;
    const c3_u4 rx = c3_rsrc(x, bytes);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, bytes, 0x00020000);

    // DMA piece i of this wave: global piece index d = 4 i' ... pieces are dealt round-robin: piece (wave, i) covers LDS KiB
    // number wv * 6 + i of the buffer, i.e. chunks u = (wv * 6 + i) * 64 + lane -> halo pixel P = u / 8, physical slot u % 8
    int hp_y[DMA_PER_WAVE], hp_x[DMA_PER_WAVE], hp_c[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int u = (wv * DMA_PER_WAVE + i) * 64 + lane;
        const int P = u >> 3, slot = u & 7;
        hp_y[i] = P < NPX ? P / HW_ : -1000000;
        hp_x[i] = P % HW_;
        hp_c[i] = slot ^ ((P >> 1) & 7);           // the logical chunk that lives in this physical slot
    }
    auto issue = [&](long t, int buf) {
        unsigned v[DMA_PER_WAVE];
        const bool live = t < ntiles;
        const int img = live ? (int)(t / tpi) : 0;
        const int rem = live ? (int)(t - (long)img * tpi) : 0;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int yy = ty * TH + hp_y[i] - 1, xx = tx * TW + hp_x[i] - 1;
            const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = (unsigned)((((long)img * H + yy) * W + xx) * 128 + hp_c[i] * 16);
            v[i] = (ok ? off : 0x80000000u) - 1024u * (i & 3);
        }
        c3_dma6(rx, lds0 + W_BYTES + buf * IN_BUF + wv * (DMA_PER_WAVE * 1024), v);
    };

    // fragment addresses that do not depend on the tile
    const int fr = lane & 15, fk = lane >> 4;
    unsigned woff[4];                               // weight fragment j (output channels 16 j .. 16 j + 15) at k step 0
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = (16 * j + fr) * 1152;
    const int wsw = (fr >> 1) & 7;                  // (n >> 1) & 7 with n = 16 j + fr: j drops out (16 j is a multiple of 16)

    long t = blockIdx.x;
    issue(t, 0);
    c3_wait_vm<0>();                                // the first tile's pieces (no stores behind them yet: the counted wait below would be short)
    __syncthreads();                                // weights in LDS (their global loads were waited for by the compiler)
    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        c3_wait_vm<4>();                            // this wave's six pieces of tile t (the four stores of the previous tile may be pending)
        __builtin_amdgcn_s_barrier();               // every wave's pieces have landed; everyone is done with the other buffer
        issue(t + gridDim.x, buf ^ 1);
        const unsigned char* in = inb + buf * IN_BUF;
        f32x4_t acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky, half = s & 1;
            bf16x8_t wf[4], pf[2];
            const int q = s * 4 + fk;               // 16-byte chunk of the weight row
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const bf16x8_t*>(wl + woff[j] + (q & ~7) * 16 + (((q & 7) ^ wsw) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = (2 * wv + i + ky) * HW_ + fr + kx;
                pf[i] = *reinterpret_cast<const bf16x8_t*>(in + P * 128 + (((half * 4 + fk) ^ ((P >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], pf[i], acc[i][j], 0, 0, 0);
        }
        // ---- epilogue: lane holds channels 16 j + 4 fk .. + 3 of pixel (row 2 wv + i, column fr): 8-byte pieces -> staging
        // [32 pixels][128 B] (chunk c of pixel p at slot c ^ (p & 7)) -> 16-byte stores
        const int img = (int)(t / tpi);
        const int rem = (int)(t - (long)img * tpi);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        unsigned pk[2][4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pk[i][j][0] = pack2bf(acc[i][j][0], acc[i][j][1]);
                pk[i][j][1] = pack2bf(acc[i][j][2], acc[i][j][3]);
                const int p = i * 16 + fr, c = 2 * j + (fk >> 1);
                *reinterpret_cast<uint2*>(stg + p * 128 + ((c ^ (p & 7)) << 4) + (fk & 1) * 8) = make_uint2(pk[i][j][0], pk[i][j][1]);
            }
        asm volatile("" ::: "memory");
        c3_u4 ov[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) ov[it] = *reinterpret_cast<const c3_u4*>(stg + (it * 64 + lane) * 16);
        // (the staged registers stay live until the read-back has returned: a ds_write may fetch its data late behind LDS-DMA)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(pk[i][j][0]), "+v"(pk[i][j][1]), "+v"(ov[0]), "+v"(ov[3])::"memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int p = it * 8 + (lane >> 3), c = (lane & 7) ^ (p & 7);
            const int yy = ty * TH + 2 * wv + (p >> 4), xx = tx * TW + (p & 15);
            const unsigned off = (unsigned)((((long)img * H + yy) * W + xx) * 128 + c * 16);
            __builtin_amdgcn_raw_buffer_store_b128(ov[it], ry, off, 0, 0);
            asm volatile("s_nop 1" ::"v"(ov[it]) : "memory");
        }
    }
    c3_wait_vm<0>();                                // no DMA may still be landing when the LDS is handed on
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution: dW[co][tap * 64 + ci] += sum over pixels of dY[p][co] * X[p + tap][ci].
//   Same tiles (8 x 16 pixels, persistent workgroups); the contraction index is the PIXEL, the memory-slow index of both
//   operands, so both fragments are `ds_read_b64_tr_b16` reads of the [pixel][128 B] images (the idiom of the TN GEMM).  The X
//   halo tile is padded to 24 pixels per row: a k step (32 pixels = 2 tile rows) then advances the pixel index by 48, which leaves
//   the swizzle term (P >> 1) & 7 alone, so the 18 fragment addresses of a lane (9 taps x 2 halves) are tile- and step-invariant
//   registers + immediates.  wave w owns input channels 16 w .. 16 w + 15: 9 taps x 4 output-channel tiles = 36 accumulator
//   tiles (144 registers) live through the whole walk; per k step 8 dY + 18 X fragment reads feed 36 MFMAs.
//   At the end every workgroup stores ONE partial [64][576] fp32; conv3_wgrad_reduce adds them into dW.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short c3_s4;
constexpr int XW = 24;                                       // padded halo row pitch (pixels)
constexpr int XBUF = 32 * 1024, DBUF = 16 * 1024;            // 10 x 24 x 128 = 30 KiB in 8 pieces per wave; 128 x 128 B
constexpr int WG_LDS = 2 * XBUF + 2 * DBUF;                  // 96 KiB

__device__ __forceinline__ void c3_dma4(c3_u4 rs, unsigned dst, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
    unsigned keep;
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[c0], %[rs], 0 offen lds\n\t"
                 "buffer_load_dwordx4 %[c1], %[rs], 0 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %[c2], %[rs], 0 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %[c3], %[rs], 0 offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep)
                 : [c0] "v"(v0), [c1] "v"(v1), [c2] "v"(v2), [c3] "v"(v3), [rs] "s"(rs), [d] "s"(dst)
                 : "memory");
}

__device__ __forceinline__ void c3_dma2(c3_u4 rs, unsigned dst, unsigned v0, unsigned v1) {
    unsigned keep;
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[c0], %[rs], 0 offen lds\n\t"
                 "buffer_load_dwordx4 %[c1], %[rs], 0 offen offset:1024 lds\n\t"
                 "s_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep)
                 : [c0] "v"(v0), [c1] "v"(v1), [rs] "s"(rs), [d] "s"(dst)
                 : "memory");
}

__global__ __launch_bounds__(256, 1) void conv3_c64_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                 float* __restrict__ part, int nimg, int H, int W, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int tpi = tiles_x * tiles_y;
    const long ntiles = (long)nimg * tpi;
    const c3_u4 rx = c3_rsrc(x, bytes), rd = c3_rsrc(dy, bytes);

    // DMA: X piece i (8 per wave) covers chunks u = (8 wv + i) * 64 + lane of the padded halo image: pixel P = u / 8 = hy * 24 + hx;
    // dY piece i (4 per wave): chunks u = (4 wv + i) * 64 + lane, pixel p = u / 8 of the 8 x 16 tile
    int xy[8], xx_[8], xc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int u = (wv * 8 + i) * 64 + lane;
        const int P = u >> 3, hy = P / XW, hx = P - hy * XW;
        xy[i] = (hy < HH && hx < HW_) ? hy : -1000000;
        xx_[i] = hx;
        xc[i] = (u & 7) ^ ((P >> 1) & 7);
    }
    int dpy[4], dpx[4], dc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = (wv * 4 + i) * 64 + lane;
        const int p = u >> 3;
        dpy[i] = p >> 4;
        dpx[i] = p & 15;
        dc[i] = (u & 7) ^ ((p >> 1) & 7);
    }
    auto issue = [&](long t, int buf) {
        const bool live = t < ntiles;
        const int img = live ? (int)(t / tpi) : 0;
        const int rem = live ? (int)(t - (long)img * tpi) : 0;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        unsigned v[8], q[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int yy = ty * TH + xy[i] - 1, xx = tx * TW + xx_[i] - 1;
            const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = (unsigned)((((long)img * H + yy) * W + xx) * 128 + xc[i] * 16);
            v[i] = (ok ? off : 0x80000000u) - 1024u * (i & 3);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = (unsigned)((((long)img * H + ty * TH + dpy[i]) * W + tx * TW + dpx[i]) * 128 + dc[i] * 16);
            q[i] = (live ? off : 0x80000000u) - 1024u * i;
        }
        const unsigned xb = lds0 + buf * XBUF + wv * 8192;
        c3_dma4(rx, xb, v[0], v[1], v[2], v[3]);
        c3_dma4(rx, xb + 4096, v[4], v[5], v[6], v[7]);
        c3_dma4(rd, lds0 + 2 * XBUF + buf * DBUF + wv * 4096, q[0], q[1], q[2], q[3]);
    };

    // fragment addresses (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns, 8 bytes per lane, and hands lane i column i)
    const int g = lane >> 4, rlo = (lane & 15) >> 2, csub = (lane & 3) >> 1, bsub = 8 * (lane & 1);
    unsigned da[4][2];                              // dY fragment of output-channel tile jt, half hf, at k step 0 (+ 4096 per k step)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int p = 8 * g + rlo + 4 * hf;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) da[jt][hf] = p * 128 + (((2 * jt + csub) ^ ((p >> 1) & 7)) << 4) + bsub;
    }
    unsigned xa[9][2];                              // X fragment (this wave's 16 input channels) of tap, half, at k step 0 (+ 6144 per k step)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int kk = 8 * g + rlo + 4 * hf;
            const int P = ((kk >> 4) + tap / 3) * XW + (kk & 15) + tap % 3;
            xa[tap][hf] = P * 128 + (((2 * wv + csub) ^ ((P >> 1) & 7)) << 4) + bsub;
        }

    f32x4_t acc[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[tap][jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    long t = blockIdx.x;
    issue(t, 0);
    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        c3_wait_vm<0>();                            // this wave's pieces of tile t (nothing else is in flight: no stores in the loop)
        __builtin_amdgcn_s_barrier();               // everyone's pieces have landed; everyone is done with the other buffers
        issue(t + gridDim.x, buf ^ 1);
        const unsigned char* xi = smem + buf * XBUF;
        const unsigned char* di = smem + 2 * XBUF + buf * DBUF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            c3_s4 df[4][2], xf[9][2];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    df[jt][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3_s4*)(di + da[jt][hf] + ks * 4096));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    xf[tap][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3_s4*)(xi + xa[tap][hf] + ks * (2 * XW * 128)));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(&xf[tap][0]);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[tap][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&df[jt][0]), b, acc[tap][jt], 0, 0, 0);
            }
        }
    }
    c3_wait_vm<0>();
    // partial of this workgroup: lane holds D[co = 16 jt + 4 (lane >> 4) + r][ci = 16 wv + (lane & 15)] of every tap
    float* pp = part + (long)blockIdx.x * (64 * 576);
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[(16 * jt + 4 * fk + r) * 576 + tap * 64 + 16 * wv + fr] = acc[tap][jt][r];
}

// ------------------------------------------------------------------------------------------------
// The same weight gradient for the 3 x 3 / STRIDE-2 convolution of the deep stem (ga_cswin.py:470: Conv2d(64, 64, 3, 2, 1), 112 -> 56):
// dW[co][tap * 64 + ci] += sum over output pixels of dY[p][co] * X[2 p + tap - 1][ci].  Tiles of 8 x 8 output pixels; the X halo
// tile is 17 x 17 input pixels at a pitch of 20 (a k step = 32 output pixels = 4 tile rows advances the input pixel index by
// 8 x 20 = 160: the swizzle term (P >> 1) & 7 stays, and consecutive output pixels sit two input pixels apart, so the four rows of
// a transposing read land in four different swizzle classes).  Everything else -- roles of the waves, accumulators, partials --
// as above.  The gather form of gemm_tn ran this launch at 137 TFLOP/s (0.43 ms on the tail of the backward).
// ------------------------------------------------------------------------------------------------
constexpr int T2 = 8, H2 = 2 * T2 + 1, XW2 = 20;             // output tile edge, halo edge, halo row pitch (pixels)
constexpr int XBUF2 = 48 * 1024, DBUF2 = 8 * 1024;           // 17 x 20 x 128 B = 42.5 KiB in 12 pieces per wave; 64 x 128 B
constexpr int WG_LDS2 = 2 * XBUF2 + 2 * DBUF2;               // 112 KiB

__global__ __launch_bounds__(256, 1) void conv3s2_c64_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                   float* __restrict__ part, int nimg, int H, int W, unsigned xbytes,
                                                                   unsigned ybytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int OH = H >> 1, OW = W >> 1;
    const int tiles_x = OW / T2, tiles_y = OH / T2;
    const int tpi = tiles_x * tiles_y;
    const long ntiles = (long)nimg * tpi;
    const c3_u4 rx = c3_rsrc(x, xbytes), rd = c3_rsrc(dy, ybytes);

    // DMA: X piece i (12 per wave) covers chunks u = (12 wv + i) * 64 + lane of the padded halo image: pixel P = u / 8 = hy * 20 + hx;
    // dY piece i (2 per wave): chunks u = (2 wv + i) * 64 + lane, pixel p = u / 8 = row * 8 + column of the 8 x 8 tile
    int xy[12], xx_[12], xc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int u = (wv * 12 + i) * 64 + lane;
        const int P = u >> 3, hy = P / XW2, hx = P - hy * XW2;
        xy[i] = (hy < H2 && hx < H2) ? hy : -1000000;
        xx_[i] = hx;
        xc[i] = (u & 7) ^ ((P >> 1) & 7);
    }
    int dpy[2], dpx[2], dc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = (wv * 2 + i) * 64 + lane;
        const int p = u >> 3;
        dpy[i] = p >> 3;
        dpx[i] = p & 7;
        dc[i] = (u & 7) ^ ((p >> 1) & 7);
    }
    auto issue = [&](long t, int buf) __attribute__((always_inline)) {
        const bool live = t < ntiles;
        const int img = live ? (int)(t / tpi) : 0;
        const int rem = live ? (int)(t - (long)img * tpi) : 0;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        unsigned v[12], q[2];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int yy = 2 * ty * T2 + xy[i] - 1, xx = 2 * tx * T2 + xx_[i] - 1;
            const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = (unsigned)((((long)img * H + yy) * W + xx) * 128 + xc[i] * 16);
            v[i] = (ok ? off : 0x80000000u) - 1024u * (i & 3);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = (unsigned)((((long)img * OH + ty * T2 + dpy[i]) * OW + tx * T2 + dpx[i]) * 128 + dc[i] * 16);
            q[i] = (live ? off : 0x80000000u) - 1024u * i;
        }
        const unsigned xb = lds0 + buf * XBUF2 + wv * 12288;
        c3_dma4(rx, xb, v[0], v[1], v[2], v[3]);
        c3_dma4(rx, xb + 4096, v[4], v[5], v[6], v[7]);
        c3_dma4(rx, xb + 8192, v[8], v[9], v[10], v[11]);
        c3_dma2(rd, lds0 + 2 * XBUF2 + buf * DBUF2 + wv * 2048, q[0], q[1]);
    };

    // fragment addresses (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns, 8 bytes per lane, and hands lane i column i)
    const int g = lane >> 4, rlo = (lane & 15) >> 2, csub = (lane & 3) >> 1, bsub = 8 * (lane & 1);
    unsigned da[4][2];                              // dY fragment of output-channel tile jt, half hf, at k step 0 (+ 4096 per k step)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int p = 8 * g + rlo + 4 * hf;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) da[jt][hf] = p * 128 + (((2 * jt + csub) ^ ((p >> 1) & 7)) << 4) + bsub;
    }
    unsigned xa[9][2];                              // X fragment (this wave's 16 input channels) of tap, half, at k step 0 (+ 160 x 128 per k step)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int kk = 8 * g + rlo + 4 * hf;    // output pixel of the k step: row kk >> 3, column kk & 7
            const int P = (2 * (kk >> 3) + tap / 3) * XW2 + 2 * (kk & 7) + tap % 3;
            xa[tap][hf] = P * 128 + (((2 * wv + csub) ^ ((P >> 1) & 7)) << 4) + bsub;
        }

    f32x4_t acc[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[tap][jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    long t = blockIdx.x;
    issue(t, 0);
    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        c3_wait_vm<0>();                            // this wave's pieces of tile t (nothing else is in flight: no stores in the loop)
        __builtin_amdgcn_s_barrier();               // everyone's pieces have landed; everyone is done with the other buffers
        issue(t + gridDim.x, buf ^ 1);
        const unsigned char* xi = smem + buf * XBUF2;
        const unsigned char* di = smem + 2 * XBUF2 + buf * DBUF2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            c3_s4 df[4][2], xf[9][2];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    df[jt][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3_s4*)(di + da[jt][hf] + ks * 4096));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    xf[tap][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3_s4*)(xi + xa[tap][hf] + ks * (8 * XW2 * 128)));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(&xf[tap][0]);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[tap][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(&df[jt][0]), b, acc[tap][jt], 0, 0, 0);
            }
        }
    }
    c3_wait_vm<0>();
    float* pp = part + (long)blockIdx.x * (64 * 576);
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[(16 * jt + 4 * fk + r) * 576 + tap * 64 + 16 * wv + fr] = acc[tap][jt][r];
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the FIRST convolution of the deep stem (ga_cswin.py:464: Conv2d(3, 64, 3, 2, 1)) on the NHWC8 image copy:
// dW[co][tap * 8 + ch] += sum over output pixels of dY[p][co] * X8[2 p + tap - 1][ch] -- 3.2 M pixels, a 64 x 72 result, 411 + 205 MB
// of operands: HBM-bound (0.13 ms); the gather form of gemm_tn took 0.35-0.37 ms on the very tail of the backward.
//   Tiles of 8 x 16 output pixels (dY [128 pixels][128 B], swizzled as above; X8 halo 17 x 33 input pixels of 16 bytes at a pitch of
//   36), double-buffered by LDS-DMA, 52 KiB per workgroup -> three workgroups per CU.  wave w owns output channels 16 w .. + 15; the
//   B operand of (tap row ky, column pair cp) is a transposing read of 4 output pixels x [2 adjacent input pixels x 8 channels]
//   (32 contiguous bytes): columns 0-7 = tap kx = 2 cp, columns 8-15 = tap 2 cp + 1 (cp = 1: kx = 3 does not exist, those columns
//   are dropped) -> 6 MFMAs per k step (32 output pixels) and wave.  One fp32 partial [64][72] per workgroup + reduce.
// ------------------------------------------------------------------------------------------------
constexpr int XW0 = 36;                                      // halo row pitch (pixels)
constexpr int XBUF0 = 16 * 1024, DBUF0 = 16 * 1024;          // 17 x 36 x 16 B = 9.6 KiB in 4 pieces per wave (the rest out of range); 128 x 128 B
constexpr int WG_LDS0 = 2 * XBUF0 + 2 * DBUF0;               // 64 KiB

__global__ __launch_bounds__(256, 2) void conv0_c8_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                float* __restrict__ part, int nimg, int H, int W, unsigned xbytes,
                                                                unsigned ybytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int OH = H >> 1, OW = W >> 1;
    const int tiles_x = OW / TW, tiles_y = OH / TH;
    const int tpi = tiles_x * tiles_y;
    const long ntiles = (long)nimg * tpi;
    const c3_u4 rx = c3_rsrc(x, xbytes), rd = c3_rsrc(dy, ybytes);
    // DMA: X piece i (4 per wave): chunk u = (4 wv + i) * 64 + lane = halo pixel P = hy * 36 + hx (one 16-byte pixel per chunk);
    // dY piece i (4 per wave): chunks u = (4 wv + i) * 64 + lane, pixel p = u / 8 of the 8 x 16 tile
    int xy[4], xx_[4], dpy[4], dpx[4], dc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = (wv * 4 + i) * 64 + lane, hy = P / XW0, hx = P - hy * XW0;
        xy[i] = (hy < 2 * TH + 1 && hx < 2 * TW + 1) ? hy : -1000000;
        xx_[i] = hx;
        const int u = (wv * 4 + i) * 64 + lane, p = u >> 3;
        dpy[i] = p >> 4;
        dpx[i] = p & 15;
        dc[i] = (u & 7) ^ ((p >> 1) & 7);
    }
    auto issue = [&](long t, int buf) __attribute__((always_inline)) {
        const bool live = t < ntiles;
        const int img = live ? (int)(t / tpi) : 0;
        const int rem = live ? (int)(t - (long)img * tpi) : 0;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        unsigned v[4], q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = 2 * ty * TH + xy[i] - 1, xx = 2 * tx * TW + xx_[i] - 1;
            const bool ok = live && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            v[i] = (ok ? (unsigned)((((long)img * H + yy) * W + xx) * 16) : 0x80000000u) - 1024u * i;
            const unsigned off = (unsigned)((((long)img * OH + ty * TH + dpy[i]) * OW + tx * TW + dpx[i]) * 128 + dc[i] * 16);
            q[i] = (live ? off : 0x80000000u) - 1024u * i;
        }
        c3_dma4(rx, lds0 + buf * XBUF0 + wv * 4096, v[0], v[1], v[2], v[3]);
        c3_dma4(rd, lds0 + 2 * XBUF0 + buf * DBUF0 + wv * 4096, q[0], q[1], q[2], q[3]);
    };
    const int g = lane >> 4, rlo = (lane & 15) >> 2, csub = (lane & 3) >> 1, bsub = 8 * (lane & 1);
    unsigned da[2], xa[6][2];                       // dY fragment of this wave's channel tile; X fragment of (ky, cp); per half, at k step 0
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int p = 8 * g + rlo + 4 * hf;         // output pixel of the k step: row p >> 4, column p & 15
        da[hf] = p * 128 + (((2 * wv + csub) ^ ((p >> 1) & 7)) << 4) + bsub;
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            const int ky = nt >> 1, cp = nt & 1;
            const int P = (2 * (p >> 4) + ky) * XW0 + 2 * (p & 15) + 2 * cp;
            xa[nt][hf] = P * 16 + (lane & 3) * 8;
        }
    }
    f32x4_t acc[6];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    long t = blockIdx.x;
    issue(t, 0);
    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        c3_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        issue(t + gridDim.x, buf ^ 1);
        const unsigned char* xi = smem + buf * XBUF0;
        const unsigned char* di = smem + 2 * XBUF0 + buf * DBUF0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {            // 32 output pixels = 2 tile rows = 4 input rows
            c3_s4 df[2], xf[6][2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                df[hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3_s4*)(di + da[hf] + ks * 4096));
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
                    xf[nt][hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3_s4*)(xi + xa[nt][hf] + ks * (4 * XW0 * 16)));
            }
            const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(&df[0]);
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, *reinterpret_cast<const bf16x8_t*>(&xf[nt][0]), acc[nt], 0, 0, 0);
        }
    }
    c3_wait_vm<0>();
    // lane holds D[co = 16 wv + 4 (lane >> 4) + r][column lane & 15 of (ky, cp)]: column = (pixel e = c >> 3, channel c & 7), tap kx = 2 cp + e
    float* pp = part + (long)blockIdx.x * (64 * 72);
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int ky = nt >> 1, kx = 2 * (nt & 1) + (fr >> 3);
        if (kx < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[(16 * wv + 4 * fk + r) * 72 + (3 * ky + kx) * 8 + (fr & 7)] = acc[nt][r];
        }
    }
}

// dW[n][k] += alpha * sum_s part[s][n][k] for the [64][72] partials of the kernel above
__global__ __launch_bounds__(256) void conv0_wgrad_reduce(const float* __restrict__ part, int nparts, float alpha, float* __restrict__ dW,
                                                          long ldw) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // over 64 * 72
    if (i >= 64 * 72) return;
    const int s0 = blockIdx.y * 16, s1 = min(nparts, s0 + 16);
    float a = 0.f;
#pragma unroll 8
    for (int s = s0; s < s1; ++s) a += part[(long)s * (64 * 72) + i];
    const int n = i / 72, k = i - n * 72;
    atomicAdd(dW + (long)n * ldw + k, alpha * a);
}

// dW[n][k] += alpha * sum_s part[s][n][k]: blockIdx.y sums 16 partials and adds its share with one atomic per element
__global__ __launch_bounds__(256) void conv3_wgrad_reduce(const float* __restrict__ part, int nparts, float alpha, float* __restrict__ dW,
                                                          long ldw) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // over 64 * 576
    if (i >= 64 * 576) return;
    const int s0 = blockIdx.y * 16, s1 = min(nparts, s0 + 16);
    float a = 0.f;
#pragma unroll 8
    for (int s = s0; s < s1; ++s) a += part[(long)s * (64 * 576) + i];
    const int n = i / 576, k = i - n * 576;
    atomicAdd(dW + (long)n * ldw + k, alpha * a);
}

// ------------------------------------------------------------------------------------------------
// First convolution of GA-CSWin's deep stem (ga_cswin.py:464: Conv2d(3, 64, 3, 2, 1), no bias) on the NHWC8 copy of the image
// (3 channels + 5 zeros = 16 bytes per pixel): 3.2 M output pixels x 64 channels from K = 27 -- HBM-bound (205 MB in, 411 MB out
// at batch 256), the gather GEMM ran it at 60 TFLOP/s (0.49 ms; its byte floor is 0.13 ms).
//   No LDS, no barrier: a k step is one tap ROW (3 taps x 8 channels + 8 zeros = 32), so the B fragment of output pixel
//   (oy, ox0 + (lane & 15)) is ONE 16-byte pixel per lane -- (2 oy + ky - 1, 2 ox + (lane >> 4) - 1), lanes 48-63 and pixels
//   outside the image an out-of-range buffer offset = zeros -- and the 12 weight fragments (3 tap rows x 4 channel tiles) stay in
//   registers for the whole kernel.  D: lane = 4 consecutive channels 16 j + 4 (lane >> 4) + r of pixel lane & 15; lane pairs
//   (lane ^ 16) swap 8-byte halves so that every lane ends with two whole 16-byte chunks -> 16-byte NHWC stores.
//   wave = NG groups of 16 pixels of one output row per iteration (3 NG loads in flight).
// ------------------------------------------------------------------------------------------------
constexpr int C0_NG = 4;

__global__ __launch_bounds__(256) void conv0_c8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wmat, long ldb,
                                                       bf16_t* __restrict__ y, int nimg, int H, int W, unsigned in_bytes, unsigned out_bytes) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, g = lane >> 4;
    const int OH = H >> 1, OW = W >> 1;
    const int gpr = (OW + 15) / 16;                                 // 16-pixel groups per output row
    const long ngroups = (long)nimg * OH * gpr;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, out_bytes, 0x00020000);
    // weights: fragment (ky, j) = row 16 j + p of the [64][72] matrix, its 8 elements of tap (ky, kx = g); g == 3: zeros
    bf16x8_t wf[3][4];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c3_u4 v = {0u, 0u, 0u, 0u};
            if (g < 3) v = *reinterpret_cast<const c3_u4*>(wmat + (long)(16 * j + p) * ldb + (3 * ky + g) * 8);
            wf[ky][j] = *reinterpret_cast<const bf16x8_t*>(&v);
        }
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long q0 = wave0 * C0_NG; q0 < ngroups; q0 += nwaves * C0_NG) {
        c3_u4 xin[C0_NG][3];
        unsigned obase[C0_NG];
#pragma unroll
        for (int u = 0; u < C0_NG; ++u) {
            const long q = q0 + u;
            const bool live = q < ngroups;
            const long row = live ? q / gpr : 0;                    // (img, oy)
            const int gx = live ? (int)(q - row * gpr) : 0;
            const int oy = (int)(row % OH);
            const long img = row / OH;
            const int ox = gx * 16 + p;
            const int ix = 2 * ox + g - 1;
            const bool okx = live && g < 3 && ox < OW && (unsigned)ix < (unsigned)W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy + ky - 1;
                const bool ok = okx && (unsigned)iy < (unsigned)H;
                const unsigned off = ok ? (unsigned)(((img * H + iy) * W + ix) * 16) : 0x80000000u;
                xin[u][ky] = __builtin_bit_cast(c3_u4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            }
            obase[u] = (live && ox < OW) ? (unsigned)(((img * OH + oy) * OW + ox) * 128) : 0x80000000u;
        }
#pragma unroll
        for (int u = 0; u < C0_NG; ++u) {
            f32x4_t acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(&xin[u][ky]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][j], b, acc[j], 0, 0, 0);
            }
            // lane pair (g even, g odd): the even lane assembles the chunk of tile j = 2 t, the odd lane that of tile 2 t + 1
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned e0 = pack2bf(acc[2 * t][0], acc[2 * t][1]), e1 = pack2bf(acc[2 * t][2], acc[2 * t][3]);
                const unsigned o0 = pack2bf(acc[2 * t + 1][0], acc[2 * t + 1][1]), o1 = pack2bf(acc[2 * t + 1][2], acc[2 * t + 1][3]);
                const bool odd = g & 1;
                // what I send: the half the partner assembles (even lane sends its tile-(2t+1) half, odd lane its tile-2t half)
                const unsigned s0 = odd ? e0 : o0, s1 = odd ? e1 : o1;
                const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
                c3_u4 ov;
                if (!odd) { ov[0] = e0; ov[1] = e1; ov[2] = r0; ov[3] = r1; }      // channels 16 (2t) + 8 (g >> 1) .. + 7
                else { ov[0] = r0; ov[1] = r1; ov[2] = o0; ov[3] = o1; }            // channels 16 (2t + 1) + 8 (g >> 1) .. + 7
                const unsigned c = 2 * (2 * t + (odd ? 1 : 0)) + (g >> 1);
                __builtin_amdgcn_raw_buffer_store_b128(ov, ry, obase[u] + c * 16u, 0, 0);
                asm volatile("s_nop 1" ::"v"(ov) : "memory");
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ConvNeXt stem (ga_convnext.py:431-434 / timm ConvNeXt stem: Conv2d(3, C, 4, 4) + LayerNorm over the channels) in one pass from
// the fp32 NCHW image: 4 x 4 / stride-4 patches (K = 48 = (c, ky, kx)) x C = 16 NT output channels, + bias -> `pre` (bf16, kept for
// the LayerNorm backward) -> LayerNorm of the ROUNDED values (what the two-launch path normalises) -> y, mean, rstd.
//   HBM-bound: 154 MB in, 2 x 154 MB out at batch 256; the gather GEMM + LayerNorm launches took 0.18 + 0.08 ms (K = 48 at 40 TFLOP/s,
//   and the LayerNorm re-read `pre`).  Same structure as the kernel above: no LDS, a lane's pixel operand of a k step is two float4
//   (rows ky, ky + 1 of one channel, 4 kx each) packed to 8 bf16, the 2 NT weight fragments stay in registers; a pixel's C channels
//   end up on the four lanes l, l + 16, l + 32, l + 48 (4 consecutive channels per tile each), so the LayerNorm sums are two
//   `__shfl_xor`; lane pairs swap 8-byte halves for 16-byte stores.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void stem4_ln_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wmat, long ldw,
                                                       const float* __restrict__ bias, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16_t* __restrict__ pre, bf16_t* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int nimg, int H, int W,
                                                       float eps) {
    constexpr int C = 16 * NT;
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, g = lane >> 4;
    const int OH = H >> 2, OW = W >> 2;
    const long M = (long)nimg * OH * OW;
    const long ngroups = (M + 15) / 16;
    // weight fragment (s, j): row 16 j + p, elements 32 s + 8 g .. + 7 (zeros beyond k = 48)
    bf16x8_t wf[2][NT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            c3_u4 v = {0u, 0u, 0u, 0u};
            if (32 * s2 + 8 * g < 48) v = *reinterpret_cast<const c3_u4*>(wmat + (long)(16 * j + p) * ldw + 32 * s2 + 8 * g);
            wf[s2][j] = *reinterpret_cast<const bf16x8_t*>(&v);
        }
    float bi[NT][4], ga[NT][4], be[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * j + 4 * g + r;
            bi[j][r] = bias ? bias[c] : 0.f;
            ga[j][r] = gamma[c];
            be[j][r] = beta[c];
        }
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long q = wave0; q < ngroups; q += nwaves) {
        const long m = q * 16 + p;
        const bool live = m < M;
        const long mm = live ? m : M - 1;
        const int ox = (int)(mm % OW);
        const long t = mm / OW;
        const int oy = (int)(t % OH);
        const long b = t / OH;
        bf16x8_t xb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int k0 = 32 * s2 + 8 * g;
            c3_u4 v = {0u, 0u, 0u, 0u};
            if (k0 < 48) {
                const int c = k0 >> 4, ky = (k0 >> 2) & 3;
                const float* src = x + ((b * 3 + c) * H + 4 * oy + ky) * (long)W + 4 * ox;
                const float4 a = *reinterpret_cast<const float4*>(src);
                const float4 a2 = *reinterpret_cast<const float4*>(src + W);
                v[0] = pack2bf(a.x, a.y);
                v[1] = pack2bf(a.z, a.w);
                v[2] = pack2bf(a2.x, a2.y);
                v[3] = pack2bf(a2.z, a2.w);
            }
            xb[s2] = *reinterpret_cast<const bf16x8_t*>(&v);
        }
        f32x4_t acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], xb[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], xb[1], acc[j], 0, 0, 0);
        }
        // + bias, round to bf16 (`pre`), LayerNorm statistics of the rounded values over the pixel's C channels (4 lanes)
        unsigned pk[NT][2];
        float vr[NT][4];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            pk[j][0] = pack2bf(acc[j][0] + bi[j][0], acc[j][1] + bi[j][1]);
            pk[j][1] = pack2bf(acc[j][2] + bi[j][2], acc[j][3] + bi[j][3]);
            vr[j][0] = __uint_as_float(pk[j][0] << 16);
            vr[j][1] = __uint_as_float(pk[j][0] & 0xffff0000u);
            vr[j][2] = __uint_as_float(pk[j][1] << 16);
            vr[j][3] = __uint_as_float(pk[j][1] & 0xffff0000u);
            sum += (vr[j][0] + vr[j][1]) + (vr[j][2] + vr[j][3]);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mu = sum * (1.f / C);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dlt = vr[j][r] - mu;
                sq = fmaf(dlt, dlt, sq);
            }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rs = rsqrtf(sq * (1.f / C) + eps);
        unsigned yk[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaf((vr[j][r] - mu) * rs, ga[j][r], be[j][r]);
            yk[j][0] = pack2bf(o[0], o[1]);
            yk[j][1] = pack2bf(o[2], o[3]);
        }
        if (live && g == 0) {
            mean[m] = mu;
            rstd[m] = rs;
        }
        // lane pair (g even, g odd): the even lane assembles the 16-byte chunk of tile 2 t, the odd lane that of tile 2 t + 1
        const bool odd = g & 1;
#pragma unroll
        for (int t2 = 0; t2 < NT / 2; ++t2) {
            const unsigned c = 2 * (2 * t2 + (odd ? 1 : 0)) + (g >> 1);
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const unsigned e0 = which ? yk[2 * t2][0] : pk[2 * t2][0], e1 = which ? yk[2 * t2][1] : pk[2 * t2][1];
                const unsigned o0 = which ? yk[2 * t2 + 1][0] : pk[2 * t2 + 1][0], o1 = which ? yk[2 * t2 + 1][1] : pk[2 * t2 + 1][1];
                const unsigned s0 = odd ? e0 : o0, s1 = odd ? e1 : o1;
                const unsigned r0 = (unsigned)__shfl_xor((int)s0, 16), r1 = (unsigned)__shfl_xor((int)s1, 16);
                c3_u4 ov;
                if (!odd) { ov[0] = e0; ov[1] = e1; ov[2] = r0; ov[3] = r1; }
                else { ov[0] = r0; ov[1] = r1; ov[2] = o0; ov[3] = o1; }
                if (live) *reinterpret_cast<c3_u4*>((which ? y : pre) + m * C + c * 8) = ov;
            }
        }
    }
}

}  // namespace

extern "C" int ga_stem4_ln_fwd(const float* x, const void* W, int64_t ldw, const float* bias, const float* gamma, const float* beta,
                               void* pre, void* y, float* mean, float* rstd, int B, int H, int W_, int C, float eps,
                               ga_stream_t stream) {
    GA_REQUIRE(x && W && gamma && beta && pre && y && mean && rstd && B > 0 && H > 0 && W_ > 0, "ga_stem4_ln_fwd: null / empty argument");
    GA_REQUIRE(H % 4 == 0 && W_ % 4 == 0 && (C == 96 || C == 128) && ldw >= 48 && ldw % 8 == 0,
               "ga_stem4_ln_fwd: needs H, W multiples of 4, C = 96 or 128, ldw >= 48 and a multiple of 8 (H=%d W=%d C=%d ldw=%ld)", H, W_, C,
               (long)ldw);
    GA_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(pre) |
                 reinterpret_cast<uintptr_t>(y)) & 15) == 0, "ga_stem4_ln_fwd: operands must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int cus = 256;
    ga_device_info(&cus, nullptr, nullptr);
    const long ngroups = ((long)B * (H / 4) * (W_ / 4) + 15) / 16;
    const int grid = (int)std::max<long>(1, std::min<long>((ngroups + 3) / 4, (long)cus * 8));
    if (C == 96)
        hipLaunchKernelGGL(stem4_ln_kernel<6>, dim3(grid), dim3(256), 0, s, x, (const bf16_t*)W, (long)ldw, bias, gamma, beta, (bf16_t*)pre,
                           (bf16_t*)y, mean, rstd, B, H, W_, eps);
    else
        hipLaunchKernelGGL(stem4_ln_kernel<8>, dim3(grid), dim3(256), 0, s, x, (const bf16_t*)W, (long)ldw, bias, gamma, beta, (bf16_t*)pre,
                           (bf16_t*)y, mean, rstd, B, H, W_, eps);
    return ga_check_launch("ga_stem4_ln_fwd");
}

// ga_gemm's GA_A_CONV3S2 product for the 3 -> 64-channel first convolution on the NHWC8 image: returns 1 if it took the launch
int ga_conv0_c8_try(const ga_gemm_desc* d, hipStream_t s) {
    if (!GA_KNOB("CONV0_DIRECT", 1)) return 0;
    if (d->dtype != GA_BF16 || d->a_kind != GA_A_CONV3S2 || d->a_C != 8 || d->N != 64 || d->K != 72 || d->batch != 1) return 0;
    if (d->c_kind != GA_C_PLAIN || d->c_f32 || d->ldc != 64 || d->bias || d->R || d->H || d->C2 || d->rowscale || d->colsum || d->colsumsq ||
        d->act != GA_ACT_NONE || d->a_act != GA_ACT_NONE || d->relu_after || d->alpha != 1.0f)
        return 0;
    if (d->a_H % 2 != 0 || d->a_W % 2 != 0 || d->ldb % 8 != 0 || ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B) |
                                                                   reinterpret_cast<uintptr_t>(d->C)) & 15))
        return 0;
    const long ohw = (long)(d->a_H / 2) * (d->a_W / 2);
    if (d->M % ohw != 0) return 0;
    const long nimg = d->M / ohw;
    const long in_bytes = nimg * d->a_H * d->a_W * 16, out_bytes = d->M * 128;
    if (in_bytes >= (1L << 31) || out_bytes >= (1L << 31)) return 0;
    int cus = 256;
    ga_device_info(&cus, nullptr, nullptr);
    const long ngroups = nimg * (d->a_H / 2) * ((d->a_W / 2 + 15) / 16);
    const int grid = (int)std::max<long>(1, std::min<long>((ngroups + 4 * C0_NG - 1) / (4 * C0_NG), (long)cus * 8));
    hipLaunchKernelGGL(conv0_c8_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)d->A, (const bf16_t*)d->B, (long)d->ldb, (bf16_t*)d->C,
                       (int)nimg, d->a_H, d->a_W, (unsigned)in_bytes, (unsigned)out_bytes);
    return 1;
}

// ga_gemm's GA_A_CONV3 product, plain epilogue, bf16, 64 input channels, N = 64, maps of 8 x 16 tiles: returns 1 if it took the launch
int ga_conv3_c64_try(const ga_gemm_desc* d, hipStream_t s) {
    if (!GA_KNOB("CONV3_DIRECT", 1)) return 0;
    if (d->dtype != GA_BF16 || d->a_kind != GA_A_CONV3 || d->a_C != 64 || d->N != 64 || d->K != 576 || d->batch != 1) return 0;
    if (d->c_kind != GA_C_PLAIN || d->c_f32 || d->ldc != 64 || d->bias || d->R || d->H || d->C2 || d->rowscale || d->colsum || d->colsumsq ||
        d->act != GA_ACT_NONE || d->a_act != GA_ACT_NONE || d->relu_after || d->alpha != 1.0f)
        return 0;
    if (d->a_H % TH != 0 || d->a_W % TW != 0 || d->ldb % 8 != 0) return 0;
    const long hw = (long)d->a_H * d->a_W;
    if (d->M % hw != 0) return 0;
    const long nimg = d->M / hw;
    const long bytes = d->M * 128;
    if (bytes >= (1L << 31)) return 0;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) == hipSuccess;
    if (!attr_ok) return 0;
    int cus = 256;
    ga_device_info(&cus, nullptr, nullptr);
    const long ntiles = nimg * (d->a_H / TH) * (d->a_W / TW);
    const int grid = (int)std::min<long>(ntiles, cus);
    hipLaunchKernelGGL(conv3_c64_kernel, dim3(grid), dim3(256), LDS_TOTAL, s, (const bf16_t*)d->A, (const bf16_t*)d->B, (long)d->ldb,
                       (bf16_t*)d->C, (int)nimg, d->a_H, d->a_W, (unsigned)bytes);
    return 1;
}

// ga_wgrad's GA_A_CONV3 product for the same layer (bf16, 64 -> 64 channels, accumulate into dW, no bias): workgroups / bytes of
// partial sums it needs (0: does not apply)
static int conv3_wgrad_wgs(const ga_wgrad_desc* d) {
    if (!GA_KNOB("CONV3_DIRECT", 1)) return 0;
    if (d->dtype != GA_BF16 || d->x_kind != GA_A_CONV3 || d->x_C != 64 || d->N != 64 || d->K != 576 || d->batch != 1 || d->dbias ||
        d->x_act != GA_ACT_NONE || d->ldy != 64 || !d->accumulate)
        return 0;
    if (d->x_H % TH != 0 || d->x_W % TW != 0 || (long)d->M % ((long)d->x_H * d->x_W) != 0 || (long)d->M * 128 >= (1L << 31)) return 0;
    int cus = 256;
    ga_device_info(&cus, nullptr, nullptr);
    const long ntiles = (long)d->M / (TH * TW);
    return (int)std::min<long>(ntiles, cus);
}
// ... and for the stride-2 convolution (GA_A_CONV3S2: x_H x x_W is the INPUT map, M the output pixels)
static int conv3s2_wgrad_wgs(const ga_wgrad_desc* d) {
    if (!GA_KNOB("CONV3_DIRECT", 1)) return 0;
    if (d->dtype != GA_BF16 || d->x_kind != GA_A_CONV3S2 || d->x_C != 64 || d->N != 64 || d->K != 576 || d->batch != 1 || d->dbias ||
        d->x_act != GA_ACT_NONE || d->ldy != 64 || !d->accumulate)
        return 0;
    if (d->x_H % (2 * T2) != 0 || d->x_W % (2 * T2) != 0) return 0;
    const long ohw = (long)(d->x_H / 2) * (d->x_W / 2);
    if ((long)d->M % ohw != 0 || (long)d->M * 4 * 128 >= (1L << 31)) return 0;
    int cus = 256;
    ga_device_info(&cus, nullptr, nullptr);
    const long ntiles = (long)d->M / (T2 * T2);
    return (int)std::min<long>(ntiles, cus);
}
// ... and for the first convolution (GA_A_CONV3S2 on the 8-channel image copy)
static int conv0_wgrad_wgs(const ga_wgrad_desc* d) {
    if (!GA_KNOB("CONV0_DIRECT", 1)) return 0;
    if (d->dtype != GA_BF16 || d->x_kind != GA_A_CONV3S2 || d->x_C != 8 || d->N != 64 || d->K != 72 || d->batch != 1 || d->dbias ||
        d->x_act != GA_ACT_NONE || d->ldy != 64 || !d->accumulate)
        return 0;
    if (d->x_H % (2 * TH) != 0 || d->x_W % (2 * TW) != 0) return 0;
    const long ohw = (long)(d->x_H / 2) * (d->x_W / 2);
    if ((long)d->M % ohw != 0 || (long)d->M * 128 >= (1L << 31)) return 0;
    int cus = 256;
    ga_device_info(&cus, nullptr, nullptr);
    const long ntiles = (long)d->M / (TH * TW);
    return (int)std::min<long>(ntiles, 2L * cus);
}
size_t ga_conv3_c64_wgrad_workspace(const ga_wgrad_desc* d) {
    if (d->x_kind == GA_A_CONV3S2 && d->x_C == 8) return (size_t)conv0_wgrad_wgs(d) * 64 * 72 * sizeof(float);
    const int wgs = d->x_kind == GA_A_CONV3S2 ? conv3s2_wgrad_wgs(d) : conv3_wgrad_wgs(d);
    return (size_t)wgs * 64 * 576 * sizeof(float);
}

int ga_conv0_c8_wgrad_try(const ga_wgrad_desc* d, hipStream_t s) {
    const int wgs = conv0_wgrad_wgs(d);
    if (!wgs || !d->workspace || (size_t)d->ws_bytes < (size_t)wgs * 64 * 72 * sizeof(float)) return 0;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv0_c8_wgrad_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS0) == hipSuccess;
    if (!attr_ok) return 0;
    const long ohw = (long)(d->x_H / 2) * (d->x_W / 2);
    const long nimg = d->M / ohw;
    float* part = reinterpret_cast<float*>(d->workspace);
    hipLaunchKernelGGL(conv0_c8_wgrad_kernel, dim3(wgs), dim3(256), WG_LDS0, s, (const bf16_t*)d->Y, (const bf16_t*)d->X, part, (int)nimg,
                       d->x_H, d->x_W, (unsigned)(nimg * d->x_H * d->x_W * 16), (unsigned)((long)d->M * 128));
    hipLaunchKernelGGL(conv0_wgrad_reduce, dim3(cdiv(64 * 72, 256), cdiv(wgs, 16)), dim3(256), 0, s, part, wgs, d->alpha, d->dW, (long)d->ldw);
    return 1;
}

int ga_conv3s2_c64_wgrad_try(const ga_wgrad_desc* d, hipStream_t s) {
    const int wgs = conv3s2_wgrad_wgs(d);
    if (!wgs || !d->workspace || (size_t)d->ws_bytes < (size_t)wgs * 64 * 576 * sizeof(float)) return 0;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3s2_c64_wgrad_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS2) == hipSuccess;
    if (!attr_ok) return 0;
    const long ohw = (long)(d->x_H / 2) * (d->x_W / 2);
    const long nimg = d->M / ohw;
    float* part = reinterpret_cast<float*>(d->workspace);
    hipLaunchKernelGGL(conv3s2_c64_wgrad_kernel, dim3(wgs), dim3(256), WG_LDS2, s, (const bf16_t*)d->Y, (const bf16_t*)d->X, part, (int)nimg,
                       d->x_H, d->x_W, (unsigned)(nimg * d->x_H * d->x_W * 128), (unsigned)((long)d->M * 128));
    hipLaunchKernelGGL(conv3_wgrad_reduce, dim3(cdiv(64 * 576, 256), cdiv(wgs, 16)), dim3(256), 0, s, part, wgs, d->alpha, d->dW, (long)d->ldw);
    return 1;
}

int ga_conv3_c64_wgrad_try(const ga_wgrad_desc* d, hipStream_t s) {
    const int wgs = conv3_wgrad_wgs(d);
    if (!wgs || !d->workspace || (size_t)d->ws_bytes < (size_t)wgs * 64 * 576 * sizeof(float)) return 0;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_c64_wgrad_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS) == hipSuccess;
    if (!attr_ok) return 0;
    const long hw = (long)d->x_H * d->x_W;
    float* part = reinterpret_cast<float*>(d->workspace);
    hipLaunchKernelGGL(conv3_c64_wgrad_kernel, dim3(wgs), dim3(256), WG_LDS, s, (const bf16_t*)d->Y, (const bf16_t*)d->X, part,
                       (int)(d->M / hw), d->x_H, d->x_W, (unsigned)((long)d->M * 128));
    hipLaunchKernelGGL(conv3_wgrad_reduce, dim3(cdiv(64 * 576, 256), cdiv(wgs, 16)), dim3(256), 0, s, part, wgs, d->alpha, d->dW, (long)d->ldw);
    return 1;
}
