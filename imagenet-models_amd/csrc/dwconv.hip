// Depthwise 7x7 convolution (pad 3, stride 1), NHWC, for gfx950.  HBM/LDS-bound VALU kernels -- no MFMA:
// 49 MAC per element, a depthwise product as a GEMM would waste 1 - 1/C of the matrix core.
//
//  * workgroup = (image, TH x TW output tile, slice of <=64 channels); the input tile with its 3-pixel halo is
//    staged once in LDS ([pixel][channel], channel-contiguous so global loads are 128 B per pixel);
//  * work item = (output row, strip of 7 output pixels, 4 channels): a 13-pixel input row segment is held in
//    registers and reused by the 7 taps x 7 outputs of that row -> 13 LDS reads per 196 FMAs;
//  * forward and backward-data are the same kernel (backward-data = flipped taps, + fused residual add);
//  * backward-weight keeps the 49 x 64 partial sums of a channel slice in registers while a persistent
//    workgroup walks over many tiles, so only nblocks x 49 x 64 fp32 atomics reach memory.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

int num_cus() {
    static int n = [] {
        int c = 256;
        ga_device_info(&c, nullptr, nullptr);
        return c;
    }();
    return n;
}

constexpr int kCSF = 32;  // channels per slice, forward / backward-data kernel (LDS 32 KiB -> 4-5 workgroups per CU)
constexpr int kCSW = 64;  // channels per slice, backward-weight kernel (224 of 256 threads hold accumulators)
constexpr int kCG = 4;    // channels per work item

template <typename T> struct vec4;  // 4 channels of T
template <> struct vec4<float> { typedef float4 type; };
template <> struct vec4<bf16_t> { typedef uint2 type; };

__device__ __forceinline__ void cvt4(const float4& v, float f[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
__device__ __forceinline__ void cvt4(const uint2& v, float f[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

// stage a (TH+6)x(TW+6) x cs tile (zero outside the image) of NHWC tensor `src` into LDS [pix][CS];
// 16-byte global loads (8 bf16 / 4 fp32 channels per thread), constant-divisor index math
template <typename T, int TH, int TW, int CS>
__device__ __forceinline__ void stage_halo(const T* src, T* lds, long img_base, int H, int W, int C, int y0, int x0,
                                           int c0, int cs, int tid, int nthreads) {
    constexpr int PW = TW + 6, PH = TH + 6;
    constexpr int EPC = 16 / (int)sizeof(T);          // channels per 16-byte piece
    constexpr int PPP = CS / EPC;                     // pieces per pixel (full slice)
    if (cs % EPC == 0) {
        const int npp = cs / EPC;
        for (int i = tid; i < PH * PW * PPP; i += nthreads) {
            const int pc = i % PPP, p = i / PPP;
            if (pc >= npp) continue;
            const int py = p / PW, px = p - py * PW;
            const int y = y0 + py - 3, x = x0 + px - 3;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                v = *reinterpret_cast<const uint4*>(src + (img_base + (long)y * W + x) * C + c0 + pc * EPC);
            *reinterpret_cast<uint4*>(lds + p * CS + pc * EPC) = v;
        }
    } else {   // ragged slice (cs multiple of 4 only): 4-channel pieces
        const int cgs = cs / kCG;
        for (int i = tid; i < PH * PW * cgs; i += nthreads) {
            const int cg = i % cgs, p = i / cgs;
            const int py = p / PW, px = p - py * PW;
            const int y = y0 + py - 3, x = x0 + px - 3;
            typename vec4<T>::type v;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                v = *reinterpret_cast<const typename vec4<T>::type*>(src + (img_base + (long)y * W + x) * C + c0 + cg * kCG);
            else
                memset(&v, 0, sizeof(v));
            *reinterpret_cast<typename vec4<T>::type*>(lds + p * CS + cg * kCG) = v;
        }
    }
}

// y = bias + conv7x7(x, w)  [+ res];  flip selects the transposed (backward-data) taps
template <typename T, int TH, int TW, int NT>
__global__ __launch_bounds__(NT) void dwconv7_kernel(const T* __restrict__ x, const float* __restrict__ w49,
                                                     const float* __restrict__ bias, const T* __restrict__ res,
                                                     T* __restrict__ y, int H, int W, int C, int flip,
                                                     T* __restrict__ y2, const float* __restrict__ y2scale) {
    constexpr int PW = TW + 6, PH = TH + 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);                                    // [PH*PW][kCSF]
    float* ws = reinterpret_cast<float*>(smem + PH * PW * kCSF * sizeof(T)); // [49][kCSF]
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    // channel slice fastest: the slices of one spatial tile run together and share the tile's cache lines in L2
    const int slices = (C + kCSF - 1) / kCSF;
    int t = blockIdx.x / slices;
    const int c0 = (blockIdx.x - t * slices) * kCSF;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const long b = t / tiles_y;
    const int cs = min(kCSF, C - c0);
    const int y0 = ty * TH, x0 = tx * TW;
    const long img = b * H * W;

    stage_halo<T, TH, TW, kCSF>(x, xs, img, H, W, C, y0, x0, c0, cs, tid, NT);
    for (int i = tid; i < 49 * cs; i += NT) {
        const int tap = i / cs, c = i - tap * cs;
        ws[tap * kCSF + c] = w49[(flip ? 48 - tap : tap) * C + c0 + c];
    }
    __syncthreads();

    const int cgs = cs / kCG;
    constexpr int XS = TW / 7;                 // strips per row
    const int nitems = TH * XS * cgs;
    for (int it = tid; it < nitems; it += NT) {
        const int cg = it % cgs;
        int r = it / cgs;
        const int oy = r % TH, xsi = r / TH;   // row-fastest item order (see bank note in DESIGN.md)
        const int ox0 = xsi * 7;
        float acc[7][4];
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = bias[c0 + cg * kCG + c];
        }
#pragma unroll
        for (int o = 0; o < 7; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[o][c] = bv[c];
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
            float in[13][4];
            const T* rowp = xs + ((oy + ky) * PW + ox0) * kCSF + cg * kCG;
#pragma unroll
            for (int i = 0; i < 13; ++i) cvt4(*reinterpret_cast<const typename vec4<T>::type*>(rowp + i * kCSF), in[i]);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float4 wv = *reinterpret_cast<const float4*>(ws + (ky * 7 + kx) * kCSF + cg * kCG);
#pragma unroll
                for (int o = 0; o < 7; ++o) {
                    acc[o][0] = fmaf(in[o + kx][0], wv.x, acc[o][0]);
                    acc[o][1] = fmaf(in[o + kx][1], wv.y, acc[o][1]);
                    acc[o][2] = fmaf(in[o + kx][2], wv.z, acc[o][2]);
                    acc[o][3] = fmaf(in[o + kx][3], wv.w, acc[o][3]);
                }
            }
        }
        const int yy = y0 + oy;
        if (yy < H) {
#pragma unroll
            for (int o = 0; o < 7; ++o) {
                const int xx = x0 + ox0 + o;
                if (xx < W) {
                    const long off = (img + (long)yy * W + xx) * C + c0 + cg * kCG;
                    if (res) {
                        float rv[4];
                        load4(res + off, rv);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[o][c] += rv[c];
                    }
                    store4(y + off, acc[o]);
                    if (y2) {   // second output: the stored value times a per-image factor (the next block's DropPath scale)
                        const float sc = y2scale[b];
                        float a2[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) a2[c] = elt<T>::round(acc[o][c]) * sc;
                        store4(y2 + off, a2);
                    }
                }
            }
        }
    }
}

// dw49[tap][c] += sum_{b,y,x} dy[b,y,x,c] * x[b,y+ky-3,x+kx-3,c];   dbias[c] += sum dy
// persistent: blockIdx.x walks tiles with stride gridDim.x, blockIdx.y = channel slice.
template <typename T, int TH, int TW, int NT>
__global__ __launch_bounds__(NT) void dwconv7_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           float* __restrict__ part, int B, int H, int W, int C) {
    constexpr int PW = TW + 6, PH = TH + 6;
    constexpr int RS = TH / 7;  // row groups of 7 output rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);                                   // [PH*PW][kCSW]
    T* ds = reinterpret_cast<T*>(smem + PH * PW * kCSW * sizeof(T));       // [TH*TW][kCSW]
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long ntiles = (long)B * tiles_x * tiles_y;
    const int c0 = blockIdx.y * kCSW;
    const int cs = min(kCSW, C - c0);
    const int cgs = cs / kCG;
    // item = (cg, ky, row group): keeps acc[7 kx][4 ch] (+ 4 bias sums) in registers across tiles
    const int nitems = cgs * 7 * RS;
    const bool active = tid < nitems;
    const int cg = tid % cgs;
    const int ky = (tid / cgs) % 7;
    const int rg = tid / (cgs * 7);
    float acc[7][4], bsum[4];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] = 0.f;

    // bf16 slices made of whole 16-byte pieces: the next tile's pieces are fetched into registers while this tile is
    // multiplied (a load -> LDS-store loop exposes the memory latency once per iteration, 12x per tile)
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int PPP = kCSW / EPC;
    constexpr int NX = (PH * PW * PPP + NT - 1) / NT, ND = (TH * TW * PPP + NT - 1) / NT;
    constexpr bool kPrefetch = sizeof(T) == 2;
    const bool fast = kPrefetch && cs % EPC == 0;
    uint4 rx[kPrefetch ? NX : 1], rd[kPrefetch ? ND : 1];
    auto fetch = [&](long t) {
        long tt = t;
        const int tx = (int)(tt % tiles_x); tt /= tiles_x;
        const int ty = (int)(tt % tiles_y);
        const long img = (tt / tiles_y) * H * W;
        const int y0 = ty * TH, x0 = tx * TW;
        const int npp = cs / EPC;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + j * NT;
            const int pc = i % PPP, p = i / PPP;
            const int py = p / PW, px = p - py * PW;
            const int y = y0 + py - 3, xx = x0 + px - 3;
            rx[j] = make_uint4(0, 0, 0, 0);
            if (p < PH * PW && pc < npp && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
                rx[j] = *reinterpret_cast<const uint4*>(x + (img + (long)y * W + xx) * C + c0 + pc * EPC);
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + j * NT;
            const int pc = i % PPP, p = i / PPP;
            const int py = p / TW, px = p - py * TW;
            const int y = y0 + py, xx = x0 + px;
            rd[j] = make_uint4(0, 0, 0, 0);
            if (p < TH * TW && pc < npp && y < H && xx < W)
                rd[j] = *reinterpret_cast<const uint4*>(dy + (img + (long)y * W + xx) * C + c0 + pc * EPC);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + j * NT;
            if (i < PH * PW * PPP) *reinterpret_cast<uint4*>(xs + (i / PPP) * kCSW + (i % PPP) * EPC) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + j * NT;
            if (i < TH * TW * PPP) *reinterpret_cast<uint4*>(ds + (i / PPP) * kCSW + (i % PPP) * EPC) = rd[j];
        }
    };
    if constexpr (kPrefetch) {
        if (fast && (long)blockIdx.x < ntiles) fetch(blockIdx.x);
    }

    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();  // previous tile fully consumed
        if (fast) {
            if constexpr (kPrefetch) {
                commit();
                __syncthreads();
                if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
            }
        } else {
            long tt = t;
            const int tx = (int)(tt % tiles_x); tt /= tiles_x;
            const int ty = (int)(tt % tiles_y);
            const long b = tt / tiles_y;
            const int y0 = ty * TH, x0 = tx * TW;
            const long img = b * H * W;
            stage_halo<T, TH, TW, kCSW>(x, xs, img, H, W, C, y0, x0, c0, cs, tid, NT);
            for (int i = tid; i < TH * TW * cgs; i += NT) {
                const int g = i % cgs, p = i / cgs;
                const int py = p / TW, px = p - py * TW;
                const int yy = y0 + py, xx = x0 + px;
                typename vec4<T>::type v;
                if (yy < H && xx < W)
                    v = *reinterpret_cast<const typename vec4<T>::type*>(dy + (img + (long)yy * W + xx) * C + c0 + g * kCG);
                else
                    memset(&v, 0, sizeof(v));
                *reinterpret_cast<typename vec4<T>::type*>(ds + p * kCSW + g * kCG) = v;
            }
            __syncthreads();
        }
        if (active) {
#pragma unroll 1
            for (int r = 0; r < 7; ++r) {
                const int oy = rg * 7 + r;
#pragma unroll 1
                for (int xs0 = 0; xs0 < TW; xs0 += 7) {
                    float in[13][4], g[7][4];
                    const T* rowp = xs + ((oy + ky) * PW + xs0) * kCSW + cg * kCG;
#pragma unroll
                    for (int i = 0; i < 13; ++i)
                        cvt4(*reinterpret_cast<const typename vec4<T>::type*>(rowp + i * kCSW), in[i]);
                    const T* dp = ds + (oy * TW + xs0) * kCSW + cg * kCG;
#pragma unroll
                    for (int i = 0; i < 7; ++i) cvt4(*reinterpret_cast<const typename vec4<T>::type*>(dp + i * kCSW), g[i]);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                        for (int o = 0; o < 7; ++o)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[kx][c] = fmaf(g[o][c], in[o + kx][c], acc[kx][c]);
                    if (ky == 0) {
#pragma unroll
                        for (int o = 0; o < 7; ++o)
#pragma unroll
                            for (int c = 0; c < 4; ++c) bsum[c] += g[o][c];
                    }
                }
            }
        }
    }
    // partial sums of this workgroup: part[blockIdx.x * RS + rg][50][C] (rows 0..48 taps, row 49 the bias sums), plain stores.
    // (Hundreds of workgroups adding into the same 49 x C floats with atomics serialise in L2: that was 10x the
    // time of the arithmetic.)  dwconv7_wgrad_reduce sums the partials.
    if (active) {
        float* pp = part + ((long)blockIdx.x * RS + rg) * 50 * C + c0 + cg * kCG;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) store4(pp + (long)(ky * 7 + kx) * C, acc[kx]);
        if (ky == 0) store4(pp + 49L * C, bsum);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 backward-weight on v_dot2_f32_bf16.  The reduction runs over pixels, so with the tile stored per CHANNEL
// as rows of consecutive x (two pixels per dword) one dot2 does two multiply-adds straight from the bf16 bits:
//     acc[kx] += (dy[x], dy[x+1]) . (in[x+kx], in[x+kx+1])
// -- no bf16 -> fp32 unpacking (it was half of the VALU work of the fp32-FMA form) and fp32 accumulation.  Odd kx
// need the input pairs shifted by one pixel: one v_alignbit per pair and row.
//   lane = channel plane (64 planes per workgroup), wave = (ky, group of 7 output rows); acc[7 kx] per thread.
//   LDS plane strides are odd multiples of 16 B: the 64 lanes' ds_read_b128 fall on distinct 16-B slots, and with
//   plane(channel c) = (c & 7) * 8 + (c >> 3) the transposing ds_write_b32 of the staging pass are conflict-free
//   while its global loads stay 128 B contiguous per pixel.
// ------------------------------------------------------------------------------------------------
constexpr int odd16(int bytes) { return (((bytes + 15) / 16) | 1) * 16; }

template <int TH, int TW> struct DWD {
    static constexpr int TWP = (TW + 1) & ~1, NP = TWP / 2;          // dy row: pixels (even), pairs
    static constexpr int PW = TWP + 6, PH = TH + 6, NPX = PW / 2;    // input tile with halo
    static constexpr int NPR = NP + 3;                               // input pairs a row item needs
    static constexpr int ROWX = (PW * 2 + 15) / 16 * 16, ROWD = (TWP * 2 + 15) / 16 * 16;
    static constexpr int PLX = odd16(PH * ROWX), PLD = odd16(TH * ROWD);
    static constexpr int RS = TH / 7, NW = 7 * RS, NT = 64 * NW;
    static constexpr int LDS = 64 * (PLX + PLD);
    static constexpr int XU = PH * NPX * 8, DU = TH * NP * 8;        // staging units: (row, pixel pair, 8 channels)
    static constexpr int NXU = (XU + NT - 1) / NT, NDU = (DU + NT - 1) / NT;
};

typedef __attribute__((ext_vector_type(2))) __bf16 dw_bf16x2;
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const dw_bf16x2*>(&a), *reinterpret_cast<const dw_bf16x2*>(&b),
                                           c, false);
}

template <int TH, int TW>
__global__ __launch_bounds__((DWD<TH, TW>::NT)) void dwconv7_wgrad_dot2_kernel(const bf16_t* __restrict__ dy,
                                                                             const bf16_t* __restrict__ x,
                                                                             float* __restrict__ part, int B, int H,
                                                                             int W, int C) {
    using G = DWD<TH, TW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                       // [64 planes][PH rows of ROWX bytes]
    unsigned char* ds = smem + 64 * G::PLX;         // [64 planes][TH rows of ROWD bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ky = wave % 7, rg = wave / 7;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long ntiles = (long)B * tiles_x * tiles_y;
    const int c0 = blockIdx.y * 64;
    const int cs = min(64, C - c0);
    const int ncg = cs >> 3;                        // 8-channel groups in this slice (C % 8 == 0)
    const int ch = (lane & 7) * 8 + (lane >> 3);    // channel held by plane `lane`

    // staging units of this thread: (row, pixel pair, 8-channel group) -- the same for every tile, decoded once
    uint4 rxa[G::NXU], rxb[G::NXU], rda[G::NDU], rdb[G::NDU];
    int xrow[G::NXU], xcol[G::NXU], xsrc[G::NXU], xdst[G::NXU];     // tile-relative y / x of the even pixel, offsets
    int drow[G::NDU], dcol[G::NDU], dsrc[G::NDU], ddst[G::NDU];
#pragma unroll
    for (int j = 0; j < G::NXU; ++j) {
        const int u = tid + j * G::NT;
        const int cgi = u & 7, q = u >> 3;
        const int pr = q % G::NPX, row = q / G::NPX;
        const bool live = u < G::XU && cgi < ncg;
        xrow[j] = live ? row - 3 : -(1 << 20);                     // dead units fail the row test of every tile
        xcol[j] = 2 * pr - 3;
        xsrc[j] = ((row - 3) * W + 2 * pr - 3) * C + cgi * 8;
        xdst[j] = cgi * G::PLX + row * G::ROWX + pr * 4;
    }
#pragma unroll
    for (int j = 0; j < G::NDU; ++j) {
        const int u = tid + j * G::NT;
        const int cgi = u & 7, q = u >> 3;
        const int pr = q % G::NP, row = q / G::NP;
        const bool live = u < G::DU && cgi < ncg;
        drow[j] = live ? row : (1 << 20);
        dcol[j] = 2 * pr;
        dsrc[j] = (row * W + 2 * pr) * C + cgi * 8;
        ddst[j] = cgi * G::PLD + row * G::ROWD + pr * 4;
    }
    auto fetch = [&](long t) {
        long tt = t;
        const int tx = (int)(tt % tiles_x); tt /= tiles_x;
        const int ty = (int)(tt % tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const long base = ((tt / tiles_y) * H * W + (long)y0 * W + x0) * C + c0;   // tile origin pixel, slice channel 0
#pragma unroll
        for (int j = 0; j < G::NXU; ++j) {
            const int y = y0 + xrow[j], xa = x0 + xcol[j];
            rxa[j] = make_uint4(0, 0, 0, 0);
            rxb[j] = make_uint4(0, 0, 0, 0);
            if ((unsigned)y < (unsigned)H) {
                const bf16_t* src = x + base + xsrc[j];
                if ((unsigned)xa < (unsigned)W) rxa[j] = *reinterpret_cast<const uint4*>(src);
                if ((unsigned)(xa + 1) < (unsigned)W) rxb[j] = *reinterpret_cast<const uint4*>(src + C);
            }
        }
#pragma unroll
        for (int j = 0; j < G::NDU; ++j) {
            const int y = y0 + drow[j], xa = x0 + dcol[j];
            rda[j] = make_uint4(0, 0, 0, 0);
            rdb[j] = make_uint4(0, 0, 0, 0);
            if (y < H) {
                const bf16_t* src = dy + base + dsrc[j];
                if (xa < W && dcol[j] < TW) rda[j] = *reinterpret_cast<const uint4*>(src);
                if (xa + 1 < W && dcol[j] + 1 < TW) rdb[j] = *reinterpret_cast<const uint4*>(src + C);
            }
        }
    };
    // commit: two pixels x 8 channels -> 8 dwords (the pixel pair of one channel), one per channel plane
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < G::NXU; ++j) {
            if (xrow[j] > -(1 << 19)) {
                unsigned char* base = xs + xdst[j];
                const unsigned wa[4] = {rxa[j].x, rxa[j].y, rxa[j].z, rxa[j].w};
                const unsigned wb[4] = {rxb[j].x, rxb[j].y, rxb[j].z, rxb[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // channel k = 2i (low halves), 2i+1 (high halves); plane = k*8 + cgi
                    *reinterpret_cast<unsigned*>(base + (2 * i) * 8 * G::PLX) = __builtin_amdgcn_perm(wb[i], wa[i], 0x05040100u);
                    *reinterpret_cast<unsigned*>(base + (2 * i + 1) * 8 * G::PLX) = __builtin_amdgcn_perm(wb[i], wa[i], 0x07060302u);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G::NDU; ++j) {
            if (drow[j] < (1 << 19)) {
                unsigned char* base = ds + ddst[j];
                const unsigned wa[4] = {rda[j].x, rda[j].y, rda[j].z, rda[j].w};
                const unsigned wb[4] = {rdb[j].x, rdb[j].y, rdb[j].z, rdb[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<unsigned*>(base + (2 * i) * 8 * G::PLD) = __builtin_amdgcn_perm(wb[i], wa[i], 0x05040100u);
                    *reinterpret_cast<unsigned*>(base + (2 * i + 1) * 8 * G::PLD) = __builtin_amdgcn_perm(wb[i], wa[i], 0x07060302u);
                }
            }
        }
    };

    float acc[7], bs = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = 0.f;
    const unsigned ones = 0x3f803f80u;              // (1.0bf16, 1.0bf16)

    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                            // previous tile fully consumed
        commit();
        __syncthreads();
        if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
        const unsigned char* xp = xs + lane * G::PLX + (rg * 7 + ky) * G::ROWX;
        const unsigned char* dp = ds + lane * G::PLD + (rg * 7) * G::ROWD;
#pragma unroll 1
        for (int r = 0; r < 7; ++r) {
            unsigned g2[(G::NP + 3) / 4 * 4], P[(G::NPR + 3) / 4 * 4 + 1], Q[G::NP + 2];
#pragma unroll
            for (int i = 0; i < (G::NP + 3) / 4; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(dp + r * G::ROWD + i * 16);
                g2[4 * i] = v.x; g2[4 * i + 1] = v.y; g2[4 * i + 2] = v.z; g2[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < (G::NPR + 3) / 4; ++i) {
                const uint4 v = *reinterpret_cast<const uint4*>(xp + r * G::ROWX + i * 16);
                P[4 * i] = v.x; P[4 * i + 1] = v.y; P[4 * i + 2] = v.z; P[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < G::NP + 2; ++j) Q[j] = __builtin_amdgcn_alignbit(P[j + 1], P[j], 16);
#pragma unroll
            for (int m = 0; m < G::NP; ++m)       // pixel pair outermost: consecutive dot2 hit different accumulators
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
                    acc[kx] = dot2bf(g2[m], (kx & 1) ? Q[m + (kx >> 1)] : P[m + (kx >> 1)], acc[kx]);
            if (ky == 0) {
#pragma unroll
                for (int m = 0; m < G::NP; ++m) bs = dot2bf(g2[m], ones, bs);
            }
        }
    }
    // the row groups of one workgroup meet in LDS, then ONE partial per workgroup: part[blockIdx.x][50][C]
    if constexpr (G::RS > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [7 ky][8][64 lanes]
        if (rg > 0) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) red[(ky * 8 + kx) * 64 + lane] = acc[kx];
            red[(ky * 8 + 7) * 64 + lane] = bs;
        }
        __syncthreads();
        if (rg == 0) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) acc[kx] += red[(ky * 8 + kx) * 64 + lane];
            bs += red[(ky * 8 + 7) * 64 + lane];
        }
    }
    if (rg == 0 && ch < cs) {
        float* pp = part + (long)blockIdx.x * 50 * C + c0 + ch;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) pp[(long)(ky * 7 + kx) * C] = acc[kx];
        if (ky == 0) pp[49L * C] = bs;
    }
}

// dw49[tap][c] += sum_g part[g][tap][c]; dbias[c] += sum_g part[g][49][c].   block = 64 columns x 16 partial ranges
__global__ __launch_bounds__(1024) void dwconv7_wgrad_reduce(const float* __restrict__ part, int nparts, int C,
                                                             float* __restrict__ dw49, float* __restrict__ dbias) {
    constexpr int NG = 16;
    __shared__ float red[NG][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + col;           // over 50 * C
    const int total = 50 * C;
    float s = 0.f;
    if (idx < total) {
        const int per = (nparts + NG - 1) / NG;
        const int g0 = grp * per, g1 = min(nparts, g0 + per);
#pragma unroll 8
        for (int g = g0; g < g1; ++g) s += part[(long)g * total + idx];
    }
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && idx < total) {
        s = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) s += red[g][col];
        if (idx < 49 * C) dw49[idx] += s;
        else if (dbias) dbias[idx - 49 * C] += s;
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward / backward-data on v_dot2_f32_bf16, same channel-plane LDS image as the backward-weight kernel above
// (lane = channel, rows of consecutive x, two pixels per dword):
//     y[x] += (in[x+kx], in[x+kx+1]) . (w[kx], w[kx+1])        kx = 0, 2, 4, 6  (w[7] = 0)
// The 7x7 taps of the lane's channel live in 28 registers as bf16 pairs (rounded once per workgroup: the bf16 mode
// rounds every GEMM weight the same way); odd output pixels use the input pairs shifted by one pixel (v_alignbit).
// 28 dot2 + 4.5 alignbit per output element, no bf16 unpacking, no weight reads from LDS, all 64 lanes busy.
//   wave = output row(s) of the tile; results go through a per-wave [pixel][64 ch] LDS strip back to NHWC 16-byte
//   stores (+ residual for backward-data).  Persistent over tiles, next tile's pixels prefetched in registers.
// ------------------------------------------------------------------------------------------------
template <int TH, int TW> struct DWF {
    static constexpr int TWP = (TW + 1) & ~1;                        // output columns computed (even)
    static constexpr int PW = TWP + 6, PH = TH + 6, NPX = PW / 2;    // input tile with halo, pixel pairs per row
    static constexpr int NPR = NPX + 1;                              // dwords of a row an item touches (incl. zero pad)
    static constexpr int ROWX = (NPR * 4 + 15) / 16 * 16;            // 48 (14x14) / 32 (7x7)
    static constexpr int PLX = odd16(PH * ROWX);
    static constexpr int NW = 7, NT = 64 * NW;                       // 7 waves: rows wave, wave + 7, ...
    static constexpr int STRIP = TW * 128;                           // per-wave [TW pixels][64 ch] bf16
    static constexpr int LDS = 64 * PLX + NW * STRIP;
    static constexpr int XU = PH * NPX * 8;
    static constexpr int NXU = (XU + NT - 1) / NT;
};

template <int TH, int TW>
__global__ __launch_bounds__((DWF<TH, TW>::NT)) void dwconv7_dot2_kernel(const bf16_t* __restrict__ x,
                                                                       const float* __restrict__ w49,
                                                                       const float* __restrict__ bias,
                                                                       const bf16_t* __restrict__ res,
                                                                       bf16_t* __restrict__ y, int B, int H, int W, int C,
                                                                       int flip, bf16_t* __restrict__ y2,
                                                                       const float* __restrict__ y2scale) {
    using G = DWF<TH, TW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                                   // [64 planes][PH rows of ROWX bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* strip = smem + 64 * G::PLX + wave * G::STRIP;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long ntiles = (long)B * tiles_x * tiles_y;
    const int c0 = blockIdx.y * 64;
    const int cs = min(64, C - c0);
    const int ncg = cs >> 3;
    const int ch = (lane & 7) * 8 + (lane >> 3);                // channel held by plane `lane`
    const bool ch_ok = ch < cs;

    // taps of this lane's channel as bf16 pairs (w[2p], w[2p+1]), w[7] = 0; backward-data: the flipped kernel
    unsigned wq[7][4];
    float bv = 0.f;
    {
        float wf[49];
#pragma unroll
        for (int t = 0; t < 49; ++t) wf[t] = ch_ok ? w49[(long)(flip ? 48 - t : t) * C + c0 + ch] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int p = 0; p < 4; ++p) wq[ky][p] = pack2bf(wf[ky * 7 + 2 * p], p < 3 ? wf[ky * 7 + 2 * p + 1] : 0.f);
        if (bias && ch_ok) bv = bias[c0 + ch];
    }
    // the pad dwords behind every row are multiplied by the zero tap: they must hold finite values -> zero them once
    for (int i = tid; i < 64 * G::PLX / 4; i += G::NT) reinterpret_cast<unsigned*>(xs)[i] = 0u;

    uint4 rxa[G::NXU], rxb[G::NXU];
    int xrow[G::NXU], xcol[G::NXU], xsrc[G::NXU], xdst[G::NXU];
#pragma unroll
    for (int j = 0; j < G::NXU; ++j) {
        const int u = tid + j * G::NT;
        const int cgi = u & 7, q = u >> 3;
        const int pr = q % G::NPX, row = q / G::NPX;
        const bool live = u < G::XU && cgi < ncg;
        xrow[j] = live ? row - 3 : -(1 << 20);
        xcol[j] = 2 * pr - 3;
        xsrc[j] = ((row - 3) * W + 2 * pr - 3) * C + cgi * 8;
        xdst[j] = cgi * G::PLX + row * G::ROWX + pr * 4;
    }
    auto fetch = [&](long t) {
        long tt = t;
        const int tx = (int)(tt % tiles_x); tt /= tiles_x;
        const int ty = (int)(tt % tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const long base = ((tt / tiles_y) * H * W + (long)y0 * W + x0) * C + c0;
#pragma unroll
        for (int j = 0; j < G::NXU; ++j) {
            const int yy = y0 + xrow[j], xa = x0 + xcol[j];
            rxa[j] = make_uint4(0, 0, 0, 0);
            rxb[j] = make_uint4(0, 0, 0, 0);
            if ((unsigned)yy < (unsigned)H) {
                const bf16_t* src = x + base + xsrc[j];
                if ((unsigned)xa < (unsigned)W) rxa[j] = *reinterpret_cast<const uint4*>(src);
                if ((unsigned)(xa + 1) < (unsigned)W) rxb[j] = *reinterpret_cast<const uint4*>(src + C);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < G::NXU; ++j) {
            if (xrow[j] > -(1 << 19)) {
                unsigned char* base = xs + xdst[j];
                const unsigned wa[4] = {rxa[j].x, rxa[j].y, rxa[j].z, rxa[j].w};
                const unsigned wb[4] = {rxb[j].x, rxb[j].y, rxb[j].z, rxb[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<unsigned*>(base + (2 * i) * 8 * G::PLX) = __builtin_amdgcn_perm(wb[i], wa[i], 0x05040100u);
                    *reinterpret_cast<unsigned*>(base + (2 * i + 1) * 8 * G::PLX) = __builtin_amdgcn_perm(wb[i], wa[i], 0x07060302u);
                }
            }
        }
    };

    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                            // previous tile fully consumed (and the pad zeroing, first time)
        commit();
        __syncthreads();
        if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
        long tt = t;
        const int tx = (int)(tt % tiles_x); tt /= tiles_x;
        const int ty = (int)(tt % tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const long img = (tt / tiles_y) * H * W;
#pragma unroll 1
        for (int oy = wave; oy < TH; oy += G::NW) {
            float acc[G::TWP];
#pragma unroll
            for (int o = 0; o < G::TWP; ++o) acc[o] = bv;
            const unsigned char* xp = xs + lane * G::PLX + oy * G::ROWX;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                unsigned P[G::ROWX / 4], Q[G::NPX];
#pragma unroll
                for (int i = 0; i < G::ROWX / 16; ++i) {
                    const uint4 v = *reinterpret_cast<const uint4*>(xp + ky * G::ROWX + i * 16);
                    P[4 * i] = v.x; P[4 * i + 1] = v.y; P[4 * i + 2] = v.z; P[4 * i + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < G::NPX; ++j) Q[j] = __builtin_amdgcn_alignbit(P[j + 1], P[j], 16);
#pragma unroll
                for (int o = 0; o < G::TWP; ++o)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[o] = dot2bf((o & 1) ? Q[(o >> 1) + p] : P[(o >> 1) + p], wq[ky][p], acc[o]);
            }
            // lane's channel, TW pixels -> strip[pixel][channel]; then 16-byte NHWC pieces (+ residual) out
#pragma unroll
            for (int o = 0; o < TW; ++o)
                *reinterpret_cast<bf16_t*>(strip + o * 128 + ch * 2) = f2bf(acc[o]);
            const int yy = y0 + oy;
#pragma unroll
            for (int it = 0; it < (TW * 8 + 63) / 64; ++it) {
                const int i = lane + it * 64;
                const int px = i >> 3, pc = i & 7;
                if (i < TW * 8 && pc < ncg && yy < H && x0 + px < W) {
                    uint4 v = *reinterpret_cast<const uint4*>(strip + px * 128 + pc * 16);
                    const long off = (img + (long)yy * W + x0 + px) * C + c0 + pc * 8;
                    if (res) {
                        float a[8], r[8];
                        unpack8(v, a);
                        load8(res + off, r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] += r[e];
                        store8(y + off, a);
                        if (y2) {
                            const float sc = y2scale[img / ((long)H * W)];
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = bf2f(f2bf(a[e])) * sc;
                            store8(y2 + off, a);
                        }
                    } else {
                        *reinterpret_cast<uint4*>(y + off) = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward / backward-data of 14 x 14 tiles on the MATRIX cores.  Per channel and tap row ky, the 1-D correlation
// along x of 16 output rows at once is a small GEMM with a banded Toeplitz operand built from the 7 taps:
//     D[ox][n] += sum_j T_ky[ox][j] * IN[n + ky][j],   T_ky[ox][j] = w[ky][j - ox] for 0 <= j - ox <= 6, else 0
// i.e. one v_mfma_f32_16x16x32_bf16 (A = T_ky: 16 ox x 32 j, B = 16 input rows x 32 j of the channel plane) replaces
// 14 x 14 x 7 multiply-adds.  Only 3136 of the instruction's 16384 MACs are useful, but the matrix pipe is 16x the
// packed-VALU rate: 7 MFMAs (112 cycles) per channel tile instead of ~1600 VALU cycles in the dot2 form, and the kernel
// lands on the HBM roofline instead of the VALU's.
//   workgroup = 16 waves, 32-channel slice; wave w owns channel planes 2w, 2w+1 and keeps their 2 x 7 Toeplitz
//   fragments in registers (56 VGPRs) across a persistent walk over tiles; same transposing staging as the dot2 kernels;
//   results go through an LDS [pixel][32 ch] buffer back to 16-byte NHWC stores (+ residual for backward-data).
// ------------------------------------------------------------------------------------------------
struct DWM {
    static constexpr int CS = 32, NCG = 4;                           // channels per slice, 8-channel groups
    static constexpr int PH = 20, NPX = 10;                          // 20 x 20 input tile, 10 pixel pairs per row
    static constexpr int ROWX = 48, PLX = 976;                       // as DWF<14, 14>
    static constexpr int NT = 512, CPW = 4;                          // 8 waves, 4 channel planes per wave
    static constexpr int OROW = 14 * 64 + 16;                        // out buffer: [14 rows][14 px x 64 B] + 16 B pad per row
    static constexpr int OUT0 = CS * PLX;
    static constexpr int LDS = OUT0 + 16 * OROW;                     // 16 rows so that the garbage rows 14, 15 stay inside
    static constexpr int XU = PH * NPX * NCG;                        // 800 staging units
    static constexpr int NXU = (XU + NT - 1) / NT;
};

__global__ __launch_bounds__(512) void dwconv7_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w49,
                                                            const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ y, int B, int H, int W, int C, int flip,
                                                            bf16_t* __restrict__ y2, const float* __restrict__ y2scale) {
    using G = DWM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;
    unsigned char* ob = smem + G::OUT0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / 14, tiles_y = H / 14;
    const long ntiles = (long)B * tiles_x * tiles_y;
    const int c0 = blockIdx.y * G::CS;
    const int cs = min(G::CS, C - c0);
    const int ncg = cs >> 3;

    // Toeplitz fragments of this wave's four channels: lane holds T[ox = lane & 15][j = 8 * (lane >> 4) + i], i < 8
    bf16x8_t tq[G::CPW][7];
    float bv[G::CPW];
    int chl[G::CPW];
#pragma unroll
    for (int q = 0; q < G::CPW; ++q) {
        const int p = G::CPW * wave + q;               // plane; channel = (p % 4) * 8 + p / 4
        chl[q] = (p & 3) * 8 + (p >> 2);
        const bool ok = chl[q] < cs;
        const int ox = lane & 15, j0 = 8 * (lane >> 4);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int dd = j0 + i - ox;
                const int tap = ky * 7 + dd;
                t[i] = (ok && ox < 14 && dd >= 0 && dd <= 6) ? w49[(long)(flip ? 48 - tap : tap) * C + c0 + chl[q]] : 0.f;
            }
            unsigned u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = pack2bf(t[2 * i], t[2 * i + 1]);
            tq[q][ky] = *reinterpret_cast<const bf16x8_t*>(u);
            __builtin_amdgcn_sched_barrier(0);     // one fragment's loads at a time (224 hoisted loads spill)
        }
        bv[q] = (bias && ok) ? bias[c0 + chl[q]] : 0.f;
    }
    // every LDS byte a fragment read can touch must be finite (it meets a zero of T): zero the whole allocation once
    for (int i = tid; i < G::LDS / 4; i += G::NT) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // staging units of this thread (tile-invariant): (row, pixel pair, 8-channel group)
    uint4 rxa[G::NXU], rxb[G::NXU];
    int xrow[G::NXU], xcol[G::NXU], xsrc[G::NXU], xdst[G::NXU];
#pragma unroll
    for (int j = 0; j < G::NXU; ++j) {
        const int u = tid + j * G::NT;
        const int cgi = u & 3, q = u >> 2;
        const int pr = q % G::NPX, row = q / G::NPX;
        const bool live = u < G::XU && cgi < ncg;
        xrow[j] = live ? row - 3 : -(1 << 20);
        xcol[j] = 2 * pr - 3;
        xsrc[j] = ((row - 3) * W + 2 * pr - 3) * C + cgi * 8;
        xdst[j] = cgi * G::PLX + row * G::ROWX + pr * 4;
    }
    auto fetch = [&](long t) {
        long tt = t;
        const int tx = (int)(tt % tiles_x); tt /= tiles_x;
        const int ty = (int)(tt % tiles_y);
        const int y0 = ty * 14, x0 = tx * 14;
        const long base = ((tt / tiles_y) * H * W + (long)y0 * W + x0) * C + c0;
#pragma unroll
        for (int j = 0; j < G::NXU; ++j) {
            const int yy = y0 + xrow[j], xa = x0 + xcol[j];
            rxa[j] = make_uint4(0, 0, 0, 0);
            rxb[j] = make_uint4(0, 0, 0, 0);
            if ((unsigned)yy < (unsigned)H) {
                const bf16_t* src = x + base + xsrc[j];
                if ((unsigned)xa < (unsigned)W) rxa[j] = *reinterpret_cast<const uint4*>(src);
                if ((unsigned)(xa + 1) < (unsigned)W) rxb[j] = *reinterpret_cast<const uint4*>(src + C);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < G::NXU; ++j) {
            if (xrow[j] > -(1 << 19)) {
                unsigned char* base = xs + xdst[j];
                const unsigned wa[4] = {rxa[j].x, rxa[j].y, rxa[j].z, rxa[j].w};
                const unsigned wb[4] = {rxb[j].x, rxb[j].y, rxb[j].z, rxb[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // channel k = 2i / 2i+1 of the group -> plane k * 4 + cgi
                    *reinterpret_cast<unsigned*>(base + (2 * i) * 4 * G::PLX) = __builtin_amdgcn_perm(wb[i], wa[i], 0x05040100u);
                    *reinterpret_cast<unsigned*>(base + (2 * i + 1) * 4 * G::PLX) = __builtin_amdgcn_perm(wb[i], wa[i], 0x07060302u);
                }
            }
        }
    };

    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                            // planes and out buffer of the previous tile fully consumed
        commit();
        __syncthreads();
        if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll
        for (int q = 0; q < G::CPW; ++q) {
            f32x4_t acc = {bv[q], bv[q], bv[q], bv[q]};
            const unsigned char* bp = xs + (G::CPW * wave + q) * G::PLX + (lane & 15) * G::ROWX + (lane >> 4) * 16;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const uint4 bfrag = *reinterpret_cast<const uint4*>(bp + ky * G::ROWX);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq[q][ky], *reinterpret_cast<const bf16x8_t*>(&bfrag), acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // one channel's 7 fragment reads in flight at a time (28 VGPRs)
            // D[ox = 4 * (lane >> 4) + r][n = lane & 15] -> out[n][ox][channel]
            const int n = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = 4 * (lane >> 4) + r;
                if (ox < 14) *reinterpret_cast<bf16_t*>(ob + n * G::OROW + ox * 64 + chl[q] * 2) = f2bf(acc[r]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (14 * 14 * 4 + G::NT - 1) / G::NT; ++it) {   // 16-byte NHWC pieces: (row, pixel, 8-channel group)
            const int id = tid + it * G::NT;
            const int pc = id & 3, px = (id >> 2) % 14, rw = (id >> 2) / 14;
            if (id < 14 * 14 * 4 && pc < ncg) {
                long tt = t;
                const int tx = (int)(tt % tiles_x); tt /= tiles_x;
                const int ty = (int)(tt % tiles_y);
                const long off = ((tt / tiles_y) * H * W + (long)(ty * 14 + rw) * W + tx * 14 + px) * C + c0 + pc * 8;
                uint4 v = *reinterpret_cast<const uint4*>(ob + rw * G::OROW + px * 64 + pc * 16);
                if (res) {
                    float a[8], rr[8];
                    unpack8(v, a);
                    load8(res + off, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += rr[e];
                    store8(y + off, a);
                    if (y2) {
                        const float sc = y2scale[tt / tiles_y];
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = bf2f(f2bf(a[e])) * sc;
                        store8(y2 + off, a);
                    }
                } else {
                    *reinterpret_cast<uint4*>(y + off) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward / backward-data, register-sliding NHWC form (no transposition, no workgroup barriers).
//   lane = one CHANNEL PAIR (one dword of the NHWC row) of a strip of 4 output columns; the wave marches down the rows
//   of its segment.  Per input row it takes 10 dwords (one per input column; columns outside the image read 0),
//   converts them to 2 x fp32 once, and feeds them to the SEVEN output rows the input row belongs to: 196 v_pk_fma_f32
//   (2 channels each) per row, with the weights (49 x 2 fp32) and the 7 x 4 x 2 accumulators of the output rows in
//   flight in registers.  The accumulator slot of output row q is q % 7: the march is unrolled by 7 rows, so every slot
//   index is a compile-time constant; R (rows per segment) is a multiple of 7, so the march is head (rows 0 .. 6: tap
//   rows 0 .. r), steady blocks of 7 rows (all seven tap rows, no checks at all) and tail (rows R .. R + 5: tap rows
//   t + 1 .. 6) -- straight-line code around one loop, exactly the 49 multiply-adds per output element of the
//   convolution (plus zero columns at the left / right edge of the image).
//   Bound: packed-fp32 VALU rate -- the kernel with every memory operation removed runs 87 us for 256 x 56 x 56 x 96
//   (220 VALU instructions per row and wave; two waves per SIMD at 236 VGPRs) next to an HBM time of 62 us.
//   Data movement is shaped by the cost of a vector-memory INSTRUCTION (address path: ~16 clocks per 64 lanes whatever
//   the width; the dword-per-lane version of this kernel spent 40 % of its time there):
//     * input rows reach the lanes through a per-wave LDS ring filled by `buffer_load_dwordx4 ... lds` THREE rows ahead
//       of their use (two waves per SIMD cannot hide HBM latency with registers one row ahead).  One instruction
//       moves 4 columns x 16 channel-pair quads: lane L of instruction k fetches the 16 bytes of quad L % 16 at
//       column 4 k + L / 16, which land at ring offset 256 (4 k + L / 16) + 16 (L % 16): the [column][lane] dword
//       image the lanes read back.  3 instructions per row instead of 10 (+ 1 for the residual row);
//     * output rows leave through a 1 KiB per-wave staging piece ([column][lane] dwords in, 16 bytes per lane out):
//       ONE 16-byte store per lane and row (+ 1 for the second output).
//   Every row issues the same number of vector-memory operations (rows outside the image / segment and output rows
//   not yet complete use a buffer resource of zero records: they fetch zeros / drop the store but still count), so
//   the counted s_waitcnt in front of the ring read is exact.
//   Addressing: voffset = per-lane offset (constant down the march), soffset = row base (scalar).
//   Work unit = (image, segment of R output rows); lanes of a unit = strips x channel pairs, padded to whole waves.
// ------------------------------------------------------------------------------------------------
typedef float dw_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned dw_u4;

__device__ __forceinline__ dw_u4 dw_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    dw_u4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int N> __device__ __forceinline__ void dw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ring row = 3 x 1 KiB (input columns 0 .. 11, 10 used) [+ 1 KiB: the 4 residual columns].  c0..c2 / cr are the lane offsets
// minus the instruction offset (which is added to the memory address as well as to the LDS address).
__device__ __forceinline__ void dw_dma_x(dw_u4 rs, unsigned so, unsigned dst, unsigned c0, unsigned c1, unsigned c2) {
    unsigned keep;
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[c0], %[rs], %[so] offen lds\n\t"
                 "buffer_load_dwordx4 %[c1], %[rs], %[so] offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %[c2], %[rs], %[so] offen offset:2048 lds\n\t"
                 "s_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep)
                 : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [rs] "s"(rs), [so] "s"(so), [d] "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void dw_dma_res(dw_u4 rs, unsigned so, unsigned dst, unsigned cr) {
    unsigned keep;
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %[c], %[rs], %[so] offen offset:3072 lds\n\t"
                 "s_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep)
                 : [c] "v"(cr), [rs] "s"(rs), [so] "s"(so), [d] "s"(dst)
                 : "memory");
}

#ifndef DW_DBG
#define DW_DBG 0                                              // timing variants (results wrong): 1 no DMA, 2 no stores, 4 no ring reads
#endif

template <bool RES, bool Y2> struct DWR {
    static constexpr int TW = 4, NC = TW + 6;
    static constexpr int P = 3, D = 4;                        // rows ahead, ring depth
    static constexpr int GC = 3 + (RES ? 1 : 0);              // DMA operations per row group
    static constexpr int ROWB = GC * 1024;
    static constexpr int ST = (DW_DBG & 8) ? 4 : (Y2 ? 2 : 1); // stores per row
    static constexpr int STAGE = D * ROWB;                    // 1 KiB output staging piece behind the ring
    static constexpr int WAVE_LDS = D * ROWB + 1024, LDS = 4 * WAVE_LDS;
    static constexpr int N_STEADY = (P - 1) * GC + P * ST;    // operations younger than row group r + 1 at the wait of row r >= P
};


template <bool RES, bool Y2>
__global__ __launch_bounds__(256, 2) void dwconv7_rs_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w49,
                                                            const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ y, int H, int W, int C, int flip,
                                                            bf16_t* __restrict__ y2, const float* __restrict__ y2scale,
                                                            int R, int nseg, int nstrips, int wpu, int total_waves,
                                                            unsigned bytes) {
    using G = DWR<RES, Y2>;
    constexpr int TW = G::TW, NC = G::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (gw >= total_waves) return;
    const int CP = C >> 1;
    const int unit = gw / wpu, wi = gw - unit * wpu;
    const int img = unit / nseg, seg = unit - img * nseg;
    const int item = wi * 64 + lane;
    const int strip = item / CP;
    const bool active = strip < nstrips;
    const int cp = active ? item - strip * CP : 0;
    const int oy0 = seg * R;

    unsigned char* wsm = smem + wv * G::WAVE_LDS;
    const unsigned wring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wv * G::WAVE_LDS;
    const unsigned* ring = reinterpret_cast<const unsigned*>(wsm) + lane;
    unsigned* stage_w = reinterpret_cast<unsigned*>(wsm + G::STAGE) + lane;                  // [column][lane] dwords in
    const uint4* stage_r = reinterpret_cast<const uint4*>(wsm + G::STAGE) + lane;            // 16 bytes per lane out
    const dw_u4 rx = dw_make_rsrc(x, bytes), rres = dw_make_rsrc(RES ? res : x, bytes);
    const __amdgpu_buffer_rsrc_t ry = dw_rsrc(y, bytes), ry_off = dw_rsrc(y, 0), ry2 = dw_rsrc(Y2 ? y2 : y, bytes),
                                 ry2_off = dw_rsrc(Y2 ? y2 : y, 0);
    // the 16-byte pieces this lane moves: channel-pair quad lane % 16 of the wave (4 consecutive lanes: C % 8 == 0 keeps a
    // quad inside one strip), at column (lane / 16) + 4 k relative to the strip's first input column
    unsigned cx[3], cr, cs;
    {
        const int qitem = wi * 64 + 4 * (lane & 15);
        const int qstrip = qitem / CP;
        const int qcp = qitem - qstrip * CP;
        const bool qa = qstrip < nstrips;
        auto off = [&](int col) {                  // col: image column
            return (qa && (unsigned)col < (unsigned)W) ? (unsigned)(col * C + 2 * qcp) * 2u : 0x80000000u;
        };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = 4 * k + (lane >> 4);
            cx[k] = (j < NC ? off(qstrip * TW + j - 3) : 0x80000000u) - 1024u * k;
        }
        cs = off(qstrip * TW + (lane >> 4));       // output / residual column (lane / 16) of the quad's strip
        cr = cs - 3072u;
    }
    dw_f2 w[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) w[t] = *reinterpret_cast<const dw_f2*>(w49 + (long)(flip ? 48 - t : t) * C + 2 * cp);
    dw_f2 bv = {0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const dw_f2*>(bias + 2 * cp);
    const float sc2 = Y2 ? y2scale[img] : 0.f;
    // every register load of the kernel is complete before the first DMA: from here on the compiler has nothing to wait for
    // on the vector-memory counter, and the hand-counted waits below see only the ring's DMA and the stores
#pragma unroll
    for (int t = 0; t < 49; ++t) asm volatile("" : "+v"(w[t]));
    asm volatile("" : "+v"(bv));
    dw_wait_vm<0>();

    const unsigned rowb = (unsigned)W * C * 2u;
    const unsigned img_base = (unsigned)img * H * rowb;
    const int iy0 = oy0 - 3, nr = R + 6;
    auto issue = [&](int r) {                      // row group r: input row iy0 + r (and the residual of output row r - 6)
        const int iy = iy0 + r;
        const bool v = r < nr && (unsigned)iy < (unsigned)H;
        dw_u4 rs = rx;
        rs[2] = v ? bytes : 0u;
        const unsigned dst = wring + (unsigned)(r & (G::D - 1)) * G::ROWB;
        dw_dma_x(rs, v ? img_base + iy * rowb : 0u, dst, cx[0], cx[1], cx[2]);
        if constexpr (RES) {
            const int qd = r - 6;
            const bool vr = qd >= 0 && qd < R;
            dw_u4 rq = rres;
            rq[2] = vr ? bytes : 0u;
            dw_dma_res(rq, vr ? img_base + (oy0 + qd) * rowb : 0u, dst, cr);
        }
    };
    unsigned raw[NC] = {}, rsd[TW] = {};           // ring row r + 1 (and its residual columns), read while row r is multiplied
    auto read_row = [&](int r) {
        const unsigned* rowp = ring + (r & (G::D - 1)) * (G::ROWB / 4);
#pragma unroll
        for (int j = 0; j < NC; ++j) raw[j] = rowp[j * 64];
        if constexpr (RES) {
#pragma unroll
            for (int o = 0; o < TW; ++o) rsd[o] = rowp[(12 + o) * 64];
        }
    };
    dw_f2 in[NC];
    unsigned rs[TW];
    auto unpack_row = [&]() {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            in[j].x = __uint_as_float(raw[j] << 16);
            in[j].y = __uint_as_float(raw[j] & 0xffff0000u);
        }
        if constexpr (RES) {
#pragma unroll
            for (int o = 0; o < TW; ++o) rs[o] = rsd[o];
        }
    };
#pragma unroll
    for (int r = 0; r <= G::P; ++r) issue(r);
    dw_wait_vm<G::P * G::GC>();
    read_row(0);
    unpack_row();

    dw_f2 acc[7][TW];
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int o = 0; o < TW; ++o) acc[s][o] = bv;

    // One input row r of the march (slot base RR = r % 7, compile-time): tap rows KLO .. KHI are the ones whose output row lies
    // in the segment.  Operations younger than group r + 1 at its wait: P - 1 groups and min(r, P) store sets.
    auto row_iter = [&](int r, auto rr_c, auto klo_c, auto khi_c, auto early_c, auto edge_c) {
        constexpr int RR = decltype(rr_c)::value, KLO = decltype(klo_c)::value, KHI = decltype(khi_c)::value;
        constexpr int EARLY = decltype(early_c)::value;         // r if r < P (head), else -1
        constexpr bool EDGE = decltype(edge_c)::value;          // head / tail row: the input row may lie outside the image
        if (!(DW_DBG & 1)) dw_wait_vm<(EARLY >= 0 ? (G::P - 1) * G::GC + EARLY * G::ST : G::N_STEADY)>();
        if (!(DW_DBG & 4)) read_row(r + 1);
        else {
#pragma unroll
            for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(raw[j]));      // keep the unpacking work
        }
        if (!(DW_DBG & 1)) issue(r + 1 + G::P);                // into the slot of row r, whose dwords are in `in` already
#pragma unroll
        for (int o = 0; o < TW; ++o) acc[RR][o] = bv;          // output row q = r starts in slot r % 7
        if (!EDGE || (unsigned)(iy0 + r) < (unsigned)H) {
#pragma unroll
            for (int ky = KLO; ky <= KHI; ++ky) {
                const int slot = (RR - ky + 7) % 7;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int o = 0; o < TW; ++o)
                        acc[slot][o] = __builtin_elementwise_fma(in[o + kx], w[ky * 7 + kx], acc[slot][o]);
            }
        }
        {                                                       // output row r - 6 is complete (a dropped store while r < 6)
            const int qd = r - 6;
            const bool st = qd >= 0;
            constexpr int slot = (RR + 1) % 7;
            const unsigned so = st ? img_base + (oy0 + qd) * rowb : 0u;
            const __amdgpu_buffer_rsrc_t r1 = st ? ry : ry_off, r2 = st ? ry2 : ry2_off;
            unsigned pk[TW];
#pragma unroll
            for (int o = 0; o < TW; ++o) {
                dw_f2 a = acc[slot][o];
                if constexpr (RES) {
                    a.x += __uint_as_float(rs[o] << 16);
                    a.y += __uint_as_float(rs[o] & 0xffff0000u);
                }
                pk[o] = pack2bf(a.x, a.y);
                if (DW_DBG & 8) {                                   // dword stores straight from the lanes (no staging)
                    const int ix = strip * TW + o;
                    const unsigned co = (active && ix < W) ? (unsigned)(ix * C + 2 * cp) * 2u : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b32(pk[o], r1, co, so, 0);
                } else
                    stage_w[o * 64] = pk[o];
            }
            if (DW_DBG & 8) {
                unpack_row();
                return;
            }
            asm volatile("" ::: "memory");                      // (compiler: the 16-byte read below aliases the dword writes above)
            uint4 v = *stage_r;                                 // LDS operations of one wave execute in order: no barrier
            // The staged dwords stay in their registers until the read-back has RETURNED.  Measured on gfx950: with LDS-DMA
            // in flight, a ds_write fetches the data of its last lanes late, and the VALU instruction hipcc schedules right
            // behind it (the next row's unpacking, into the same registers) reached lanes 48-63 of the staged row first.
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3])::"memory");
            if (DW_DBG & 2) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            else {
                // a 16-byte store fetches its data over several cycles: nothing may overwrite the registers in the two wait
                // states behind it (hipcc pads this hazard only for stores without an SGPR offset; on gfx950 a v_pk_mul it
                // scheduled right behind such a store reached dwords 2-3 of some lanes first)
                dw_u4 d1 = {v.x, v.y, v.z, v.w};
                __builtin_amdgcn_raw_buffer_store_b128(d1, r1, cs, so, 0);
                asm volatile("s_nop 1" ::"v"(d1) : "memory");
                if constexpr (Y2) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= sc2;
                    dw_u4 d2 = {pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7])};
                    __builtin_amdgcn_raw_buffer_store_b128(d2, r2, cs, so, 0);
                    asm volatile("s_nop 1" ::"v"(d2) : "memory");
                }
            }
        }
        unpack_row();                                           // ring row r + 1 -> registers of the next iteration
    };
#define DW_IC(v) std::integral_constant<int, (v)>{}
#define DW_HEAD(r) row_iter(r, DW_IC(r), DW_IC(0), DW_IC(r), DW_IC((r) < G::P ? (r) : -1), std::true_type{})
#define DW_BODY(rr) row_iter(r0 + rr, DW_IC(rr), DW_IC(0), DW_IC(6), DW_IC(-1), std::false_type{})
#define DW_TAIL(t) row_iter(R + t, DW_IC(t), DW_IC(t + 1), DW_IC(6), DW_IC(-1), std::true_type{})
    DW_HEAD(0); DW_HEAD(1); DW_HEAD(2); DW_HEAD(3); DW_HEAD(4); DW_HEAD(5); DW_HEAD(6);
    for (int r0 = 7; r0 < R; r0 += 7) {
        DW_BODY(0); DW_BODY(1); DW_BODY(2); DW_BODY(3); DW_BODY(4); DW_BODY(5); DW_BODY(6);
    }
    DW_TAIL(0); DW_TAIL(1); DW_TAIL(2); DW_TAIL(3); DW_TAIL(4); DW_TAIL(5);
#undef DW_HEAD
#undef DW_BODY
#undef DW_TAIL
#undef DW_IC
    dw_wait_vm<0>();                               // no DMA may still be landing when the LDS is handed to the next workgroup
}

template <bool RES, bool Y2>
int launch_dwconv_rs_t(const void* x, const float* w49, const float* bias, const void* res, void* y, int B, int H, int W, int C,
                       int flip, hipStream_t s, void* y2, const float* y2scale) {
    using G = DWR<RES, Y2>;
    auto k = dwconv7_rs_kernel<RES, Y2>;
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
    if (!attr_ok) {
        ga_set_error("dwconv7: cannot reserve %d B of LDS", G::LDS);
        return GA_ERR_HIP;
    }
    const int nstrips = cdiv(W, G::TW);
    const int wpu = cdiv(nstrips * (C / 2), 64);
    // rows per segment: a multiple of 7 that divides H -- the longest march (least halo re-reading, fewest wave starts) that
    // still gives every SIMD its two waves (measured at 256 x 56 x 56 x 96: R = 56 0.101 ms, 28 0.118, 14 0.130)
    int R = GA_KNOB("DW_RS_ROWS", 0);
    if (R <= 0 || R % 7 != 0 || H % R != 0) {
        const long slots = 8L * num_cus();
        R = 7;
        for (int d = 1; d <= H / 7; ++d) {
            if ((H / 7) % d != 0) continue;
            const int r = H / d;                   // candidates from the longest down
            if (r % 7 == 0 && (long)B * d * wpu >= slots) {
                R = r;
                break;
            }
        }
    }
    const int nseg = H / R;
    const long waves = (long)B * nseg * wpu;
    const unsigned bytes = (unsigned)((long)B * H * W * C * 2);
    hipLaunchKernelGGL(k, dim3((unsigned)cdiv(waves, 4L)), dim3(256), G::LDS, s, (const bf16_t*)x, w49, bias, (const bf16_t*)res,
                       (bf16_t*)y, H, W, C, flip, (bf16_t*)y2, y2scale, R, nseg, nstrips, wpu, (int)waves, bytes);
    return ga_check_launch("ga_dwconv7");
}

int launch_dwconv_rs(const void* x, const float* w49, const float* bias, const void* res, void* y, int B, int H, int W, int C,
                     int flip, hipStream_t s, void* y2, const float* y2scale) {
#define DW_RS_GO(RES, Y2) return launch_dwconv_rs_t<RES, Y2>(x, w49, bias, res, y, B, H, W, C, flip, s, y2, y2scale)
    const bool r = res != nullptr, t = y2 != nullptr;
    if (r && t) DW_RS_GO(true, true);
    if (r) DW_RS_GO(true, false);
    if (t) DW_RS_GO(false, true);
    DW_RS_GO(false, false);
#undef DW_RS_GO
}

// ------------------------------------------------------------------------------------------------
// bf16 backward-weight, the same register-sliding march: dW[ky][kx][c] = sum dy[oy][ox][c] x[oy + ky - 3][ox + kx - 3][c].
//   lane = channel pair of a strip of 4 output columns; it keeps its 49 x 2 partial sums (and the bias sums) in registers
//   over EVERY unit (image, row segment) it walks; per input row: the row's 10 dwords and the dy row that starts there come
//   from the LDS-DMA ring (4 instructions per row), dy rows of the 7 output rows in flight stay unpacked in registers
//   (slot q % 7), 196 v_pk_fma_f32 per row -- the forward kernel's arithmetic with the roles of weights and accumulators
//   swapped, no stores in the loop.
//   Grid: persistent, workgroup b serves wave position wi = b % wpu of the units (all four waves of a workgroup share the
//   lane -> (strip, channel pair) map, each walks its own units); at the end the four waves meet in LDS and the
//   workgroup writes ONE partial of 100 x 64 floats; dwconv7_wgrad_rs_reduce adds the partials of all workgroups and
//   strips in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
struct DWW {
    static constexpr int TW = 4, NC = 10;
    static constexpr int P = 3, D = 4, GC = 4;                // rows ahead, ring depth, DMA operations per row (3 x + 1 dy)
    static constexpr int ROWB = GC * 1024, WAVE_LDS = D * ROWB, LDS = 4 * WAVE_LDS;     // 64 KiB per workgroup
    static constexpr int NV = 100;                            // values per lane: 49 taps x 2 channels + 2 bias sums
};

__global__ __launch_bounds__(256, 2) void dwconv7_wgrad_rs_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  float* __restrict__ part, int H, int W, int C, int R, int nseg,
                                                                  int nstrips, int wpu, int nunits, unsigned bytes) {
    using G = DWW;
    constexpr int TW = G::TW, NC = G::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = blockIdx.x % wpu, kb = blockIdx.x / wpu, nbw = gridDim.x / wpu;      // gridDim.x is a multiple of wpu
    const int CP = C >> 1;

    unsigned char* wsm = smem + wv * G::WAVE_LDS;
    const unsigned wring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wv * G::WAVE_LDS;
    const unsigned* ring = reinterpret_cast<const unsigned*>(wsm) + lane;
    const dw_u4 rx = dw_make_rsrc(x, bytes), rdy = dw_make_rsrc(dy, bytes);
    unsigned cx[3], cd;
    {
        const int qitem = wi * 64 + 4 * (lane & 15);
        const int qstrip = qitem / CP;
        const int qcp = qitem - qstrip * CP;
        const bool qa = qstrip < nstrips;
        auto off = [&](int col) {
            return (qa && (unsigned)col < (unsigned)W) ? (unsigned)(col * C + 2 * qcp) * 2u : 0x80000000u;
        };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = 4 * k + (lane >> 4);
            cx[k] = (j < NC ? off(qstrip * TW + j - 3) : 0x80000000u) - 1024u * k;
        }
        cd = off(qstrip * TW + (lane >> 4)) - 3072u;
    }
    dw_f2 acc[49], bs = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 49; ++t) acc[t] = dw_f2{0.f, 0.f};

    const unsigned rowb = (unsigned)W * C * 2u;
    const int nr = R + 6;
    for (int unit = kb * 4 + wv; unit < nunits; unit += 4 * nbw) {
        const int img = unit / nseg, seg = unit - img * nseg;
        const int oy0 = seg * R, iy0 = oy0 - 3;
        const unsigned img_base = (unsigned)img * H * rowb;
        auto issue = [&](int r) {                  // row group r: input row iy0 + r and dy row oy0 + r
            const int iy = iy0 + r;
            const bool v = r < nr && (unsigned)iy < (unsigned)H;
            dw_u4 rs = rx;
            rs[2] = v ? bytes : 0u;
            const unsigned dst = wring + (unsigned)(r & (G::D - 1)) * G::ROWB;
            dw_dma_x(rs, v ? img_base + iy * rowb : 0u, dst, cx[0], cx[1], cx[2]);
            const bool vd = r < R;
            dw_u4 rq = rdy;
            rq[2] = vd ? bytes : 0u;
            dw_dma_res(rq, vd ? img_base + (oy0 + r) * rowb : 0u, dst, cd);
        };
        unsigned raw[NC] = {}, rawd[TW] = {};
        auto read_row = [&](int r) {
            const unsigned* rowp = ring + (r & (G::D - 1)) * (G::ROWB / 4);
#pragma unroll
            for (int j = 0; j < NC; ++j) raw[j] = rowp[j * 64];
#pragma unroll
            for (int o = 0; o < TW; ++o) rawd[o] = rowp[(12 + o) * 64];
        };
        dw_f2 in[NC], g[7][TW];                    // input row r; dy rows of the 7 output rows in flight (slot q % 7)
#pragma unroll
        for (int q = 0; q < 7; ++q)
#pragma unroll
            for (int o = 0; o < TW; ++o) g[q][o] = dw_f2{0.f, 0.f};
        unsigned gd[TW];
        auto unpack_row = [&]() {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                in[j].x = __uint_as_float(raw[j] << 16);
                in[j].y = __uint_as_float(raw[j] & 0xffff0000u);
            }
#pragma unroll
            for (int o = 0; o < TW; ++o) gd[o] = rawd[o];
        };
#pragma unroll
        for (int r = 0; r <= G::P; ++r) issue(r);
        dw_wait_vm<G::P * G::GC>();
        read_row(0);
        unpack_row();

        auto row_iter = [&](int r, auto rr_c, auto klo_c, auto khi_c, auto edge_c) {
            constexpr int RR = decltype(rr_c)::value, KLO = decltype(klo_c)::value, KHI = decltype(khi_c)::value;
            constexpr bool EDGE = decltype(edge_c)::value;
            dw_wait_vm<(G::P - 1) * G::GC>();                  // ring row r + 1 has landed (the loop has no other memory operation)
            read_row(r + 1);
            issue(r + 1 + G::P);
#pragma unroll
            for (int o = 0; o < TW; ++o) {                      // dy row q = r enters slot r % 7 (zeros beyond the segment)
                g[RR][o].x = __uint_as_float(gd[o] << 16);
                g[RR][o].y = __uint_as_float(gd[o] & 0xffff0000u);
                if (KLO == 0) bs += g[RR][o];
            }
            if (!EDGE || (unsigned)(iy0 + r) < (unsigned)H) {
#pragma unroll
                for (int ky = KLO; ky <= KHI; ++ky) {
                    const int slot = (RR - ky + 7) % 7;
#pragma unroll
                    for (int o = 0; o < TW; ++o)               // column outermost: consecutive FMAs hit different accumulators
#pragma unroll
                        for (int kx = 0; kx < 7; ++kx)
                            acc[ky * 7 + kx] = __builtin_elementwise_fma(g[slot][o], in[o + kx], acc[ky * 7 + kx]);
                }
            }
            unpack_row();
        };
#define DW_IC(v) std::integral_constant<int, (v)>{}
#define DW_HEAD(r) row_iter(r, DW_IC(r), DW_IC(0), DW_IC(r), std::true_type{})
#define DW_BODY(rr) row_iter(r0 + rr, DW_IC(rr), DW_IC(0), DW_IC(6), std::false_type{})
#define DW_TAIL(t) row_iter(R + t, DW_IC(t), DW_IC(t + 1), DW_IC(6), std::true_type{})
        DW_HEAD(0); DW_HEAD(1); DW_HEAD(2); DW_HEAD(3); DW_HEAD(4); DW_HEAD(5); DW_HEAD(6);
        for (int r0 = 7; r0 < R; r0 += 7) {
            DW_BODY(0); DW_BODY(1); DW_BODY(2); DW_BODY(3); DW_BODY(4); DW_BODY(5); DW_BODY(6);
        }
        DW_TAIL(0); DW_TAIL(1); DW_TAIL(2); DW_TAIL(3); DW_TAIL(4); DW_TAIL(5);
#undef DW_HEAD
#undef DW_BODY
#undef DW_TAIL
#undef DW_IC
        dw_wait_vm<0>();                           // the ring is re-used by the next unit / by the reduction below
    }
    // the four waves meet in LDS ([value][lane] floats, 25 KiB per wave image): 2 + 3 -> 0 + 1, then 1 -> 0
    float* red = reinterpret_cast<float*>(smem);
    auto dump = [&](int slot) {
        float* p = red + slot * (G::NV * 64) + lane;
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            p[(2 * t) * 64] = acc[t].x;
            p[(2 * t + 1) * 64] = acc[t].y;
        }
        p[98 * 64] = bs.x;
        p[99 * 64] = bs.y;
    };
    auto add = [&](int slot) {
        const float* p = red + slot * (G::NV * 64) + lane;
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            acc[t].x += p[(2 * t) * 64];
            acc[t].y += p[(2 * t + 1) * 64];
        }
        bs.x += p[98 * 64];
        bs.y += p[99 * 64];
    };
    __syncthreads();
    if (wv >= 2) dump(wv - 2);
    __syncthreads();
    if (wv < 2) add(wv);
    __syncthreads();
    if (wv == 1) dump(0);
    __syncthreads();
    if (wv == 0) {
        add(0);
        float* pp = part + (long)blockIdx.x * (G::NV * 64) + lane;
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            pp[(2 * t) * 64] = acc[t].x;
            pp[(2 * t + 1) * 64] = acc[t].y;
        }
        pp[98 * 64] = bs.x;
        pp[99 * 64] = bs.y;
    }
}

// dw49[tap][c] += sum over strips and workgroups of the partials; dbias[c] likewise.  workgroup = (64 channel pairs, value v of
// the 100) x 16 groups of workgroups; fixed summation order (deterministic).  (4 groups: 20 us for the 13 MB of a 56 x 56 x 96
// launch -- 168 dependent-latency loads per thread)
__global__ __launch_bounds__(1024) void dwconv7_wgrad_rs_reduce(const float* __restrict__ part, int nbw, int wpu, int nstrips, int C,
                                                                float* __restrict__ dw49, float* __restrict__ dbias) {
    constexpr int NG = 16;
    __shared__ float red[NG][64];
    const int CP = C >> 1;
    const int cpl = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int cp = blockIdx.x * 64 + cpl, v = blockIdx.y;
    float sum = 0.f;
    if (cp < CP) {
        const int per = (nbw + NG - 1) / NG, k0 = kg * per, k1 = min(nbw, k0 + per);
        for (int st = 0; st < nstrips; ++st) {
            const int item = st * CP + cp, wi = item >> 6, l = item & 63;
            const float* p = part + (long)wi * (DWW::NV * 64) + v * 64 + l;
#pragma unroll 4
            for (int k = k0; k < k1; ++k) sum += p[(long)k * wpu * (DWW::NV * 64)];
        }
    }
    red[kg][cpl] = sum;
    __syncthreads();
    if (kg == 0 && cp < CP) {
        sum = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) sum += red[g][cpl];
        const int tap = v >> 1, e = v & 1;
        if (tap < 49) dw49[(long)tap * C + 2 * cp + e] += sum;
        else if (dbias) dbias[2 * cp + e] += sum;
    }
}

// geometry of the register-sliding backward-weight form: rows per segment, workgroups (a multiple of wpu), or 0 if it does not apply
inline bool dww_rs_geometry(int B, int H, int W, int C, int* R_out, int* wpu_out, int* nb_out) {
    if (C % 8 != 0 || H % 7 != 0 || (long)B * H * W * C * 2 >= (1L << 31)) return false;
    const int nstrips = cdiv(W, DWW::TW);
    const int wpu = cdiv(nstrips * (C / 2), 64);
    const int nbw = std::max(1, 2 * num_cus() / wpu);          // two workgroups per CU
    // rows per segment: the longest march that still gives each of the 4 * nbw waves of a lane map about two units
    int R = GA_KNOB("DWW_RS_ROWS", 0);
    if (R <= 0 || R % 7 != 0 || H % R != 0) {
        R = 7;
        for (int d = 1; d <= H / 7; ++d) {
            if ((H / 7) % d != 0) continue;
            const int r = H / d;
            if (r % 7 == 0 && (long)B * d >= 8L * nbw) {
                R = r;
                break;
            }
        }
    }
    *R_out = R;
    *wpu_out = wpu;
    *nb_out = nbw * wpu;
    return true;
}

int launch_dwconv_wgrad_rs(const void* dy, const void* x, float* dw49, float* dbias, int B, int H, int W, int C, float* part,
                           hipStream_t s) {
    int R, wpu, nb;
    dww_rs_geometry(B, H, W, C, &R, &wpu, &nb);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv7_wgrad_rs_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, DWW::LDS) == hipSuccess;
    if (!attr_ok) {
        ga_set_error("dwconv7_wgrad: cannot reserve %d B of LDS", DWW::LDS);
        return GA_ERR_HIP;
    }
    const int nseg = H / R, nstrips = cdiv(W, DWW::TW);
    const unsigned bytes = (unsigned)((long)B * H * W * C * 2);
    hipLaunchKernelGGL(dwconv7_wgrad_rs_kernel, dim3(nb), dim3(256), DWW::LDS, s, (const bf16_t*)dy, (const bf16_t*)x, part, H, W, C, R,
                       nseg, nstrips, wpu, B * nseg, bytes);
    hipLaunchKernelGGL(dwconv7_wgrad_rs_reduce, dim3(cdiv(C / 2, 64), DWW::NV), dim3(1024), 0, s, part, nb / wpu, wpu, nstrips, C, dw49,
                       dbias);
    return ga_check_launch("ga_dwconv7_bwd_weight");
}

template <typename T>
int launch_dwconv(const void* x, const float* w49, const float* bias, const void* res, void* y, int B, int H, int W,
                  int C, int flip, hipStream_t s, void* y2 = nullptr, const float* y2scale = nullptr) {
    if constexpr (sizeof(T) == 2) {
        // register-sliding form (default): C a multiple of 8, H a multiple of 7, any W, tensors below 2 GiB (32-bit buffer offsets)
        if (C % 8 == 0 && H % 7 == 0 && (long)B * H * W * C * 2 < (1L << 31) && (long)B * (H / 7) * cdiv(W, 4) * C < (1L << 31) &&
            GA_KNOB("DW_RS", 1))
            return launch_dwconv_rs(x, w49, bias, res, y, B, H, W, C, flip, s, y2, y2scale);
        if (C % 8 == 0) {
            using G14 = DWF<14, 14>;
            using G7 = DWF<7, 7>;
            auto k14 = dwconv7_dot2_kernel<14, 14>;
            auto k7 = dwconv7_dot2_kernel<7, 7>;
            constexpr int lds14 = G14::LDS;
            static const bool attr_ok =
                hipFuncSetAttribute(reinterpret_cast<const void*>(k14), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds14) == hipSuccess;
            if (!attr_ok) {
                ga_set_error("dwconv7: cannot reserve %d B of LDS", lds14);
                return GA_ERR_HIP;
            }
            const int sl = cdiv(C, 64);
            const bool big = H % 14 == 0 && W % 14 == 0;
            const long ntiles = big ? (long)B * (H / 14) * (W / 14) : (long)B * cdiv(H, 7) * cdiv(W, 7);
            const int gx = (int)std::min<long>(ntiles, std::max(1, (big ? 2 : 4) * num_cus() / sl));
            const int use_mfma = GA_KNOB("DW_MFMA", 1);     // 0: never, 1: heuristic (default), 2: every 14 x 14-tiled launch
            // the MFMA form pays a long prologue (Toeplitz fragments) per workgroup: ahead of the dot2 form only when
            // a workgroup walks many tiles (56 x 56 maps, also as half batches: 0.169 vs 0.192 ms; 28 x 28 and 14 x 14:
            // 5-10 % behind)
            const bool many = H * W >= 56 * 56 && ntiles * cdiv(C, DWM::CS) >= 16L * num_cus();
            if (big && (use_mfma == 2 || (use_mfma == 1 && many))) {
                const int slm = cdiv(C, DWM::CS);
                // grid.x a multiple of 8: workgroup (x, y) then sits on XCD x % 8 for every slice y, so the 32-channel slices of
                // one tile (64 of the 128 bytes of every line each) share one L2 instead of fetching the line once per slice
                const int xcd8 = GA_KNOB("DW_XCD8", 1);
                int gxm = (int)std::min<long>(ntiles, std::max(1, num_cus() / slm));
                if (xcd8 && gxm >= 16) gxm = gxm / 8 * 8;
                hipLaunchKernelGGL(dwconv7_mfma_kernel, dim3(gxm, slm), dim3(DWM::NT), DWM::LDS, s, (const bf16_t*)x, w49, bias,
                                   (const bf16_t*)res, (bf16_t*)y, B, H, W, C, flip, (bf16_t*)y2, y2scale);
            } else if (big)
                hipLaunchKernelGGL(k14, dim3(gx, sl), dim3(G14::NT), G14::LDS, s, (const bf16_t*)x, w49, bias,
                                   (const bf16_t*)res, (bf16_t*)y, B, H, W, C, flip, (bf16_t*)y2, y2scale);
            else
                hipLaunchKernelGGL(k7, dim3(gx, sl), dim3(G7::NT), G7::LDS, s, (const bf16_t*)x, w49, bias,
                                   (const bf16_t*)res, (bf16_t*)y, B, H, W, C, flip, (bf16_t*)y2, y2scale);
            return ga_check_launch("ga_dwconv7");
        }
    }
    const int slices = cdiv(C, kCSF);
    if (H % 14 == 0 && W % 14 == 0) {
        constexpr int TH = 14, TW = 14, NT = 256;
        const size_t lds = (TH + 6) * (TW + 6) * kCSF * sizeof(T) + 49 * kCSF * 4;
        auto k = dwconv7_kernel<T, TH, TW, NT>;
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                ga_set_error("dwconv7: cannot reserve %zu B of LDS", lds);
                return GA_ERR_HIP;
            }
            once = true;
        }
        dim3 grid(B * (H / TH) * (W / TW) * slices);
        hipLaunchKernelGGL(k, grid, dim3(NT), lds, s, (const T*)x, w49, bias, (const T*)res, (T*)y, H, W, C, flip, (T*)y2, y2scale);
    } else {
        constexpr int TH = 7, TW = 7, NT = 128;
        const size_t lds = (TH + 6) * (TW + 6) * kCSF * sizeof(T) + 49 * kCSF * 4;
        dim3 grid(B * cdiv(H, TH) * cdiv(W, TW) * slices);
        hipLaunchKernelGGL((dwconv7_kernel<T, TH, TW, NT>), grid, dim3(NT), lds, s, (const T*)x, w49, bias,
                           (const T*)res, (T*)y, H, W, C, flip, (T*)y2, y2scale);
    }
    return ga_check_launch("ga_dwconv7");
}

void dwconv_wgrad_geometry(int B, int H, int W, int C, bool bf16, int* gx_out, int* nparts_out) {
    const int slices = cdiv(C, kCSW);
    const bool big = H % 14 == 0 && W % 14 == 0;
    const long ntiles = big ? (long)B * (H / 14) * (W / 14) : (long)B * cdiv(H, 7) * cdiv(W, 7);
    const bool dot2 = bf16 && C % 8 == 0;   // dot2 form: 89 KB of LDS at 14x14 -> one workgroup per CU
    const int gx = (int)std::min<long>(ntiles, std::max(1, (big ? (dot2 ? 1 : 2) : 4) * num_cus() / slices));
    *gx_out = gx;
    *nparts_out = gx * (big ? 2 : 1);          // one partial per (workgroup, group of 7 output rows)
}

// bytes of partial sums the backward-weight launch of this geometry needs (the larger of the forms that may serve it)
size_t dwconv_wgrad_ws_bytes(int B, int H, int W, int C, bool bf16) {
    int gx, nparts;
    dwconv_wgrad_geometry(B, H, W, C, bf16, &gx, &nparts);
    size_t need = (size_t)nparts * 50 * C * sizeof(float);
    int R, wpu, nb;
    if (bf16 && dww_rs_geometry(B, H, W, C, &R, &wpu, &nb)) need = std::max(need, (size_t)nb * DWW::NV * 64 * sizeof(float));
    return need;
}

template <typename T>
int launch_dwconv_wgrad(const void* dy, const void* x, float* dw49, float* dbias, int B, int H, int W, int C,
                        float* part, size_t ws_bytes, hipStream_t s) {
    const int slices = cdiv(C, kCSW);
    const bool big = H % 14 == 0 && W % 14 == 0;
    const bool dot2 = sizeof(T) == 2 && C % 8 == 0;
    int gx, nparts;
    dwconv_wgrad_geometry(B, H, W, C, sizeof(T) == 2, &gx, &nparts);
    const size_t need = dwconv_wgrad_ws_bytes(B, H, W, C, sizeof(T) == 2);
    if (!part || ws_bytes < need) {
        ga_set_error("ga_dwconv7_bwd_weight: needs %zu B of caller-provided workspace (ga_dwconv7_bwd_weight_workspace), got %zu",
                     need, part ? ws_bytes : (size_t)0);
        return GA_ERR_BAD_ARG;
    }
    if constexpr (sizeof(T) == 2) {
        int rsR, rsW, rsN;
        // register-sliding form: 0 never, 1 (default) on 28 x 28 maps and larger (256 x 56 x 56 x 96: 0.134 vs 0.182 ms, 28 x 28 x 192:
        // 0.071 vs 0.078; on the smaller maps its per-unit start-up and the 13 MB of partials eat the gain: 14 x 14 0.047 vs 0.044),
        // 2 wherever it applies
        const int use_rs = GA_KNOB("DWW_RS", 1);
        if (use_rs && (use_rs == 2 || H * W >= 28 * 28) && dww_rs_geometry(B, H, W, C, &rsR, &rsW, &rsN))
            return launch_dwconv_wgrad_rs(dy, x, dw49, dbias, B, H, W, C, part, s);
        if (dot2) {
            auto k14 = dwconv7_wgrad_dot2_kernel<14, 14>;
            auto k7 = dwconv7_wgrad_dot2_kernel<7, 7>;
            constexpr int lds14 = DWD<14, 14>::LDS;
            static const bool attr_ok =
                hipFuncSetAttribute(reinterpret_cast<const void*>(k14), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds14) == hipSuccess;
            if (!attr_ok) {
                ga_set_error("dwconv7_wgrad: cannot reserve %d B of LDS", lds14);
                return GA_ERR_HIP;
            }
            using G14 = DWD<14, 14>;
            using G7 = DWD<7, 7>;
            if (big)
                hipLaunchKernelGGL(k14, dim3(gx, slices), dim3(G14::NT), G14::LDS, s, (const bf16_t*)dy,
                                   (const bf16_t*)x, part, B, H, W, C);
            else
                hipLaunchKernelGGL(k7, dim3(gx, slices), dim3(G7::NT), G7::LDS, s, (const bf16_t*)dy,
                                   (const bf16_t*)x, part, B, H, W, C);
            hipLaunchKernelGGL(dwconv7_wgrad_reduce, dim3(cdiv(50 * C, 64)), dim3(1024), 0, s, part, gx, C, dw49, dbias);
            return ga_check_launch("ga_dwconv7_bwd_weight");
        }
    }
    if (big) {
        constexpr int TH = 14, TW = 14, NT = 256;
        const size_t lds = ((TH + 6) * (TW + 6) + TH * TW) * kCSW * sizeof(T);
        auto k = dwconv7_wgrad_kernel<T, TH, TW, NT>;
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                ga_set_error("dwconv7_wgrad: cannot reserve %zu B of LDS", lds);
                return GA_ERR_HIP;
            }
            once = true;
        }
        hipLaunchKernelGGL(k, dim3(gx, slices), dim3(NT), lds, s, (const T*)dy, (const T*)x, part, B, H, W, C);
    } else {
        constexpr int TH = 7, TW = 7, NT = 128;
        const size_t lds = ((TH + 6) * (TW + 6) + TH * TW) * kCSW * sizeof(T);
        hipLaunchKernelGGL((dwconv7_wgrad_kernel<T, TH, TW, NT>), dim3(gx, slices), dim3(NT), lds, s, (const T*)dy,
                           (const T*)x, part, B, H, W, C);
    }
    hipLaunchKernelGGL(dwconv7_wgrad_reduce, dim3(cdiv(50 * C, 64)), dim3(1024), 0, s, part, nparts, C, dw49, dbias);
    return ga_check_launch("ga_dwconv7_bwd_weight");
}

}  // namespace

extern "C" int ga_dwconv7_fwd(const void* x, const float* w49, const float* bias, void* y, int B, int H, int W, int C,
                              int dtype, ga_stream_t stream) {
    GA_REQUIRE(x && w49 && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ga_dwconv7_fwd: bad args (C%%4)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? launch_dwconv<bf16_t>(x, w49, bias, nullptr, y, B, H, W, C, 0, s)
                            : launch_dwconv<float>(x, w49, bias, nullptr, y, B, H, W, C, 0, s);
}

extern "C" int ga_dwconv7_bwd_data(const void* dy, const float* w49, const void* res, void* dx, int B, int H, int W,
                                   int C, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dy && w49 && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ga_dwconv7_bwd_data: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? launch_dwconv<bf16_t>(dy, w49, nullptr, res, dx, B, H, W, C, 1, s)
                            : launch_dwconv<float>(dy, w49, nullptr, res, dx, B, H, W, C, 1, s);
}

extern "C" int ga_dwconv7_bwd_data2(const void* dy, const float* w49, const void* res, void* dx, void* dx2,
                                    const float* scale2, int B, int H, int W, int C, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dy && w49 && res && dx && dx2 && scale2 && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0,
               "ga_dwconv7_bwd_data2: bad args (the second output needs the residual form)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? launch_dwconv<bf16_t>(dy, w49, nullptr, res, dx, B, H, W, C, 1, s, dx2, scale2)
                            : launch_dwconv<float>(dy, w49, nullptr, res, dx, B, H, W, C, 1, s, dx2, scale2);
}

extern "C" size_t ga_dwconv7_bwd_weight_workspace(int B, int H, int W, int C, int dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return dwconv_wgrad_ws_bytes(B, H, W, C, dtype == GA_BF16);
}

extern "C" int ga_dwconv7_bwd_weight(const void* dy, const void* x, float* dw49, float* dbias, int B, int H, int W,
                                     int C, int dtype, void* workspace, size_t ws_bytes, ga_stream_t stream) {
    GA_REQUIRE(dy && x && dw49 && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ga_dwconv7_bwd_weight: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == GA_BF16 ? launch_dwconv_wgrad<bf16_t>(dy, x, dw49, dbias, B, H, W, C, (float*)workspace, ws_bytes, s)
                            : launch_dwconv_wgrad<float>(dy, x, dw49, dbias, B, H, W, C, (float*)workspace, ws_bytes, s);
}
