// Pooling-transformer pieces (/root/reference/MAP/models/map_pit.py) outside the ViT block:
//   conv_head_pooling (:58-68): depthwise 3x3 / stride 2 / pad 1 convolution with a channel multiplier (groups = Cin,
//     Cout = mult * Cin; PiT: mult = 2) on NHWC token maps -- forward, data gradient, weight / bias gradient;
//   general bilinear resize (align_corners = False, no antialias) of an NHWC map into a column slice of the MultiScale concat
//     buffer (map.py:322-333: 27 x 27 -> 14 x 14 is not an integer factor) and its backward.
// The maps are small (<= 27 x 27 x 576 per image): plain HBM-bound kernels, 8 channels per thread.
#include <algorithm>
#include <stdlib.h>
#include "common.h"

namespace {

// y[b, oy, ox, co] = bias[co] + sum_{ky,kx} w[co][ky*3+kx] * x[b, 2oy-1+ky, 2ox-1+kx, co / mult]
template <typename T>
__global__ __launch_bounds__(256) void dwpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, int B, int H, int W, int Cin, int mult, int Ho, int Wo) {
    const int Co = Cin * mult, C8 = Co / 8;
    const long n = (long)B * Ho * Wo * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int co = (int)(i % C8) * 8;
        const long pix = i / C8;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias[co + j];
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const T* xp = x + (((long)b * H + iy) * W + ix) * Cin;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(w[(co + j) * 9 + ky * 3 + kx], elt<T>::ld(xp + (co + j) / mult), acc[j]);
            }
        }
        store8(y + pix * Co + co, acc);
    }
}

// dx[b, iy, ix, ci] = sum over (oy, ky): 2oy-1+ky = iy, (ox, kx) likewise, m < mult:  w[ci*mult+m][ky*3+kx] * dy[b, oy, ox, ci*mult+m]
template <typename T>
__global__ __launch_bounds__(256) void dwpool_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int B,
                                                              int H, int W, int Cin, int mult, int Ho, int Wo) {
    const int Co = Cin * mult, C8 = Cin / 8;
    const long n = (long)B * H * W * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % C8) * 8;
        const long pix = i / C8;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < 3; ++ky) {
            const int t = iy + 1 - ky;
            if (t < 0 || (t & 1)) continue;
            const int oy = t >> 1;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int s = ix + 1 - kx;
                if (s < 0 || (s & 1)) continue;
                const int ox = s >> 1;
                if (ox >= Wo) continue;
                const T* gp = dy + (((long)b * Ho + oy) * Wo + ox) * Co;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    for (int m = 0; m < mult; ++m) {
                        const int co = (ci + j) * mult + m;
                        acc[j] = fmaf(w[co * 9 + ky * 3 + kx], elt<T>::ld(gp + co), acc[j]);
                    }
            }
        }
        store8(dx + pix * Cin + ci, acc);
    }
}

// dw[co][tap] += sum dy[b,oy,ox,co] * x[b, 2oy-1+ky, 2ox-1+kx, co/mult];  db[co] += sum dy.  thread = 8 output channels x a strided
// set of output pixels; per-thread partial sums, then fp32 atomics (gridDim.y pixel groups x Co/8 chunks)
template <typename T>
__global__ __launch_bounds__(256) void dwpool_bwd_weight_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw,
                                                                float* __restrict__ db, int B, int H, int W, int Cin, int mult, int Ho, int Wo) {
    const int Co = Cin * mult, C8 = Co / 8;
    const int chunk = (blockIdx.x * 256 + threadIdx.x) % C8, pg = (blockIdx.x * 256 + threadIdx.x) / C8;
    const int npg = (gridDim.x * 256) / C8;
    if (pg >= npg) return;
    const int co = chunk * 8;
    float acc[10][8];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    const long npix = (long)B * Ho * Wo;
    for (long pix = pg; pix < npix; pix += npg) {
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
        float g[8];
        load8(dy + pix * Co + co, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[9][j] += g[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const T* xp = x + (((long)b * H + iy) * W + ix) * Cin;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] = fmaf(g[j], elt<T>::ld(xp + (co + j) / mult), acc[ky * 3 + kx][j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(dw + (co + j) * 9 + k, acc[k][j]);
        atomicAdd(db + co + j, acc[9][j]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// channel multiplier 1 / 2 (PiT: 2): weights staged once per workgroup in LDS as [tap][Cout] (+ bias row), 16-byte / 8-byte
// vector loads of the input channels an 8-channel output chunk needs
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_w(float* wl, const float* __restrict__ w, const float* __restrict__ bias, int Co) {
    for (int i = threadIdx.x; i < Co * 9; i += 256) wl[(i % 9) * Co + i / 9] = w[i];
    if (bias)
        for (int i = threadIdx.x; i < Co; i += 256) wl[9 * Co + i] = bias[i];
    __syncthreads();
}

// the 8 output channels co .. co+7 read input channels co/MULT .. : xv[j] = x[(co + j) / MULT]
template <typename T, int MULT>
__device__ __forceinline__ void load_in8(const T* xp, int co, float xv[8]) {
    if constexpr (MULT == 1) {
        load8(xp + co, xv);
    } else {
        float h[4];
        load4(xp + co / 2, h);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = h[j >> 1];
    }
}

template <typename T, int MULT>
__global__ __launch_bounds__(256) void dwpool_fwd_v(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                    T* __restrict__ y, int B, int H, int W, int Cin, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int Co = Cin * MULT, C8 = Co / 8;
    stage_w(wl, w, bias, Co);
    const long n = (long)B * Ho * Wo * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int co = (int)(i % C8) * 8;
        const long pix = i / C8;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
        float acc[8];
        load8(wl + 9 * Co + co, acc);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    float xv[8], wv[8];
                    load_in8<T, MULT>(x + (((long)b * H + iy) * W + ix) * Cin, co, xv);
                    load8(wl + (ky * 3 + kx) * Co + co, wv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(wv[j], xv[j], acc[j]);
                }
            }
        }
        store8(y + pix * Co + co, acc);
    }
}

template <typename T, int MULT>
__global__ __launch_bounds__(256) void dwpool_bwd_data_v(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int B, int H,
                                                         int W, int Cin, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int Co = Cin * MULT, C8 = Cin / 8;
    stage_w(wl, w, nullptr, Co);
    const long n = (long)B * H * W * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % C8) * 8;
        const long pix = i / C8;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int t = iy + 1 - ky, oy = t >> 1;
            if (t < 0 || (t & 1) || oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int u = ix + 1 - kx, ox = u >> 1;
                if (u < 0 || (u & 1) || ox >= Wo) continue;
                const T* gp = dy + (((long)b * Ho + oy) * Wo + ox) * Co + ci * MULT;
                const float* wp = wl + (ky * 3 + kx) * Co + ci * MULT;
#pragma unroll
                for (int m = 0; m < MULT; ++m) {
                    float g[8], wv[8];
                    load8(gp + 8 * m, g);
                    load8(wp + 8 * m, wv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[(8 * m + j) / MULT] = fmaf(wv[j], g[j], acc[(8 * m + j) / MULT]);
                }
            }
        }
        store8(dx + pix * Cin + ci, acc);
    }
}

// a WAVE per (8-channel output chunk, pixel group): lanes = 64 output pixels, per-lane partial sums, one shuffle reduction and 80
// atomics per wave
template <typename T, int MULT>
__global__ __launch_bounds__(256) void dwpool_bwd_weight_v(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ db, int B, int H, int W, int Cin, int Ho, int Wo, int npg) {
    const int Co = Cin * MULT, C8 = Co / 8;
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int chunk = gw % C8, pg = gw / C8;
    if (pg >= npg) return;
    const int co = chunk * 8;
    float acc[10][8];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    const long npix = (long)B * Ho * Wo;
    for (long pix = (long)pg * 64 + lane; pix < npix; pix += (long)npg * 64) {
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
        float g[8];
        load8(dy + pix * Co + co, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[9][j] += g[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    float xv[8];
                    load_in8<T, MULT>(x + (((long)b * H + iy) * W + ix) * Cin, co, xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] = fmaf(g[j], xv[j], acc[ky * 3 + kx][j]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[k][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[k][j] = v;
        }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < 9; ++k) atomicAdd(dw + (co + j) * 9 + k, acc[k][j]);
            atomicAdd(db + co + j, acc[9][j]);
        }
    }
}

// bilinear source taps of output index o (PyTorch upsample_bilinear2d, align_corners = False): src = max(0, (o + .5) * in/out - .5)
__device__ __forceinline__ void bil_taps(int o, int in, int out, int& i0, int& i1, float& w1) {
    const float s = fmaxf(0.f, (o + 0.5f) * ((float)in / (float)out) - 0.5f);
    i0 = min((int)s, in - 1);
    i1 = min(i0 + 1, in - 1);
    w1 = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_concat_fwd_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int Hin, int Win, int C,
                                                                int Hout, int Wout, long ldd, int c_off) {
    const int C8 = C / 8;
    const long n = (long)B * Hout * Wout * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long pix = i / C8;
        const int ox = (int)(pix % Wout), oy = (int)((pix / Wout) % Hout), b = (int)(pix / ((long)Wout * Hout));
        int y0, y1, x0, x1;
        float wy, wx;
        bil_taps(oy, Hin, Hout, y0, y1, wy);
        bil_taps(ox, Win, Wout, x0, x1, wx);
        const T* base = src + (long)b * Hin * Win * C + c;
        float a[8], bq[8], cq[8], dq[8], v[8];
        load8(base + ((long)y0 * Win + x0) * C, a);
        load8(base + ((long)y0 * Win + x1) * C, bq);
        load8(base + ((long)y1 * Win + x0) * C, cq);
        load8(base + ((long)y1 * Win + x1) * C, dq);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (1.f - wy) * ((1.f - wx) * a[j] + wx * bq[j]) + wy * ((1.f - wx) * cq[j] + wx * dq[j]);
        store8(dst + pix * ldd + c_off + c, v);
    }
}

// gather form of the backward: every input pixel collects from the (few) output pixels whose taps touch it
template <typename T>
__global__ __launch_bounds__(256) void resize_concat_bwd_kernel(const T* __restrict__ dcat, T* __restrict__ dsrc, int B, int Hin, int Win, int C,
                                                                int Hout, int Wout, long ldd, int c_off) {
    const int C8 = C / 8;
    const long n = (long)B * Hin * Win * C8;
    const float sy = (float)Hout / (float)Hin, sx = (float)Wout / (float)Win;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long pix = i / C8;
        const int ix = (int)(pix % Win), iy = (int)((pix / Win) % Hin), b = (int)(pix / ((long)Win * Hin));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int oy_lo = max(0, (int)floorf((iy - 1) * sy) - 1), oy_hi = min(Hout - 1, (int)ceilf((iy + 2) * sy) + 1);
        const int ox_lo = max(0, (int)floorf((ix - 1) * sx) - 1), ox_hi = min(Wout - 1, (int)ceilf((ix + 2) * sx) + 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float wy;
            bil_taps(oy, Hin, Hout, y0, y1, wy);
            const float fy = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
            if (fy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float wx;
                bil_taps(ox, Win, Wout, x0, x1, wx);
                const float fx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
                if (fx == 0.f) continue;
                float g[8];
                load8(dcat + (((long)b * Hout + oy) * Wout + ox) * ldd + c_off + c, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(fy * fx, g[j], acc[j]);
            }
        }
        store8(dsrc + pix * C + c, acc);
    }
}

// x0[b][p] = tok[b][p] + pos[p]  (pos fp32 [Np][C]: the NCHW pos_embed of map_pit.py:190-191 transposed once per step)
template <typename T>
__global__ __launch_bounds__(256) void pos_add_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ pos, T* __restrict__ x0, int B,
                                                          int Np, int C) {
    const int C8 = C / 8;
    const long n = (long)B * Np * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long row = i / C8;
        float v[8], pv[8];
        load8(tok + row * C + c, v);
        load8(pos + (row % Np) * C + c, pv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += pv[j];
        store8(x0 + row * C + c, v);
    }
}

// dpos[p][c] = sum_b dx0[b][p][c]   (overwrites; fixed summation order)
template <typename T>
__global__ __launch_bounds__(256) void pos_add_bwd_kernel(const T* __restrict__ dx0, float* __restrict__ dpos, int B, int Np, int C) {
    const int C8 = C / 8;
    const long n = (long)Np * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long p = i / C8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            float v[8];
            load8(dx0 + ((long)b * Np + p) * C + c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        store8(dpos + p * C + c, acc);
    }
}

// dst[b][p][:] = scale * src[b][:]   (gradient of x.mean([-2, -1]): every pixel of image b receives dsrc[b] / HW)
template <typename T>
__global__ __launch_bounds__(256) void rows_bcast_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int HW, int C, float scale) {
    const int C8 = C / 8;
    const long n = (long)B * HW * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long row = i / C8;
        float v[8];
        load8(src + (row / HW) * C + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= scale;
        store8(dst + row * C + c, v);
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool fast_path() {       // GAEXT_DWPOOL_SIMPLE=1: the plain any-multiplier kernels (diagnostics / tests)
    return !GA_KNOB("DWPOOL_SIMPLE", 0);
}
int grid_for(long n) { return (int)std::max<long>(1, std::min<long>(8192, (n + 255) / 256)); }

}  // namespace

#define PIT_DISPATCH_V(dtype, KERNEL, MULT, grid, lds, s, ...)                                                   \
    do {                                                                                                         \
        if ((dtype) == GA_BF16) { using T = bf16_t; hipLaunchKernelGGL((KERNEL<T, MULT>), dim3(grid), dim3(256), lds, s, __VA_ARGS__); } \
        else { using T = float; hipLaunchKernelGGL((KERNEL<T, MULT>), dim3(grid), dim3(256), lds, s, __VA_ARGS__); }                     \
    } while (0)

#define PIT_DISPATCH(dtype, KERNEL, grid, s, ...)                                                              \
    do {                                                                                                       \
        if ((dtype) == GA_BF16) { using T = bf16_t; hipLaunchKernelGGL(KERNEL<T>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); } \
        else { using T = float; hipLaunchKernelGGL(KERNEL<T>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); }     \
    } while (0)

extern "C" int ga_dwpool_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cin, int mult, int dtype,
                             ga_stream_t stream) {
    GA_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0 && Cin > 0 && mult >= 1 && (Cin * mult) % 8 == 0 && aligned16(y),
               "ga_dwpool_fwd: bad args (Cout must be a multiple of 8)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n8 = (long)B * Ho * Wo * Cin * mult / 8;
    const size_t lds = (size_t)Cin * mult * 10 * sizeof(float);
    if ((mult == 1 || mult == 2) && Cin % 8 == 0 && lds <= 64 * 1024 && aligned16(x) && fast_path()) {
        const int grid = (int)std::max<long>(1, std::min<long>(2048, (n8 + 255) / 256));
        if (mult == 1) PIT_DISPATCH_V(dtype, dwpool_fwd_v, 1, grid, lds, s, (const T*)x, w, bias, (T*)y, B, H, W, Cin, Ho, Wo);
        else PIT_DISPATCH_V(dtype, dwpool_fwd_v, 2, grid, lds, s, (const T*)x, w, bias, (T*)y, B, H, W, Cin, Ho, Wo);
        return ga_check_launch("ga_dwpool_fwd");
    }
    PIT_DISPATCH(dtype, dwpool_fwd_kernel, grid_for(n8), s, (const T*)x, w, bias, (T*)y, B, H, W, Cin, mult, Ho, Wo);
    return ga_check_launch("ga_dwpool_fwd");
}

extern "C" int ga_dwpool_bwd_data(const void* dy, const float* w, void* dx, int B, int H, int W, int Cin, int mult, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dy && w && dx && B > 0 && Cin % 8 == 0 && mult >= 1 && aligned16(dx), "ga_dwpool_bwd_data: bad args (Cin %% 8)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n8 = (long)B * H * W * Cin / 8;
    const size_t lds = (size_t)Cin * mult * 9 * sizeof(float);
    if ((mult == 1 || mult == 2) && lds <= 64 * 1024 && aligned16(dy) && fast_path()) {
        const int grid = (int)std::max<long>(1, std::min<long>(2048, (n8 + 255) / 256));
        if (mult == 1) PIT_DISPATCH_V(dtype, dwpool_bwd_data_v, 1, grid, lds, s, (const T*)dy, w, (T*)dx, B, H, W, Cin, Ho, Wo);
        else PIT_DISPATCH_V(dtype, dwpool_bwd_data_v, 2, grid, lds, s, (const T*)dy, w, (T*)dx, B, H, W, Cin, Ho, Wo);
        return ga_check_launch("ga_dwpool_bwd_data");
    }
    PIT_DISPATCH(dtype, dwpool_bwd_data_kernel, grid_for(n8), s, (const T*)dy, w, (T*)dx, B, H, W, Cin, mult, Ho, Wo);
    return ga_check_launch("ga_dwpool_bwd_data");
}

extern "C" int ga_dwpool_bwd_weight(const void* dy, const void* x, float* dw, float* db, int B, int H, int W, int Cin, int mult, int dtype,
                                    ga_stream_t stream) {
    GA_REQUIRE(dy && x && dw && db && B > 0 && (Cin * mult) % 8 == 0 && aligned16(dy), "ga_dwpool_bwd_weight: bad args");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, C8 = Cin * mult / 8;
    // about 128 pixel groups per channel chunk, whole workgroups
    const int blocks = std::max(1, (C8 * 128 + 255) / 256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((mult == 1 || mult == 2) && Cin % 8 == 0 && aligned16(x) && fast_path()) {
        const long npix = (long)B * Ho * Wo;
        const int npg = (int)std::max<long>(1, std::min<long>((npix + 63) / 64, std::max(1, 4096 / C8)));
        const int grid = (C8 * npg + 3) / 4;
        if (mult == 1) PIT_DISPATCH_V(dtype, dwpool_bwd_weight_v, 1, grid, 0, s, (const T*)dy, (const T*)x, dw, db, B, H, W, Cin, Ho, Wo, npg);
        else PIT_DISPATCH_V(dtype, dwpool_bwd_weight_v, 2, grid, 0, s, (const T*)dy, (const T*)x, dw, db, B, H, W, Cin, Ho, Wo, npg);
        return ga_check_launch("ga_dwpool_bwd_weight");
    }
    PIT_DISPATCH(dtype, dwpool_bwd_weight_kernel, blocks, s, (const T*)dy, (const T*)x, dw, db, B, H, W, Cin, mult, Ho, Wo);
    return ga_check_launch("ga_dwpool_bwd_weight");
}

extern "C" int ga_resize_concat_fwd(const void* src, void* dst, int B, int Hin, int Win, int C, int Hout, int Wout, int64_t ldd, int c_off,
                                    int dtype, ga_stream_t stream) {
    GA_REQUIRE(src && dst && B > 0 && C % 8 == 0 && c_off % 8 == 0 && ldd % 8 == 0 && aligned16(src) && aligned16(dst), "ga_resize_concat_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PIT_DISPATCH(dtype, resize_concat_fwd_kernel, grid_for((long)B * Hout * Wout * C / 8), s, (const T*)src, (T*)dst, B, Hin, Win, C, Hout, Wout,
                 (long)ldd, c_off);
    return ga_check_launch("ga_resize_concat_fwd");
}

extern "C" int ga_resize_concat_bwd(const void* dcat, void* dsrc, int B, int Hin, int Win, int C, int Hout, int Wout, int64_t ldd, int c_off,
                                    int dtype, ga_stream_t stream) {
    GA_REQUIRE(dcat && dsrc && B > 0 && C % 8 == 0 && c_off % 8 == 0 && ldd % 8 == 0 && aligned16(dcat) && aligned16(dsrc), "ga_resize_concat_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PIT_DISPATCH(dtype, resize_concat_bwd_kernel, grid_for((long)B * Hin * Win * C / 8), s, (const T*)dcat, (T*)dsrc, B, Hin, Win, C, Hout, Wout,
                 (long)ldd, c_off);
    return ga_check_launch("ga_resize_concat_bwd");
}

extern "C" int ga_pos_add_fwd(const void* tok, const float* pos, void* x0, int B, int Np, int C, int dtype, ga_stream_t stream) {
    GA_REQUIRE(tok && pos && x0 && B > 0 && Np > 0 && C > 0 && C % 8 == 0 && aligned16(tok) && aligned16(pos) && aligned16(x0),
               "ga_pos_add_fwd: bad args (C %% 8, 16-byte alignment)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PIT_DISPATCH(dtype, pos_add_fwd_kernel, grid_for((long)B * Np * C / 8), s, (const T*)tok, pos, (T*)x0, B, Np, C);
    return ga_check_launch("ga_pos_add_fwd");
}

extern "C" int ga_pos_add_bwd(const void* dx0, float* dpos, int B, int Np, int C, int dtype, ga_stream_t stream) {
    GA_REQUIRE(dx0 && dpos && B > 0 && Np > 0 && C > 0 && C % 8 == 0 && aligned16(dx0) && aligned16(dpos),
               "ga_pos_add_bwd: bad args (C %% 8, 16-byte alignment)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PIT_DISPATCH(dtype, pos_add_bwd_kernel, grid_for((long)Np * C / 8), s, (const T*)dx0, dpos, B, Np, C);
    return ga_check_launch("ga_pos_add_bwd");
}

extern "C" int ga_rows_bcast(const void* src, void* dst, int B, int HW, int C, float scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(src && dst && B > 0 && HW > 0 && C > 0 && C % 8 == 0 && aligned16(src) && aligned16(dst), "ga_rows_bcast: bad args (C %% 8, 16-byte alignment)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PIT_DISPATCH(dtype, rows_bcast_kernel, grid_for((long)B * HW * C / 8), s, (const T*)src, (T*)dst, B, HW, C, scale);
    return ga_check_launch("ga_rows_bcast");
}
