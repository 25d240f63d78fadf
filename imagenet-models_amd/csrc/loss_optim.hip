// GA training loss (fused forward + gradient), top-k metric and the flat fused optimizer step (gfx950).
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = -3.0e38f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, red[i]);
    return s;
}

// One workgroup per sample.  logits fp32 [K][B][NC].
//   loss = sum_k L(out_k, y) + lam * sum_k KL_mean(log_softmax(out_k) || log_softmax(mean_j out_j))   (GA/train.py:735-745)
//   kind 0: cross entropy with label smoothing `smooth` (mean over B); kind 1: BCE-with-logits on smoothed one-hot
//   targets (timm BinaryCrossEntropy, mean over B*NC).
//   dlogits[k][b][c] (T) = d loss / d out_k * gscale     (mean target is detached, as in the reference)
template <typename T>
__global__ __launch_bounds__(256) void ga_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                      float* __restrict__ loss, T* __restrict__ dlogits, int K, int B,
                                                      int NC, float lam, int kind, float smooth, float gscale,
                                                      const float* __restrict__ avg, T* __restrict__ davg,
                                                      const float* __restrict__ dense, float thr) {
    extern __shared__ float sm[];  // mean logits -> r[NC]; red[8]
    float* r = sm;
    float* red = sm + NC;
    const long b = blockIdx.x;
    const int y = dense ? -1 : (int)target[b];
    // dense targets (mixup / cutmix: SoftTargetCrossEntropy, or BinaryCrossEntropy on the mixed targets): row sum for the
    // cross-entropy gradient (p * sum_t - t); thr >= 0 binarises the BCE target (timm BinaryCrossEntropy target_threshold)
    const float* trow = dense ? dense + b * NC : nullptr;
    float tsum = 1.f;
    if (dense && kind == 0) {
        float ts = 0.f;
        for (int c = threadIdx.x; c < NC; c += 256) ts += trow[c];
        tsum = block_sum(ts, red);
    }
    const float invB = 1.f / (float)B, invBN = 1.f / ((float)B * (float)NC);
    // r = softmax(mean_k out_k)
    float mx = -3.0e38f;
    for (int c = threadIdx.x; c < NC; c += 256) {
        float m = 0.f;
        for (int k = 0; k < K; ++k) m += logits[((long)k * B + b) * NC + c];
        m /= (float)K;
        r[c] = m;
        mx = fmaxf(mx, m);
    }
    mx = block_max(mx, red);
    float se = 0.f;
    for (int c = threadIdx.x; c < NC; c += 256) se += __expf(r[c] - mx);
    se = block_sum(se, red);
    const float lse_r = mx + __logf(se);
    for (int c = threadIdx.x; c < NC; c += 256) r[c] = r[c] - lse_r;  // log r
    __syncthreads();
    float total = 0.f;  // this thread's share of the loss
    for (int k = 0; k < K; ++k) {
        const float* o = logits + ((long)k * B + b) * NC;
        float m2 = -3.0e38f;
        for (int c = threadIdx.x; c < NC; c += 256) m2 = fmaxf(m2, o[c]);
        m2 = block_max(m2, red);
        float s2 = 0.f;
        for (int c = threadIdx.x; c < NC; c += 256) s2 += __expf(o[c] - m2);
        s2 = block_sum(s2, red);
        const float lse = m2 + __logf(s2);
        const float off = smooth / (float)NC, on = 1.f - smooth + off;
        for (int c = threadIdx.x; c < NC; c += 256) {
            const float logp = o[c] - lse, p = __expf(logp);
            const float logr = r[c], rr = __expf(logr);
            // KL term: exp(target) * (target - input), mean over B*NC
            total += lam * rr * (logr - logp) * invBN;
            float g = lam * invBN * (p - rr);
            float t = trow ? trow[c] : ((c == y) ? on : off);
            if (kind == 1 && thr >= 0.f) t = t > thr ? 1.f : 0.f;
            if (kind == 0) {
                total += -t * logp * invB;
                g += (p * tsum - t) * invB;
            } else {
                const float x = o[c];
                // BCE with logits: max(x,0) - x*t + log(1 + exp(-|x|))
                total += (fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)))) * invBN;
                g += (1.f / (1.f + __expf(-x)) - t) * invBN;
            }
            if (dlogits) elt<T>::st(dlogits + ((long)k * B + b) * NC + c, g * gscale);
        }
        if (avg) {   // MAP self-distillation term (MAP/train.py:815-816): KL_sum(log_softmax(avg_k) || log_softmax(out_k).detach()) / (B*NC)
            const float* a = avg + ((long)k * B + b) * NC;
            float m3 = -3.0e38f;
            for (int c = threadIdx.x; c < NC; c += 256) m3 = fmaxf(m3, a[c]);
            m3 = block_max(m3, red);
            float s3 = 0.f;
            for (int c = threadIdx.x; c < NC; c += 256) s3 += __expf(a[c] - m3);
            s3 = block_sum(s3, red);
            const float lse_a = m3 + __logf(s3);
            for (int c = threadIdx.x; c < NC; c += 256) {
                const float logp = o[c] - lse, p = __expf(logp), loga = a[c] - lse_a;
                total += p * (logp - loga) * invBN;
                if (davg) elt<T>::st(davg + ((long)k * B + b) * NC + c, (__expf(loga) - p) * invBN * gscale);
            }
        }
    }
    total = block_sum(total, red);
    if (threadIdx.x == 0) atomicAdd(loss, total);
}

// sum over heads + top-k (k <= 8) with lowest-index tie-break; one wave per sample.
__global__ __launch_bounds__(64) void heads_topk_kernel(const float* __restrict__ logits, int K, int B, int NC, int topk,
                                                        float* __restrict__ out_sum, int64_t* __restrict__ out_idx) {
    extern __shared__ float sv[];  // [NC]
    const long b = blockIdx.x;
    const int lane = threadIdx.x;
    for (int c = lane; c < NC; c += 64) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += logits[((long)k * B + b) * NC + c];
        if (out_sum) out_sum[b * NC + c] = s;
        sv[c] = s != s ? INFINITY : s;   // torch.topk ranks NaN largest (a diverged run must still give valid indices)
    }
    __syncthreads();
    for (int t = 0; t < topk; ++t) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < NC; c += 64) {
            const float v = sv[c];
            if (v != v) continue;        // already taken
            if (v > best || (v == best && c < bi)) {
                best = v;
                bi = c;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            out_idx[b * topk + t] = bi;
            if (bi < NC) sv[bi] = __builtin_nanf("");
        }
        __syncthreads();
    }
}

// hp = {lr, weight_decay, momentum|beta1, beta2, eps, bias_corr1, bias_corr2, first_step}
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, const float* __restrict__ hp, long n,
                                                  int nesterov, float wd_mult) {
    const float lr = hp[0], wd = hp[1] * wd_mult, mu = hp[2], first = hp[7];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i] + wd * p[i];
        float bi = first != 0.f ? gi : mu * buf[i] + gi;
        buf[i] = bi;
        gi = nesterov ? gi + mu * bi : bi;
        p[i] -= lr * gi;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ hp, long n, float wd_mult) {
    const float lr = hp[0], wd = hp[1] * wd_mult, b1 = hp[2], b2 = hp[3], eps = hp[4], bc1 = hp[5], bc2 = hp[6];
    const float step = lr / bc1, rbc2 = rsqrtf(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float pi = p[i] * (1.f - lr * wd);
        p[i] = pi - step * mi / (sqrtf(vi) * rbc2 + eps);
    }
}

}  // namespace

extern "C" int ga_loss_fwd_bwd(const float* logits, const int64_t* target, float* loss, void* dlogits, int K, int B,
                               int NC, float lam, int kind, float smoothing, float grad_scale, int dtype,
                               ga_stream_t stream) {
    return ga_map_loss_fwd_bwd(logits, nullptr, target, loss, dlogits, nullptr, K, B, NC, lam, kind, smoothing, grad_scale, dtype,
                               stream);
}

extern "C" int ga_map_loss_fwd_bwd(const float* org, const float* avg, const int64_t* target, float* loss, void* dorg, void* davg,
                                   int K, int B, int NC, float lam, int kind, float smoothing, float grad_scale, int dtype,
                                   ga_stream_t stream) {
    return ga_loss_dense_fwd_bwd(org, avg, target, nullptr, loss, dorg, davg, K, B, NC, lam, kind, smoothing, -1.f, grad_scale, dtype,
                                 stream);
}

extern "C" int ga_loss_dense_fwd_bwd(const float* org, const float* avg, const int64_t* target, const float* dense, float* loss,
                                     void* dorg, void* davg, int K, int B, int NC, float lam, int kind, float smoothing,
                                     float bce_threshold, float grad_scale, int dtype, ga_stream_t stream) {
    GA_REQUIRE(org && (target || dense) && !(target && dense) && loss && K >= 1 && B >= 1 && NC >= 1 && (kind == 0 || kind == 1),
               "ga_loss_fwd_bwd: bad args (exactly one of the class-index and the dense target)");
    GA_REQUIRE(NC <= 36000, "ga_loss_fwd_bwd: num_classes=%d exceeds the LDS row buffer (36000)", NC);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (NC + 8) * sizeof(float);
    if (lds > 65536) {
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(ga_loss_kernel<bf16_t>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                               hipFuncSetAttribute(reinterpret_cast<const void*>(ga_loss_kernel<float>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        GA_REQUIRE(ok, "ga_loss_fwd_bwd: cannot reserve %zu B of LDS", lds);
    }
    if (dtype == GA_BF16)
        hipLaunchKernelGGL(ga_loss_kernel<bf16_t>, dim3(B), dim3(256), lds, s, org, target, loss, (bf16_t*)dorg, K,
                           B, NC, lam, kind, smoothing, grad_scale, avg, (bf16_t*)davg, dense, bce_threshold);
    else
        hipLaunchKernelGGL(ga_loss_kernel<float>, dim3(B), dim3(256), lds, s, org, target, loss, (float*)dorg, K, B,
                           NC, lam, kind, smoothing, grad_scale, avg, (float*)davg, dense, bce_threshold);
    return ga_check_launch("ga_loss_fwd_bwd");
}

extern "C" int ga_heads_topk(const float* logits, int K, int B, int NC, int topk, float* out_sum, int64_t* out_idx,
                             ga_stream_t stream) {
    GA_REQUIRE(logits && out_idx && K >= 1 && topk >= 1 && topk <= NC, "ga_heads_topk: bad args");
    GA_REQUIRE(NC <= 36000, "ga_heads_topk: num_classes=%d exceeds the LDS row buffer (36000)", NC);
    if ((size_t)NC * sizeof(float) > 65536) {
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_topk_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        GA_REQUIRE(ok, "ga_heads_topk: cannot reserve LDS");
    }
    hipLaunchKernelGGL(heads_topk_kernel, dim3(B), dim3(64), NC * sizeof(float), reinterpret_cast<hipStream_t>(stream),
                       logits, K, B, NC, topk, out_sum, out_idx);
    return ga_check_launch("ga_heads_topk");
}

extern "C" int ga_sgd_step(float* p, const float* g, float* buf, const float* hp, int64_t n, int nesterov,
                           float wd_mult, ga_stream_t stream) {
    GA_REQUIRE(p && g && buf && hp && n > 0, "ga_sgd_step: bad args");
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (n + 255) / 256));
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, buf, hp,
                       (long)n, nesterov, wd_mult);
    return ga_check_launch("ga_sgd_step");
}

extern "C" int ga_adamw_step(float* p, const float* g, float* m, float* v, const float* hp, int64_t n, float wd_mult,
                             ga_stream_t stream) {
    GA_REQUIRE(p && g && m && v && hp && n > 0, "ga_adamw_step: bad args");
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (n + 255) / 256));
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, hp,
                       (long)n, wd_mult);
    return ga_check_launch("ga_adamw_step");
}

// ------------------------------------------------------------------------------------------------
// gradient clipping on the flat gradient buffer (timm dispatch_clip_grad, GA/train.py:325: 'norm' and 'value')
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = x[n4 * 4 + threadIdx.x];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// g *= min(1, max_norm / (sqrt(sumsq) + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, long n, const float* __restrict__ sumsq,
                                                         float max_norm) {
    const float coef = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    if (coef >= 1.f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] *= coef;
}
__global__ __launch_bounds__(256) void clip_value_kernel(float* __restrict__ g, long n, float v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] = fminf(fmaxf(g[i], -v), v);
}
}  // namespace

extern "C" int ga_sumsq_f32(const float* x, int64_t n, float* out, ga_stream_t stream) {
    GA_REQUIRE(x && out && n > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "ga_sumsq_f32: bad args (x 16-byte aligned)");
    const int blocks = (int)std::max<long>(1, std::min<long>(2048, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)n, out);
    return ga_check_launch("ga_sumsq_f32");
}

extern "C" int ga_clip_grad_f32(float* g, int64_t n, const float* sumsq, float limit, int mode, ga_stream_t stream) {
    GA_REQUIRE(g && n > 0 && limit > 0.f && (mode == 0 ? sumsq != nullptr : mode == 1), "ga_clip_grad_f32: bad args");
    const int blocks = (int)std::max<long>(1, std::min<long>(4096, (n + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (mode == 0) hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, s, g, (long)n, sumsq, limit);
    else hipLaunchKernelGGL(clip_value_kernel, dim3(blocks), dim3(256), 0, s, g, (long)n, limit);
    return ga_check_launch("ga_clip_grad_f32");
}

// ------------------------------------------------------------------------------------------------
// mixup / cutmix of one batch on the device (timm.data.Mixup, mode 'batch', as GA/train.py:544-557,727-728 applies it):
//   mixup:  out[b] = x[b] * lam + x[B-1-b] * (1 - lam)          (two rounded products, one rounded sum: as torch's mul_ / add_)
//   cutmix: out[b] = x[B-1-b] inside the box [yl, yh) x [xl, xh), x[b] outside
//   target: out[b][c] = lam * onehot_s(t[b])[c] + (1 - lam) * onehot_s(t[B-1-b])[c],  onehot_s: on = 1 - s + s/NC, off = s/NC
// lam and the box are drawn on the host (numpy, as timm does)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void mixup_batch_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int CH, int H,
                                                          int W, float lam, float oml, int cutmix, int yl, int yh, int xl, int xh) {
#pragma clang fp contract(off)      // two rounded products and a rounded sum, as torch's mul_ / add_ (hipcc would fuse them into an FMA)
    const long per = (long)CH * H * W, n = (long)B * per;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / per, r = i - b * per;
        const long j = (B - 1 - b) * per + r;
        if (cutmix) {
            const int px = (int)(r % W), py = (int)((r / W) % H);
            out[i] = (py >= yl && py < yh && px >= xl && px < xh) ? x[j] : x[i];
        } else {
            out[i] = x[i] * lam + x[j] * oml;
        }
    }
}

__global__ __launch_bounds__(256) void mixup_target_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int B, int NC,
                                                           float lam, float oml, float on, float off) {
#pragma clang fp contract(off)
    const long n = (long)B * NC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / NC), c = (int)(i - (long)b * NC);
        const float y1 = c == (int)t[b] ? on : off, y2 = c == (int)t[B - 1 - b] ? on : off;
        out[i] = y1 * lam + y2 * oml;
    }
}

// adaptive gradient clipping (timm.utils.agc.adaptive_clip_grad, reached through dispatch_clip_grad(mode='agc'),
// GA/train.py:752-756): per UNIT (row of a >= 2-d parameter, the whole tensor otherwise)
//   max_norm = max(|p|_2, eps) * clip_factor;   g *= max_norm / max(|g|_2, 1e-6)  where |g|_2 >= max_norm
// one wave per unit; units = device table {offset, length} into the flat parameter / gradient buffers
__global__ __launch_bounds__(256) void agc_kernel(const float* __restrict__ p, float* __restrict__ g, const int64_t* __restrict__ units,
                                                  int nunits, float clip_factor, float eps) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= nunits) return;
    const long off = units[2 * u], len = units[2 * u + 1];
    float sp = 0.f, sg = 0.f;
    for (long i = lane; i < len; i += 64) {
        const float a = p[off + i], b = g[off + i];
        sp = fmaf(a, a, sp);
        sg = fmaf(b, b, sg);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sp += __shfl_xor(sp, o, 64);
        sg += __shfl_xor(sg, o, 64);
    }
    const float max_norm = fmaxf(sqrtf(sp), eps) * clip_factor, gn = sqrtf(sg);
    if (gn < max_norm) return;
    const float sc = max_norm / fmaxf(gn, 1e-6f);
    for (long i = lane; i < len; i += 64) g[off + i] *= sc;
}
}  // namespace

// uint8 NCHW batch -> normalised fp32 NCHW: out = (float(x) - mean[c]) / std[c] -- what timm's PrefetchLoader does on the device
// to the uint8 batches of fast_collate (GA/train.py:567-595: `.float().sub_(mean).div_(std)`, mean / std already x 255)
namespace {
struct NormCh { float mean[4], std[4]; };
__global__ __launch_bounds__(256) void u8_normalize_kernel(const unsigned char* __restrict__ x, float* __restrict__ out, long n, int CH,
                                                           long HW, NormCh nc) {
    // 4 pixels per thread (HW % 4 == 0 is required): one 32-bit load, one 16-byte store
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        const int c = (int)((i / HW) % CH);
        const unsigned v = *reinterpret_cast<const unsigned*>(x + i);
        const float m = nc.mean[c], s = nc.std[c];
        *reinterpret_cast<float4*>(out + i) = make_float4(((float)(v & 255u) - m) / s, ((float)((v >> 8) & 255u) - m) / s,
                                                           ((float)((v >> 16) & 255u) - m) / s, ((float)(v >> 24) - m) / s);
    }
}
}  // namespace

extern "C" int ga_u8_normalize(const void* x, float* out, int B, int CH, int H, int W, const float* mean, const float* std,
                               ga_stream_t stream) {
    GA_REQUIRE(x && out && mean && std && B > 0 && CH > 0 && CH <= 4 && H > 0 && W > 0, "ga_u8_normalize: bad args (at most 4 channels)");
    GA_REQUIRE(((long)H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "ga_u8_normalize: H*W must be a multiple of 4 and the buffers 4 / 16-byte aligned");
    NormCh nc;
    for (int c = 0; c < 4; ++c) {
        nc.mean[c] = c < CH ? mean[c] : 0.f;     // host arrays
        nc.std[c] = c < CH ? std[c] : 1.f;
    }
    const long n = (long)B * CH * H * W;
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(u8_normalize_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char*>(x), out, n, CH, (long)H * W, nc);
    return ga_check_launch("ga_u8_normalize");
}

extern "C" int ga_mixup_batch(const float* x, float* out, int B, int CH, int H, int W, double lam, int cutmix, int yl, int yh, int xl,
                              int xh, ga_stream_t stream) {
    GA_REQUIRE(x && out && x != out && B > 0 && CH > 0 && H > 0 && W > 0, "ga_mixup_batch: bad args (out of place only)");
    GA_REQUIRE(!cutmix || (0 <= yl && yl <= yh && yh <= H && 0 <= xl && xl <= xh && xh <= W), "ga_mixup_batch: box outside the image");
    const long n = (long)B * CH * H * W;
    const int blocks = (int)std::max<long>(1, std::min<long>(8192, (n + 255) / 256));
    hipLaunchKernelGGL(mixup_batch_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out, B, CH, H, W, (float)lam,
                       (float)(1.0 - lam), cutmix, yl, yh, xl, xh);
    return ga_check_launch("ga_mixup_batch");
}

extern "C" int ga_mixup_target(const int64_t* target, float* out, int B, int NC, double lam, double smoothing, ga_stream_t stream) {
    GA_REQUIRE(target && out && B > 0 && NC > 0, "ga_mixup_target: bad args");
    const long n = (long)B * NC;
    const int blocks = (int)std::max<long>(1, std::min<long>(4096, (n + 255) / 256));
    hipLaunchKernelGGL(mixup_target_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), target, out, B, NC, (float)lam,
                       (float)(1.0 - lam), (float)(1.0 - smoothing + smoothing / NC), (float)(smoothing / NC));
    return ga_check_launch("ga_mixup_target");
}

extern "C" int ga_agc_clip(const float* params, float* grads, const int64_t* units, int nunits, float clip_factor, float eps,
                           ga_stream_t stream) {
    GA_REQUIRE(params && grads && units && nunits > 0 && clip_factor > 0.f, "ga_agc_clip: bad args");
    hipLaunchKernelGGL(agc_kernel, dim3((nunits + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, grads, units,
                       nunits, clip_factor, eps);
    return ga_check_launch("ga_agc_clip");
}

// ------------------------------------------------------------------------------------------------
// LAMB on the flat buffers (timm.optim.Lamb as used by the published GA recipes, GA/README.md:26 `--opt lamb`):
// per-TENSOR trust ratios need per-tensor norms, so the flat buffer is walked in chunks that never straddle a tensor
// (device table {offset, length, tensor id, weight-decay flag}); stage 1 updates the moments, writes the update
// direction and accumulates sum p^2 / sum u^2 per tensor, stage 2 applies p -= lr * trust * u.
//   hp (device): [lr, wd, beta1, beta2, eps, 1-beta1^t, 1-beta2^t, beta3, max_grad_norm]
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void lamb_stage1_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          float* __restrict__ u, const float* __restrict__ hp,
                                                          const float* __restrict__ gsumsq, const int* __restrict__ chunks,
                                                          float* __restrict__ norms) {
    __shared__ float red[2][4];
    const int* c = chunks + 4 * blockIdx.x;
    const long off = c[0];
    const int len = c[1], tid = c[2];
    const float wd = c[3] ? hp[1] : 0.f;
    const float b1 = hp[2], b2 = hp[3], eps = hp[4], ibc1 = 1.f / hp[5], isbc2 = rsqrtf(hp[6]), b3 = hp[7];
    const float gn = sqrtf(*gsumsq);
    const float iclip = gn > hp[8] ? hp[8] / gn : 1.f;
    float sp = 0.f, su = 0.f;
    for (int i = threadIdx.x; i < len; i += 256) {
        const long e = off + i;
        const float gi = g[e] * iclip, pi = p[e];
        const float mi = b1 * m[e] + b3 * gi;
        const float vi = b2 * v[e] + (1.f - b2) * gi * gi;
        m[e] = mi;
        v[e] = vi;
        const float up = (mi * ibc1) / (sqrtf(vi) * isbc2 + eps) + wd * pi;
        u[e] = up;
        sp = fmaf(pi, pi, sp);
        su = fmaf(up, up, su);
    }
    sp = wave_sum(sp);
    su = wave_sum(su);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sp;
        red[1][threadIdx.x >> 6] = su;
    }
    __syncthreads();
    if (threadIdx.x < 2)
        atomicAdd(norms + 2 * tid + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void lamb_stage2_kernel(float* __restrict__ p, const float* __restrict__ u,
                                                          const float* __restrict__ hp, const int* __restrict__ chunks,
                                                          const float* __restrict__ norms) {
    const int* c = chunks + 4 * blockIdx.x;
    const long off = c[0];
    const int len = c[1], tid = c[2];
    float trust = 1.f;
    if (c[3] && hp[1] != 0.f) {     // trust ratio only where weight decay applies (timm: weight_decay != 0 or always_adapt)
        const float wn = sqrtf(norms[2 * tid]), un = sqrtf(norms[2 * tid + 1]);
        trust = (wn > 0.f && un > 0.f) ? wn / un : 1.f;
    }
    const float step = hp[0] * trust;
    for (int i = threadIdx.x; i < len; i += 256) p[off + i] -= step * u[off + i];
}
}  // namespace

extern "C" int ga_lamb_stage1(const float* p, const float* g, float* m, float* v, float* u, const float* hp,
                              const float* gsumsq, const int* chunks, int nchunks, float* norms, ga_stream_t stream) {
    GA_REQUIRE(p && g && m && v && u && hp && gsumsq && chunks && norms && nchunks > 0, "ga_lamb_stage1: bad args");
    hipLaunchKernelGGL(lamb_stage1_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, u,
                       hp, gsumsq, chunks, norms);
    return ga_check_launch("ga_lamb_stage1");
}

extern "C" int ga_lamb_stage2(float* p, const float* u, const float* hp, const int* chunks, int nchunks, const float* norms,
                              ga_stream_t stream) {
    GA_REQUIRE(p && u && hp && chunks && norms && nchunks > 0, "ga_lamb_stage2: bad args");
    hipLaunchKernelGGL(lamb_stage2_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, u, hp,
                       chunks, norms);
    return ga_check_launch("ga_lamb_stage2");
}


// ------------------------------------------------------------------------------------------------
// DropPath masks (timm DropPath behind GA/ga_convnext.py:111, ga_cswin.py:209-210): out[s][b] = Bernoulli(keep[s]) / keep[s]
// for every stochastic-depth site s and sample b, from a counter-based generator (splitmix64 of seed, call counter and
// element index): one workgroup, so the call counter in device memory can be advanced by the kernel itself.
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(1024) void drop_path_sample_kernel(float* __restrict__ out, const float* __restrict__ keep, int sites,
                                                                int B, unsigned long long seed, unsigned long long* counter) {
    const unsigned long long call = *counter;
    const int n = sites * B;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const unsigned long long h = splitmix64(splitmix64(seed ^ (call * 0xD1342543DE82EF95ull)) + (unsigned long long)i);
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);     // 24 random bits -> [0, 1)
        const float k = keep[i / B];
        out[i] = u < k ? 1.0f / k : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) *counter = call + 1;
}
}  // namespace

extern "C" int ga_drop_path_sample(float* out, const float* keep, int sites, int B, uint64_t seed, uint64_t* counter,
                                   ga_stream_t stream) {
    GA_REQUIRE(out && keep && counter && sites > 0 && B > 0, "ga_drop_path_sample: bad args");
    hipLaunchKernelGGL(drop_path_sample_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), out, keep, sites,
                       B, (unsigned long long)seed, reinterpret_cast<unsigned long long*>(counter));
    return ga_check_launch("ga_drop_path_sample");
}

// ------------------------------------------------------------------------------------------------
// dropout masks (nn.Dropout inside the MAP head, map.py:82-83,54): out[i] = Bernoulli(keep) / keep for n elements, same
// counter-based generator as the DropPath sampler; any grid size -- the call counter is advanced by a second 1-thread launch
// that runs after every workgroup of the first has read it (stream order).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long n, float keep, unsigned long long seed,
                                                           const unsigned long long* __restrict__ counter) {
    const unsigned long long base = splitmix64(seed ^ (*counter * 0xD1342543DE82EF95ull));
    const float inv = 1.0f / keep;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned long long h = splitmix64(base + (unsigned long long)i);
        out[i] = (float)(h >> 40) * (1.0f / 16777216.0f) < keep ? inv : 0.0f;
    }
}
__global__ void bump_counter_kernel(unsigned long long* counter) { *counter += 1; }
}  // namespace

extern "C" int ga_dropout_mask_sample(float* out, int64_t n, float keep, uint64_t seed, uint64_t* counter, ga_stream_t stream) {
    GA_REQUIRE(out && counter && n > 0 && keep > 0.f && keep <= 1.f, "ga_dropout_mask_sample: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)std::max<long>(1, std::min<long>(2048, (n + 255) / 256));
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, s, out, (long)n, keep, (unsigned long long)seed,
                       reinterpret_cast<const unsigned long long*>(counter));
    hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned long long*>(counter));
    return ga_check_launch("ga_dropout_mask_sample");
}
