/* libgaext -- C ABI of the MI355X (gfx950) kernels for the GA-ConvNeXt / GA / MAP training hot path.
 *
 * The reference (Lab-LVM/imagenet-models) has NO native layer: its hot path is the list of ATen ops that
 * GA/ga_convnext.py, GA/train.py execute.  Each entry point below replaces the ATen
 * op(s) named in its comment (file:line = /root/reference/...).  Conventions:
 *
 *  - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (the library never
 *    allocates, frees or synchronises: the two entry points that reduce per-workgroup partial results take a
 *    caller-owned workspace sized by their ga_*_workspace() query); every call only ENQUEUES work on the
 *    hipStream_t passed last.
 *  - activations are NHWC ("channels last"), i.e. 2-D row-major [rows = B*H*W, C]; `dtype` selects the
 *    element type of activations and of the prepared ("effective") weight copies: GA_F32 = the parity math
 *    mode, GA_BF16 = the throughput mode (fp32 accumulation, fp32 statistics, fp32 master weights/gradients).
 *  - return 0 on success; <0 on error (GA_ERR_*), message retrievable with ga_last_error (thread-local).
 *  - 16-byte alignment of every activation / weight-copy pointer and leading dimensions that are multiples of
 *    8 elements are required (GA_ERR_BAD_ARG otherwise).
 */
#ifndef GAEXT_H
#define GAEXT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ga_stream_t; /* hipStream_t */

enum { GA_F32 = 0, GA_BF16 = 1 };
enum { GA_OK = 0, GA_ERR_BAD_ARG = -1, GA_ERR_UNSUPPORTED = -2, GA_ERR_HIP = -3 };

/* activation codes */
enum { GA_ACT_NONE = 0, GA_ACT_GELU = 1, GA_ACT_RELU = 2 };

/* A-operand gather kinds (implicit im2col): how row m / column k of the GEMM's A matrix map to memory */
enum {
    GA_A_PLAIN = 0,      /* A[m*lda + k]                                                              */
    GA_A_PATCH2 = 1,     /* NHWC [B,H,W,C] -> rows (b,oy,ox) of the 2x2/s2 patches, k = (ky,kx,c)    (ga_convnext.py:127) */
    GA_A_STEM4_NCHW = 2, /* fp32 NCHW [B,3,H,W] -> rows (b,oy,ox) of the 4x4/s4 patches, k=(c,ky,kx) (ga_convnext.py:357) */
    GA_A_CONV3 = 3,      /* NHWC [B,H,W,C], 3x3 pad 1 stride 1, k = (ky,kx,c)                         (ga_convnext.py:273) */
    GA_A_CONV3S2 = 4,    /* NHWC [B,H,W,C], 3x3 pad 1 stride 2 -> rows (b,oy,ox) of the [(H+1)/2, (W+1)/2] output,
                            k = (ky,kx,c): the deep stem and Merge_Block convs        (ga_cswin.py:464,474,256) */
    GA_A_NEIGH2 = 5      /* NHWC [B,H,W,C], rows (b,y,x), k = (ay,ax,c) reads pixel (y+ay, x+ax), ay,ax in {0,1}, zero
                            beyond the map: with ga_conv3s2_dgrad_prep's operand and GA_C_UNPATCH2 it is the
                            data-gradient (transposed convolution) of GA_A_CONV3S2 */
};
/* C-output kinds */
enum {
    GA_C_PLAIN = 0,
    GA_C_UNPATCH2 = 1 /* rows (b,oy,ox), cols (ky,kx,c) scattered back to NHWC [B,H,W,C] (dgrad of the 2x2/s2 conv) */
};

int ga_version(void);
/* copies the calling thread's last error message (NUL-terminated) into buf; returns its length */
int ga_last_error(char* buf, size_t n);
/* number of compute units / LDS per CU etc. of the current device (used for grid sizing); 0 on success */
int ga_device_info(int* num_cu, int* lds_bytes, int* wave_size);

/* Tuning knobs (kernel-form selection; no reference counterpart -- the reference delegates kernel choice to cuDNN / cuBLAS
 * heuristics behind torch.backends.cudnn.benchmark, GA/train.py:400).  Each knob NAME takes its value from the environment
 * variable GAEXT_<NAME> once, when the dispatch code first asks for it, or from ga_set_knob(); no launch ever reads the
 * environment.  ga_unset_knob() returns a knob to its built-in default.  ga_config_string() writes "libgaext <version> gfx950"
 * followed by every knob that is NOT at its default as " NAME=value(env|api)" (NUL-terminated, truncated to n) and returns
 * the untruncated length -- a bench / parity record shows what actually ran.  Result-changing debug switches do not exist in a
 * release build (they need -DGAEXT_DEBUG, which ga_config_string reports as DEBUG-BUILD). */
int ga_set_knob(const char* name, int value);
int ga_unset_knob(const char* name);
int ga_config_string(char* buf, size_t n);

/* ------------------------------------------------------------------------------------------------------------
 * ga_gemm:  C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] )          (both operands K-contiguous)
 * replaces: nn.Linear / 1x1, 2x2-s2, 4x4-s4 and 3x3 nn.Conv2d forward AND their data-gradients
 *           (ga_convnext.py:94,127,163-167,202-205,259-283,357,407,418,422) -- dgrad uses the transposed
 *           weight copy made by ga_weight_prep, so it is the same NT product.
 * epilogue order: v = alpha*acc; v += bias[n]; (C2 store); v = act(v); v *= dact(H[m][n]) (GELU'); v *= rowscale[m / rows_per_scale];
 *                 v += R[m][n]; if (relu_after) v = max(v,0); colsum[n] += v; colsumsq[n] += v*v; store.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, K;
    int batch;             /* grid z (>=1) */
    int dtype;             /* GA_F32 / GA_BF16: type of A, B, R, H (and of C unless c_f32) */
    /* A */
    const void* A;
    int64_t lda, strideA;
    int a_batch_mod;       /* A batch index = z % a_batch_mod (0: = z) */
    int a_kind;            /* GA_A_* */
    int a_H, a_W, a_C;     /* input dims for the gather kinds */
    int a_act;             /* GA_ACT_*: applied to A elements while staging (GELU of a stored pre-activation) */
    /* B (weights, [N][K] K-contiguous) */
    const void* B;
    int64_t ldb, strideB;
    /* C */
    void* C;
    int64_t ldc, strideC;
    int c_kind;            /* GA_C_* */
    int c_H, c_W, c_C;     /* NHWC dims of the scatter target (GA_C_UNPATCH2) */
    int c_f32;             /* 1: C is fp32 regardless of dtype */
    void* C2;              /* optional second output (dtype, layout of C): see c2_mode */
    int c2_mode;           /* 1: the pre-activation value (alpha*acc + bias); 2: act'(pre-activation) (GELU') -- lets the
                              forward fc1 store gelu(h) AND gelu'(h) so that backward never re-evaluates erf */
    /* epilogue */
    float alpha;
    const float* bias;     /* [N] or NULL */
    int64_t strideBias;
    int act;               /* GA_ACT_* */
    const void* H;         /* GELU-backward: multiply by gelu'(H[m][n]) (dtype), or NULL */
    int64_t ldh, strideH;
    int h_is_deriv;        /* 1: H already holds the derivative (stored by c2_mode 2): multiply by H[m][n] itself */
    const float* rowscale; /* per-sample scale (DropPath mask / keep) or NULL */
    int rows_per_scale;
    const void* R;         /* residual (dtype) or NULL */
    int64_t ldr, strideR;
    int relu_after;
    float* colsum;         /* [N] fp32 atomics, or NULL  (BatchNorm batch statistics / bias gradients) */
    float* colsumsq;       /* [N] fp32 atomics, or NULL */
    int64_t strideCol;
} ga_gemm_desc;
int ga_gemm(const ga_gemm_desc* d, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * ga_wgrad:  dW[z][n][k] (+)= alpha * sum_m Y[z][m][n] * X[z][m][k]              (reduction over the ROW index)
 * replaces: the weight-gradient of every Linear/Conv2d above (ATen convolution_backward / mm in autograd) and,
 *           with Y == X, the Gram product X.X^T of GA_ConvNeXt.get_gram (ga_convnext.py:459-460).
 * The M range is split over `split_m` workgroups that combine with fp32 atomics (split_m == 1: plain store when
 * accumulate == 0).  dbias[n] += sum_m Y[m][n] when dbias != NULL.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, K;
    int batch;
    int dtype;
    const void* Y;
    int64_t ldy, strideY;
    const void* X;
    int64_t ldx, strideX;
    int x_batch_mod;       /* X batch index = z % x_batch_mod (0: = z) */
    int x_kind;            /* GA_A_* gather of X rows */
    int x_H, x_W, x_C;
    int x_act;             /* GA_ACT_GELU: X := gelu(X) while staging */
    float* dW;             /* fp32 [N][ldw] */
    int64_t ldw, strideW;
    float* dbias;          /* fp32 [N] or NULL */
    int64_t strideDbias;
    float alpha;
    int split_m;           /* >=1 */
    int accumulate;        /* 1: add into dW (atomics); 0 with split_m==1: overwrite */
    void* workspace;       /* caller-owned device scratch of ga_wgrad_workspace(d) bytes (16-byte aligned), or NULL: the wide */
    int64_t ws_bytes;      /* form then combines its row splits with fp32 atomics instead of partial tiles + a reduce launch */
} ga_wgrad_desc;
int ga_wgrad(const ga_wgrad_desc* d, ga_stream_t stream);
/* bytes of scratch ga_wgrad would like for this descriptor (0: none).  Concurrent calls (different streams) need distinct
 * workspaces; calls on one stream may share one. */
size_t ga_wgrad_workspace(const ga_wgrad_desc* d);

/* ------------------------------------------------------------------------------------------------------------
 * Weight preparation: fp32 master conv/linear weight [G*Co][Ci][KH][KW]  ->  "effective" copies in `dtype`
 *   out [G][Co][ldo]  with k = (ky,kx,ci)   : out = rs[n] * w * cs[ci]          (B operand of the forward GEMM)
 *   outT[G][KH*KW*Ci][ldt] (co contiguous)  : transposed copy                   (B operand of the dgrad GEMM);
 *        flip=1: outT is [G][Ci][ldt >= KH*KW*Co] with column ((KH-1-ky)*KW + KW-1-kx)*Co + co
 *        (dgrad of a stride-1 'same' conv as a GA_A_CONV3 product over dY).
 *   stem=1: k = (ci,ky,kx) (the NCHW patch order of GA_A_STEM4_NCHW).
 * Folds LayerNorm scale (cs) / LayerScale gamma (rs) into the weights (ga_convnext.py:93,95,106,110).
 * Pads [.., ldo) / [.., ldt) with zeros.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const float* w;
    int G, Co, Ci, KH, KW;
    const float* rs;       /* [G*Co] or NULL (indexed by the SOURCE row) */
    const float* cs;       /* [Ci] or NULL */
    const int* row_perm;   /* [G*Co] or NULL: effective row n is master row row_perm[n] (channel_shuffle, ga_convnext.py:217,557) */
    int dtype;
    void* out;  int64_t ldo;   /* may be NULL */
    void* outT; int64_t ldt;   /* may be NULL */
    int flip;
    int stem;
    int t_cols;            /* columns of every outT row this job writes (incl. zero padding), 0 = ldt: several jobs
                            * may fill column ranges of ONE stacked transposed operand of row stride ldt */
} ga_wprep_desc;
int ga_weight_prep(const ga_wprep_desc* d, ga_stream_t stream);
/* be[n] = rs[m] * (b[m] + sum_c W[m][c] * v[c]), m = row_perm ? row_perm[n] : n   (W fp32 [N][C]; b, rs, v may be NULL) */
int ga_bias_fold(const float* W, const float* b, const float* rs, const float* v, const int* row_perm, float* be, int N,
                 int C, ga_stream_t stream);
/* Inverse of the folding for gradients. G = d(effective weight) fp32 [N][ldg] with k=(ky,kx,ci); gb = d(effective bias),
 * where We = rs[n]*W*cs[ci] and be[n] = rs[n]*(b[n] + sum_c W[n][c]*v[c]):
 *   dW[n][ci][ky][kx] += rs[n]*(G*cs[ci] + gb[n]*v[ci]);  d_cs[ci] += sum rs[n]*G*W;
 *   d_rs[n] += sum_k G*W*cs[ci] + gb[n]*(b[n] + sum_c W*v);  d_v[ci] += sum_n gb[n]*rs[n]*W[n][ci];  db[n] += rs[n]*gb[n].
 * cs / v / d_cs / d_v need KH=KW=1.  Any output may be NULL. */
typedef struct {
    const float* G; int64_t ldg;
    const float* gb;
    const float* W; const float* b;
    const float* rs; const float* cs; const float* v;
    const int* row_perm;   /* rows of G / gb are effective rows; master row = row_perm[n] */
    int N, Ci, KH, KW;
    int stem;
    float* dW; float* db; float* d_rs; float* d_cs; float* d_v;
} ga_wunfold_desc;
int ga_weight_unfold(const ga_wunfold_desc* d, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Depthwise 7x7 (pad 3) convolution, NHWC  (ga_convnext.py:92,100)
 *   w49: fp32 [49][C] (tap-major copy of conv_dw.weight), bias fp32 [C]
 * ------------------------------------------------------------------------------------------------------------ */
int ga_dwconv7_fwd(const void* x, const float* w49, const float* bias, void* y, int B, int H, int W, int C,
                   int dtype, ga_stream_t stream);
/* dx = res + conv7x7(dy, flipped w)   (res may be NULL) */
int ga_dwconv7_bwd_data(const void* dy, const float* w49, const void* res, void* dx, int B, int H, int W, int C,
                        int dtype, ga_stream_t stream);
/* the same with a second output dx2[b,...] = dx[b,...] (as stored) * scale2[b]: the next block's DropPath row scale
 * (x.div(keep) * mask, timm DropPath behind GA/ga_convnext.py:111) applied where dx is produced instead of by a
 * separate pass over it; res is required */
int ga_dwconv7_bwd_data2(const void* dy, const float* w49, const void* res, void* dx, void* dx2, const float* scale2,
                         int B, int H, int W, int C, int dtype, ga_stream_t stream);
/* dw49[49][C] += sum dy * shifted x ; dbias[C] += sum dy   (per-workgroup partial sums in the CALLER-provided workspace of
 * ga_dwconv7_bwd_weight_workspace(...) bytes, then one reduction launch; deterministic for a fixed grid) */
size_t ga_dwconv7_bwd_weight_workspace(int B, int H, int W, int C, int dtype);
int ga_dwconv7_bwd_weight(const void* dy, const void* x, float* dw49, float* dbias, int B, int H, int W, int C,
                          int dtype, void* workspace, size_t ws_bytes, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim of [rows][C]  (F.layer_norm / LayerNorm2d, ga_convnext.py:51-67,93,233,237)
 *   fwd: y = (x-mean)*rstd [*w + b];  saves mean/rstd fp32 [rows] (each may be NULL).
 *   bwd: xhat = x_is_normalized ? x : (x-mean)*rstd;
 *        dx = rstd * (g*w - mean_c(g*w) - xhat*mean_c(g*w*xhat)) (+ dres);  dw[c] += sum g*xhat; db[c] += sum g
 * ------------------------------------------------------------------------------------------------------------ */
int ga_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t rows,
                     int C, float eps, int dtype, ga_stream_t stream);
int ga_layernorm_bwd(const void* g, const void* x, const float* mean, const float* rstd, const float* w,
                     const void* dres, void* dx, float* dw, float* db, int64_t rows, int C, int x_is_normalized,
                     int dtype, ga_stream_t stream);
/* the same with a second output dx2[row] = (dx[row] as stored) * scale2[row / rows_per_scale]: the DropPath-scaled copy the next
 * block of the backward chain would otherwise make in a pass of its own (timm DropPath in Block.forward: x + drop_path(f(x))) */
int ga_layernorm_bwd_dp(const void* g, const void* x, const float* mean, const float* rstd, const float* w, const void* dres,
                        void* dx, float* dw, float* db, int64_t rows, int C, int x_is_normalized, void* dx2, const float* scale2,
                        int64_t rows_per_scale, int dtype, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * BatchNorm2d, train mode = per-process batch statistics (ga_convnext.py:261,270,276,283,409,420).
 * The column sums come from ga_gemm's colsum/colsumsq epilogue.
 *   finalize: mean/var (biased) -> scale = w*rstd, shift = b - mean*scale; running stats updated with the unbiased
 *             variance and `momentum`; training=0 builds scale/shift from the running stats instead.
 *   affine_act: y = (x*scale[c] + shift[c]) * rowscale[row/rows_per_scale] (+ res) (ReLU)   (scale/shift/rowscale NULL ok;
 *               rowscale = the DropPath factor of the Bottleneck main branch, ga_convnext.py:310-311)
 *   bwd_reduce: s1[c] += sum g, s2[c] += sum g*xhat, g = dy * (y_relu > 0 if given) * rowscale
 *   bwd_apply:  dx = w*rstd*(g - s1/n - xhat*s2/n)
 * ------------------------------------------------------------------------------------------------------------ */
int ga_bn_finalize(const float* sum, const float* sumsq, int64_t n, const float* w, const float* b, float eps,
                   float momentum, float* running_mean, float* running_var, float* mean_out, float* rstd_out,
                   float* scale, float* shift, int C, int training, ga_stream_t stream);
/* ldx / lddx: row stride (elements) of x / dx when they are column slices of a wider matrix (the five heads'
 * gram_contraction outputs come from ONE GEMM); 0 = C.  Every other operand is contiguous [rows][C]. */
int ga_affine_act(const void* x, const float* scale, const float* shift, const void* res, const float* rowscale,
                  int64_t rows_per_scale, void* y, int64_t rows, int C, int relu, int dtype, int64_t ldx, ga_stream_t stream);
int ga_bn_bwd_reduce(const void* dy, const void* y_relu, const void* x, const float* mean, const float* rstd,
                     const float* rowscale, int64_t rows_per_scale, float* s1, float* s2, int64_t rows, int C, int dtype,
                     int64_t ldx, ga_stream_t stream);
int ga_bn_bwd_apply(const void* dy, const void* y_relu, const void* x, const float* mean, const float* rstd,
                    const float* w, const float* s1, const float* s2, const float* rowscale, int64_t rows_per_scale,
                    int64_t n, void* dx, int64_t rows, int C, int dtype, int64_t ldx, int64_t lddx, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-scale aggregate (ga_convnext.py:396-397,479-483): write pool(src) into channels [c_off, c_off+C) of the
 * concat buffer dst[B,Hout,Wout,ldd].  mode 0: f x f average (AdaptiveAvgPool2d to 14; f=1 is a copy);
 * mode 1: bilinear x2, align_corners=False (nn.Upsample).   bwd: dsrc = (dres +) pool^T(dcat slice).
 * ------------------------------------------------------------------------------------------------------------ */
int ga_pool_concat_fwd(const void* src, void* dst, int B, int Hin, int Win, int C, int Hout, int Wout, int ldd,
                       int c_off, int mode, int dtype, ga_stream_t stream);
int ga_pool_concat_bwd(const void* dcat, const void* dres, void* dsrc, int B, int Hin, int Win, int C, int Hout,
                       int Wout, int ldd, int c_off, int mode, int dtype, ga_stream_t stream);

/* Squeeze-excite (timm SEModule via create_attn('se'), ga_convnext.py:279,305):
 *   ga_spatial_sum: out[b][c] = scale * sum_hw a[b,hw,c] (* b2[b,hw,c] if given)      (fp32 [B][C])
 *   ga_se_mlp_fwd:  gate = sigmoid(W2 relu(W1 s + b1) + b2)  (W1 [R][C], W2 [C][R], fp32 master weights)
 *   ga_se_mlp_bwd:  ds = ds_scale * d(pooled mean), parameter gradients accumulated with atomics
 *   ga_chan_scale:  y[b,hw,c] = x*g[b][c] + add[b][c]   (g/add may be NULL) */
int ga_spatial_sum(const void* a, const void* b2, float* out, int B, int HW, int C, float scale, int dtype,
                   ga_stream_t stream);
int ga_se_mlp_fwd(const float* s, const float* W1, const float* b1, const float* W2, const float* b2, float* hid,
                  float* gate, int B, int C, int R, ga_stream_t stream);
int ga_se_mlp_bwd(const float* dgate, const float* gate, const float* hid, const float* s, const float* W1,
                  const float* W2, float* ds, float ds_scale, float* dW1, float* db1, float* dW2, float* db2, int B,
                  int C, int R, ga_stream_t stream);
int ga_chan_scale(const void* x, const float* g, const float* add, void* y, int B, int HW, int C, int dtype,
                  ga_stream_t stream);

/* Gram vector (GA_ConvNeXt.get_gram, ga_convnext.py:452-467; index order :424-430): the fp32 Gram matrices
 * G[B][C][C] (from ga_wgrad with Y == X) -> row-major upper-triangular entries, L2-normalised (eps 1e-12), stored in
 * the grouped layout [B][groups][Kp] (Kp >= ntri/groups, zero padded) read by the grouped embedding GEMM.
 * bwd: S[B][C][C] = symmetric gradient of the raw Gram entries (diagonal doubled) so that dX = alpha * X . S */
int ga_gram_pack_fwd(const float* G, void* out, float* inv_norm, int B, int C, int groups, int Kp, int dtype,
                     ga_stream_t stream);
int ga_gram_pack_bwd(const void* dvec, const void* vhat, const float* inv_norm, void* S, int B, int C, int groups,
                     int Kp, int dtype, ga_stream_t stream);

/* Class attention with ONE query token (ClassAttn / LayerScaleBlockClassAttn, ga_convnext.py:153-187,244-248):
 *   token_cat: u[B][N+1][C] = cat(cls[B][C], tok[B][N][C]);  token_split: dcls (+)= du[:,0], dtok (+)= du[:,1:]
 *   class_attn: q [B][E], kv [B*(N)][2E] (k | v), E = heads*hd <= 64 per head; P fp32 [B][heads][N] saved. */
int ga_token_cat(const void* cls, const void* tok, void* u, int B, int N, int C, int dtype, ga_stream_t stream);
int ga_token_split(const void* du, void* dcls, void* dtok, int B, int N, int C, int acc_cls, int acc_tok, int dtype,
                   ga_stream_t stream);
int ga_class_attn_fwd(const void* q, const void* kv, void* out, float* P, int B, int N, int heads, int hd, float scale,
                      int dtype, ga_stream_t stream);
int ga_class_attn_bwd(const void* dout, const void* q, const void* kv, const float* P, void* dq, void* dkv, int B, int N,
                      int heads, int hd, float scale, int dtype, ga_stream_t stream);
/* split form: the class-token row and the N-1 image-token rows of k|v in separate arrays (kv_cls [B][2E], kv_tok
 * [B][N-1][2E]; N counts the class token).  The image tokens' LayerNorm (LayerScaleBlockClassAttn.norm1 on cat(x_cls, x),
 * ga_convnext.py:244-246) is row-wise, so the normalised image tokens are shared by all heads and never concatenated. */
/* tok_ld: row stride (elements) of kv_tok / dkv_tok, 0 = 2E: the five heads' k|v rows may be column slices of ONE
 * [B*(N-1)][5*2E] matrix produced by a single GEMM over the shared normalised tokens. */
int ga_class_attn_fwd2(const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, void* out, float* P, int B, int N,
                       int heads, int hd, float scale, int dtype, ga_stream_t stream);
int ga_class_attn_bwd2(const void* dout, const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, const float* P,
                       void* dq, void* dkv_cls, void* dkv_tok, int B, int N, int heads, int hd, float scale, int dtype,
                       ga_stream_t stream);

/* GA training loss, fused forward + gradient (GA/train.py:735-745):
 *   loss += sum_k L(out_k, y) + lam * sum_k KL_mean(log_softmax(out_k) || log_softmax(mean_j out_j))  (mean detached)
 *   kind 0: (label-smoothed) cross entropy; kind 1: timm BinaryCrossEntropy on smoothed one-hot targets.
 *   logits fp32 [K][B][NC]; target int64 [B]; *loss must be zeroed by the caller; dlogits (dtype) = grad * grad_scale */
int ga_loss_fwd_bwd(const float* logits, const int64_t* target, float* loss, void* dlogits, int K, int B, int NC,
                    float lam, int kind, float smoothing, float grad_scale, int dtype, ga_stream_t stream);
/* validate(): output = sum_k out_k.float(); top-k indices, ties -> lowest index  (GA/train.py:848-860) */
int ga_heads_topk(const float* logits, int K, int B, int NC, int topk, float* out_sum, int64_t* out_idx,
                  ga_stream_t stream);

/* Flat fused optimizer steps over fp32 [n] (train.py:769; torch.optim.SGD(nesterov) / torch.optim.AdamW semantics).
 * hp (device, fp32[8]) = {lr, weight_decay, momentum|beta1, beta2, eps, 1-beta1^t, 1-beta2^t, first_step};
 * wd_mult = 0 for the no-decay segment (timm: ndim<=1 or *.bias). */
int ga_sgd_step(float* p, const float* g, float* buf, const float* hp, int64_t n, int nesterov, float wd_mult,
                ga_stream_t stream);
int ga_adamw_step(float* p, const float* g, float* m, float* v, const float* hp, int64_t n, float wd_mult,
                  ga_stream_t stream);

/* Batched variants: `jobs_dev` is a DEVICE-resident array of n descriptors (same structs as above); one launch
 * processes them all (grid.y = job).  Used for the ~270 small per-parameter jobs of a training step. */
typedef struct {
    int kind;              /* 0: y[i] += a*x[i] (n elems); 1: y[c][r] (+)= x[r][c] (x is [R][C]); 2: bias fold (R rows, C cols):
                              y[n] = rs[m]*(b[m] + sum_c x[m][c]*v[c]), m = row_perm ? row_perm[n] : n */
    float* y; const float* x; float a; int64_t n;
    int R, C, accumulate;
    const float* b; const float* rs; const float* v; const int* row_perm;
} ga_small_desc;
int ga_weight_prep_batch(const ga_wprep_desc* jobs_dev, int n, ga_stream_t stream);
int ga_weight_unfold_batch(const ga_wunfold_desc* jobs_dev, int n, ga_stream_t stream);
int ga_small_batch(const ga_small_desc* jobs_dev, int n, ga_stream_t stream);

/* LAMB (timm.optim.Lamb, the optimizer of the published GA recipes: GA/README.md:26) on the flat buffers.
 *   chunks: int32 [nchunks][4] = {element offset, length, tensor id, weight-decay flag}; a chunk never straddles a tensor.
 *   hp (device fp32[9]): lr, weight_decay, beta1, beta2, eps, 1-beta1^t, 1-beta2^t, beta3, max_grad_norm.
 *   stage1: g' = g / max(|g|_2 / max_grad_norm, 1) (|g|_2^2 = *gsumsq, from ga_sumsq_f32); m, v updated;
 *           u = (m/bc1) / (sqrt(v)/sqrt(bc2) + eps) + wd*p;  norms[2t] += sum p^2, norms[2t+1] += sum u^2 (caller zeroes norms)
 *   stage2: p -= lr * trust_t * u, trust_t = |p_t| / |u_t| where weight decay applies (both > 0), else 1. */
int ga_lamb_stage1(const float* p, const float* g, float* m, float* v, float* u, const float* hp, const float* gsumsq,
                   const int* chunks, int nchunks, float* norms, ga_stream_t stream);
int ga_lamb_stage2(float* p, const float* u, const float* hp, const int* chunks, int nchunks, const float* norms,
                   ga_stream_t stream);

/* gradient clipping on the flat fp32 gradient buffer (timm dispatch_clip_grad via NativeScaler, GA/train.py:312-333):
 *   ga_sumsq_f32: *out += sum x^2 (caller zeroes *out; under DDP call after the all-reduce);
 *   ga_clip_grad_f32 mode 0 ('norm'): g *= min(1, limit / (sqrt(*sumsq) + 1e-6));  mode 1 ('value'): clamp to [-limit, limit] */
int ga_sumsq_f32(const float* x, int64_t n, float* out, ga_stream_t stream);
int ga_clip_grad_f32(float* g, int64_t n, const float* sumsq, float limit, int mode, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * GA-CSWin: cross-shaped stripe-window attention with LePE (LePEAttention.forward, ga_cswin.py:110-136; get_lepe
 * :95-108; img2windows / windows2img :215-233) for the 1 or 2 branches of a CSWinBlock (:202-207) in ONE launch.
 *   qkv [B*L][ldq]: q | k | v column blocks of width C (tokens in image raster order, L = reso*reso);
 *   branch i owns channels [i*C/nbranch, (i+1)*C/nbranch) of each block, heads/nbranch heads, stripes Hs[i] x Ws[i];
 *   out[b, t, ch] = softmax_j((q_t * scale) . k_j) v_j + conv3x3_depthwise(v)(t)   over the tokens j of t's stripe,
 *   the 3x3 (lepe_w[i]: fp32 [C/nbranch][9] = get_v.weight, lepe_b[i]) zero padded at the STRIPE border.
 *   head_dim in {8, 16, 32}; Hs*Ws <= 128.  bf16 with head_dim 32 runs on MFMA, everything else on a generic fp32 form.
 * bwd: dqkv [B*L][ldq] = d(q | k | v) from dout [B*L][ldo]  (dv includes the LePE transpose).  With a caller-owned `lepe_ws` of
 *   ga_cswin_attn_bwd_workspace(d) bytes (16-byte aligned; 0 = this shape runs the generic form, pass NULL) the kernel also
 *   leaves the per-(window, head) partial LePE weight gradients there from the tiles it already holds in LDS;
 *   ga_cswin_lepe_wgrad_reduce then adds their sum into dw_i / db_i.  The buffer may be reused once the reduce has run.
 * lepe_wgrad: the unfused form of the same (any shape): dw_i[ch][tap] += sum dout * shifted v, db_i[ch] += sum dout  (fp32 atomics). */
typedef struct {
    int B, reso, C, heads, nbranch;
    int Hs[2], Ws[2];
    const float* lepe_w[2];
    const float* lepe_b[2];
    float scale;
    int dtype;
    const void* qkv; int64_t ldq;
    void* out; int64_t ldo;
} ga_cswin_attn_desc;
int ga_cswin_attn_fwd(const ga_cswin_attn_desc* d, ga_stream_t stream);
size_t ga_cswin_attn_bwd_workspace(const ga_cswin_attn_desc* d);
int ga_cswin_attn_bwd(const ga_cswin_attn_desc* d, const void* dout, void* dqkv, void* lepe_ws, size_t ws_bytes, ga_stream_t stream);
int ga_cswin_lepe_wgrad_reduce(const ga_cswin_attn_desc* d, const void* lepe_ws, float* dw0, float* db0, float* dw1, float* db1,
                               ga_stream_t stream);
int ga_cswin_lepe_wgrad(const ga_cswin_attn_desc* d, const void* dout, float* dw0, float* db0, float* dw1, float* db1,
                        ga_stream_t stream);
/* deep-stem helpers (ga_cswin.py:463-477):
 *   ga_nchw3_to_nhwc8: fp32 NCHW [B,3,H,W] -> NHWC [B,H,W,8] in `dtype`, channels 3..7 zero (the first 3x3/s2 conv then is
 *                      a GA_A_CONV3S2 gather with C = 8);
 *   ga_convw_pack:     fp32 [Co][Ci][taps] -> out[co][tap*Cp + ci] in `dtype` (zero for ci >= Ci and up to ldo);
 *   ga_convw_unpack_grad: dW[co][ci][tap] += G[co][tap*Cp + ci];
 *   ga_conv3s2_dgrad_prep: fp32 [Co][Ci][3][3] -> out[(py,px,ci)][(ay,ax,co)] (4*Ci rows, ldo >= 4*Co), see GA_A_NEIGH2 */
/* LayerNorm(affine) -> GELU(erf) between the stem convs (ga_cswin.py:466-468,471-473): y = gelu(LN(x)*w + b);
 * bwd: dx = LN'(g * gelu'(LN(x)*w + b)), dw / db accumulated (fp32 atomics).  C a power of two in [8, 512]. */
int ga_layernorm_gelu_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t rows,
                          int C, float eps, int dtype, ga_stream_t stream);
int ga_layernorm_gelu_bwd(const void* g, const void* x, const float* mean, const float* rstd, const float* w, const float* b,
                          void* dx, float* dw, float* db, int64_t rows, int C, int dtype, ga_stream_t stream);
int ga_nchw3_to_nhwc8(const float* x, void* y, int B, int H, int W, int dtype, ga_stream_t stream);
int ga_convw_pack(const float* w, void* out, int Co, int Ci, int taps, int Cp, int64_t ldo, int dtype, ga_stream_t stream);
int ga_convw_unpack_grad(const float* G, float* dW, int Co, int Ci, int taps, int Cp, int64_t ldg, ga_stream_t stream);
int ga_conv3s2_dgrad_prep(const float* w, void* out, int Co, int Ci, int64_t ldo, int dtype, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * MAP head (MAP/models/map.py).  The GEMM-shaped parts go through ga_gemm / ga_wgrad; these are the rest.
 *   ga_gram_pack_fwd2 / _bwd2: ga_gram_pack_* with GramToken's token interleave (map.py:225-227): packed upper-triangular
 *       entry t is stored at (t % ntok) * (ntri / ntok) + t / ntok before the grouped layout is applied (ntok = 1: identity).
 *   ga_map_tokens_fwd: e [B][C*T] (channel c*T + t, the bp_reduction output, map.py:231-232) -> tok [B][T (+1)][C];
 *       add_mean: the extra row is the mean over the T tokens (CAP's self-distillation token, map.py:273-275).  _bwd: its transpose.
 *   ga_class_attn_mt_*: ClassAttention with T <= 4 query tokens (map.py:118-144, in_dim == dim branch; `interactive`: ga_class_attn_mt_ia_* below):
 *       q [B][T][E], kv_cls [B][T][2E] (k | v of the class rows), kv_tok = k | v rows of the N - T image tokens (row stride
 *       tok_ld); P [B][T][heads][N] fp32 = softmax, saved; mask (fp32, same shape, or NULL) = attention dropout mask
 *       (already divided by keep); out [B][T][E].  bwd overwrites dq, dkv_cls and the dkv_tok rows (stride dtok_ld).
 *   ga_map_loss_fwd_bwd: MAP/train.py:792-839 (distill_tokens == 0): ga_loss_fwd_bwd on the org logits plus, per group,
 *       KL_sum(log_softmax(avg_k) || log_softmax(org_k).detach()) / (B*NC); davg = its gradient (avg == NULL: the GA loss).
 *   ga_gelu_fwd / _bwd: y = gelu(x) (erf), dx = dy * gelu'(x)   (MultiScale ConvNormAct with non_linearity = GELU, map.py:331)
 *   ga_relu_drop: out = a * m, deriv = (a > 0) * m  for a = relu output, m = fp32 dropout mask or NULL (GroupConvMlp, act ReLU)
 *   ga_mask_mul: y = x * m (+ res);   ga_copy2d: dst[r][0..cols) (+)= src[r][0..cols) with row strides (elements) */
int ga_gram_pack_fwd2(const float* G, void* out, float* inv_norm, int B, int C, int groups, int Kp, int ntok, int dtype,
                      ga_stream_t stream);
int ga_gram_pack_bwd2(const void* dvec, const void* vhat, const float* inv_norm, void* S, int B, int C, int groups, int Kp,
                      int ntok, int dtype, ga_stream_t stream);
int ga_map_tokens_fwd(const void* e, void* tok, int B, int C, int T, int add_mean, int dtype, ga_stream_t stream);
int ga_map_tokens_bwd(const void* dtok, void* de, int B, int C, int T, int add_mean, int dtype, ga_stream_t stream);
int ga_class_attn_mt_fwd(const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, void* out, float* P,
                         const float* mask, int B, int T, int N, int heads, int hd, float scale, int dtype, ga_stream_t stream);
int ga_class_attn_mt_bwd(const void* dout, const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, const float* P,
                         const float* mask, void* dq, void* dkv_cls, void* dkv_tok, int64_t dtok_ld, int B, int T, int N,
                         int heads, int hd, float scale, int dtype, ga_stream_t stream);
/* ClassAttention with `interactive` = True (map.py:96-98,130-136): W1, W2 [heads][heads] + b1, b2 [heads] fp32 mix the HEADS of
 * the scores before and of the probabilities after the softmax: U = S + W1 S + b1, A = softmax(U), Pm = A + W2 A + b2.
 * Same operand layouts as ga_class_attn_mt_*; P saves A.  _bwd also ACCUMULATES dW1, db1, dW2, db2 (fp32). */
int ga_class_attn_mt_ia_fwd(const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, void* out, float* P,
                            const float* mask, const float* W1, const float* b1, const float* W2, const float* b2, int B, int T, int N,
                            int heads, int hd, float scale, int dtype, ga_stream_t stream);
int ga_class_attn_mt_ia_bwd(const void* dout, const void* q, const void* kv_cls, const void* kv_tok, int64_t tok_ld, const float* P,
                            const float* mask, const float* W1, const float* W2, const float* b2, void* dq, void* dkv_cls,
                            void* dkv_tok, int64_t dtok_ld, float* dW1, float* db1, float* dW2, float* db2, int B, int T, int N,
                            int heads, int hd, float scale, int dtype, ga_stream_t stream);
int ga_map_loss_fwd_bwd(const float* org, const float* avg, const int64_t* target, float* loss, void* dorg, void* davg, int K,
                        int B, int NC, float lam, int kind, float smoothing, float grad_scale, int dtype, ga_stream_t stream);
/* the same losses on DENSE targets [B][NC] fp32 (mixup / cutmix, GA/train.py:616-621: SoftTargetCrossEntropy for kind 0 --
 * mean_b sum_c -t log_softmax(x) -- and BinaryCrossEntropy on the mixed targets for kind 1); exactly one of `target` (class
 * indices, with `smoothing`) and `dense` is given; bce_threshold >= 0 binarises the BCE target (--bce-target-thresh), < 0: off.
 * avg / davg as ga_map_loss_fwd_bwd (NULL for the GA loss). */
int ga_loss_dense_fwd_bwd(const float* org, const float* avg, const int64_t* target, const float* dense, float* loss, void* dorg,
                          void* davg, int K, int B, int NC, float lam, int kind, float smoothing, float bce_threshold,
                          float grad_scale, int dtype, ga_stream_t stream);
/* timm.data.Mixup (mode 'batch') on the device, lam / box drawn by the host as timm does (GA/train.py:544-557,727-728):
 *   ga_mixup_batch:  fp32 NCHW x -> out (out of place): mixup out[b] = x[b]*lam + x[B-1-b]*(1-lam), or cutmix (x[B-1-b] inside
 *                    the box [yl,yh) x [xl,xh));
 *   ga_mixup_target: class indices -> dense [B][NC]: lam * onehot_s(t[b]) + (1-lam) * onehot_s(t[B-1-b]).
 * lam / smoothing are doubles: 1 - lam and the on / off values are formed in double and rounded to fp32 once, as torch does. */
int ga_mixup_batch(const float* x, float* out, int B, int CH, int H, int W, double lam, int cutmix, int yl, int yh, int xl, int xh,
                   ga_stream_t stream);
int ga_mixup_target(const int64_t* target, float* out, int B, int NC, double lam, double smoothing, ga_stream_t stream);
/* uint8 NCHW batch (timm fast_collate) -> fp32 NCHW, out = (x - mean[c]) / std[c]; mean / std are HOST arrays of CH <= 4 floats
 * (already scaled by 255 as timm's PrefetchLoader holds them, GA/train.py:567-595); H*W a multiple of 4 */
int ga_u8_normalize(const void* x, float* out, int B, int CH, int H, int W, const float* mean, const float* std, ga_stream_t stream);
/* adaptive gradient clipping (timm adaptive_clip_grad, clip_mode 'agc'): units = int64 {offset, length} pairs into the flat fp32
 * parameter / gradient buffers (a row of a >= 2-d parameter or a whole <= 1-d one); per unit
 * g *= max(|p|, eps) * clip_factor / max(|g|, 1e-6) where |g| exceeds max(|p|, eps) * clip_factor */
int ga_agc_clip(const float* params, float* grads, const int64_t* units, int nunits, float clip_factor, float eps, ga_stream_t stream);
int ga_gelu_fwd(const void* x, void* y, int64_t n, int dtype, ga_stream_t stream);
int ga_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, ga_stream_t stream);
int ga_relu_drop(const void* a, const float* mask, void* out, void* deriv, int64_t n, int dtype, ga_stream_t stream);
int ga_mask_mul(const void* x, const float* mask, const void* res, void* y, int64_t n, int dtype, ga_stream_t stream);
int ga_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int cols, int accumulate, int dtype,
              ga_stream_t stream);

/* DropPath masks of a training step (timm DropPath: ga_convnext.py:96,111; ga_cswin.py:181,209-210):
 *   out[s][b] = Bernoulli(keep[s]) / keep[s]  for `sites` stochastic-depth sites x B samples, from a counter-based
 *   generator keyed by (seed, *counter, element); *counter (device memory) is advanced by one per call. */
int ga_drop_path_sample(float* out, const float* keep, int sites, int B, uint64_t seed, uint64_t* counter, ga_stream_t stream);

/* nn.Dropout masks of the MAP head (map.py:54,82-83): out[i] = Bernoulli(keep) / keep, i < n; *counter advanced by one */
int ga_dropout_mask_sample(float* out, int64_t n, float keep, uint64_t seed, uint64_t* counter, ga_stream_t stream);

/* Fused MLP bodies of the narrow stages (bf16, C in {96, 192}, H = 4C; ga_mlp_supported tells): the [M][H] hidden activation
 * stays on chip.  Replaces the fc1 / fc2 pair of ga_gemm launches of a ConvNeXt Block (ga_convnext.py:86-101) and, backward,
 * the dgrad2 / dgrad1 pair:
 *   ga_mlp_fwd: Y = R + rowscale[m / rows_per_scale] * (gelu(X W1^T + b1) W2^T + b2)      (R, rowscale optional)
 *               W1 [H][ldw1] (rows = hidden), W2 [C][ldw2] (rows = output channel): the effective weights of ga_weight_prep;
 *   ga_mlp_bwd: Hd = X W1^T + b1 re-computed;  A = gelu(Hd) -> A [M][lda];  DH = (DY W2) * gelu'(Hd) -> DH [M][lddh];
 *               DX = DH W1 -> DX [M][lddx].  W2T [H][ldw2t] = W2 transposed (rows = hidden), W1T [C][ldw1t] = W1 transposed.
 *               The two weight-gradient GEMMs (ga_wgrad) then read A / DH exactly as they read the stored tensors before.
 * GELU is the tanh form the bf16 fc1 epilogue of ga_gemm uses (|error| <= 5e-4, below the bf16 rounding of the result). */
typedef struct {
    const void* X; int64_t ldx;
    const void* W1; int64_t ldw1; const float* b1;
    const void* W2; int64_t ldw2; const float* b2;
    const void* R; int64_t ldr;
    const float* rowscale; int rows_per_scale;
    void* Y; int64_t ldy;
    int64_t M; int C, H, dtype;
} ga_mlp_desc;
typedef struct {
    const void* X; int64_t ldx;
    const void* DY; int64_t lddy;
    const void* W1; int64_t ldw1; const float* b1;
    const void* W2T; int64_t ldw2t;
    const void* W1T; int64_t ldw1t;
    void* A; int64_t lda;
    void* DH; int64_t lddh;
    void* DX; int64_t lddx;
    int64_t M; int C, H, dtype;
} ga_mlp_bwd_desc;
int ga_mlp_supported(int C, int H, int dtype);
int ga_mlp_fwd(const ga_mlp_desc* d, ga_stream_t stream);
int ga_mlp_bwd(const ga_mlp_bwd_desc* d, ga_stream_t stream);

/* Alignment-free forms for the odd-width variants (ga_convnext_*_688, base_976: 86 / 172 / 122 / 244 channels per group are off
 * the 16-byte grid of the MFMA kernels).  The heads' grouped 1x1 convolutions act on one token per image (rows = batch):
 *   ga_small_linear_fwd: Y[r][g*Ng + n] = R + rowscale[r / rps] * col_scale[.] * (sum_k A[r][col(g*a_gstride + k)] W[g*Ng + n][k] + bias[.])
 *     A / Y / R / Yraw in `dtype`, element-wise loads (any alignment); W [groups*Ng][Kg], bias, col_scale: the fp32 MASTER
 *     parameters, no weight preparation; col(c) = a_perm ? a_perm[c] : c (the channel_shuffle of GroupConvMlp, ga_convnext.py:557-566);
 *     Yraw (optional, leading dimension ldy) keeps the value before col_scale for the col_scale gradient.
 *   ga_small_linear_bwd (same descriptor): dA (+)= dYeff W;  dW += dYeff^T A;  dbias += colsum(dYeff);  dcol_scale += colsum(dY * rowscale * Yraw)
 *     with dYeff = dY * rowscale * col_scale; any of dA / dW / dbias / dcol_scale may be NULL.
 *   ga_colstats: sum[c] += sum_r x[r][c], sumsq[c] += sum_r x^2 (the BatchNorm batch statistics ga_gemm's epilogue otherwise delivers).
 *   ga_pad_copy_f32: dst[r][0..cols) (+)= src[r][0..cols), row strides lds / ldd in elements: parameters of the 172 / 244-wide stage-4
 *     Bottleneck into zero-padded 176 / 248-wide buffers for the MFMA kernels, and their gradients back. */
typedef struct {
    int rows, groups, Ng, Kg;
    const void* A; int64_t lda; int64_t a_gstride; const int* a_perm;
    const float* W; const float* bias; const float* col_scale;
    const float* rowscale; int rows_per_scale;
    const void* R; int64_t ldr;
    void* Y; int64_t ldy; void* Yraw;
    int dtype;
} ga_small_linear_desc;
int ga_small_linear_fwd(const ga_small_linear_desc* d, ga_stream_t stream);
int ga_small_linear_bwd(const ga_small_linear_desc* d, const void* dY, void* dA, int accumulate_dA, float* dW, float* dbias,
                        float* dcol_scale, ga_stream_t stream);
int ga_colstats(const void* x, int64_t ld, int rows, int C, float* sum, float* sumsq, int dtype, ga_stream_t stream);
int ga_pad_copy_f32(const float* src, float* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd, int accumulate, ga_stream_t stream);
/* ConvNeXt stem in one pass (timm ConvNeXt stem / ga_convnext.py:431-434: Conv2d(3, C, 4, 4) + LayerNorm over the channels), bf16:
 *   x fp32 NCHW [B][3][H][W];  W bf16 [C][ldw], k = (c, ky, kx) as ga_weight_prep(stem = 1) writes it;  bias / gamma / beta fp32 [C];
 *   pre [B*H/4*W/4][C] = conv + bias (kept for ga_layernorm_bwd), y = LayerNorm(pre) * gamma + beta, mean / rstd per pixel.
 *   Equals ga_gemm(GA_A_STEM4_NCHW, bias) followed by ga_layernorm_fwd on the rounded `pre`.  C = 96 or 128. */
int ga_stem4_ln_fwd(const float* x, const void* W, int64_t ldw, const float* bias, const float* gamma, const float* beta, void* pre,
                    void* y, float* mean, float* rstd, int B, int H, int W_, int C, float eps, ga_stream_t stream);
/* grouped 1x1 convolution as ONE dense GEMM: the weight [R][cg] of a conv with ng groups (rg output rows and cg input channels per
 * group; R = a multiple of rg * ng: several such weights stacked) <-> its block-diagonal image [R][ld] (row r's values at columns
 * ((r / rg) % ng) * cg ..., zeros elsewhere, zero-initialised by the caller).  to_diag = 1 writes the block-diagonal side, 0 reads it back ((+)= with accumulate).  GA-CSWin's grouped gram_contraction
 * (ga_cswin.py:559-561: 8 groups of 64 -> 24 channels, five heads) runs as one 960 x 512 product instead of 40 products with N = 24. */
int ga_blockdiag_f32(const float* src, float* dst, int64_t R, int rg, int ng, int cg, int64_t ld, int to_diag, int accumulate,
                     ga_stream_t stream);
/* two-level group padding of an fp32 parameter matrix [R][C] <-> [R/RG*RGp][C/CG*CGp] (rows in groups of RG padded to RGp, columns
 * in groups of CG padded to CGp; the padded side is zero-initialised by the caller): unpad = 0 writes the padded matrix, unpad = 1
 * reads it back ((+)= with accumulate).  The grouped one-token layers of the odd-width variants (GroupConvMlp of ga_convnext_*_688 /
 * base_976, ga_convnext.py:190-222: 172 / 244 channels per group) run on the MFMA kernels over such copies. */
int ga_pad_groups_f32(const float* src, float* dst, int64_t R, int64_t C, int RG, int RGp, int CG, int CGp, int unpad, int accumulate,
                      ga_stream_t stream);
/* the same element-wise strided copy for activations in `dtype` (group compaction: [rows*groups][88] -> [rows*groups][86]) */
int ga_pad_copy(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd, int accumulate, int dtype,
                ga_stream_t stream);

/* Global multi-head self-attention (timm vision_transformer.Attention inside `Block`, MAP/models/map_pit.py:14,35-44):
 *   qkv [B*N][ldq]: q | k | v column blocks of width C = H*hd (head h = columns h*hd.. of each block); out [B*N][ldo];
 *   out = softmax(q k^T * scale) v per (image, head);  lse [B][H][N] fp32 = row log-sum-exp, kept for the backward pass.
 * bf16 with hd in {16, 32, 48, 64} runs flash-style on MFMA (64-query workgroups, 64-key blocks streamed through LDS, online softmax);
 * everything else (fp32 parity mode, other hd <= 128) on a plain fp32 form meant for small batches.
 * bwd: dqkv [B*N][ldq] = d(q | k | v) from dout [B*N][ldo]; needs `out` and `lse` of the forward pass and a caller-owned
 * workspace of ga_attn_bwd_workspace(d) bytes (delta[b][h][q] = dout . out). */
typedef struct {
    int B, N, H, hd;
    float scale;
    int dtype;
    const void* qkv; int64_t ldq;
    void* out; int64_t ldo;
    float* lse;
} ga_attn_desc;
int ga_attn_fwd(const ga_attn_desc* d, ga_stream_t stream);
size_t ga_attn_bwd_workspace(const ga_attn_desc* d);
int ga_attn_bwd(const ga_attn_desc* d, const void* dout, void* dqkv, void* workspace, size_t ws_bytes, ga_stream_t stream);
/* ViT stem (timm PatchEmbed + cls_token / pos_embed; the P x P stride-P patch convolution becomes a ga_gemm over patches):
 *   ga_patchify:      fp32 NCHW [B,CH,H,W] -> [B*(H/P)*(W/P)][CH*P*P] in `dtype`, k = (c, ky, kx) (the conv weight's flattened order)
 *   ga_vit_embed_fwd: x0[b][0] = cls + pos[0];  x0[b][1+p] = tok[b][p] + pos[1+p]      (cls [C], pos [Np+1][C] fp32)
 *   ga_vit_embed_bwd: dtok = dx0[b][1+p];  dcls += sum_b dx0[b][0];  dpos[t] += sum_b dx0[b][t] */
int ga_patchify(const float* x, void* out, int B, int CH, int H, int W, int P, int dtype, ga_stream_t stream);
int ga_vit_embed_fwd(const void* tok, const float* cls, const float* pos, void* x0, int B, int Np, int C, int dtype, ga_stream_t stream);
int ga_vit_embed_bwd(const void* dx0, void* dtok, float* dcls, float* dpos, int B, int Np, int C, int dtype, ga_stream_t stream);

/* Pooling transformer (PiT) pieces around the ViT blocks -- /root/reference/MAP/models/map_pit.py
 *   ga_patchify_strided: conv_embedding (:71-81) as im2col for a patch convolution whose stride S differs from the patch size P
 *                        (16 / 8 in map_pit_s: overlapping patches): [B*gh*gw][CH*P*P], gh = (H - P) / S + 1; P % 8 == 0, S % 4 == 0
 *   ga_pos_add_fwd/bwd:  x0[b][p] = tok[b][p] + pos[p]  (:190-191; pos fp32 [Np][C], the NCHW parameter transposed);
 *                        dpos[p] = sum_b dx0[b][p]  (overwrites)
 *   ga_dwpool_*:         conv_head_pooling (:58-68): depthwise 3 x 3 / stride 2 / pad 1 convolution with a channel multiplier
 *                        (groups = Cin, Cout = mult * Cin) on NHWC maps [B][H][W][Cin] -> [B][Ho][Wo][Cout], Ho = (H - 1) / 2 + 1;
 *                        w fp32 [Cout][9] (the (Cout, 1, 3, 3) parameter), bias fp32 [Cout].  bwd_weight ACCUMULATES into dw / db.
 *   ga_resize_concat_*:  bilinear resize (align_corners = False, no antialias; any Hin x Win -> Hout x Wout) of an NHWC map into
 *                        columns [c_off, c_off + C) of the MultiScale concat buffer (row stride ldd); map.py:322-333 reduces
 *                        PiT's 27 x 27 maps to 14 x 14 this way.  bwd writes dsrc (gather form, no atomics). */
int ga_patchify_strided(const float* x, void* out, int B, int CH, int H, int W, int P, int S, int dtype, ga_stream_t stream);
int ga_pos_add_fwd(const void* tok, const float* pos, void* x0, int B, int Np, int C, int dtype, ga_stream_t stream);
int ga_pos_add_bwd(const void* dx0, float* dpos, int B, int Np, int C, int dtype, ga_stream_t stream);
int ga_dwpool_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cin, int mult, int dtype,
                  ga_stream_t stream);
int ga_dwpool_bwd_data(const void* dy, const float* w, void* dx, int B, int H, int W, int Cin, int mult, int dtype, ga_stream_t stream);
int ga_dwpool_bwd_weight(const void* dy, const void* x, float* dw, float* db, int B, int H, int W, int Cin, int mult, int dtype,
                         ga_stream_t stream);
int ga_resize_concat_fwd(const void* src, void* dst, int B, int Hin, int Win, int C, int Hout, int Wout, int64_t ldd, int c_off,
                         int dtype, ga_stream_t stream);
int ga_resize_concat_bwd(const void* dcat, void* dsrc, int B, int Hin, int Win, int C, int Hout, int Wout, int64_t ldd, int c_off,
                         int dtype, ga_stream_t stream);
/* dst[b][p][:] = scale * src[b][:], p < HW: the gradient of the global average pool of the plain ConvNeXt head
 * (MAP/models/map_convnext.py:134-135, `x.mean([-2, -1])`; scale = 1 / HW) */
int ga_rows_bcast(const void* src, void* dst, int B, int HW, int C, float scale, int dtype, ga_stream_t stream);

/* small fp32 / elementwise utilities */
int ga_memset(void* p, int value, size_t bytes, ga_stream_t stream); /* hipMemsetAsync on `stream` */
int ga_transpose_f32(const float* in, float* out, int R, int C, int accumulate, ga_stream_t stream); /* out[c][r] (+)= in[r][c] */
int ga_axpy_f32(float* y, const float* x, float a, int64_t n, ga_stream_t stream);
int ga_lerp_f32(float* y, const float* x, float w, int64_t n, ga_stream_t stream); /* y += w*(x-y): ModelEmaV2 update, GA/train.py:499 */
int ga_rowscale(const void* x, const float* s, void* y, int64_t n, int64_t elems_per_scale, int dtype,
                ga_stream_t stream);
int ga_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, ga_stream_t stream);
int ga_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, ga_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gradient exchange over RCCL / xGMI.  replaces: NativeDDP(model, device_ids=[local_rank]) and its bucket reducer
 * (GA/train.py:514, MAP/train.py:576), the per-forward buffer broadcast (`broadcast_buffers`), timm distribute_bn
 * (GA/train.py:665-674).  One communicator per process (one process per GPU).  The 128-byte id comes from rank 0
 * (ga_comm_unique_id) and reaches the other ranks through the caller's side channel (torch.distributed's store in
 * imagenet_models_amd.comm).  librccl is resolved at run time: GA_ERR_UNSUPPORTED when it cannot be loaded.
 * Every call only enqueues on `stream`; buffers are caller-owned device memory.
 *   ga_allreduce_bucket:      grads[0..n) := scale * sum over ranks, in place.  wire_dtype GA_F32: ncclAllReduce on the slice;
 *                             GA_BF16: packed to bf16 in `workspace` (ga_allreduce_workspace(n, GA_BF16) bytes), reduced in
 *                             bf16, unpacked -- half the wire bytes, bf16 summation (an option, not the default).
 *   ga_reduce_scatter_bucket: shard[0..n_per_rank) := scale * sum over ranks of grads[rank*n_per_rank ..] (grads holds
 *                             world * n_per_rank elements); ga_allgather_bucket is its inverse.  Together they are the
 *                             all-reduce with room for a sharded (ZeRO-1) optimizer step in between.
 *   ga_comm_broadcast:        n_f32 floats from `root` (initial parameters, BatchNorm running statistics).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct ga_comm* ga_comm_t;
int ga_comm_unique_id(void* id128);
int ga_comm_init(ga_comm_t* comm, int rank, int world, const void* id128);
int ga_comm_destroy(ga_comm_t comm);
int ga_comm_info(ga_comm_t comm, int* rank, int* world);
size_t ga_allreduce_workspace(int64_t n, int wire_dtype);
int ga_allreduce_bucket(ga_comm_t comm, float* grads, int64_t n, int wire_dtype, float scale, void* workspace, size_t ws_bytes,
                        ga_stream_t stream);
int ga_reduce_scatter_bucket(ga_comm_t comm, const float* grads, float* shard, int64_t n_per_rank, float scale, ga_stream_t stream);
int ga_allgather_bucket(ga_comm_t comm, const float* shard, float* full, int64_t n_per_rank, ga_stream_t stream);
int ga_comm_broadcast(ga_comm_t comm, void* buf, int64_t n_f32, int root, ga_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAEXT_H */
