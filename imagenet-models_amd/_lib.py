"""ctypes binding of csrc/libgaext.so (C ABI declared in include/gaext.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAEXT_LIB: another build of the same library (A/B timing of kernel variants on ONE box; boxes differ by several %)
LIB_PATH = os.environ.get('GAEXT_LIB') or os.path.join(_HERE, 'csrc', 'libgaext.so')

GA_F32, GA_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
A_PLAIN, A_PATCH2, A_STEM4_NCHW, A_CONV3, A_CONV3S2, A_NEIGH2 = 0, 1, 2, 3, 4, 5
C_PLAIN, C_UNPATCH2 = 0, 1

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ('M', i32), ('N', i32), ('K', i32), ('batch', i32), ('dtype', i32),
        ('A', vp), ('lda', i64), ('strideA', i64), ('a_batch_mod', i32), ('a_kind', i32),
        ('a_H', i32), ('a_W', i32), ('a_C', i32), ('a_act', i32),
        ('B', vp), ('ldb', i64), ('strideB', i64),
        ('C', vp), ('ldc', i64), ('strideC', i64), ('c_kind', i32),
        ('c_H', i32), ('c_W', i32), ('c_C', i32), ('c_f32', i32), ('C2', vp), ('c2_mode', i32),
        ('alpha', f32), ('bias', vp), ('strideBias', i64), ('act', i32),
        ('H', vp), ('ldh', i64), ('strideH', i64), ('h_is_deriv', i32),
        ('rowscale', vp), ('rows_per_scale', i32),
        ('R', vp), ('ldr', i64), ('strideR', i64), ('relu_after', i32),
        ('colsum', vp), ('colsumsq', vp), ('strideCol', i64),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ('M', i32), ('N', i32), ('K', i32), ('batch', i32), ('dtype', i32),
        ('Y', vp), ('ldy', i64), ('strideY', i64),
        ('X', vp), ('ldx', i64), ('strideX', i64), ('x_batch_mod', i32), ('x_kind', i32),
        ('x_H', i32), ('x_W', i32), ('x_C', i32), ('x_act', i32),
        ('dW', vp), ('ldw', i64), ('strideW', i64),
        ('dbias', vp), ('strideDbias', i64),
        ('alpha', f32), ('split_m', i32), ('accumulate', i32), ('workspace', vp), ('ws_bytes', i64),
    ]


class WprepDesc(C.Structure):
    _fields_ = [
        ('w', vp), ('G', i32), ('Co', i32), ('Ci', i32), ('KH', i32), ('KW', i32),
        ('rs', vp), ('cs', vp), ('row_perm', vp), ('dtype', i32),
        ('out', vp), ('ldo', i64), ('outT', vp), ('ldt', i64), ('flip', i32), ('stem', i32), ('t_cols', i32),
    ]


class WunfoldDesc(C.Structure):
    _fields_ = [
        ('G', vp), ('ldg', i64), ('gb', vp), ('W', vp), ('b', vp), ('rs', vp), ('cs', vp), ('v', vp), ('row_perm', vp),
        ('N', i32), ('Ci', i32), ('KH', i32), ('KW', i32), ('stem', i32),
        ('dW', vp), ('db', vp), ('d_rs', vp), ('d_cs', vp), ('d_v', vp),
    ]


class SmallDesc(C.Structure):
    _fields_ = [('kind', i32), ('y', vp), ('x', vp), ('a', f32), ('n', i64), ('R', i32), ('C', i32), ('accumulate', i32),
                ('b', vp), ('rs', vp), ('v', vp), ('row_perm', vp)]


class CswinAttnDesc(C.Structure):
    _fields_ = [('B', i32), ('reso', i32), ('C', i32), ('heads', i32), ('nbranch', i32),
                ('Hs', i32 * 2), ('Ws', i32 * 2), ('lepe_w', vp * 2), ('lepe_b', vp * 2),
                ('scale', f32), ('dtype', i32), ('qkv', vp), ('ldq', i64), ('out', vp), ('ldo', i64)]


class MlpDesc(C.Structure):
    _fields_ = [('X', vp), ('ldx', i64), ('W1', vp), ('ldw1', i64), ('b1', vp), ('W2', vp), ('ldw2', i64), ('b2', vp),
                ('R', vp), ('ldr', i64), ('rowscale', vp), ('rows_per_scale', i32), ('Y', vp), ('ldy', i64),
                ('M', i64), ('C', i32), ('H', i32), ('dtype', i32)]


class MlpBwdDesc(C.Structure):
    _fields_ = [('X', vp), ('ldx', i64), ('DY', vp), ('lddy', i64), ('W1', vp), ('ldw1', i64), ('b1', vp),
                ('W2T', vp), ('ldw2t', i64), ('W1T', vp), ('ldw1t', i64), ('A', vp), ('lda', i64), ('DH', vp), ('lddh', i64),
                ('DX', vp), ('lddx', i64), ('M', i64), ('C', i32), ('H', i32), ('dtype', i32)]


class SmallLinearDesc(C.Structure):
    _fields_ = [('rows', i32), ('groups', i32), ('Ng', i32), ('Kg', i32), ('A', vp), ('lda', i64), ('a_gstride', i64), ('a_perm', vp),
                ('W', vp), ('bias', vp), ('col_scale', vp), ('rowscale', vp), ('rows_per_scale', i32), ('R', vp), ('ldr', i64),
                ('Y', vp), ('ldy', i64), ('Yraw', vp), ('dtype', i32)]


class AttnDesc(C.Structure):
    _fields_ = [('B', i32), ('N', i32), ('H', i32), ('hd', i32), ('scale', f32), ('dtype', i32), ('qkv', vp), ('ldq', i64),
                ('out', vp), ('ldo', i64), ('lse', vp)]


_SIGS = {
    'ga_version': ([], i32),
    'ga_last_error': ([C.c_char_p, C.c_size_t], i32),
    'ga_device_info': ([C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)], i32),
    'ga_comm_unique_id': ([vp], i32),
    'ga_comm_init': ([C.POINTER(vp), i32, i32, vp], i32),
    'ga_comm_destroy': ([vp], i32),
    'ga_comm_info': ([vp, C.POINTER(i32), C.POINTER(i32)], i32),
    'ga_allreduce_workspace': ([i64, i32], C.c_size_t),
    'ga_allreduce_bucket': ([vp, vp, i64, i32, f32, vp, C.c_size_t, vp], i32),
    'ga_reduce_scatter_bucket': ([vp, vp, vp, i64, f32, vp], i32),
    'ga_allgather_bucket': ([vp, vp, vp, i64, vp], i32),
    'ga_comm_broadcast': ([vp, vp, i64, i32, vp], i32),
    'ga_set_knob': ([C.c_char_p, i32], i32),
    'ga_unset_knob': ([C.c_char_p], i32),
    'ga_config_string': ([C.c_char_p, C.c_size_t], i32),
    'ga_gemm': ([C.POINTER(GemmDesc), vp], i32),
    'ga_wgrad': ([C.POINTER(WgradDesc), vp], i32),
    'ga_weight_prep': ([C.POINTER(WprepDesc), vp], i32),
    'ga_bias_fold': ([vp, vp, vp, vp, vp, vp, i32, i32, vp], i32),
    'ga_weight_unfold': ([C.POINTER(WunfoldDesc), vp], i32),
    'ga_dwconv7_fwd': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_dwconv7_bwd_data': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_dwconv7_bwd_data2': ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_dwconv7_bwd_weight': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, C.c_size_t, vp], i32),
    'ga_dwconv7_bwd_weight_workspace': ([i32, i32, i32, i32, i32], C.c_size_t),
    'ga_wgrad_workspace': ([C.POINTER(WgradDesc)], C.c_size_t),
    'ga_layernorm_fwd': ([vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp], i32),
    'ga_layernorm_bwd': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp], i32),
    'ga_layernorm_bwd_dp': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, i64, i32, vp], i32),
    'ga_bn_finalize': ([vp, vp, i64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, i32, i32, vp], i32),
    'ga_affine_act': ([vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i64, vp], i32),
    'ga_bn_bwd_reduce': ([vp, vp, vp, vp, vp, vp, i64, vp, vp, i64, i32, i32, i64, vp], i32),
    'ga_bn_bwd_apply': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp, i64, i32, i32, i64, i64, vp], i32),
    'ga_pool_concat_fwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_pool_concat_bwd': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_spatial_sum': ([vp, vp, vp, i32, i32, i32, f32, i32, vp], i32),
    'ga_se_mlp_fwd': ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp], i32),
    'ga_se_mlp_bwd': ([vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, vp], i32),
    'ga_chan_scale': ([vp, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_gram_pack_fwd': ([vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_gram_pack_bwd': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_token_cat': ([vp, vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_token_split': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_class_attn_fwd': ([vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_class_attn_bwd': ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_class_attn_fwd2': ([vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_class_attn_bwd2': ([vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_loss_fwd_bwd': ([vp, vp, vp, vp, i32, i32, i32, f32, i32, f32, f32, i32, vp], i32),
    'ga_heads_topk': ([vp, i32, i32, i32, i32, vp, vp, vp], i32),
    'ga_sgd_step': ([vp, vp, vp, vp, i64, i32, f32, vp], i32),
    'ga_adamw_step': ([vp, vp, vp, vp, vp, i64, f32, vp], i32),
    'ga_weight_prep_batch': ([vp, i32, vp], i32),
    'ga_weight_unfold_batch': ([vp, i32, vp], i32),
    'ga_small_batch': ([vp, i32, vp], i32),
    'ga_cswin_attn_fwd': ([C.POINTER(CswinAttnDesc), vp], i32),
    'ga_cswin_attn_bwd_workspace': ([C.POINTER(CswinAttnDesc)], C.c_size_t),
    'ga_cswin_attn_bwd': ([C.POINTER(CswinAttnDesc), vp, vp, vp, C.c_size_t, vp], i32),
    'ga_cswin_lepe_wgrad_reduce': ([C.POINTER(CswinAttnDesc), vp, vp, vp, vp, vp, vp], i32),
    'ga_cswin_lepe_wgrad': ([C.POINTER(CswinAttnDesc), vp, vp, vp, vp, vp, vp], i32),
    'ga_layernorm_gelu_fwd': ([vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp], i32),
    'ga_layernorm_gelu_bwd': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp], i32),
    'ga_nchw3_to_nhwc8': ([vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_convw_pack': ([vp, vp, i32, i32, i32, i32, i64, i32, vp], i32),
    'ga_convw_unpack_grad': ([vp, vp, i32, i32, i32, i32, i64, vp], i32),
    'ga_conv3s2_dgrad_prep': ([vp, vp, i32, i32, i64, i32, vp], i32),
    'ga_gram_pack_fwd2': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_gram_pack_bwd2': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_map_tokens_fwd': ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_map_tokens_bwd': ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    'ga_class_attn_mt_fwd': ([vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_class_attn_mt_bwd': ([vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_class_attn_mt_ia_fwd': ([vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp], i32),
    'ga_class_attn_mt_ia_bwd': ([vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32,
                                 vp], i32),
    'ga_map_loss_fwd_bwd': ([vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, f32, f32, i32, vp], i32),
    'ga_gelu_fwd': ([vp, vp, i64, i32, vp], i32),
    'ga_gelu_bwd': ([vp, vp, vp, i64, i32, vp], i32),
    'ga_relu_drop': ([vp, vp, vp, vp, i64, i32, vp], i32),
    'ga_mask_mul': ([vp, vp, vp, vp, i64, i32, vp], i32),
    'ga_copy2d': ([vp, i64, vp, i64, i64, i32, i32, i32, vp], i32),
    'ga_dropout_mask_sample': ([vp, i64, f32, C.c_uint64, vp, vp], i32),
    'ga_drop_path_sample': ([vp, vp, i32, i32, C.c_uint64, vp, vp], i32),
    'ga_loss_dense_fwd_bwd': ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, f32, f32, f32, i32, vp], i32),
    'ga_u8_normalize': ([vp, vp, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp], i32),
    'ga_mixup_batch': ([vp, vp, i32, i32, i32, i32, C.c_double, i32, i32, i32, i32, i32, vp], i32),
    'ga_mixup_target': ([vp, vp, i32, i32, C.c_double, C.c_double, vp], i32),
    'ga_agc_clip': ([vp, vp, vp, i32, f32, f32, vp], i32),
    'ga_small_linear_fwd': ([C.POINTER(SmallLinearDesc), vp], i32),
    'ga_small_linear_bwd': ([C.POINTER(SmallLinearDesc), vp, vp, i32, vp, vp, vp, vp], i32),
    'ga_colstats': ([vp, i64, i32, i32, vp, vp, i32, vp], i32),
    'ga_pad_copy_f32': ([vp, vp, i64, i64, i64, i64, i32, vp], i32),
    'ga_pad_copy': ([vp, vp, i64, i64, i64, i64, i32, i32, vp], i32),
    'ga_stem4_ln_fwd': ([vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp], i32),
    'ga_blockdiag_f32': ([vp, vp, i64, i32, i32, i32, i64, i32, i32, vp], i32),
    'ga_pad_groups_f32': ([vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_attn_fwd': ([C.POINTER(AttnDesc), vp], i32),
    'ga_attn_bwd_workspace': ([C.POINTER(AttnDesc)], C.c_size_t),
    'ga_attn_bwd': ([C.POINTER(AttnDesc), vp, vp, vp, C.c_size_t, vp], i32),
    'ga_patchify': ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_vit_embed_fwd': ([vp, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_vit_embed_bwd': ([vp, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_patchify_strided': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_pos_add_fwd': ([vp, vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_pos_add_bwd': ([vp, vp, i32, i32, i32, i32, vp], i32),
    'ga_dwpool_fwd': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_dwpool_bwd_data': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_dwpool_bwd_weight': ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ga_resize_concat_fwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, i32, vp], i32),
    'ga_resize_concat_bwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, i32, vp], i32),
    'ga_rows_bcast': ([vp, vp, i32, i32, i32, f32, i32, vp], i32),
    'ga_mlp_supported': ([i32, i32, i32], i32),
    'ga_mlp_fwd': ([C.POINTER(MlpDesc), vp], i32),
    'ga_mlp_bwd': ([C.POINTER(MlpBwdDesc), vp], i32),
    'ga_memset': ([vp, i32, C.c_size_t, vp], i32),
    'ga_transpose_f32': ([vp, vp, i32, i32, i32, vp], i32),
    'ga_axpy_f32': ([vp, vp, f32, i64, vp], i32),
    'ga_lerp_f32': ([vp, vp, f32, i64, vp], i32),
    'ga_sumsq_f32': ([vp, i64, vp, vp], i32),
    'ga_lamb_stage1': ([vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp], i32),
    'ga_lamb_stage2': ([vp, vp, vp, vp, i32, vp, vp], i32),
    'ga_clip_grad_f32': ([vp, i64, vp, f32, i32, vp], i32),
    'ga_rowscale': ([vp, vp, vp, i64, i64, i32, vp], i32),
    'ga_cast_from_f32': ([vp, vp, i64, i32, vp], i32),
    'ga_cast_to_f32': ([vp, vp, i64, i32, vp], i32),
}

_lib = None


def load():
    """Load libgaext.so (built by __graft_entry__.build() / `make -C csrc`). Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'libgaext.so not found at {LIB_PATH}: build it with `make -C {os.path.dirname(LIB_PATH)}` '
                           f'(there is no CPU fallback for the product path)')
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def last_error():
    buf = C.create_string_buffer(512)
    load().ga_last_error(buf, 512)
    return buf.value.decode(errors='replace')


def config_string():
    """library version + every tuning knob that is not at its default (include/gaext.h: ga_config_string)"""
    buf = C.create_string_buffer(1024)
    load().ga_config_string(buf, 1024)
    s = buf.value.decode(errors='replace')
    # the host-side scheduling switches the engines read from the environment when they are built (engine.py: lanes, chains)
    host = sorted(k for k in os.environ if k.startswith(('GAEXT_', 'GA_FUSED_MLP')) and k != 'GAEXT_LIB')
    known = {'GAEXT_ASYNC_WGRAD', 'GAEXT_FWD_SPLIT', 'GAEXT_PAR_BRANCH', 'GAEXT_FWD_SKEW', 'GAEXT_FUSE_DP', 'GAEXT_HEAD_STREAMS',
             'GA_FUSED_MLP', 'GAEXT_SMALL_SINGLE', 'GAEXT_SYNC_DEBUG'}
    for k in host:
        if k in known:
            s += f' host:{k}={os.environ[k]}'
    if os.environ.get('GAEXT_LIB'):
        s += f' lib={os.environ["GAEXT_LIB"]}'
    return s


class knobs:
    """context manager: `with knobs(NT_DMA=2, NT_PP=0): ...` sets library tuning knobs (ga_set_knob) and restores the defaults
    (ga_unset_knob) on exit -- the kernel-form switches the tests and A/B tools use; the environment is read once only"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        lib = load()
        for k, v in self.kv.items():
            check(lib.ga_set_knob(k.encode(), int(v)), f'ga_set_knob({k})')
        return self

    def __exit__(self, *exc):
        lib = load()
        for k in self.kv:
            lib.ga_unset_knob(k.encode())
        return False


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'libgaext {what} failed (code {rc}): {last_error()}')


def exported_symbols():
    return sorted(_SIGS)
