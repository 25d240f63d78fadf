"""GPU: the GA-CSWin kernels through the C ABI (ops.Plan, eager) against CPU references:
  * stripe attention + LePE fwd / bwd / LePE weight gradient vs the oracle restatement of LePEAttention
    (oracle.ga_cswin_oracle.lepe_attention, pinned against /root/reference/GA/ga_cswin.py:59-136 by
    tests/golden/cswin_modules.npz) and vs the committed reference vectors themselves;
  * the 3x3 / stride-2 conv gather (GA_A_CONV3S2), its data gradient (GA_A_NEIGH2 + GA_C_UNPATCH2) and weight
    gradient vs F.conv2d + autograd (deep stem / Merge_Block, ga_cswin.py:253-268,462-477).
Tolerances: fp32 2e-4, bf16 2e-2 of the tensor's max."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from test_kernels_gpu import assert_close, gen, rnd, tol

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _imp():
    from imagenet_models_amd import ops
    return ops


def _oracle():
    from oracle import ga_cswin_oracle as O
    return O


def _attn_case(dt, B, reso, C, heads, stripes, seed, force_simple=False):
    """one CSWinBlock attention (1 or 2 branches) fwd + bwd + lepe wgrad on the GPU vs the oracle on the CPU"""
    ops, O = _imp(), _oracle()
    g = gen(seed)
    L = reso * reso
    nb = len(stripes)
    cb = C // nb
    qkv_c, qkv_g = rnd((B * L, 3 * C), dt, g)
    do_c, do_g = rnd((B * L, C), dt, g)
    lw = [(torch.randn(cb, 1, 3, 3, generator=g) / 3).contiguous() for _ in range(nb)]
    lb = [torch.randn(cb, generator=g) * 0.1 for _ in range(nb)]
    hd = C // heads
    # ---- CPU reference
    q3 = qkv_c.reshape(B, L, 3, C).permute(2, 0, 1, 3).clone().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in lw]
    bs = [b.clone().requires_grad_(True) for b in lb]
    outs = []
    for i, (hs, ws_) in enumerate(stripes):
        sl = slice(i * cb, (i + 1) * cb)
        if nb == 1:
            idx, split = -1, reso
        elif hs == reso:
            idx, split = 0, ws_
        else:
            idx, split = 1, hs
        outs.append(O.lepe_attention(q3[0][:, :, sl], q3[1][:, :, sl], q3[2][:, :, sl], ws[i], bs[i], reso, idx, split,
                                     heads // nb))
    ref = torch.cat(outs, dim=2).reshape(B * L, C)
    ref.backward(do_c)
    dqkv_ref = q3.grad.permute(1, 2, 0, 3).reshape(B * L, 3 * C)
    # ---- GPU
    if force_simple:
        from imagenet_models_amd import _lib
        _lib.load().ga_set_knob(b'CSWIN_MFMA', 0)
    out = torch.empty(B * L, C, dtype=dt, device='cuda')
    dqkv = torch.zeros(B * L, 3 * C, dtype=dt, device='cuda')
    lwg = [w.cuda() for w in lw]
    lbg = [b.cuda() for b in lb]
    dw = [torch.zeros_like(w) for w in lwg]
    db = [torch.zeros_like(b) for b in lbg]
    p = ops.Plan(eager=True)
    d = p.cswin_desc(qkv_g, out, B, reso, C, heads, stripes, list(zip(lwg, lbg)), hd ** -0.5, ops.ga_dtype(dt))
    p.cswin_attn_fwd(d)
    need = ops.cswin_attn_bwd_workspace(d)
    if need:                                             # MFMA form: the LePE weight gradient comes out of the same kernel
        lws = torch.empty(need // 4, device='cuda')
        p.cswin_attn_bwd(d, do_g, dqkv, lepe_ws=lws)
        p.cswin_lepe_wgrad_reduce(d, lws, list(zip(dw, db)))
        dw2 = [torch.zeros_like(w) for w in lwg]
        db2 = [torch.zeros_like(b) for b in lbg]
        p.cswin_lepe_wgrad(d, do_g, list(zip(dw2, db2)))  # ... and must agree with the unfused kernel
        torch.cuda.synchronize()
        for i in range(nb):
            assert_close(dw[i], dw2[i].cpu(), 1e-3, f'fused vs unfused lepe dw{i}')
            assert_close(db[i], db2[i].cpu(), 1e-3, f'fused vs unfused lepe db{i}')
    else:
        p.cswin_attn_bwd(d, do_g, dqkv)
        p.cswin_lepe_wgrad(d, do_g, list(zip(dw, db)))
    torch.cuda.synchronize()
    t = tol(dt)
    assert_close(out, ref, t, 'attn out')
    for j, nm in enumerate('qkv'):
        assert_close(dqkv[:, j * C:(j + 1) * C], dqkv_ref[:, j * C:(j + 1) * C], t * 1.5, f'd{nm}')
    for i in range(nb):
        assert_close(dw[i], ws[i].grad, t * 2, f'lepe dw{i}')
        assert_close(db[i], bs[i].grad, t * 2, f'lepe db{i}')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('case', [
    # (B, reso, C, heads, stripes)
    (2, 14, 64, 2, [(14, 7), (7, 14)]),      # head_dim 32: the MFMA form in bf16 (7 tiles, N = 98)
    (2, 7, 64, 2, [(7, 7)]),                 # last stage: one branch, N = 49 (4 tiles)
    (1, 28, 128, 4, [(28, 2), (2, 28)]),     # stage 2 shape: N = 56, 2 heads per branch
    (1, 56, 64, 2, [(56, 1), (1, 56)]),      # stage 1 shape: N = 56 column / row stripes, 1 head per branch
    (2, 14, 32, 4, [(14, 7), (7, 14)]),      # head_dim 8 (generic form)
    (2, 14, 96, 6, [(14, 7), (7, 14)]),      # head_dim 16
])
def test_stripe_attention_vs_oracle(dt, case):
    _attn_case(dt, *case, seed=11)


def test_stripe_attention_generic_form_bf16_hd32():
    """the generic (non-MFMA) kernel in bf16 at head_dim 32, so both forms are pinned on the same shape"""
    try:
        _attn_case(torch.bfloat16, 2, 14, 64, 2, [(14, 7), (7, 14)], seed=12, force_simple=True)
    finally:
        from imagenet_models_amd import _lib
        _lib.load().ga_unset_knob(b'CSWIN_MFMA')


@pytest.mark.parametrize('name', ['lepe_v', 'lepe_h', 'lepe_full', 'lepe_s1', 'lepe_s2h'])
def test_stripe_attention_vs_reference_vectors(name):
    """fp32 kernels against the vectors the REAL LePEAttention produced (tests/golden/cswin_modules.npz)"""
    ops = _imp()
    z = np.load(os.path.join(GOLDEN, 'cswin_modules.npz'))
    reso, idx, split, dim, heads = [int(v) for v in z[f'{name}.cfg']]
    g = torch.Generator().manual_seed(4321 + len(name))
    qkv = torch.randn(3, 2, reso * reso, dim, generator=g)
    g2 = torch.Generator().manual_seed(4321 + 100 + len(name))
    gy = torch.randn(2, reso * reso, dim, generator=g2)
    B, L = 2, reso * reso
    stripe = (reso, reso) if idx == -1 else ((reso, split) if idx == 0 else (split, reso))
    mat = qkv.permute(1, 2, 0, 3).reshape(B * L, 3 * dim).contiguous().cuda()
    out = torch.empty(B * L, dim, device='cuda')
    dqkv = torch.zeros(B * L, 3 * dim, device='cuda')
    w = torch.from_numpy(z[f'{name}.w']).cuda()
    b = torch.from_numpy(z[f'{name}.b']).cuda()
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    p = ops.Plan(eager=True)
    d = p.cswin_desc(mat, out, B, reso, dim, heads, [stripe], [(w, b)], (dim // heads) ** -0.5, ops.GA_F32)
    p.cswin_attn_fwd(d)
    p.cswin_attn_bwd(d, gy.reshape(B * L, dim).cuda(), dqkv)
    p.cswin_lepe_wgrad(d, gy.reshape(B * L, dim).cuda(), [(dw, db)])
    assert_close(out.reshape(B, L, dim), torch.from_numpy(z[f'{name}.y']), 2e-4, 'y')
    want = torch.from_numpy(z[f'{name}.dqkv']).permute(1, 2, 0, 3).reshape(B * L, 3 * dim)
    assert_close(dqkv, want, 3e-4, 'dqkv')
    assert_close(dw, torch.from_numpy(z[f'{name}.dw']), 3e-4, 'dw')
    assert_close(db, torch.from_numpy(z[f'{name}.db']), 3e-4, 'db')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('shape', [(2, 16, 16, 16, 24), (2, 28, 28, 32, 64), (1, 14, 14, 64, 128)])
def test_conv3x3_stride2_fwd_dgrad_wgrad(dt, shape):
    """Merge_Block conv (ga_cswin.py:256): NHWC gather GEMM, transposed-conv data gradient, weight gradient"""
    ops = _imp()
    B, H, W, Ci, Co = shape
    OH, OW = H // 2, W // 2
    g = gen(5)
    x_c, x_g = rnd((B, H, W, Ci), dt, g)
    w_c = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(dt).float()
    bias = torch.randn(Co, generator=g)
    gy_c, gy_g = rnd((B * OH * OW, Co), dt, g)
    xr = x_c.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w_c.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, bias, stride=2, padding=1)
    y.backward(gy_c.reshape(B, OH, OW, Co).permute(0, 3, 1, 2))
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    Wf = torch.empty(Co, 9 * Ci, dtype=dt, device='cuda')
    p.convw_pack(w_c.cuda(), Wf, Co, Ci, 9, Ci, 9 * Ci, ga)
    out = torch.empty(B * OH * OW, Co, dtype=dt, device='cuda')
    p.gemm(x_g, Wf, out, B * OH * OW, Co, 9 * Ci, ga, a_kind=ops.A_CONV3S2, a_dims=(H, W, Ci), bias=bias.cuda())
    assert_close(out, y.permute(0, 2, 3, 1).reshape(-1, Co), tol(dt), 'conv3s2 fwd')
    # data gradient
    Bt = torch.empty(4 * Ci, 4 * Co, dtype=dt, device='cuda')
    p.conv3s2_dgrad_prep(w_c.cuda(), Bt, Co, Ci, 4 * Co, ga)
    dx = torch.empty(B * H * W, Ci, dtype=dt, device='cuda')
    p.gemm(gy_g, Bt, dx, B * OH * OW, 4 * Ci, 4 * Co, ga, a_kind=ops.A_NEIGH2, a_dims=(OH, OW, Co), c_kind=ops.C_UNPATCH2,
           c_dims=(H, W, Ci))
    assert_close(dx, xr.grad.permute(0, 2, 3, 1).reshape(-1, Ci), tol(dt), 'conv3s2 dgrad')
    # weight gradient
    G = torch.zeros(Co, 9 * Ci, device='cuda')
    dbias = torch.zeros(Co, device='cuda')
    p.wgrad(gy_g, x_g, G, B * OH * OW, Co, 9 * Ci, ga, x_kind=ops.A_CONV3S2, x_dims=(H, W, Ci), dbias=dbias)
    dW = torch.zeros(Co, Ci, 3, 3, device='cuda')
    p.convw_unpack_grad(G, dW, Co, Ci, 9, Ci, 9 * Ci)
    assert_close(dW, wr.grad, tol(dt, 2), 'conv3s2 wgrad')
    assert_close(dbias, gy_c.sum(0), tol(dt, 2), 'conv3s2 dbias')


@pytest.mark.parametrize('shape', [(2, 28, 28, 32, 64), (1, 14, 14, 64, 128), (3, 56, 56, 64, 64), (8, 112, 112, 64, 64), (5, 14, 30, 96, 32)])
def test_conv3x3_stride2_dgrad_on_the_ring_form(shape, knobs):
    """data gradient of the 3 x 3 / stride-2 convs (ga_cswin.py:256,470) with the 2 x 2 neighbourhood rows (GA_A_NEIGH2) fetched by the
    3-slot ring form: two runs of 2C elements one map row apart, neighbours beyond the right / lower edge as zeros -- against
    F.conv2d's gradient and bit-for-bit against the register-staged gather form"""
    ops = _imp()
    B, H, W, Ci, Co = shape
    OH, OW = H // 2, W // 2
    dt = torch.bfloat16
    g = gen(15)
    w_c = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(dt).float()
    gy_c, gy_g = rnd((B * OH * OW, Co), dt, g)
    xr = torch.zeros(B, Ci, H, W, requires_grad=True)
    F.conv2d(xr, w_c, None, stride=2, padding=1).backward(gy_c.reshape(B, OH, OW, Co).permute(0, 3, 1, 2))
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    Bt = torch.empty(4 * Ci, 4 * Co, dtype=dt, device='cuda')
    p.conv3s2_dgrad_prep(w_c.cuda(), Bt, Co, Ci, 4 * Co, ga)
    outs = []
    for ring in (1, 0):
        knobs(NT_R3=15 if ring else 0)
        dx = torch.full((B * H * W, Ci), 7.0, dtype=dt, device='cuda')
        p.gemm(gy_g, Bt, dx, B * OH * OW, 4 * Ci, 4 * Co, ga, a_kind=ops.A_NEIGH2, a_dims=(OH, OW, Co), c_kind=ops.C_UNPATCH2,
               c_dims=(H, W, Ci))
        assert_close(dx, xr.grad.permute(0, 2, 3, 1).reshape(-1, Ci), tol(dt), f'conv3s2 dgrad ring={ring}')
        outs.append(dx.float().cpu())
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-2 * outs[1].abs().max().item()


@pytest.mark.parametrize('shape', [(2, 28, 28, 32, 64), (1, 14, 14, 64, 128), (3, 56, 56, 64, 64), (8, 112, 112, 64, 64), (5, 14, 30, 96, 40),
                                   (2, 4, 4, 32, 16)])
def test_conv3x3_stride2_forward_on_the_ring_form(shape, knobs):
    """the 3 x 3 / stride-2 convs (ga_cswin.py:256,470; GA_A_CONV3S2) with their 9C-element rows fetched by the 3-slot ring form as
    three runs of 3C elements (buffer base moved W + 1 pixels down, lane offset at the centre tap, taps left of / above the map
    out of range) -- against F.conv2d and the register-staged gather form"""
    ops = _imp()
    B, H, W, Ci, Co = shape
    OH, OW = H // 2, W // 2
    dt = torch.bfloat16
    g = gen(16)
    x_c, x_g = rnd((B, H, W, Ci), dt, g)
    w_c = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(dt).float()
    bias = torch.randn(Co, generator=g)
    y = F.conv2d(x_c.permute(0, 3, 1, 2), w_c, bias, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    Wf = torch.empty(Co, 9 * Ci, dtype=dt, device='cuda')
    p.convw_pack(w_c.cuda(), Wf, Co, Ci, 9, Ci, 9 * Ci, ga)
    outs = []
    for ring in (1, 0):
        knobs(NT_R3=15 if ring else 0)
        out = torch.full((B * OH * OW, Co), 7.0, dtype=dt, device='cuda')
        p.gemm(x_g, Wf, out, B * OH * OW, Co, 9 * Ci, ga, a_kind=ops.A_CONV3S2, a_dims=(H, W, Ci), bias=bias.cuda())
        assert_close(out, y, tol(dt), f'conv3s2 fwd ring={ring}')
        outs.append(out.float().cpu())
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-2 * outs[1].abs().max().item()


@pytest.mark.parametrize('geom', [(2, 32, 32), (1, 16, 48), (3, 112, 112), (40, 112, 112)])     # (the last: more tiles than workgroups)
def test_conv3x3_stride2_wgrad_direct_form_64_channels(geom, knobs):
    """weight gradient of the stem's Conv2d(64, 64, 3, 2, 1) (ga_cswin.py:470) on the direct kernel (8 x 8 output tiles, 17 x 17 halo,
    transposing LDS reads; csrc/conv3.hip) against F.conv2d's gradient and against the gather form of gemm_tn; accumulates into dW"""
    ops = _imp()
    B, H, W = geom
    Ci = Co = 64
    OH, OW = H // 2, W // 2
    dt = torch.bfloat16
    g = gen(18)
    x_c, x_g = rnd((B, H, W, Ci), dt, g)
    gy_c, gy_g = rnd((B * OH * OW, Co), dt, g)
    wr = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x_c.permute(0, 3, 1, 2), wr, None, stride=2, padding=1).backward(gy_c.reshape(B, OH, OW, Co).permute(0, 3, 1, 2))
    ga = ops.ga_dtype(dt)
    outs = []
    for direct in (1, 0):
        knobs(CONV3_DIRECT=direct)
        p = ops.Plan(eager=True)
        G = torch.full((Co, 9 * Ci), 0.5, device='cuda')            # the launch ADDS its result
        p.wgrad(gy_g, x_g, G, B * OH * OW, Co, 9 * Ci, ga, x_kind=ops.A_CONV3S2, x_dims=(H, W, Ci))
        dW = torch.zeros(Co, Ci, 3, 3, device='cuda')
        p.convw_unpack_grad(G - 0.5, dW, Co, Ci, 9, Ci, 9 * Ci)
        assert_close(dW, wr.grad, tol(dt, 2), f'conv3s2 wgrad direct={direct}')
        outs.append(dW.cpu())
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-2 * outs[1].abs().max().item()


@pytest.mark.parametrize('geom', [(2, 32, 32), (1, 20, 36), (3, 224, 224), (1, 2, 2), (1, 16, 96), (20, 224, 224)])
def test_stem_first_conv_direct_form_64_channels(geom, knobs):
    """stage1_conv_embed.0 at its real width (3 -> 64, 3x3 s2, no bias, ga_cswin.py:464): the direct kernel (one NHWC8 pixel per lane
    and tap row, weights in registers) against F.conv2d and against the gather GEMM"""
    ops = _imp()
    B, H, W = geom
    Co = 64
    dt = torch.bfloat16
    g = gen(17)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(Co, 3, 3, 3, generator=g) / math.sqrt(27)
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    x8 = torch.empty(B * H * W, 8, dtype=dt, device='cuda')
    p.nchw3_to_nhwc8(x.cuda(), x8, B, H, W, ga)
    Wf = torch.empty(Co, 72, dtype=dt, device='cuda')
    p.convw_pack(w.cuda(), Wf, Co, 3, 9, 8, 72, ga)
    OH, OW = H // 2, W // 2
    ref = F.conv2d(x.to(dt).float(), w.to(dt).float(), None, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    outs = []
    for direct in (1, 0):
        knobs(CONV0_DIRECT=direct)
        out = torch.full((B * OH * OW, Co), 7.0, dtype=dt, device='cuda')
        p.gemm(x8, Wf, out, B * OH * OW, Co, 72, ga, a_kind=ops.A_CONV3S2, a_dims=(H, W, 8))
        assert_close(out, ref, tol(dt), f'stem conv0 direct={direct}')
        outs.append(out.float().cpu())
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-2 * max(outs[1].abs().max().item(), 1e-3)
    # its weight gradient: the direct kernel needs maps of 16 x 32-pixel input tiles (other maps stay on the gather form of gemm_tn)
    gy_c, gy_g = rnd((B * OH * OW, Co), dt, g)
    wr = w.to(dt).float().clone().requires_grad_(True)
    F.conv2d(x.to(dt).float(), wr, None, stride=2, padding=1).backward(gy_c.reshape(B, OH, OW, Co).permute(0, 3, 1, 2))
    gs = []
    for direct in (1, 0):
        knobs(CONV0_DIRECT=direct)
        G = torch.full((Co, 72), 0.25, device='cuda')               # the launch ADDS its result
        p.wgrad(gy_g, x8, G, B * OH * OW, Co, 72, ga, x_kind=ops.A_CONV3S2, x_dims=(H, W, 8))
        dW = torch.zeros(Co, 3, 3, 3, device='cuda')
        p.convw_unpack_grad(G - 0.25, dW, Co, 3, 9, 8, 72)
        assert_close(dW, wr.grad, tol(dt, 2), f'stem conv0 wgrad direct={direct}')
        gs.append(dW.cpu())
    assert (gs[0] - gs[1]).abs().max().item() <= 2e-2 * max(gs[1].abs().max().item(), 1e-3)


@pytest.mark.parametrize('dt', DT)
def test_stem_first_conv_from_nchw(dt):
    """stage1_conv_embed.0 (3 -> E, 3x3 s2, no bias, ga_cswin.py:464): NCHW fp32 input packed to NHWC8, then the gather GEMM"""
    ops = _imp()
    B, H, W, Co = 2, 32, 32, 16
    g = gen(6)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(Co, 3, 3, 3, generator=g) / math.sqrt(27)
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    x8 = torch.empty(B * H * W, 8, dtype=dt, device='cuda')
    p.nchw3_to_nhwc8(x.cuda(), x8, B, H, W, ga)
    Wf = torch.empty(Co, 72, dtype=dt, device='cuda')
    p.convw_pack(w.cuda(), Wf, Co, 3, 9, 8, 72, ga)
    OH, OW = H // 2, W // 2
    out = torch.empty(B * OH * OW, Co, dtype=dt, device='cuda')
    p.gemm(x8, Wf, out, B * OH * OW, Co, 72, ga, a_kind=ops.A_CONV3S2, a_dims=(H, W, 8))
    xr = x.to(dt).float()
    ref = F.conv2d(xr, w.to(dt).float(), None, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    assert_close(out, ref, tol(dt), 'stem conv0')
    gy_c, gy_g = rnd((B * OH * OW, Co), dt, g)
    G = torch.zeros(Co, 72, device='cuda')
    p.wgrad(gy_g, x8, G, B * OH * OW, Co, 72, ga, x_kind=ops.A_CONV3S2, x_dims=(H, W, 8))
    dW = torch.zeros(Co, 3, 3, 3, device='cuda')
    p.convw_unpack_grad(G, dW, Co, 3, 9, 8, 72)
    wr = w.to(dt).float().clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=2, padding=1).backward(gy_c.reshape(B, OH, OW, Co).permute(0, 3, 1, 2))
    assert_close(dW, wr.grad, tol(dt, 2), 'stem conv0 wgrad')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('C', [16, 64])
def test_layernorm_gelu(dt, C):
    """LayerNorm(1e-5, affine) -> GELU between the deep-stem convs (ga_cswin.py:466-468)"""
    ops = _imp()
    rows = 3 * 37
    g = gen(8)
    x_c, x_g = rnd((rows, C), dt, g, 1.5)
    gy_c, gy_g = rnd((rows, C), dt, g)
    w = torch.rand(C, generator=g) * 0.4 + 0.8
    b = torch.randn(C, generator=g) * 0.1
    xr, wr, br = x_c.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.gelu(F.layer_norm(xr, (C,), wr, br, 1e-5))
    y.backward(gy_c)
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    out = torch.empty(rows, C, dtype=dt, device='cuda')
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    p.layernorm_gelu_fwd(x_g, w.cuda(), b.cuda(), out, mean, rstd, rows, C, 1e-5, ga)
    assert_close(out, y, tol(dt), 'ln+gelu fwd')
    dx = torch.empty(rows, C, dtype=dt, device='cuda')
    dw, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    p.layernorm_gelu_bwd(gy_g, x_g, mean, rstd, w.cuda(), b.cuda(), dx, dw, db, rows, C, ga)
    assert_close(dx, xr.grad, tol(dt, 1.5), 'ln+gelu dx')
    assert_close(dw, wr.grad, tol(dt, 2), 'ln+gelu dw')
    assert_close(db, br.grad, tol(dt, 2), 'ln+gelu db')
