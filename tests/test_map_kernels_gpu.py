"""GPU: the MAP-head kernels through the C ABI (ops.Plan, eager) against PyTorch CPU fp32 references of the same ops
(the formulas of /root/reference/MAP/models/map.py and MAP/train.py:792-839 as restated in oracle/map_oracle.py).
Tolerances: fp32 2e-4, bf16 2e-2 of the tensor's max."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import assert_close, gen, rnd, tol

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _imp():
    from imagenet_models_amd import ops
    return ops


@pytest.mark.parametrize('dt', DT)
def test_multiscale_resize_modes(dt):
    """MultiScale.forward (map.py:326-329): larger maps are REDUCED by bilinear interpolation (56 -> 14, 28 -> 14), the
    smaller one ENLARGED by adaptive_avg_pool2d (7 -> 14); forward and backward vs autograd"""
    ops = _imp()
    B, C = 2, 16
    g = gen(3)
    for Hin, mode in ((56, 2), (28, 2), (7, 3)):
        x_c, x_g = rnd((B, Hin, Hin, C), dt, g)
        xr = x_c.permute(0, 3, 1, 2).clone().requires_grad_(True)
        y = F.adaptive_avg_pool2d(xr, (14, 14)) if Hin < 14 else F.interpolate(xr, size=(14, 14), mode='bilinear')
        gy_c, gy_g = rnd((B, 14, 14, C + 8), dt, g)
        y.backward(gy_c[..., 8:].permute(0, 3, 1, 2))
        dst = torch.zeros(B, 14, 14, C + 8, dtype=dt, device='cuda')
        p = ops.Plan(eager=True)
        p.pool_concat_fwd(x_g, dst, B, Hin, Hin, C, 14, 14, C + 8, 8, mode, ops.ga_dtype(dt))
        assert_close(dst[..., 8:], y.permute(0, 2, 3, 1), tol(dt), f'resize fwd {Hin}')
        dsrc = torch.empty(B, Hin, Hin, C, dtype=dt, device='cuda')
        p.pool_concat_bwd(gy_g, None, dsrc, B, Hin, Hin, C, 14, 14, C + 8, 8, mode, ops.ga_dtype(dt))
        assert_close(dsrc, xr.grad.permute(0, 2, 3, 1), tol(dt), f'resize bwd {Hin}')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('cfg', [(48, 4, 3), (64, 8, 2), (64, 8, 1)])
def test_gram_pack_with_token_interleave(dt, cfg):
    """GramToken.forward (map.py:217-227): triu gather, L2 normalise, (b, -1, T) -> (b, T, -1) interleave, grouped layout"""
    ops = _imp()
    C, groups, T = cfg
    B = 3
    g = gen(4)
    x = torch.randn(B, C, 20, generator=g)
    G = (x @ x.transpose(1, 2)).contiguous()
    ntri = C * (C + 1) // 2
    Kg = ntri // groups
    Kp = (Kg + 7) // 8 * 8
    Gr = G.clone().requires_grad_(True)
    iu = torch.triu_indices(C, C)
    v = F.normalize(Gr.reshape(B, C * C)[:, iu[0] * C + iu[1]], dim=-1)
    v = v.reshape(B, -1, T).permute(0, 2, 1).reshape(B, ntri)
    gy = torch.randn(B, ntri, generator=g).to(dt).float()
    v.backward(gy)
    out = torch.full((B, groups * Kp), 7.0, dtype=dt, device='cuda')
    inv = torch.empty(B, device='cuda')
    p = ops.Plan(eager=True)
    p.gram_pack_fwd2(G.cuda(), out, inv, B, C, groups, Kp, T, ops.ga_dtype(dt))
    got = out.float().cpu().reshape(B, groups, Kp)
    assert_close(got[:, :, :Kg].reshape(B, ntri), v, tol(dt), 'gram vec')
    assert float(got[:, :, Kg:].abs().max()) == 0.0 if Kp > Kg else True
    dvec = torch.zeros(B, groups, Kp)
    dvec[:, :, :Kg] = gy.reshape(B, groups, Kg)
    S = torch.empty(B, C, C, dtype=dt, device='cuda')
    p.gram_pack_bwd2(dvec.to(dt).cuda(), out, inv, S, B, C, groups, Kp, T, ops.ga_dtype(dt))
    # S is the symmetric gradient of the raw Gram entries with the diagonal doubled: dG_full = triu(S)+... compare via dX = X.S
    dG = Gr.grad                                                   # gradient wrt the upper-triangular entries only
    want = dG + dG.transpose(1, 2)                                  # symmetrised, diagonal doubled
    assert_close(S, want, tol(dt, 2), 'gram S')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('T', [2, 3])
def test_map_tokens(dt, T):
    ops = _imp()
    B, C = 5, 64
    g = gen(5)
    e_c, e_g = rnd((B, C * T), dt, g)
    er = e_c.clone().requires_grad_(True)
    tok = er.reshape(B, C, T).permute(0, 2, 1)
    tok = torch.cat([tok, tok.mean(dim=1, keepdim=True)], dim=1)
    gy_c, gy_g = rnd((B, T + 1, C), dt, g)
    tok.backward(gy_c)
    out = torch.empty(B, T + 1, C, dtype=dt, device='cuda')
    p = ops.Plan(eager=True)
    p.map_tokens_fwd(e_g, out, B, C, T, True, ops.ga_dtype(dt))
    assert_close(out, tok, tol(dt), 'tokens')
    de = torch.empty(B, C * T, dtype=dt, device='cuda')
    p.map_tokens_bwd(gy_g, de, B, C, T, True, ops.ga_dtype(dt))
    assert_close(de, er.grad, tol(dt), 'dtokens')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('case', [(3, 2 + 1, 196, 12, 32, False), (2, 3 + 1, 196, 8, 8, False), (3, 3, 49, 4, 16, True)])
def test_multi_token_class_attention(dt, case):
    """ClassAttention.forward, in_dim == dim branch (map.py:118-144): T query tokens against T class rows + Nt image rows;
    optional attention-dropout mask; forward and backward vs autograd"""
    ops = _imp()
    B, T, Nt, heads, hd, use_mask = case
    E, N = heads * hd, T + Nt
    g = gen(6)
    q_c, q_g = rnd((B, T, E), dt, g)
    kvc_c, kvc_g = rnd((B, T, 2 * E), dt, g)
    ld = 2 * E + 16                                        # token rows are column slices of a wider matrix
    tokbuf = (torch.randn(B * Nt, ld, generator=g)).to(dt)
    kvt_c = tokbuf.float()[:, :2 * E].reshape(B, Nt, 2 * E)
    do_c, do_g = rnd((B, T, E), dt, g)
    mask = None
    if use_mask:
        mask = (torch.rand(B, T, heads, N, generator=g) < 0.9).float() / 0.9
    qr, kcr, ktr = q_c.clone().requires_grad_(True), kvc_c.clone().requires_grad_(True), kvt_c.clone().requires_grad_(True)
    kv = torch.cat([kcr, ktr], dim=1)
    qh = qr.reshape(B, T, heads, hd).permute(0, 2, 1, 3) * hd ** -0.5
    kh = kv[..., :E].reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    vh = kv[..., E:].reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    a = (qh @ kh.transpose(-2, -1)).softmax(-1)                      # (B, heads, T, N)
    am = a * mask.permute(0, 2, 1, 3) if mask is not None else a
    o = (am @ vh).transpose(1, 2).reshape(B, T, E)
    o.backward(do_c)
    out = torch.empty(B, T, E, dtype=dt, device='cuda')
    P = torch.empty(B, T, heads, N, device='cuda')
    mg = mask.cuda() if mask is not None else None
    p = ops.Plan(eager=True)
    tok_g = tokbuf.cuda()
    p.class_attn_mt_fwd(q_g, kvc_g, tok_g, ld, out, P, mg, B, T, N, heads, hd, hd ** -0.5, ops.ga_dtype(dt))
    assert_close(out, o, tol(dt), 'mt attn out')
    assert_close(P, a.permute(0, 2, 1, 3), tol(dt), 'mt attn P')
    dq = torch.empty(B, T, E, dtype=dt, device='cuda')
    dkc = torch.empty(B, T, 2 * E, dtype=dt, device='cuda')
    dtok = torch.zeros(B * Nt, ld, dtype=dt, device='cuda')
    p.class_attn_mt_bwd(do_g, q_g, kvc_g, tok_g, ld, P, mg, dq, dkc, dtok, ld, B, T, N, heads, hd, hd ** -0.5, ops.ga_dtype(dt))
    assert_close(dq, qr.grad, tol(dt, 1.5), 'dq')
    assert_close(dkc, kcr.grad, tol(dt, 1.5), 'dkv_cls')
    assert_close(dtok[:, :2 * E].reshape(B, Nt, 2 * E), ktr.grad, tol(dt, 1.5), 'dkv_tok')
    assert float(dtok[:, 2 * E:].abs().max()) == 0.0


@pytest.mark.parametrize('kind', ['ce', 'bce'])
def test_map_multi_group_loss(kind):
    """MAP/train.py:792-839 (distill_tokens == 0): value and both gradients vs the oracle's restatement under autograd"""
    from oracle import map_oracle as O
    ops = _imp()
    K, B, NC = 4, 6, 40
    g = gen(7)
    org = torch.randn(K, B, NC, generator=g)
    avg = torch.randn(K, B, NC, generator=g)
    tgt = torch.randint(0, NC, (B,), generator=g)
    orr, avr = org.clone().requires_grad_(True), avg.clone().requires_grad_(True)
    want = O.multi_group_loss([[orr[k], avr[k]] for k in range(K)], tgt, -0.8, kind, 0.1)
    want.backward()
    loss = torch.zeros(1, device='cuda')
    dorg = torch.empty(K, B, NC, device='cuda')
    davg = torch.empty(K, B, NC, device='cuda')
    ops.Plan(eager=True).map_loss_fwd_bwd(org.cuda(), avg.cuda(), tgt.cuda(), loss, dorg, davg, K, B, NC, -0.8,
                                          0 if kind == 'ce' else 1, 0.1, 1.0, ops.GA_F32)
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    assert_close(dorg, orr.grad, 2e-4, 'dorg')
    assert_close(davg, avr.grad, 2e-4, 'davg')


@pytest.mark.parametrize('dt', DT)
def test_elementwise_helpers(dt):
    ops = _imp()
    g = gen(8)
    n = 8 * 123
    x_c, x_g = rnd((n,), dt, g, 2.0)
    gy_c, gy_g = rnd((n,), dt, g)
    ga = ops.ga_dtype(dt)
    p = ops.Plan(eager=True)
    y = torch.empty(n, dtype=dt, device='cuda')
    p.gelu_fwd(x_g, y, n, ga)
    xr = x_c.clone().requires_grad_(True)
    F.gelu(xr).backward(gy_c)
    assert_close(y, F.gelu(x_c), tol(dt), 'gelu')
    dx = torch.empty(n, dtype=dt, device='cuda')
    p.gelu_bwd(gy_g, x_g, dx, n, ga)
    assert_close(dx, xr.grad, tol(dt), 'gelu bwd')
    m = (torch.rand(n, generator=g) < 0.9).float() / 0.9
    a_c = F.relu(x_c)
    out, der = torch.empty(n, dtype=dt, device='cuda'), torch.empty(n, dtype=dt, device='cuda')
    p.relu_drop(a_c.to(dt).cuda(), m.cuda(), out, der, n, ga)
    assert_close(out, a_c * m, tol(dt), 'relu*mask')
    assert_close(der, (a_c > 0).float() * m, tol(dt), 'relu deriv')
    p.mask_mul(x_g, m.cuda(), gy_g, out, n, ga)
    assert_close(out, x_c * m + gy_c, tol(dt), 'mask_mul')
    src_c, src_g = rnd((7, 48), dt, g)
    dst = torch.zeros(7, 32, dtype=dt, device='cuda')
    p.copy2d(src_g[:, 8:], 48, dst[:, 8:], 32, 7, 24, ga)
    assert_close(dst[:, 8:], src_c[:, 8:32], tol(dt), 'copy2d')
    assert float(dst[:, :8].abs().max()) == 0.0
    p.copy2d(src_g[:, 8:], 48, dst[:, 8:], 32, 7, 24, ga, accumulate=True)
    assert_close(dst[:, 8:], 2 * src_c[:, 8:32], tol(dt), 'copy2d acc')


def test_topk_with_nan_and_all_inf_rows():
    """a diverged run (NaN / -inf logits) must still give valid, distinct indices like torch.topk"""
    ops = _imp()
    K, B, NC = 2, 4, 24
    g = gen(9)
    logits = torch.randn(K, B, NC, generator=g)
    logits[:, 1, :] = float('-inf')
    logits[0, 2, 5] = float('nan')
    logits[:, 3, :] = float('nan')
    idx = torch.empty(B, 5, dtype=torch.int64, device='cuda')
    ops.Plan(eager=True).heads_topk(logits.cuda(), K, B, NC, 5, None, idx)
    idx = idx.cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < NC
    for b in range(B):
        assert len(set(idx[b].tolist())) == 5
    assert torch.equal(idx[0], logits.sum(0)[0].topk(5)[1])
    assert int(idx[2, 0]) == 5            # the NaN entry ranks first, as in torch.topk
