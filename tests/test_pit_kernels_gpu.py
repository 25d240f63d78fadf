"""The pooling-transformer kernels of csrc/pit.hip (map_pit.py conv_embedding / pos_embed / conv_head_pooling; the general bilinear
resize of map.py:322-333) against torch on the same inputs, through the C ABI.  fp32 mode 1e-5; bf16 2e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from imagenet_models_amd import ops
    return ops


def err(a, b):
    return float((a.float().cpu() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


@pytest.mark.parametrize('dt,tol', [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize('B,H,P,S', [(2, 64, 16, 8), (1, 224, 16, 8), (2, 48, 16, 16), (1, 40, 8, 4)])
def test_patchify_strided(dt, tol, B, H, P, S):
    ops = _ops()
    x = torch.randn(B, 3, H, H, generator=torch.Generator().manual_seed(H))
    g = (H - P) // S + 1
    out = torch.empty(B * g * g, 3 * P * P, dtype=dt, device='cuda')
    p = ops.Plan(eager=True)
    p.patchify_strided(x.cuda(), out, P, S, ops.ga_dtype(dt))
    torch.cuda.synchronize()
    ref = F.unfold(x, P, stride=S).transpose(1, 2).reshape(B * g * g, 3 * P * P)
    assert err(out, ref) <= tol
    # the patch convolution itself = rows @ W^T
    w = torch.randn(24, 3, P, P, generator=torch.Generator().manual_seed(1))
    conv = F.conv2d(x, w, stride=S).permute(0, 2, 3, 1).reshape(B * g * g, 24)
    assert err(out.float().cpu() @ w.reshape(24, -1).t(), conv) < max(tol, 1e-5) * 5


@pytest.mark.parametrize('dt,tol', [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_pos_add(dt, tol):
    ops = _ops()
    B, Np, C = 3, 49, 48
    g = torch.Generator().manual_seed(0)
    tok, pos = torch.randn(B * Np, C, generator=g).to(dt), torch.randn(Np, C, generator=g)
    x0 = torch.empty(B * Np, C, dtype=dt, device='cuda')
    dpos = torch.full((Np, C), 7.0, device='cuda')
    p = ops.Plan(eager=True)
    p.pos_add_fwd(tok.cuda(), pos.cuda(), x0, B, Np, C, ops.ga_dtype(dt))
    p.pos_add_bwd(tok.cuda(), dpos, B, Np, C, ops.ga_dtype(dt))
    torch.cuda.synchronize()
    assert err(x0.reshape(B, Np, C), tok.float().reshape(B, Np, C) + pos) <= tol
    assert err(dpos, tok.float().reshape(B, Np, C).sum(0)) < 1e-5


@pytest.mark.parametrize('dt,tol', [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('simple', [0, 1])
@pytest.mark.parametrize('B,H,Cin,mult', [(2, 7, 48, 2), (2, 27, 144, 2), (1, 14, 288, 2), (3, 4, 96, 1), (2, 5, 8, 3), (70, 14, 16, 2)])
def test_dwpool(dt, tol, B, H, Cin, mult, simple, knobs):
    """simple = 1: the any-multiplier kernels; 0: the LDS-weight / wave-per-chunk forms multipliers 1 and 2 take"""
    knobs(DWPOOL_SIMPLE=simple)
    ops = _ops()
    g = torch.Generator().manual_seed(H + Cin)
    Co, Ho = Cin * mult, (H - 1) // 2 + 1
    x = torch.randn(B, H, H, Cin, generator=g).to(dt)
    w, b = torch.randn(Co, 1, 3, 3, generator=g) * 0.3, torch.randn(Co, generator=g)
    dy = torch.randn(B, Ho, Ho, Co, generator=g).to(dt)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, stride=2, padding=1, groups=Cin)
    assert ref.shape[2] == Ho
    ref.backward(dy.float().permute(0, 3, 1, 2))
    y = torch.empty(B * Ho * Ho, Co, dtype=dt, device='cuda')
    dx = torch.empty(B * H * H, Cin, dtype=dt, device='cuda')
    dw, db = torch.ones(Co, 1, 3, 3, device='cuda'), torch.ones(Co, device='cuda')        # accumulated into
    p = ops.Plan(eager=True)
    gd = ops.ga_dtype(dt)
    p.dwpool_fwd(x.cuda(), w.cuda(), b.cuda(), y, B, H, H, Cin, mult, gd)
    p.dwpool_bwd_data(dy.cuda(), w.cuda(), dx, B, H, H, Cin, mult, gd)
    p.dwpool_bwd_weight(dy.cuda(), x.cuda(), dw, db, B, H, H, Cin, mult, gd)
    torch.cuda.synchronize()
    assert err(y.reshape(B, Ho, Ho, Co), ref.detach().permute(0, 2, 3, 1)) < tol
    assert err(dx.reshape(B, H, H, Cin), xr.grad.permute(0, 2, 3, 1)) < tol
    assert err(dw - 1, wr.grad) < 1e-4 and err(db - 1, br.grad) < 1e-4


@pytest.mark.parametrize('dt,tol', [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('B,Hin,Hout,C', [(2, 27, 14, 144), (2, 7, 4, 48), (1, 28, 14, 16), (1, 9, 5, 8), (2, 14, 14, 8), (1, 5, 9, 8)])
def test_resize_concat(dt, tol, B, Hin, Hout, C):
    ops = _ops()
    g = torch.Generator().manual_seed(Hin)
    ctot, off = C + 24, 16
    x = torch.randn(B, Hin, Hin, C, generator=g).to(dt)
    dcat = torch.randn(B * Hout * Hout, ctot, generator=g).to(dt)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = F.interpolate(xr, size=(Hout, Hout), mode='bilinear')
    ref.backward(dcat.float().reshape(B, Hout, Hout, ctot)[..., off:off + C].permute(0, 3, 1, 2))
    cat = torch.full((B * Hout * Hout, ctot), -5.0, dtype=dt, device='cuda')
    dsrc = torch.empty(B * Hin * Hin, C, dtype=dt, device='cuda')
    p = ops.Plan(eager=True)
    gd = ops.ga_dtype(dt)
    p.resize_concat_fwd(x.cuda(), cat, B, Hin, Hin, C, Hout, Hout, ctot, off, gd)
    p.resize_concat_bwd(dcat.cuda(), dsrc, B, Hin, Hin, C, Hout, Hout, ctot, off, gd)
    torch.cuda.synchronize()
    c = cat.float().cpu().reshape(B, Hout, Hout, ctot)
    assert err(c[..., off:off + C], ref.detach().permute(0, 2, 3, 1)) < tol
    assert bool((c[..., :off] == -5).all()) and bool((c[..., off + C:] == -5).all())       # the other columns are untouched
    assert err(dsrc.reshape(B, Hin, Hin, C), xr.grad.permute(0, 2, 3, 1)) < tol
