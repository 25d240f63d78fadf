"""Global multi-head attention (csrc/attn.hip; timm vision_transformer.Attention inside Block, MAP/models/map_pit.py:14,35-44)
against a plain torch evaluation: fp32 generic form (2e-4) and the bf16 MFMA flash form (2e-2), forward and backward, sequence
lengths that are not multiples of the 64-row blocks (577 = ViT-B/16 @ 384 with the class token)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from imagenet_models_amd import ops
    return ops


def _case(dt, B, N, H, hd, tol, force_simple=False):
    ops = _ops()
    g = torch.Generator().manual_seed(N + H)
    C = H * hd
    qkv = (torch.randn(B * N, 3 * C, generator=g) * 0.7).to(dt)
    do = torch.randn(B * N, C, generator=g).to(dt)
    x = qkv.float().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)       # (3, B, H, N, hd)
    att = torch.softmax(x[0] @ x[1].transpose(-1, -2) * hd ** -0.5, dim=-1)
    ref = (att @ x[2]).transpose(1, 2).reshape(B * N, C)
    ref.backward(do.float())
    dref = x.grad.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * C)
    from imagenet_models_amd import _lib
    if force_simple:
        _lib.load().ga_set_knob(b'ATTN_MFMA', 0)
    try:
        out = torch.empty(B * N, C, dtype=dt, device='cuda')
        lse = torch.empty(B, H, N, device='cuda')
        dqkv = torch.zeros(B * N, 3 * C, dtype=dt, device='cuda')
        ws = torch.empty(B * H * N, device='cuda')
        p = ops.Plan(eager=True)
        d = p.attn_desc(qkv.cuda(), out, lse, B, N, H, hd, hd ** -0.5, ops.ga_dtype(dt))
        p.attn_fwd(d)
        p.attn_bwd(d, do.cuda(), dqkv, ws)
        torch.cuda.synchronize()
    finally:
        _lib.load().ga_unset_knob(b'ATTN_MFMA')

    def err(a, b):
        return float((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    e_o = err(out, ref.detach())
    lse_ref = torch.logsumexp(x[0] @ x[1].transpose(-1, -2) * hd ** -0.5, dim=-1).detach()
    e_l = err(lse, lse_ref)
    e = [err(dqkv[:, j * C:(j + 1) * C], dref[:, j * C:(j + 1) * C]) for j in range(3)]
    print(f'[{dt} B{B} N{N} H{H} hd{hd} simple={force_simple}] out {e_o:.2e} lse {e_l:.2e} dq/dk/dv {e}')
    assert e_o < tol and e_l < max(tol, 1e-3) and max(e) < 2 * tol, (e_o, e_l, e)


@pytest.mark.parametrize('B,N,H,hd', [(2, 577, 3, 64), (1, 64, 2, 64), (2, 197, 2, 64), (1, 130, 1, 64),
                                      (2, 729, 3, 48), (2, 196, 6, 48), (3, 49, 12, 48), (1, 70, 2, 32), (2, 100, 5, 16)])   # PiT: head_dim 48
def test_attention_bf16_mfma(B, N, H, hd):
    _case(torch.bfloat16, B, N, H, hd, 2e-2)


@pytest.mark.parametrize('dt,tol', [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('B,N,H,hd', [(2, 197, 2, 64), (1, 50, 3, 32), (1, 577, 1, 64)])
def test_attention_generic_form(dt, tol, B, N, H, hd):
    _case(dt, B, N, H, hd, tol, force_simple=(dt == torch.bfloat16))
