"""Fused MLP bodies (ga_mlp_fwd / ga_mlp_bwd, csrc/mlp.hip) against a plain fp32 torch evaluation of the same formulas on the
bf16-rounded operands (Block.forward of /root/reference/GA/ga_convnext.py:86-101 with the LayerNorm affine / gamma folded into
the weights).  Tolerance: bf16 storage of the hidden tile and of the results, 2e-2 of the tensor max."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from imagenet_models_amd import ops
    return ops


def gelu_tanh(x):
    return torch.nn.functional.gelu(x, approximate='tanh')


def close(got, ref, tol, what):
    err = float((got.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))
    assert err < tol, (what, err)
    return err


@pytest.mark.parametrize('C', [96, 192])
@pytest.mark.parametrize('M,rps', [(256, 64), (1000, 250), (128 * 37 + 5, 4741)])
def test_mlp_fwd(C, M, rps):
    ops = _ops()
    H = 4 * C
    g = torch.Generator().manual_seed(C + M)
    bf = lambda t: t.to(torch.bfloat16)
    X, R = bf(torch.randn(M, C, generator=g)), bf(torch.randn(M, C, generator=g))
    W1, W2 = bf(torch.randn(H, C, generator=g) * C ** -0.5), bf(torch.randn(C, H, generator=g) * H ** -0.5)
    b1, b2 = torch.randn(H, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    rs = (torch.rand((M + rps - 1) // rps, generator=g) < 0.7).float() / 0.7
    assert ops.mlp_supported(C, H, ops.GA_BF16) and not ops.mlp_supported(C, H, ops.GA_F32) and not ops.mlp_supported(384, 1536, ops.GA_BF16)
    hid = bf(gelu_tanh(X.float() @ W1.float().T + b1)).float()
    ref = R.float() + rs.repeat_interleave(rps)[:M, None] * (hid @ W2.float().T + b2)
    Y = torch.empty(M, C, dtype=torch.bfloat16, device='cuda')
    p = ops.Plan(eager=True)
    p.mlp_fwd(X.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda(), Y, M, C, ops.GA_BF16, R=R.cuda(), rowscale=rs.cuda(), rows_per_scale=rps)
    torch.cuda.synchronize()
    close(Y, ref, 2e-2, 'y')
    # no residual / row scale
    Y2 = torch.empty_like(Y)
    p.mlp_fwd(X.cuda(), W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda(), Y2, M, C, ops.GA_BF16)
    torch.cuda.synchronize()
    close(Y2, hid @ W2.float().T + b2, 2e-2, 'y (plain)')


@pytest.mark.parametrize('C', [96, 192])
@pytest.mark.parametrize('M', [256, 128 * 21 + 77])
def test_mlp_bwd(C, M):
    ops = _ops()
    H = 4 * C
    g = torch.Generator().manual_seed(7 * C + M)
    bf = lambda t: t.to(torch.bfloat16)
    X, DY = bf(torch.randn(M, C, generator=g)), bf(torch.randn(M, C, generator=g))
    W1, W2 = bf(torch.randn(H, C, generator=g) * C ** -0.5), bf(torch.randn(C, H, generator=g) * H ** -0.5)
    b1 = torch.randn(H, generator=g) * 0.1
    pre = (X.float() @ W1.float().T + b1).requires_grad_(True)
    a = gelu_tanh(pre)
    a.backward(DY.float() @ W2.float())
    dh = bf(pre.grad).float()
    dx = dh @ W1.float()
    A = torch.empty(M, H, dtype=torch.bfloat16, device='cuda')
    DH, DX = torch.empty_like(A), torch.empty(M, C, dtype=torch.bfloat16, device='cuda')
    p = ops.Plan(eager=True)
    p.mlp_bwd(X.cuda(), DY.cuda(), W1.cuda(), b1.cuda(), W2.T.contiguous().cuda(), W1.T.contiguous().cuda(), A, DH, DX, M, C, ops.GA_BF16)
    torch.cuda.synchronize()
    close(A, a.detach(), 2e-2, 'a')
    close(DH, dh, 2e-2, 'dh')
    close(DX, dx, 2e-2, 'dx')
