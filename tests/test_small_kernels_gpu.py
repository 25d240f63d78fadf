"""csrc/small.hip: the alignment-free one-token grouped linear layers (odd-width variants, ga_convnext.py:190-222,418-420) and the
pad copy, against plain torch fp32 evaluations of the same formulas.  Shapes deliberately off the 16-byte grid."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from imagenet_models_amd import ops
    return ops


@pytest.mark.parametrize('dt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('rows,groups,Ng,Kg,perm', [(37, 8, 86, 579, False), (64, 4, 172, 688, True), (5, 1, 7, 3, False)])
def test_small_linear_fwd_bwd(dt, tol, rows, groups, Ng, Kg, perm):
    ops = _ops()
    g = torch.Generator().manual_seed(rows + Ng)
    Kt, Nt = groups * Kg, groups * Ng
    pad = 4
    A = torch.randn(rows, Kt + pad, generator=g).to(dt)              # leading dimension != K, group stride = Kg
    W = torch.randn(Nt, Kg, generator=g) * Kg ** -0.5
    b, cs = torch.randn(Nt, generator=g) * 0.1, torch.rand(Nt, generator=g) + 0.5
    rs = (torch.rand(rows, generator=g) < 0.7).float() / 0.7
    R = torch.randn(rows, Nt, generator=g).to(dt)
    dY = torch.randn(rows, Nt, generator=g).to(dt)
    p_idx = torch.randperm(Kt, generator=g).to(torch.int32) if perm else None
    # reference
    Af = A.float()[:, :Kt].clone().requires_grad_(True)
    Wf, bf_, csf = W.clone().requires_grad_(True), b.clone().requires_grad_(True), cs.clone().requires_grad_(True)
    Ain = Af[:, p_idx.long()] if perm else Af
    raw = torch.cat([Ain[:, k * Kg:(k + 1) * Kg] @ Wf[k * Ng:(k + 1) * Ng].T for k in range(groups)], 1) + bf_
    y = R.float() + rs[:, None] * (csf * raw)
    y.backward(dY.float())
    # kernels
    Ad, Wd, Yd, Yraw = A.cuda(), W.cuda(), torch.empty(rows, Nt, dtype=dt, device='cuda'), torch.empty(rows, Nt, dtype=dt, device='cuda')
    p = ops.Plan(eager=True)
    d = p.small_linear_desc(Ad, Wd, Yd, rows, groups, Ng, Kg, ops.ga_dtype(dt), lda=Kt + pad, a_gstride=Kg, ldy=Nt, bias=b.cuda(),
                            a_perm=p_idx.cuda() if perm else None, col_scale=cs.cuda(), rowscale=rs.cuda(), rows_per_scale=1,
                            R=R.cuda(), ldr=Nt, Yraw=Yraw)
    p.small_linear_fwd(d)
    dA = torch.zeros(rows, Kt + pad, dtype=dt, device='cuda')
    dW, db, dcs = torch.zeros_like(Wd), torch.zeros(Nt, device='cuda'), torch.zeros(Nt, device='cuda')
    p.small_linear_bwd(d, dY.cuda(), dA=dA, dW=dW, dbias=db, dcol_scale=dcs)
    torch.cuda.synchronize()

    def close(got, ref, what, t=tol):
        err = float((got.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))
        assert err < t, (what, err)
    close(Yd, y.detach(), 'y')
    close(Yraw, raw.detach(), 'yraw')
    close(dA[:, :Kt], Af.grad, 'dA')
    assert float(dA[:, Kt:].abs().max()) == 0.0                      # the pad columns are never written
    close(dW, Wf.grad, 'dW', tol * 2)
    close(db, bf_.grad, 'db', tol * 2)
    close(dcs, csf.grad, 'dcs', tol * 4)


def test_colstats_and_pad_copy():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, 91, generator=g)
    s, q = torch.zeros(86, device='cuda'), torch.zeros(86, device='cuda')
    ops.Plan(eager=True).colstats(x.cuda(), 91, 50, 86, s, q, ops.GA_F32)
    assert torch.allclose(s.cpu(), x[:, :86].sum(0), atol=1e-4) and torch.allclose(q.cpu(), (x[:, :86] ** 2).sum(0), atol=1e-3)
    src = torch.randn(172, 9 * 172, generator=g)
    dst = torch.zeros(176, 9 * 176, device='cuda')
    p = ops.Plan(eager=True)
    p.pad_copy_f32(src.cuda(), dst, 172, 9 * 172, 9 * 172, 9 * 176)
    want = torch.zeros(176, 9 * 176)
    want[:172, :9 * 172] = src
    assert torch.equal(dst.cpu(), want)
    back = torch.ones(172, 9 * 172, device='cuda')
    p.pad_copy_f32(dst, back, 172, 9 * 172, 9 * 176, 9 * 172, accumulate=True)
    assert torch.equal(back.cpu(), src + 1)


@pytest.mark.parametrize('R,C,RG,RGp,CG,CGp', [(2752, 172, 688, 704, 172, 176), (688, 688, 172, 176, 172, 176), (1, 688, 1, 1, 172, 176),
                                                (6, 10, 3, 5, 5, 5)])
def test_pad_groups_round_trip(R, C, RG, RGp, CG, CGp):
    """ga_pad_groups_f32: rows in groups of RG -> RGp, columns in groups of CG -> CGp (the GroupConvMlp parameters of the odd-width
    variants, ga_convnext.py:190-222): exact copy of the real part, zeros elsewhere, and the way back with accumulation"""
    ops = _ops()
    g = torch.Generator().manual_seed(R + C)
    src = torch.randn(R, C, generator=g)
    Rp, Cp = R // RG * RGp, C // CG * CGp
    dst = torch.zeros(Rp, Cp, device='cuda')
    p = ops.Plan(eager=True)
    p.pad_groups_f32(src.cuda(), dst, R, C, RG, RGp, CG, CGp)
    want = torch.zeros(R // RG, RGp, C // CG, CGp)
    want[:, :RG, :, :CG] = src.view(R // RG, RG, C // CG, CG)
    assert torch.equal(dst.cpu(), want.view(Rp, Cp))
    back = torch.ones(R, C, device='cuda')
    p.pad_groups_f32(dst, back, R, C, RG, RGp, CG, CGp, unpad=True, accumulate=True)
    assert torch.equal(back.cpu(), src + 1)


@pytest.mark.parametrize('R,rg,ng,cg,ld', [(192, 24, 8, 64, 512), (960, 24, 8, 64, 520), (12, 2, 3, 3, 9)])
def test_blockdiag_round_trip(R, rg, ng, cg, ld):
    """ga_blockdiag_f32: the weight of a grouped 1x1 convolution [R][cg] <-> its block-diagonal image [R][ld] (GA-CSWin's grouped
    gram_contraction as one dense product, ga_cswin.py:559-561)"""
    ops = _ops()
    g = torch.Generator().manual_seed(R + cg)
    src = torch.randn(R, cg, generator=g)
    dst = torch.zeros(R, ld, device='cuda')
    p = ops.Plan(eager=True)
    p.blockdiag_f32(src.cuda(), dst, R, rg, ng, cg, ld)
    want = torch.zeros(R, ld)
    for r in range(R):
        want[r, ((r // rg) % ng) * cg:((r // rg) % ng) * cg + cg] = src[r]
    assert torch.equal(dst.cpu(), want)
    back = torch.ones(R, cg, device='cuda')
    p.blockdiag_f32(dst, back, R, rg, ng, cg, ld, to_diag=False, accumulate=True)
    assert torch.equal(back.cpu(), src + 1)
