"""GPU: every libgaext kernel family through the C ABI (ops.Plan, eager) against a plain PyTorch fp32 CPU
reference of the same op.  Tolerances: fp32 math mode 2e-4 (relative to the tensor's max), bf16 mode 2e-2
(inputs are rounded to bf16 on both sides; outputs are bf16)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _imp():
    from imagenet_models_amd import ops
    return ops


def tol(dt, scale=1.0):
    return (2e-4 if dt == torch.float32 else 2e-2) * scale


def rnd(shape, dt, g, scale=1.0):
    """random tensor already representable in dt (so both sides see identical inputs); returns (cpu fp32, gpu dt)"""
    t = (torch.randn(shape, generator=g) * scale).to(dt)
    return t.float(), t.cuda()


def assert_close(got, ref, t, what=''):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max())
    den = max(float(ref.abs().max()), 1e-6)
    assert err <= t * den, f'{what}: max err {err:.3e} vs ref max {den:.3e} (tol {t})'


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ----------------------------------------------------------------------------------------------------------
# GEMM NT
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('shape', [(300, 200, 136), (128, 128, 64), (257, 96, 96), (70, 1000, 768), (5, 40, 8)])
def test_gemm_plain_bias_act(dt, shape):
    ops = _imp()
    M, N, K = shape
    g = gen(1)
    a, A = rnd((M, K), dt, g)
    b, B = rnd((N, K), dt, g, 1 / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    ldc = (N + 7) // 8 * 8
    for act, name in ((ops.ACT_NONE, 'none'), (ops.ACT_GELU, 'gelu'), (ops.ACT_RELU, 'relu')):
        Cout = torch.zeros(M, ldc, dtype=dt, device='cuda')
        ops.Plan(eager=True).gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), ldc=ldc, bias=bias.cuda(), act=act, alpha=0.5)
        ref = 0.5 * a @ b.t() + bias
        ref = F.gelu(ref) if act == ops.ACT_GELU else (F.relu(ref) if act == ops.ACT_RELU else ref)
        assert_close(Cout[:, :N], ref, tol(dt), f'gemm {name}')
        if ldc > N:
            assert float(Cout[:, N:].abs().max()) == 0.0  # padding untouched


@pytest.mark.parametrize('dt', DT)
def test_gemm_epilogue_fusions(dt):
    ops = _imp()
    M, N, K = 392, 160, 72   # 2 samples x 196 rows
    g = gen(2)
    a, A = rnd((M, K), dt, g)
    b, B = rnd((N, K), dt, g, 1 / math.sqrt(K))
    r, R = rnd((M, N), dt, g)
    h, H = rnd((M, N), dt, g)
    rs = torch.tensor([0.0, 1.25])
    bias = torch.randn(N, generator=g)
    # residual + rowscale + relu_after + column statistics
    Cout = torch.empty(M, N, dtype=dt, device='cuda')
    cs = torch.zeros(N, device='cuda')
    cq = torch.zeros(N, device='cuda')
    ops.Plan(eager=True).gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), rowscale=rs.cuda(),
                              rows_per_scale=196, R=R, ldr=N, relu_after=True, colsum=cs, colsumsq=cq)
    ref = F.relu((a @ b.t() + bias) * rs.repeat_interleave(196)[:, None] + r)
    assert_close(Cout, ref, tol(dt), 'res+rowscale+relu')
    assert_close(cs, ref.sum(0), tol(dt, 2), 'colsum')
    assert_close(cq, (ref * ref).sum(0), tol(dt, 2), 'colsumsq')
    # GELU-backward multiplier + fp32 output
    C32 = torch.empty(M, N, dtype=torch.float32, device='cuda')
    ops.Plan(eager=True).gemm(A, B, C32, M, N, K, ops.ga_dtype(dt), H=H, ldh=N, c_f32=True)
    hh = h.clone().requires_grad_(True)
    F.gelu(hh).sum().backward()
    assert_close(C32, (a @ b.t()) * hh.grad, tol(dt), 'gelu-bwd')
    # fc1-style double output: C = gelu(pre), C2 = gelu'(pre); then multiply-by-stored-derivative (dgrad2 style)
    Ca = torch.empty(M, N, dtype=dt, device='cuda'); Cg = torch.empty(M, N, dtype=dt, device='cuda')
    ops.Plan(eager=True).gemm(A, B, Ca, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), act=ops.ACT_GELU, C2=Cg, c2_mode=2)
    pre = (a @ b.t() + bias).requires_grad_(True)
    act = F.gelu(pre)
    act.sum().backward()
    assert_close(Ca, act, tol(dt), 'fc1 act')
    assert_close(Cg, pre.grad, tol(dt), "fc1 gelu'")
    ops.Plan(eager=True).gemm(A, B, C32, M, N, K, ops.ga_dtype(dt), H=Cg, ldh=N, h_is_deriv=True, c_f32=True)
    assert_close(C32, (a @ b.t()) * Cg.float().cpu(), tol(dt), 'mul by stored derivative')
    ops.Plan(eager=True).gemm(A, B, Ca, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), act=ops.ACT_RELU, C2=Cg, c2_mode=1)
    assert_close(Cg, a @ b.t() + bias, tol(dt), 'pre-activation copy')
    # GELU applied to A while staging
    ops.Plan(eager=True).gemm(A, B, C32, M, N, K, ops.ga_dtype(dt), a_act=ops.ACT_GELU, c_f32=True)
    ag = F.gelu(a).to(dt).float() if dt == torch.bfloat16 else F.gelu(a)
    assert_close(C32, ag @ b.t(), tol(dt), 'a_act gelu')


@pytest.mark.parametrize('dt', DT)
def test_gemm_batched_grouped(dt):
    ops = _imp()
    g = gen(3)
    Bz, M, N, K = 6, 50, 24, 40
    a, A = rnd((2, M, K), dt, g)            # A shared modulo 2
    b, Bm = rnd((Bz, N, K), dt, g)
    bias = torch.randn(Bz, N, generator=g)
    Cout = torch.empty(M, Bz * N, dtype=dt, device='cuda')  # each batch writes its own column block
    ops.Plan(eager=True).gemm(A, Bm, Cout, M, N, K, ops.ga_dtype(dt), batch=Bz, strideA=M * K, a_batch_mod=2,
                              strideB=N * K, ldc=Bz * N, strideC=N, bias=bias.cuda(), strideBias=N)
    ref = torch.cat([a[z % 2] @ b[z].t() + bias[z] for z in range(Bz)], dim=1)
    assert_close(Cout, ref, tol(dt), 'batched')


@pytest.mark.parametrize('dt', DT)
def test_gemm_patch2_and_unpatch(dt):
    ops = _imp()
    g = gen(4)
    Bn, H, W, Cc, N = 2, 8, 12, 16, 24
    x, X = rnd((Bn, H, W, Cc), dt, g)
    w, _ = rnd((N, Cc, 2, 2), dt, g, 0.2)
    Wm = w.permute(0, 2, 3, 1).reshape(N, 4 * Cc).contiguous()
    M = Bn * (H // 2) * (W // 2)
    Cout = torch.empty(M, N, dtype=dt, device='cuda')
    ops.Plan(eager=True).gemm(X, Wm.to(dt).cuda(), Cout, M, N, 4 * Cc, ops.ga_dtype(dt), a_kind=ops.A_PATCH2,
                              a_dims=(H, W, Cc))
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, stride=2).permute(0, 2, 3, 1).reshape(M, N)
    assert_close(Cout, ref, tol(dt), 'patch2 fwd')
    # dgrad: dX = unpatch(dY @ Wm)  -> B operand = Wm^T [4C][N]
    dy, DY = rnd((M, N), dt, g)
    DX = torch.zeros(Bn, H, W, Cc, dtype=dt, device='cuda')
    ops.Plan(eager=True).gemm(DY, Wm.t().contiguous().to(dt).cuda(), DX, M, 4 * Cc, N, ops.ga_dtype(dt),
                              c_kind=ops.C_UNPATCH2, c_dims=(H, W, Cc))
    xx = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.conv2d(xx, w, stride=2).backward(dy.reshape(Bn, H // 2, W // 2, N).permute(0, 3, 1, 2))
    assert_close(DX, xx.grad.permute(0, 2, 3, 1), tol(dt), 'patch2 dgrad')


@pytest.mark.parametrize('geom', [(2, 8, 12, 16, 24), (3, 28, 28, 96, 192), (5, 14, 14, 384, 768), (1, 10, 6, 48, 136), (40, 56, 56, 96, 192)])
def test_gemm_patch2_ring3_form_bf16(geom, knobs):
    """the downsample conv (2 x 2 / stride 2, ga_convnext.py:127) on the 3-slot ring form: patch rows fetched as two runs of 2C
    elements by LDS-DMA (lane offsets at the patch origin, the second run through the scalar offset), bias epilogue"""
    ops = _imp()
    knobs(NT_R3=15)
    dt = torch.bfloat16
    g = gen(14)
    Bn, H, W, Cc, N = geom
    x, X = rnd((Bn, H, W, Cc), dt, g)
    w, _ = rnd((N, Cc, 2, 2), dt, g, 0.5 / math.sqrt(Cc))
    bias = torch.randn(N, generator=g)
    Wm = w.permute(0, 2, 3, 1).reshape(N, 4 * Cc).contiguous()
    M = Bn * (H // 2) * (W // 2)
    Cout = torch.empty(M, N, dtype=dt, device='cuda')
    ops.Plan(eager=True).gemm(X, Wm.to(dt).cuda(), Cout, M, N, 4 * Cc, ops.ga_dtype(dt), a_kind=ops.A_PATCH2, a_dims=(H, W, Cc),
                              bias=bias.cuda())
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=2).permute(0, 2, 3, 1).reshape(M, N)
    assert_close(Cout, ref, 2e-2, 'patch2 on the ring form')
    # its data gradient: dX = unpatch(dY @ Wm) -- rows scattered back to the 2 x 2 pixels of the NHWC map (GA_C_UNPATCH2)
    if N % 8 == 0:
        dy, DY = rnd((M, N), dt, g)
        DX = torch.zeros(Bn, H, W, Cc, dtype=dt, device='cuda')
        ops.Plan(eager=True).gemm(DY, Wm.t().contiguous().to(dt).cuda(), DX, M, 4 * Cc, N, ops.ga_dtype(dt), c_kind=ops.C_UNPATCH2,
                                  c_dims=(H, W, Cc))
        xx = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
        F.conv2d(xx, w.to(dt).float(), stride=2).backward(dy.reshape(Bn, H // 2, W // 2, N).permute(0, 3, 1, 2))
        assert_close(DX, xx.grad.permute(0, 2, 3, 1), 2e-2, 'unpatch2 on the ring form')


@pytest.mark.parametrize('geom', [(2, 16, 32), (3, 8, 16), (1, 24, 48), (3, 112, 112)])     # (the last: more tiles than workgroups)
def test_gemm_conv3_direct_form_64_channels(geom, knobs):
    """the direct 3 x 3 / 64 -> 64 channel kernel behind ga_gemm's GA_A_CONV3 product (csrc/conv3.hip: GA-CSWin's deep stem):
    forward and backward-data against F.conv2d, and bit-identical?  no: against the implicit-GEMM form within bf16 rounding"""
    ops = _imp()
    dt = torch.bfloat16
    Bn, H, W = geom
    Cc = N = 64
    g = gen(6)
    x, X = rnd((Bn, H, W, Cc), dt, g)
    w, _ = rnd((N, Cc, 3, 3), dt, g, 0.05)
    M = Bn * H * W
    P = ops.Plan(eager=True)
    Wf = torch.empty(N, 9 * Cc, dtype=dt, device='cuda')
    WT = torch.empty(Cc, 9 * N, dtype=dt, device='cuda')
    P.weight_prep(w.cuda(), 1, N, Cc, 3, 3, ops.ga_dtype(dt), out=Wf, ldo=9 * Cc, outT=WT, ldt=9 * N, flip=True)
    xx = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.conv2d(xx, w, padding=1)
    dy, DY = rnd((M, N), dt, g)
    y.backward(dy.reshape(Bn, H, W, N).permute(0, 3, 1, 2))
    outs = {}
    for form in (1, 0):
        knobs(CONV3_DIRECT=form)
        Cout = torch.full((M, N), 777.0, dtype=dt, device='cuda')
        P.gemm(X, Wf, Cout, M, N, 9 * Cc, ops.ga_dtype(dt), a_kind=ops.A_CONV3, a_dims=(H, W, Cc))
        DX = torch.full((M, Cc), 777.0, dtype=dt, device='cuda')
        P.gemm(DY, WT, DX, M, Cc, 9 * N, ops.ga_dtype(dt), a_kind=ops.A_CONV3, a_dims=(H, W, N))
        torch.cuda.synchronize()
        assert_close(Cout, y.permute(0, 2, 3, 1).reshape(M, N), tol(dt), f'conv3 fwd (direct={form})')
        assert_close(DX, xx.grad.permute(0, 2, 3, 1).reshape(M, Cc), tol(dt), f'conv3 dgrad (direct={form})')
        # weight gradient of the same layer (ga_wgrad's GA_A_CONV3 product: conv3_c64_wgrad_kernel when direct): accumulated into dW
        dW = torch.full((N, 9 * Cc), 0.5, device='cuda')
        P.wgrad(DY, X, dW, M, N, 9 * Cc, ops.ga_dtype(dt), x_kind=ops.A_CONV3, x_dims=(H, W, Cc))
        torch.cuda.synchronize()
        wg = F.conv2d(x.permute(0, 3, 1, 2), torch.zeros(N, Cc, 3, 3, requires_grad=True), padding=1)
        ww = torch.zeros(N, Cc, 3, 3, requires_grad=True)
        F.conv2d(x.permute(0, 3, 1, 2), ww, padding=1).backward(dy.reshape(Bn, H, W, N).permute(0, 3, 1, 2))
        assert_close(dW, 0.5 + ww.grad.permute(0, 2, 3, 1).reshape(N, 9 * Cc), tol(dt, 2), f'conv3 wgrad (direct={form})')
        outs[form] = (Cout.float(), DX.float())
    assert float((outs[1][0] - outs[0][0]).abs().max()) <= 2e-2 * float(outs[0][0].abs().max())


@pytest.mark.parametrize('dt', DT)
def test_gemm_conv3_fwd_and_dgrad(dt):
    ops = _imp()
    g = gen(5)
    Bn, H, W, Cc, N = 3, 6, 5, 16, 24
    x, X = rnd((Bn, H, W, Cc), dt, g)
    w, _ = rnd((N, Cc, 3, 3), dt, g, 0.15)
    M = Bn * H * W
    P = ops.Plan(eager=True)
    Wf = torch.empty(N, 9 * Cc, dtype=dt, device='cuda')
    WT = torch.empty(Cc, 9 * N, dtype=dt, device='cuda')
    P.weight_prep(w.cuda(), 1, N, Cc, 3, 3, ops.ga_dtype(dt), out=Wf, ldo=9 * Cc, outT=WT, ldt=9 * N, flip=True)
    assert_close(Wf, w.permute(0, 2, 3, 1).reshape(N, 9 * Cc), 1e-6, 'wprep conv3')
    Cout = torch.empty(M, N, dtype=dt, device='cuda')
    P.gemm(X, Wf, Cout, M, N, 9 * Cc, ops.ga_dtype(dt), a_kind=ops.A_CONV3, a_dims=(H, W, Cc))
    xx = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.conv2d(xx, w, padding=1)
    assert_close(Cout, y.permute(0, 2, 3, 1).reshape(M, N), tol(dt), 'conv3 fwd')
    dy, DY = rnd((M, N), dt, g)
    DX = torch.empty(M, Cc, dtype=dt, device='cuda')
    P.gemm(DY, WT, DX, M, Cc, 9 * N, ops.ga_dtype(dt), a_kind=ops.A_CONV3, a_dims=(H, W, N))
    y.backward(dy.reshape(Bn, H, W, N).permute(0, 3, 1, 2))
    assert_close(DX, xx.grad.permute(0, 2, 3, 1).reshape(M, Cc), tol(dt), 'conv3 dgrad')


@pytest.mark.parametrize('dt', DT)
def test_gemm_stem(dt):
    ops = _imp()
    g = gen(6)
    Bn, H, W, N = 2, 16, 24, 32
    x = torch.randn(Bn, 3, H, W, generator=g)
    w, _ = rnd((N, 3, 4, 4), dt, g, 0.2)
    M = Bn * (H // 4) * (W // 4)
    P = ops.Plan(eager=True)
    Wf = torch.empty(N, 48, dtype=dt, device='cuda')
    P.weight_prep(w.cuda(), 1, N, 3, 4, 4, ops.ga_dtype(dt), out=Wf, ldo=48, stem=True)
    Cout = torch.empty(M, N, dtype=dt, device='cuda')
    P.gemm(x.cuda(), Wf, Cout, M, N, 48, ops.ga_dtype(dt), a_kind=ops.A_STEM4_NCHW, a_dims=(H, W, 3))
    xr = x.to(dt).float()
    ref = F.conv2d(xr, w, stride=4).permute(0, 2, 3, 1).reshape(M, N)
    assert_close(Cout, ref, tol(dt), 'stem')
    # wgrad with the same gather
    dy, DY = rnd((M, N), dt, g)
    dW = torch.zeros(N, 48, device='cuda')
    P.wgrad(DY, x.cuda(), dW, M, N, 48, ops.ga_dtype(dt), x_kind=ops.A_STEM4_NCHW, x_dims=(H, W, 3), split_m=2)
    ww = w.clone().requires_grad_(True)
    F.conv2d(xr, ww, stride=4).backward(dy.reshape(Bn, H // 4, W // 4, N).permute(0, 3, 1, 2))
    assert_close(dW, ww.grad.reshape(N, 48), tol(dt, 2), 'stem wgrad')


@pytest.mark.parametrize('geom', [(2, 16, 24, 96), (3, 224, 224, 96), (1, 4, 4, 128), (5, 36, 20, 128)])
def test_stem_conv_layernorm_one_pass(geom):
    """ga_stem4_ln_fwd (ConvNeXt stem, ga_convnext.py:431-434: Conv2d(3, C, 4, 4) + LayerNorm2d) against the two launches it replaces
    (stem gather GEMM + ga_layernorm_fwd: `pre` bit-identical, y / mean / rstd to rounding) and against torch"""
    ops = _imp()
    dt = torch.bfloat16
    g = gen(21)
    Bn, H, W, N = geom
    x = torch.randn(Bn, 3, H, W, generator=g)
    w, _ = rnd((N, 3, 4, 4), dt, g, 0.2)
    bias, gam, bet = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    M = Bn * (H // 4) * (W // 4)
    P = ops.Plan(eager=True)
    Wf = torch.empty(N, 48, dtype=dt, device='cuda')
    P.weight_prep(w.cuda(), 1, N, 3, 4, 4, ops.GA_BF16, out=Wf, ldo=48, stem=True)
    X, Bi, Ga, Be = x.cuda(), bias.cuda(), gam.cuda(), bet.cuda()
    pre2, y2 = torch.empty(M, N, dtype=dt, device='cuda'), torch.empty(M, N, dtype=dt, device='cuda')
    m2, r2 = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    P.gemm(X, Wf, pre2, M, N, 48, ops.GA_BF16, a_kind=ops.A_STEM4_NCHW, a_dims=(H, W, 3), bias=Bi)
    P.layernorm_fwd(pre2, Ga, Be, y2, m2, r2, M, N, 1e-6, ops.GA_BF16)
    pre1, y1 = torch.full((M, N), 7.0, dtype=dt, device='cuda'), torch.full((M, N), 7.0, dtype=dt, device='cuda')
    m1, r1 = torch.zeros(M, device='cuda'), torch.zeros(M, device='cuda')
    P.stem4_ln_fwd(X, Wf, 48, Bi, Ga, Be, pre1, y1, m1, r1, Bn, H, W, N, 1e-6)
    ref = F.conv2d(x.to(dt).float(), w, bias, stride=4).permute(0, 2, 3, 1).reshape(M, N)
    assert_close(pre1, ref, tol(dt), 'stem pre vs torch')
    assert_close(pre1, pre2.float().cpu(), 1e-2, 'stem pre vs the gather GEMM')
    pr = pre1.float().cpu()
    yref = F.layer_norm(pr, (N,), gam, bet, 1e-6)
    assert_close(y1, yref, 2e-2, 'stem y vs torch on the rounded pre')
    assert_close(y1, y2.float().cpu(), 2e-2, 'stem y vs ga_layernorm_fwd')
    assert_close(m1, pr.mean(1), 1e-4, 'mean')
    assert_close(r1, 1.0 / torch.sqrt(pr.var(1, unbiased=False) + 1e-6), 1e-4, 'rstd')


@pytest.mark.parametrize('form', ['dma256', 'dma128', 't256', 'pp', 'r3'])
@pytest.mark.parametrize('shape', [(70000, 384, 96), (66000, 192, 200), (65600, 768, 384), (70000, 96, 384), (66000, 512, 328),
                                   (33000, 1536, 768), (40100, 264, 520), (12500, 3072, 768), (50200, 384, 1536)])
def test_gemm_lds_dma_form_bf16(shape, form, knobs):
    """large-M bf16 launches take the 256-row LDS-DMA form (ragged M, ragged K slab, 96- and 128-wide column tiles):
    plain + column sums, fc1 (GELU and GELU' outputs), fc2 (row scale + residual), dgrad2 (x stored GELU' + sums)"""
    ops = _imp()
    if form == 'dma256':
        knobs(NT_DMA=2)      # every eligible launch, not only the shapes the heuristic picks
    elif form == 't256':                             # 256 x 256 tile, 8 waves (N % 256 == 0 and K >= 256 only)
        knobs(NT_DMA=0, NT_T256=15)
    elif form == 'pp':                               # 8-wave ping-pong form (K >= 256)
        knobs(NT_PP=15)
    elif form == 'r3':                               # 256 x 128 tile, 4 waves, 3-slot ring of 32-deep stages, two workgroups per CU
        knobs(NT_R3=15)
    else:                                            # 128 x 128 tile, 4 waves, 2-slot ring, two workgroups per CU
        knobs(NT_DMA=0, NT_DMA2=15, NT_DMA2_MINK=8)
    dt = torch.bfloat16
    M, N, K = shape
    g = gen(5)
    a, A = rnd((M, K), dt, g)
    b, B = rnd((N, K), dt, g, 1 / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    h, Hm = rnd((M, N), dt, g)
    rs = torch.rand(M // 100, generator=g) + 0.5
    base = a @ b.t() + bias
    P = ops.Plan(eager=True)
    # plain + column sums
    Cout = torch.empty(M, N, dtype=dt, device='cuda')
    cs, cq = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), colsum=cs, colsumsq=cq)
    assert_close(Cout, base, 2e-2, 'dma plain')
    assert_close(cs, base.sum(0), 2e-3, 'dma colsum')
    assert_close(cq, (base * base).sum(0), 2e-3, 'dma colsumsq')
    # fc1: gelu + gelu'
    C2 = torch.empty_like(Cout)
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), act=ops.ACT_GELU, C2=C2, c2_mode=2)
    xr = base.clone().requires_grad_(True)
    act = F.gelu(xr)
    act.sum().backward()
    assert_close(Cout, act, 2e-2, 'dma fc1 gelu')
    assert_close(C2, xr.grad, 2e-2, "dma fc1 gelu'")
    # fc2: row scale (one per 100 rows) + residual
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), rowscale=rs.cuda(), rows_per_scale=100, R=Hm, ldr=N)
    assert_close(Cout, base * rs.repeat_interleave(100)[:, None] + h, 2e-2, 'dma fc2')
    # dgrad2: multiply by the stored derivative, column sums of the result
    cs.zero_()
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), H=Hm, ldh=N, h_is_deriv=True, colsum=cs)
    ref = (a @ b.t()) * h
    assert_close(Cout, ref, 2e-2, 'dma dgrad2')
    assert_close(cs, ref.sum(0), 3e-3, 'dma dgrad2 colsum')


@pytest.mark.parametrize('shape', [(300, 136, 64, 1), (257, 96, 72, 1), (1000, 1000, 768, 1), (5, 40, 96, 1), (2600, 384, 200, 2),
                                   (131072 + 40, 128, 96, 1), (70, 1000, 1536, 1)])
def test_gemm_ring3_form_small_and_batched(shape, knobs):
    """the 3-slot ring form on shapes that stress its stream logic: fewer tiles than workgroups, one tile per workgroup, K of two
    and three stages with a ragged tail, column tiles that are mostly empty, batch > 1 (grid z), more tiles than workgroups"""
    ops = _imp()
    knobs(NT_R3=15)
    dt = torch.bfloat16
    M, N, K, Z = shape
    g = gen(9)
    a, A = rnd((Z, M, K), dt, g)
    b, B = rnd((Z, N, K), dt, g, 1 / math.sqrt(K))
    bias = torch.randn(Z, N, generator=g)
    h, Hm = rnd((Z, M, N), dt, g)
    base = a @ b.transpose(1, 2) + bias[:, None, :]
    P = ops.Plan(eager=True)
    kw = dict(batch=Z, strideA=M * K, strideB=N * K, strideC=M * N)
    Cout = torch.empty(Z, M, N, dtype=dt, device='cuda')
    cs, cq = torch.zeros(Z, N, device='cuda'), torch.zeros(Z, N, device='cuda')
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), strideBias=N, colsum=cs, colsumsq=cq, strideCol=N, **kw)
    assert_close(Cout, base, 2e-2, 'r3 plain')
    assert_close(cs, base.sum(1), 2e-3, 'r3 colsum')
    assert_close(cq, (base * base).sum(1), 2e-3, 'r3 colsumsq')
    C2 = torch.empty_like(Cout)
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), strideBias=N, act=ops.ACT_GELU, C2=C2, c2_mode=2, **kw)
    xr = base.clone().requires_grad_(True)
    act = F.gelu(xr)
    act.sum().backward()
    assert_close(Cout, act, 2e-2, 'r3 fc1 gelu')
    assert_close(C2, xr.grad, 2e-2, "r3 fc1 gelu'")
    rs = torch.rand((M + 6) // 7, generator=g) + 0.5
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), strideBias=N, rowscale=rs.cuda(), rows_per_scale=7, R=Hm, ldr=N,
           strideR=M * N, **kw)
    assert_close(Cout, base * rs.repeat_interleave(7)[:M][None, :, None] + h, 2e-2, 'r3 fc2')
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), bias=bias.cuda(), strideBias=N, R=Hm, ldr=N, strideR=M * N, **kw)
    assert_close(Cout, base + h, 2e-2, 'r3 fc2 without row scale')
    cs.zero_()
    P.gemm(A, B, Cout, M, N, K, ops.ga_dtype(dt), H=Hm, ldh=N, strideH=M * N, h_is_deriv=True, colsum=cs, strideCol=N, **kw)
    ref = (a @ b.transpose(1, 2)) * h
    assert_close(Cout, ref, 2e-2, 'r3 dgrad2')
    assert_close(cs, ref.sum(1), 3e-3, 'r3 dgrad2 colsum')


# ----------------------------------------------------------------------------------------------------------
# wgrad (TN)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('shape', [(1000, 136, 72), (4096, 384, 96), (70, 40, 128), (300, 8, 8)])
def test_wgrad_plain(dt, shape):
    ops = _imp()
    M, N, K = shape
    g = gen(7)
    y, Y = rnd((M, N), dt, g)
    x, X = rnd((M, K), dt, g)
    for split in (1, 3):
        dW = torch.full((N, K), 1.0, device='cuda')
        db = torch.zeros(N, device='cuda')
        ops.Plan(eager=True).wgrad(Y, X, dW, M, N, K, ops.ga_dtype(dt), dbias=db, split_m=split, alpha=0.5)
        assert_close(dW, 1.0 + 0.5 * y.t() @ x, tol(dt, 2), f'wgrad split {split}')
        assert_close(db, 0.5 * y.sum(0), tol(dt, 2), 'dbias')
    dW = torch.full((N, K), 7.0, device='cuda')
    ops.Plan(eager=True).wgrad(Y, X, dW, M, N, K, ops.ga_dtype(dt), split_m=1, accumulate=False, x_act=ops.ACT_GELU)
    xg = F.gelu(x).to(dt).float()
    assert_close(dW, y.t() @ xg, tol(dt, 2), 'wgrad overwrite + gelu(X)')


@pytest.mark.parametrize('shape', [(8192, 384, 1536), (8192, 96, 384), (16384, 384, 96), (8256, 200, 328), (12800, 768, 192), (8224, 136, 264)])
def test_wgrad_wide_tile_bf16(shape):
    """the 256x256-tile LDS-DMA form (bf16, M % 64 == 0, M >= 8192, accumulating output): ragged N / K, padded
    leading dimensions, bias column sums, its own choice of row splits"""
    ops = _imp()
    dt = torch.bfloat16
    M, N, K = shape
    g = gen(11)
    ldy, ldx = N + 8, K + 16
    y, _ = rnd((M, ldy), dt, g)
    x, _ = rnd((M, ldx), dt, g)
    Y, X = y.to(dt).cuda(), x.to(dt).cuda()
    y, x = y[:, :N], x[:, :K]
    ref = y.t() @ x
    for split, acc in ((2, False), (1, True)):
        dW = torch.full((N, K), 1.0, device='cuda')
        db = torch.zeros(N, device='cuda')
        ops.Plan(eager=True).wgrad(Y, X, dW, M, N, K, ops.ga_dtype(dt), ldy=ldy, ldx=ldx, dbias=db, split_m=split,
                                   accumulate=acc, alpha=0.25)
        assert_close(dW, 1.0 + 0.25 * ref, 2e-4, f'wide wgrad split {split}')
        assert_close(db, 0.25 * y.sum(0), 2e-4, 'wide dbias')


@pytest.mark.parametrize('dt', DT)
def test_wgrad_gather_kinds(dt):
    ops = _imp()
    g = gen(8)
    Bn, H, W, Cc, N = 2, 8, 12, 16, 24
    x, X = rnd((Bn, H, W, Cc), dt, g)
    # PATCH2
    M = Bn * (H // 2) * (W // 2)
    dy, DY = rnd((M, N), dt, g)
    dW = torch.zeros(N, 4 * Cc, device='cuda')
    ops.Plan(eager=True).wgrad(DY, X, dW, M, N, 4 * Cc, ops.ga_dtype(dt), x_kind=ops.A_PATCH2, x_dims=(H, W, Cc), split_m=2)
    w = torch.zeros(N, Cc, 2, 2, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2), w, stride=2).backward(dy.reshape(Bn, H // 2, W // 2, N).permute(0, 3, 1, 2))
    assert_close(dW, w.grad.permute(0, 2, 3, 1).reshape(N, 4 * Cc), tol(dt, 2), 'patch2 wgrad')
    # CONV3
    M = Bn * H * W
    dy, DY = rnd((M, N), dt, g)
    dW = torch.zeros(N, 9 * Cc, device='cuda')
    ops.Plan(eager=True).wgrad(DY, X, dW, M, N, 9 * Cc, ops.ga_dtype(dt), x_kind=ops.A_CONV3, x_dims=(H, W, Cc), split_m=3)
    w = torch.zeros(N, Cc, 3, 3, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2), w, padding=1).backward(dy.reshape(Bn, H, W, N).permute(0, 3, 1, 2))
    assert_close(dW, w.grad.permute(0, 2, 3, 1).reshape(N, 9 * Cc), tol(dt, 2), 'conv3 wgrad')


@pytest.mark.parametrize('dt', DT)
def test_wgrad_batched_gram(dt):
    ops = _imp()
    g = gen(9)
    Bn, HW, Cc = 5, 196, 32
    x, X = rnd((Bn, HW, Cc), dt, g)
    G = torch.empty(Bn, Cc, Cc, device='cuda')
    ops.Plan(eager=True).wgrad(X, X, G, HW, Cc, Cc, ops.ga_dtype(dt), batch=Bn, strideY=HW * Cc, strideX=HW * Cc,
                               strideW=Cc * Cc, split_m=1, accumulate=False, alpha=1.0 / HW)
    assert_close(G, torch.bmm(x.transpose(1, 2), x) / HW, tol(dt, 2), 'gram')


# ----------------------------------------------------------------------------------------------------------
# weight prep / unfold
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DT)
def test_weight_prep_fold_unfold(dt):
    ops = _imp()
    g = gen(10)
    N, Cc = 48, 20
    w = torch.randn(N, Cc, generator=g)
    b = torch.randn(N, generator=g)
    rs = torch.rand(N, generator=g) + 0.5
    cs = torch.rand(Cc, generator=g) + 0.5
    v = torch.randn(Cc, generator=g)
    P = ops.Plan(eager=True)
    ldo, ldt = 24, 48
    out = torch.empty(N, ldo, dtype=dt, device='cuda')
    outT = torch.empty(Cc, ldt, dtype=dt, device='cuda')
    P.weight_prep(w.cuda(), 1, N, Cc, 1, 1, ops.ga_dtype(dt), out=out, ldo=ldo, outT=outT, ldt=ldt, rs=rs.cuda(), cs=cs.cuda())
    we = rs[:, None] * w * cs[None, :]
    assert_close(out[:, :Cc], we, tol(dt, 0.5), 'fold out')
    assert float(out[:, Cc:].float().abs().max()) == 0.0
    assert_close(outT, we.t(), tol(dt, 0.5), 'fold outT')
    be = torch.empty(N, device='cuda')
    P.bias_fold(w.cuda(), b.cuda(), rs.cuda(), v.cuda(), be, N, Cc)
    assert_close(be, rs * (b + w @ v), 1e-5, 'bias fold')
    # unfold against autograd of the folding
    G = torch.randn(N, ldo, generator=g)
    gb = torch.randn(N, generator=g)
    leaf = [t.clone().requires_grad_(True) for t in (w, b, rs, cs, v)]
    lw, lb, lrs, lcs, lv = leaf
    ((lrs[:, None] * lw * lcs[None, :]) * G[:, :Cc]).sum().add((lrs * (lb + lw @ lv) * gb).sum()).backward()
    dW = torch.zeros(N, Cc, device='cuda'); db = torch.zeros(N, device='cuda')
    d_rs = torch.zeros(N, device='cuda'); d_cs = torch.zeros(Cc, device='cuda'); d_v = torch.zeros(Cc, device='cuda')
    P.weight_unfold(G.cuda(), ldo, N, Cc, gb=gb.cuda(), W=w.cuda(), b=b.cuda(), rs=rs.cuda(), cs=cs.cuda(), v=v.cuda(),
                    dW=dW, db=db, d_rs=d_rs, d_cs=d_cs, d_v=d_v)
    assert_close(dW, lw.grad, 1e-5, 'unfold dW')
    assert_close(db, lb.grad, 1e-5, 'unfold db')
    assert_close(d_cs, lcs.grad, 1e-5, 'unfold d_cs')
    assert_close(d_v, lv.grad, 1e-5, 'unfold d_v')
    assert_close(d_rs, lrs.grad, 1e-5, 'unfold d_rs')
    # grouped (G=4) transposed copy
    Gn, Co, Ci = 4, 6, 10
    wg = torch.randn(Gn * Co, Ci, generator=g)
    outT = torch.empty(Gn, Ci, 8, dtype=dt, device='cuda')
    P.weight_prep(wg.cuda(), Gn, Co, Ci, 1, 1, ops.ga_dtype(dt), outT=outT, ldt=8)
    assert_close(outT[:, :, :Co], wg.reshape(Gn, Co, Ci).transpose(1, 2), tol(dt, 0.5), 'grouped outT')
    t49 = torch.randn(12, 49, generator=g)
    o = torch.empty(49, 12, device='cuda')
    P.transpose_f32(t49.cuda(), o, 12, 49)   # in [R=12][C=49] -> out [49][12]
    assert_close(o, t49.t(), 1e-7, 'transpose')


# ----------------------------------------------------------------------------------------------------------
# depthwise 7x7
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('geom', [(2, 14, 14, 96), (1, 28, 28, 72), (3, 7, 7, 128), (2, 10, 9, 16), (1, 56, 56, 32),
                                  (2, 28, 28, 40, 'mfma'), (2, 14, 14, 96, 'lds'), (1, 56, 56, 32, 'lds'), (3, 28, 21, 48, 'rs'),
                                  (5, 14, 14, 192, 'rs'), (2, 7, 7, 768, 'rs')])
def test_dwconv7(dt, geom, knobs):
    ops = _imp()
    if len(geom) == 5:
        if geom[4] == 'mfma':     # force the matrix-core (Toeplitz) form of the bf16 forward / backward-data kernel
            knobs(DW_RS=0, DW_MFMA=2)
        elif geom[4] == 'lds':    # the LDS-staged dot2 forms (what serves maps whose height is not a multiple of 7)
            knobs(DW_RS=0, DWW_RS=0)
        else:                     # the register-sliding forms everywhere they apply (backward-weight: also on small maps)
            knobs(DWW_RS=2)
        geom = geom[:4]
    Bn, H, W, Cc = geom
    g = gen(11)
    x, X = rnd((Bn, H, W, Cc), dt, g)
    w = torch.randn(Cc, 1, 7, 7, generator=g) * 0.1
    b = torch.randn(Cc, generator=g)
    w49 = w.reshape(Cc, 49).t().contiguous().cuda()
    P = ops.Plan(eager=True)
    Y = torch.empty_like(X)
    P.dwconv7_fwd(X, w49, b.cuda(), Y, Bn, H, W, Cc, ops.ga_dtype(dt))
    xx = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ww = w.clone().requires_grad_(True)
    bb = b.clone().requires_grad_(True)
    y = F.conv2d(xx, ww, bb, padding=3, groups=Cc)
    assert_close(Y, y.permute(0, 2, 3, 1), tol(dt), 'dwconv fwd')
    dy, DY = rnd((Bn, H, W, Cc), dt, g)
    r, R = rnd((Bn, H, W, Cc), dt, g)
    y.backward(dy.permute(0, 3, 1, 2))
    DX = torch.empty_like(X)
    P.dwconv7_bwd_data(DY, w49, R, DX, Bn, H, W, Cc, ops.ga_dtype(dt))
    assert_close(DX, xx.grad.permute(0, 2, 3, 1) + r, tol(dt), 'dwconv bwd data')
    # second output: the stored dx times a per-image factor (the next block's DropPath scale), bit-identical to a
    # separate ga_rowscale pass over dx
    sc = torch.tensor(([0.0, 1.25, 2.0, 1.0, 0.5] * 2)[:Bn], device='cuda')
    DXa, DX2, REF2 = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
    P.dwconv7_bwd_data(DY, w49, R, DXa, Bn, H, W, Cc, ops.ga_dtype(dt), dx2=DX2, scale2=sc)
    P.rowscale(DXa, sc, REF2, DXa.numel(), H * W * Cc, ops.ga_dtype(dt))
    assert torch.equal(DXa, DX) and torch.equal(DX2, REF2)
    dw49 = torch.zeros(49, Cc, device='cuda')
    db = torch.zeros(Cc, device='cuda')
    P.dwconv7_bwd_weight(DY, X, dw49, db, Bn, H, W, Cc, ops.ga_dtype(dt))
    assert_close(dw49, ww.grad.reshape(Cc, 49).t(), tol(dt, 2), 'dwconv bwd weight')
    assert_close(db, bb.grad, tol(dt, 2), 'dwconv bwd bias')


# ----------------------------------------------------------------------------------------------------------
# LayerNorm / BatchNorm
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('Cc', [16, 96, 192, 384, 768, 1024])
def test_layernorm(dt, Cc):
    ops = _imp()
    g = gen(12)
    rows = 333
    x, X = rnd((rows, Cc), dt, g, 2.0)
    w = torch.rand(Cc, generator=g) + 0.5
    b = torch.randn(Cc, generator=g)
    P = ops.Plan(eager=True)
    Y = torch.empty_like(X)
    mean = torch.empty(rows, device='cuda'); rstd = torch.empty(rows, device='cuda')
    P.layernorm_fwd(X, w.cuda(), b.cuda(), Y, mean, rstd, rows, Cc, 1e-6, ops.ga_dtype(dt))
    xx = x.clone().requires_grad_(True); ww = w.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
    y = F.layer_norm(xx, (Cc,), ww, bb, 1e-6)
    assert_close(Y, y, tol(dt), 'ln fwd')
    assert_close(mean, x.mean(1), 1e-5, 'ln mean')
    dy, DY = rnd((rows, Cc), dt, g)
    r, R = rnd((rows, Cc), dt, g)
    y.backward(dy)
    DX = torch.empty_like(X)
    dw = torch.zeros(Cc, device='cuda'); db = torch.zeros(Cc, device='cuda')
    P.layernorm_bwd(DY, X, mean, rstd, w.cuda(), R, DX, dw, db, rows, Cc, False, ops.ga_dtype(dt))
    assert_close(DX, xx.grad + r, tol(dt), 'ln bwd dx')
    assert_close(dw, ww.grad, tol(dt, 2), 'ln bwd dw')
    assert_close(db, bb.grad, tol(dt, 2), 'ln bwd db')
    # no-affine forward, backward from the normalised output
    P.layernorm_fwd(X, None, None, Y, None, rstd, rows, Cc, 1e-6, ops.ga_dtype(dt))
    xx = x.clone().requires_grad_(True)
    y = F.layer_norm(xx, (Cc,), None, None, 1e-6)
    assert_close(Y, y, tol(dt), 'ln fwd noaffine')
    y.backward(dy)
    P.layernorm_bwd(DY, Y, None, rstd, None, None, DX, None, None, rows, Cc, True, ops.ga_dtype(dt))
    assert_close(DX, xx.grad, tol(dt, 2), 'ln bwd from xhat')


@pytest.mark.parametrize('dt', DT)
def test_batchnorm_train_path(dt):
    ops = _imp()
    g = gen(13)
    rows, K, Cc = 392, 40, 64
    a, A = rnd((rows, K), dt, g)
    wt, Wt = rnd((Cc, K), dt, g, 0.3)
    res, RES = rnd((rows, Cc), dt, g)
    bw = torch.rand(Cc, generator=g) + 0.5
    bb = torch.randn(Cc, generator=g)
    rm0, rv0 = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    P = ops.Plan(eager=True)
    X = torch.empty(rows, Cc, dtype=dt, device='cuda')
    s = torch.zeros(Cc, device='cuda'); q = torch.zeros(Cc, device='cuda')
    P.gemm(A, Wt, X, rows, Cc, K, ops.ga_dtype(dt), colsum=s, colsumsq=q)
    rm, rv = rm0.clone().cuda(), rv0.clone().cuda()
    mean = torch.empty(Cc, device='cuda'); rstd = torch.empty(Cc, device='cuda')
    scale = torch.empty(Cc, device='cuda'); shift = torch.empty(Cc, device='cuda')
    P.bn_finalize(s, q, rows, bw.cuda(), bb.cuda(), 1e-5, 0.1, rm, rv, mean, rstd, scale, shift, Cc, True)
    Y = torch.empty_like(X)
    P.affine_act(X, scale, shift, RES, Y, rows, Cc, True, ops.ga_dtype(dt))
    # reference on the (rounded) conv output the kernel stored
    xr = X.float().cpu().requires_grad_(True)
    bwl = bw.clone().requires_grad_(True); bbl = bb.clone().requires_grad_(True)
    rmr, rvr = rm0.clone(), rv0.clone()
    y = F.relu(F.batch_norm(xr, rmr, rvr, bwl, bbl, True, 0.1, 1e-5) + res)
    assert_close(Y, y, tol(dt, 2), 'bn fwd')
    assert_close(rm, rmr, 5e-3 if dt == torch.bfloat16 else 1e-4, 'running mean')
    assert_close(rv, rvr, 5e-3 if dt == torch.bfloat16 else 1e-4, 'running var')
    dy, DY = rnd((rows, Cc), dt, g)
    y.backward(dy)
    s1 = torch.zeros(Cc, device='cuda'); s2 = torch.zeros(Cc, device='cuda')
    P.bn_bwd_reduce(DY, Y, X, mean, rstd, s1, s2, rows, Cc, ops.ga_dtype(dt))
    DX = torch.empty_like(X)
    P.bn_bwd_apply(DY, Y, X, mean, rstd, bw.cuda(), s1, s2, rows, DX, rows, Cc, ops.ga_dtype(dt))
    assert_close(DX, xr.grad, tol(dt, 3), 'bn bwd dx')
    assert_close(s2, bwl.grad, tol(dt, 3), 'bn dw')
    assert_close(s1, bbl.grad, tol(dt, 3), 'bn db')
    # eval mode scale/shift
    P.bn_finalize(None, None, 0, bw.cuda(), bb.cuda(), 1e-5, 0.1, rm, rv, None, None, scale, shift, Cc, False)
    P.affine_act(X, scale, shift, None, Y, rows, Cc, False, ops.ga_dtype(dt))
    ye = F.batch_norm(X.float().cpu(), rm.cpu(), rv.cpu(), bw, bb, False, 0.1, 1e-5)
    assert_close(Y, ye, tol(dt), 'bn eval')


# ----------------------------------------------------------------------------------------------------------
# head pieces
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DT)
def test_pool_concat(dt):
    ops = _imp()
    g = gen(14)
    Bn = 2
    srcs = [(56, 16, 0), (28, 24, 0), (14, 32, 0), (7, 40, 1)]
    ctot = sum(c for _, c, _ in srcs)
    P = ops.Plan(eager=True)
    cat = torch.empty(Bn, 14, 14, ctot, dtype=dt, device='cuda')
    xs, refs, off = [], [], 0
    for hw, c, mode in srcs:
        x, X = rnd((Bn, hw, hw, c), dt, g)
        xs.append((x, X, hw, c, mode, off))
        P.pool_concat_fwd(X, cat, Bn, hw, hw, c, 14, 14, ctot, off, mode, ops.ga_dtype(dt))
        xn = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
        refs.append((xn, F.adaptive_avg_pool2d(xn, 14) if mode == 0 else F.interpolate(xn, scale_factor=2, mode='bilinear')))
        off += c
    ref = torch.cat([r for _, r in refs], 1)
    assert_close(cat, ref.permute(0, 2, 3, 1), tol(dt), 'aggregate fwd')
    d, D = rnd((Bn, 14, 14, ctot), dt, g)
    ref.backward(d.permute(0, 3, 1, 2))
    for (x, X, hw, c, mode, off), (xn, _) in zip(xs, refs):
        r, R = rnd((Bn, hw, hw, c), dt, g)
        DX = torch.empty_like(X)
        P.pool_concat_bwd(D, R, DX, Bn, hw, hw, c, 14, 14, ctot, off, mode, ops.ga_dtype(dt))
        assert_close(DX, xn.grad.permute(0, 2, 3, 1) + r, tol(dt), f'aggregate bwd {hw}')


@pytest.mark.parametrize('dt', DT)
def test_squeeze_excite(dt):
    ops = _imp()
    g = gen(15)
    Bn, HW, Cc, Rr = 3, 196, 64, 16
    x, X = rnd((Bn, HW, Cc), dt, g)
    W1 = torch.randn(Rr, Cc, generator=g) * 0.2; b1 = torch.randn(Rr, generator=g) * 0.1
    W2 = torch.randn(Cc, Rr, generator=g) * 0.2; b2 = torch.randn(Cc, generator=g) * 0.1
    P = ops.Plan(eager=True)
    s = torch.empty(Bn, Cc, device='cuda'); hid = torch.empty(Bn, Rr, device='cuda'); gate = torch.empty(Bn, Cc, device='cuda')
    P.spatial_sum(X, None, s, Bn, HW, Cc, 1.0 / HW, ops.ga_dtype(dt))
    P.se_mlp_fwd(s, W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda(), hid, gate, Bn, Cc, Rr)
    Z = torch.empty_like(X)
    P.chan_scale(X, gate, None, Z, Bn, HW, Cc, ops.ga_dtype(dt))
    leaves = [t.clone().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    lx, lW1, lb1, lW2, lb2 = leaves
    sm = lx.mean(1)
    gt = torch.sigmoid(F.relu(sm @ lW1.t() + lb1) @ lW2.t() + lb2)
    z = lx * gt[:, None, :]
    assert_close(Z, z, tol(dt), 'se fwd')
    dz, DZ = rnd((Bn, HW, Cc), dt, g)
    z.backward(dz)
    dgate = torch.empty(Bn, Cc, device='cuda')
    P.spatial_sum(DZ, X, dgate, Bn, HW, Cc, 1.0, ops.ga_dtype(dt))
    ds = torch.empty(Bn, Cc, device='cuda')
    dW1 = torch.zeros(Rr, Cc, device='cuda'); db1 = torch.zeros(Rr, device='cuda')
    dW2 = torch.zeros(Cc, Rr, device='cuda'); db2 = torch.zeros(Cc, device='cuda')
    P.se_mlp_bwd(dgate, gate, hid, s, W1.cuda(), W2.cuda(), ds, dW1, db1, dW2, db2, Bn, Cc, Rr, ds_scale=1.0 / HW)
    DX = torch.empty_like(X)
    P.chan_scale(DZ, gate, ds, DX, Bn, HW, Cc, ops.ga_dtype(dt))
    assert_close(DX, lx.grad, tol(dt, 2), 'se bwd dx')
    for got, leaf, nm in ((dW1, lW1, 'dW1'), (db1, lb1, 'db1'), (dW2, lW2, 'dW2'), (db2, lb2, 'db2')):
        assert_close(got, leaf.grad, tol(dt, 2), 'se ' + nm)


@pytest.mark.parametrize('dt', DT)
def test_gram_vector_fwd_bwd(dt):
    ops = _imp()
    g = gen(16)
    Bn, Hh, Cc, groups = 4, 14, 32, 8
    HW = Hh * Hh
    ntri = Cc * (Cc + 1) // 2
    Kg = ntri // groups
    Kp = (Kg + 7) // 8 * 8
    x, X = rnd((Bn, HW, Cc), dt, g)
    alpha = 1.0 / (Hh * Hh * HW)
    P = ops.Plan(eager=True)
    G = torch.empty(Bn, Cc, Cc, device='cuda')
    P.wgrad(X, X, G, HW, Cc, Cc, ops.ga_dtype(dt), batch=Bn, strideY=HW * Cc, strideX=HW * Cc, strideW=Cc * Cc,
            split_m=1, accumulate=False, alpha=alpha)
    vec = torch.empty(Bn, groups * Kp, dtype=dt, device='cuda')
    inv = torch.empty(Bn, device='cuda')
    P.gram_pack_fwd(G, vec, inv, Bn, Cc, groups, Kp, ops.ga_dtype(dt))
    xx = x.clone().requires_grad_(True)
    xc = xx.transpose(1, 2) / Hh                       # (B, C, HW) like the reference's NCHW / H
    gm = torch.bmm(xc, xc.transpose(1, 2)) / HW
    iu = torch.triu_indices(Cc, Cc)
    v = F.normalize(gm[:, iu[0], iu[1]])
    got = vec.float().cpu().reshape(Bn, groups, Kp)[:, :, :Kg].reshape(Bn, ntri)
    assert_close(got, v, tol(dt, 2), 'gram vec')
    dv = torch.randn(Bn, ntri, generator=g).to(dt).float()
    dvp = torch.zeros(Bn, groups, Kp)
    dvp[:, :, :Kg] = dv.reshape(Bn, groups, Kg)
    v.backward(dv)
    S = torch.empty(Bn, Cc, Cc, dtype=dt, device='cuda')
    P.gram_pack_bwd(dvp.to(dt).cuda().reshape(Bn, groups * Kp), vec, inv, S, Bn, Cc, groups, Kp, ops.ga_dtype(dt))
    DX = torch.empty_like(X)
    P.gemm(X, S, DX, HW, Cc, Cc, ops.ga_dtype(dt), batch=Bn, strideA=HW * Cc, strideB=Cc * Cc, strideC=HW * Cc, alpha=alpha)
    assert_close(DX, xx.grad, tol(dt, 4), 'gram bwd')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('hd', [24, 8, 48, 12])   # 12: per-head fallback form (hd % 8 != 0)
def test_class_attention(dt, hd):
    ops = _imp()
    g = gen(17)
    Bn, N, heads = 3, 197, 8
    E = heads * hd
    q, Q = rnd((Bn, E), dt, g)
    kv, KV = rnd((Bn, N, 2 * E), dt, g)
    scale = hd ** -0.5
    P = ops.Plan(eager=True)
    out = torch.empty(Bn, E, dtype=dt, device='cuda')
    Pm = torch.empty(Bn, heads, N, device='cuda')
    P.class_attn_fwd(Q, KV, out, Pm, Bn, N, heads, hd, scale, ops.ga_dtype(dt))
    lq = q.clone().requires_grad_(True); lkv = kv.clone().requires_grad_(True)
    k = lkv[:, :, :E].reshape(Bn, N, heads, hd).permute(0, 2, 1, 3)
    v = lkv[:, :, E:].reshape(Bn, N, heads, hd).permute(0, 2, 1, 3)
    qq = lq.reshape(Bn, 1, heads, hd).permute(0, 2, 1, 3) * scale
    attn = (qq @ k.transpose(-2, -1)).softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(Bn, E)
    assert_close(out, o, tol(dt, 2), 'class attn fwd')
    assert_close(Pm, attn.squeeze(2), tol(dt, 2), 'class attn P')
    do, DO = rnd((Bn, E), dt, g)
    o.backward(do)
    dq = torch.empty_like(Q); dkv = torch.empty_like(KV)
    P.class_attn_bwd(DO, Q, KV, Pm, dq, dkv, Bn, N, heads, hd, scale, ops.ga_dtype(dt))
    assert_close(dq, lq.grad, tol(dt, 3), 'class attn dq')
    assert_close(dkv, lkv.grad, tol(dt, 3), 'class attn dkv')
    # split form: class-token row and image-token rows in separate arrays (hd % 8 == 0 only)
    if hd % 8 == 0:
        KVc = KV[:, 0].contiguous(); KVt = KV[:, 1:].contiguous()
        out2 = torch.empty_like(out); Pm2 = torch.empty_like(Pm)
        P.class_attn_fwd2(Q, KVc, KVt, out2, Pm2, Bn, N, heads, hd, scale, ops.ga_dtype(dt))
        assert_close(out2, o, tol(dt, 2), 'class attn fwd (split)')
        assert_close(Pm2, attn.squeeze(2), tol(dt, 2), 'class attn P (split)')
        dq2 = torch.empty_like(Q); dkc = torch.empty_like(KVc); dkt = torch.empty_like(KVt)
        P.class_attn_bwd2(DO, Q, KVc, KVt, Pm2, dq2, dkc, dkt, Bn, N, heads, hd, scale, ops.ga_dtype(dt))
        assert_close(dq2, lq.grad, tol(dt, 3), 'class attn dq (split)')
        assert_close(torch.cat((dkc[:, None], dkt), 1), lkv.grad, tol(dt, 3), 'class attn dkv (split)')
        # token rows as a column slice of a wider matrix (row stride tok_ld)
        ld = 2 * E + 16
        wide = torch.zeros(Bn, N - 1, ld, dtype=dt, device='cuda'); wide[:, :, 8:8 + 2 * E] = KVt
        dwide = torch.zeros_like(wide)
        P.class_attn_fwd2(Q, KVc, wide[:, :, 8:], out2, Pm2, Bn, N, heads, hd, scale, ops.ga_dtype(dt), tok_ld=ld)
        assert_close(out2, o, tol(dt, 2), 'class attn fwd (strided tokens)')
        P.class_attn_bwd2(DO, Q, KVc, wide[:, :, 8:], Pm2, dq2, dkc, dwide[:, :, 8:], Bn, N, heads, hd, scale, ops.ga_dtype(dt), tok_ld=ld)
        assert_close(dwide[:, :, 8:8 + 2 * E], lkv.grad[:, 1:], tol(dt, 3), 'class attn dkv (strided tokens)')
        assert float(dwide[:, :, :8].abs().max()) == 0.0 and float(dwide[:, :, 8 + 2 * E:].abs().max()) == 0.0
    # token cat / split
    Cc = 64
    c, Cl = rnd((Bn, Cc), dt, g); t, Tk = rnd((Bn, 196, Cc), dt, g)
    U = torch.empty(Bn, 197, Cc, dtype=dt, device='cuda')
    P.token_cat(Cl, Tk, U, Bn, 196, Cc, ops.ga_dtype(dt))
    assert_close(U, torch.cat((c[:, None], t), 1), 1e-7, 'token cat')
    dc = torch.ones(Bn, Cc, dtype=dt, device='cuda'); dtk = torch.ones(Bn, 196, Cc, dtype=dt, device='cuda')
    P.token_split(U, dc, dtk, Bn, 196, Cc, False, True, ops.ga_dtype(dt))
    assert_close(dc, c, 1e-7, 'token split cls')
    assert_close(dtk, t + 1, tol(dt), 'token split tok acc')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('kind,smooth', [(0, 0.0), (0, 0.1), (1, 0.1)])
def test_ga_loss(dt, kind, smooth):
    ops = _imp()
    from oracle import ga_convnext_oracle as O
    g = gen(18)
    K, Bn, NC = 5, 6, 1000
    lg = torch.randn(K, Bn, NC, generator=g) * 2
    tgt = torch.randint(0, NC, (Bn,), generator=g)
    loss = torch.zeros(1, device='cuda')
    dl = torch.empty(K, Bn, NC, dtype=dt, device='cuda')
    ops.Plan(eager=True).loss_fwd_bwd(lg.cuda(), tgt.cuda(), loss, dl, K, Bn, NC, -0.8, kind, smooth, 1.0, ops.ga_dtype(dt))
    leaves = [lg[k].clone().requires_grad_(True) for k in range(K)]
    ref = O.ga_loss(leaves, tgt, -0.8, 'ce' if kind == 0 else 'bce', smooth)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref))
    assert_close(dl, torch.stack([l.grad for l in leaves]), 1e-4 if dt == torch.float32 else 1e-2, 'dlogits')


def test_heads_topk_bit_exact():
    ops = _imp()
    g = gen(19)
    K, Bn, NC = 5, 33, 1000
    lg = torch.randn(K, Bn, NC, generator=g)
    lg[:, 3, 17] = lg[:, 3, 400] = 50.0          # an exact tie: lowest index first
    s = torch.empty(Bn, NC, device='cuda'); idx = torch.empty(Bn, 5, dtype=torch.int64, device='cuda')
    ops.Plan(eager=True).heads_topk(lg.cuda(), K, Bn, NC, 5, s, idx)
    summed = s.cpu()                             # the kernel's own fp32 sum (k-ordered)
    ref = summed.topk(5, 1, True, True)[1]
    rows = [b for b in range(Bn) if b != 3]
    assert torch.equal(idx.cpu()[rows], ref[rows])
    assert idx.cpu()[3, 0].item() == 17 and idx.cpu()[3, 1].item() == 400
    want = lg[0]
    for k in range(1, K):
        want = want + lg[k]
    assert torch.equal(summed, want)             # same k-ordered fp32 additions as the reference loop


def test_optimizers_match_torch():
    ops = _imp()
    g = gen(20)
    n = 10007
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    for wd_mult in (1.0, 0.0):
        pt = p0.clone().requires_grad_(True)
        opt = torch.optim.SGD([pt], lr=0.1, momentum=0.9, nesterov=True, weight_decay=0.05 * wd_mult)
        p = p0.clone().cuda(); buf = torch.zeros(n, device='cuda')
        for i, gr in enumerate(grads):
            pt.grad = gr.clone(); opt.step()
            hp = torch.tensor([0.1, 0.05, 0.9, 0, 0, 0, 0, 1.0 if i == 0 else 0.0]).cuda()
            ops.Plan(eager=True).sgd_step(p, gr.cuda(), buf, hp, n, True, wd_mult)
            assert_close(p, pt.detach(), 1e-6, 'sgd')
        pt = p0.clone().requires_grad_(True)
        opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05 * wd_mult)
        p = p0.clone().cuda(); m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
        for i, gr in enumerate(grads):
            pt.grad = gr.clone(); opt.step()
            t = i + 1
            hp = torch.tensor([1e-2, 0.05, 0.9, 0.999, 1e-8, 1 - 0.9 ** t, 1 - 0.999 ** t, 0.0]).cuda()
            ops.Plan(eager=True).adamw_step(p, gr.cuda(), m, v, hp, n, wd_mult)
            assert_close(p, pt.detach(), 2e-6, 'adamw')


def test_error_path_is_loud():
    ops = _imp()
    A = torch.zeros(8, 12, device='cuda'); B = torch.zeros(8, 12, device='cuda'); Cc = torch.zeros(8, 8, device='cuda')
    with pytest.raises(RuntimeError, match='K=10'):
        ops.Plan(eager=True).gemm(A, B, Cc, 8, 8, 10, ops.GA_F32, lda=12, ldb=12)
    with pytest.raises(AssertionError):
        ops.Plan(eager=True).gemm(A.cpu(), B, Cc, 8, 8, 12, ops.GA_F32)


# ----------------------------------------------------------------------------------------------------------
# gradient clipping on the flat gradient buffer (timm dispatch_clip_grad: 'norm', 'value')
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1000003, 4096, 7])
def test_clip_grad_flat(n):
    ops = _imp()
    g = gen(23)
    base = torch.randn(n + 4, generator=g)[:n].contiguous()
    for limit in (0.5 * float(base.norm()), 10.0 * float(base.norm())):       # clips / leaves untouched
        G = base.clone().cuda()
        ss = torch.zeros(1, device='cuda')
        P = ops.Plan(eager=True)
        P.sumsq_f32(G, n, ss)
        assert abs(float(ss) - float((base.double() ** 2).sum())) <= 1e-5 * float((base.double() ** 2).sum())
        P.clip_grad_f32(G, n, ss, limit, 0)
        ref = base.clone().requires_grad_(True)
        ref.grad = base.clone()
        torch.nn.utils.clip_grad_norm_([ref], limit)
        assert_close(G, ref.grad, 1e-4, 'clip norm')   # fp32 sum of 1e6 squares: ~1e-5 relative
    G = base.clone().cuda()
    ops.Plan(eager=True).clip_grad_f32(G, n, G, 0.3, 1)
    assert torch.equal(G.cpu(), base.clamp(-0.3, 0.3))


@pytest.mark.parametrize('dt', DT)
def test_weight_prep_stacked_transposed_operand(dt):
    """several jobs fill column ranges of ONE transposed operand (t_cols): the stacked k|v weights of the five heads"""
    ops = _imp()
    g = gen(31)
    heads, Co, Ci = 3, 40, 72                      # 1x1 fast path (Ci % 4 == 0) ...
    for Ci in (72, 70):                            # ... and the generic path
        ws = [torch.randn(Co, Ci, generator=g) for _ in range(heads)]
        css = [torch.rand(Ci, generator=g) + 0.5 for _ in range(heads)]
        ldt = heads * Co
        outT = torch.full((Ci, ldt), 7.0, dtype=dt, device='cuda')
        out = torch.empty(heads * Co, Ci + (-Ci) % 8, dtype=dt, device='cuda')
        P = ops.Plan(defer_small=True)
        for k in range(heads):
            P.weight_prep(ws[k].cuda(), 1, Co, Ci, 1, 1, ops.ga_dtype(dt), out=out[k * Co:], ldo=out.shape[1],
                          outT=outT[:, k * Co:], ldt=ldt, cs=css[k].cuda(), t_cols=Co)
        P.flush()
        P.run()
        ref = torch.cat([w * c[None, :] for w, c in zip(ws, css)], 0)      # [heads*Co, Ci]
        assert_close(outT, ref.t(), tol(dt, 0.5), 'stacked outT')
        assert_close(out[:, :Ci], ref, tol(dt, 0.5), 'stacked out')


def test_drop_path_sampler_statistics():
    """ga_drop_path_sample: values in {0, 1/keep}, keep-rate within 4 sigma, a new mask every call (device-side counter)"""
    ops = _imp()
    sites, B = 7, 4096
    keep = torch.linspace(0.5, 1.0, sites)
    out = torch.zeros(sites, B, device='cuda')
    ctr = torch.zeros(1, dtype=torch.int64, device='cuda')
    p = ops.Plan(name='dp')
    p.drop_path_sample(out, keep.cuda(), sites, B, 1234, ctr)
    p.run()
    a = out.cpu().clone()
    p.run()
    b = out.cpu()
    assert int(ctr) == 2
    for s in range(sites):
        k = float(keep[s])
        assert all(v == 0.0 or abs(v - 1 / k) < 1e-6 for v in a[s].unique().tolist())
        rate = float((a[s] > 0).float().mean())
        assert abs(rate - k) <= 4 * math.sqrt(k * (1 - k) / B) + 1e-9, (s, rate, k)
    assert not torch.equal(a[0], b[0])          # keep 0.5: two calls give different masks
    assert torch.equal(a[-1], torch.ones(B))    # keep 1.0: never dropped
