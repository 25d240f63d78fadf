"""Training-path pieces the reference takes from timm (GA/train.py:544-557,616-621,727-728,752-756): mixup / cutmix on the
device, the dense-target losses (SoftTargetCrossEntropy / BinaryCrossEntropy), adaptive gradient clipping.  Checked against
oracle/mixup_oracle.py (a restatement of timm's published algorithm -- timm is not vendored in the reference and not
installed: parity UNPINNED there, see that file) and against torch formulas for the losses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracles():
    from oracle import ga_convnext_oracle as O, mixup_oracle as MO, map_oracle as MP
    return O, MO, MP


@pytest.mark.parametrize('kw', [dict(mixup_alpha=0.8, cutmix_alpha=0.0), dict(mixup_alpha=0.0, cutmix_alpha=1.0),
                                dict(mixup_alpha=0.2, cutmix_alpha=1.0), dict(mixup_alpha=0.0, cutmix_alpha=0.0, cutmix_minmax=(0.2, 0.6)),
                                dict(mixup_alpha=0.8, cutmix_alpha=1.0, prob=0.5)])
def test_mixup_matches_timm_restatement(kw):
    import imagenet_models_amd as A
    _, MO, _ = _oracles()
    NC = 37
    for seed in range(4):
        x = torch.randn(8, 3, 32, 48, generator=torch.Generator().manual_seed(seed))
        t = torch.randint(0, NC, (8,), generator=torch.Generator().manual_seed(100 + seed))
        ref = MO.Mixup(num_classes=NC, label_smoothing=0.1, rng=np.random.RandomState(seed), **kw)
        mine = A.Mixup(num_classes=NC, label_smoothing=0.1, rng=np.random.RandomState(seed), **kw)
        rx, rt = ref(x, t)
        gx, gt = mine(x.cuda(), t.cuda())
        assert torch.equal(gx.cpu(), rx), (kw, seed, mine.last)          # bit-exact: same rounding order as mul_ / add_
        assert torch.equal(gt.cpu(), rt), (kw, seed, mine.last, (gt.cpu() - rt).abs().max(), int((gt.cpu() != rt).sum()))
        assert torch.allclose(gt.sum(1).cpu(), torch.ones(8), atol=1e-6)
        lam, cut, box = mine.last
        if cut and lam != 1.0:
            assert abs(lam - (1 - (box[1] - box[0]) * (box[3] - box[2]) / (32 * 48))) < 1e-12
    off = A.Mixup(num_classes=NC, label_smoothing=0.0, rng=np.random.RandomState(0), **kw)
    off.mixup_enabled = False
    gx, gt = off(x.cuda(), t.cuda())
    assert torch.equal(gx.cpu(), x) and torch.equal(gt.cpu(), torch.nn.functional.one_hot(t, NC).float())
    with pytest.raises(NotImplementedError):
        A.Mixup(mode='elem')


@pytest.mark.parametrize('kind,thr', [('ce', None), ('bce', None), ('bce', 0.2)])
def test_ga_loss_on_dense_targets(kind, thr):
    """SoftTargetCrossEntropy / BinaryCrossEntropy (+ target threshold) summed over the heads + the GA decorrelation term"""
    import imagenet_models_amd as A
    O, MO, _ = _oracles()
    K, B, NC = 5, 6, 41
    g = torch.Generator().manual_seed(3)
    outs = [torch.randn(B, NC, generator=g) * 2 for _ in range(K)]
    dense = MO.mixup_target(torch.randint(0, NC, (B,), generator=g), NC, lam=0.37, smoothing=0.1)
    ref_in = [o.clone().requires_grad_(True) for o in outs]
    ref = O.ga_loss(ref_in, dense, -0.8, kind, 0.0, thr)
    ref.backward()
    mine_in = [o.clone().cuda().requires_grad_(True) for o in outs]
    loss = A.ga_loss(mine_in, dense.cuda(), -0.8, kind, bce_target_thresh=thr)
    loss.backward()
    assert abs(float(loss) - float(ref)) / abs(float(ref)) < 1e-5
    for a, b in zip(mine_in, ref_in):
        assert float((a.grad.cpu() - b.grad).abs().max()) < 1e-5 * float(b.grad.abs().max()) + 1e-9
    with pytest.raises(ValueError):
        A.ga_loss(mine_in, dense[:, :-1].cuda(), -0.8, kind)
    # class indices with a BCE threshold (the smoothed one-hot target binarised)
    if thr is not None:
        t = torch.randint(0, NC, (B,), generator=g)
        r2 = O.ga_loss([o.clone() for o in outs], t, -0.8, kind, 0.1, thr)
        l2 = A.ga_loss([o.cuda() for o in outs], t.cuda(), -0.8, kind, smoothing=0.1, bce_target_thresh=thr)
        assert abs(float(l2) - float(r2)) / abs(float(r2)) < 1e-5


def test_map_loss_on_dense_targets():
    import imagenet_models_amd as A
    _, MO, MP = _oracles()
    G, B, NC = 3, 4, 29
    g = torch.Generator().manual_seed(9)
    pairs = [[torch.randn(B, NC, generator=g), torch.randn(B, NC, generator=g)] for _ in range(G)]
    dense = MO.mixup_target(torch.randint(0, NC, (B,), generator=g), NC, lam=0.6, smoothing=0.1)
    ref_in = [[a.clone().requires_grad_(True), b.clone().requires_grad_(True)] for a, b in pairs]
    ref = MP.multi_group_loss(ref_in, dense, -0.8)
    ref.backward()
    mine_in = [[a.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)] for a, b in pairs]
    loss = A.map_loss(mine_in, dense.cuda(), -0.8)
    loss.backward()
    assert abs(float(loss) - float(ref)) / abs(float(ref)) < 1e-5
    for (a, b), (ra, rb) in zip(mine_in, ref_in):
        assert float((a.grad.cpu() - ra.grad).abs().max()) < 1e-5 * float(ra.grad.abs().max()) + 1e-9
        assert float((b.grad.cpu() - rb.grad).abs().max()) < 1e-5 * float(rb.grad.abs().max()) + 1e-9


def test_agc_clip_matches_timm_restatement():
    from imagenet_models_amd import ops
    _, MO, _ = _oracles()
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 3, 3, 3), (16, 40), (16,), (5, 1, 7, 7), (1,), (33, 129)]
    params = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    grads = [torch.randn(s, generator=g) * (3.0 if i % 2 else 0.001) for i, s in enumerate(shapes)]
    params[2].zero_()                                  # |p| below eps: the eps floor decides
    want = MO.adaptive_clip_grad(params, grads, clip_factor=0.02)
    flat_p = torch.cat([p.reshape(-1) for p in params]).cuda()
    flat_g = torch.cat([q.reshape(-1) for q in grads]).cuda()
    units, off = [], 0
    for p in params:
        if p.dim() > 1:
            row = p[0].numel()
            units.extend((off + r * row, row) for r in range(p.shape[0]))
        else:
            units.append((off, p.numel()))
        off += p.numel()
    u = torch.tensor(units, dtype=torch.int64).cuda()
    ops.Plan(eager=True).agc_clip(flat_p, flat_g, u, len(units), 0.02)
    got = flat_g.cpu()
    ref = torch.cat([w.reshape(-1) for w in want])
    assert float((got - ref).abs().max()) < 1e-6 * float(ref.abs().max())
    assert not torch.equal(ref, torch.cat([q.reshape(-1) for q in grads]))       # something was clipped


def test_train_step_with_mixup_and_agc():
    """TrainStep(mixup_fn=..., clip_mode='agc') on a small GA-ConvNeXt: the dense-target loss equals the oracle's on the same
    mixed batch, a step runs, the loss stays finite"""
    import imagenet_models_amd as A
    O, MO, _ = _oracles()
    cfg = O.make_cfg(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)
    m = A.GA_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], gram_dim=32, dim_embed=64, stage3_naggre=2,
                      drop_path_rate=0.0, math_mode='fp32')
    sd = O.fill_state(cfg)
    m.load_state_dict(sd)
    m = m.cuda().train()
    opt = A.create_optimizer_v2(m, opt='sgd', lr=0.01, momentum=0.9, weight_decay=1e-4)
    mix = A.Mixup(mixup_alpha=0.2, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=40, rng=np.random.RandomState(7))
    step = A.TrainStep(m, opt, 8, lam=-0.8, clip_grad=0.02, clip_mode='agc', mixup_fn=mix)
    x = O.gen_input(8, seed=2)
    t = torch.randint(0, 40, (8,), generator=torch.Generator().manual_seed(2))
    loss = float(step(x.cuda(), t.cuda()))
    ref_mix = MO.Mixup(mixup_alpha=0.2, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=40, rng=np.random.RandomState(7))
    rx, rt = ref_mix(x, t)
    new_stats = {}
    outs = O.forward(sd, rx, cfg, training=True, new_stats=new_stats)
    ref = float(O.ga_loss(outs, rt, -0.8))
    assert abs(loss - ref) / abs(ref) < 1e-3, (loss, ref)
    l2 = float(step(x.cuda(), t.cuda()))
    assert np.isfinite(l2)


def test_uint8_batches_are_normalised_on_the_device():
    """timm fast_collate hands uint8 NCHW batches to the device, PrefetchLoader does .float().sub_(mean).div_(std) there
    (GA/train.py:567-595): the model takes the uint8 tensor and produces the logits of the normalised float batch"""
    import imagenet_models_amd as A
    from imagenet_models_amd import ops
    O, _, _ = _oracles()
    g = torch.Generator().manual_seed(11)
    x8 = torch.randint(0, 256, (2, 3, 224, 224), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.485 * 255, 0.456 * 255, 0.406 * 255]).view(1, 3, 1, 1)
    std = torch.tensor([0.229 * 255, 0.224 * 255, 0.225 * 255]).view(1, 3, 1, 1)
    ref = x8.float().sub_(mean).div_(std)
    out = torch.empty(2, 3, 224, 224, device='cuda')
    ops.Plan(eager=True).u8_normalize(x8.cuda(), out, mean.flatten().tolist(), std.flatten().tolist())
    assert torch.equal(out.cpu(), ref)
    cfg = O.make_cfg(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)
    m = A.GA_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], gram_dim=32, dim_embed=64, stage3_naggre=2,
                      drop_path_rate=0.0, math_mode='fp32')
    m.load_state_dict(O.fill_state(cfg))
    m = m.cuda().eval()
    with torch.no_grad():
        a = torch.stack(m(x8.cuda())).cpu()
        b = torch.stack(m(ref.cuda())).cpu()
    assert torch.equal(a, b)
