"""GPU: the hand-maintained correspondence between `FlatModel.grad_groups()` and the marks of every engine's backward plan.

TrainStep (trainer.py) all-reduces the flat-gradient slices a group claims as soon as the backward plan has been issued up to
the group's mark (the DDP reducer of /root/reference/GA/train.py:514, overlapped with backward).  That is only correct if every
gradient of the group is FINAL at its mark: no later launch (a weight-gradient GEMM still on the asynchronous lane, a deferred
weight-unfold batch, a pooling-conv weight gradient recorded after the mark) may write it.  For every registered family this
test runs the backward plan in the same segments, snapshots the slices each mark owns when the mark is reached, finishes the
backward pass and requires the snapshots to be bit-identical to the final gradients.  One GPU, no process group."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FAMILIES = [
    ('ga_convnext_tiny_768', {}, 2),
    ('ga_CSWin_64_12211_tiny_224', {}, 2),
    ('map_convnext_tiny', {}, 2),
    ('map_vit_small_patch16_224', {}, 2),
    ('map_pit_s', {}, 2),
    ('convnext_tiny', {}, 2),
    ('ga_convnext_tiny_688', {}, 2),      # padded parameter gradients of the odd-width heads: copied back before the 'heads' mark
]


@pytest.mark.parametrize('name,kw,B', FAMILIES, ids=[f[0] for f in FAMILIES])
def test_group_slices_are_final_at_their_mark(name, kw, B):
    import imagenet_models_amd as A
    from imagenet_models_amd.trainer import make_buckets
    torch.manual_seed(0)
    m = A.create_model(name, **kw).cuda().train()
    # the reference's init leaves layer-scale at 1e-6: gradients would be denormal-small in places; any non-trivial values do
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() <= 1 and float(p.abs().max()) < 1e-3:
                p.fill_(0.1)
    eng = m.engine(B, True)
    st = m.flat_state()
    g = st['grads']
    groups = m.grad_groups()
    assert groups, f'{name}: no gradient groups declared'
    buckets = make_buckets(st, groups, 1 << 40)
    img = getattr(eng, 'img', 224) or 224
    x = torch.randn(B, 3, img, img, device='cuda')
    y = torch.randint(0, m.num_classes, (B,), device='cuda')
    m.zero_grad()
    eng.forward_loss(x, y, -0.8, 0, 0.0, 1.0)
    bwd = eng.bwd
    for mark, _ in groups:
        assert mark in bwd.marks, f'{name}: group mark {mark!r} is not recorded by the backward plan ({list(bwd.marks)})'
    snaps, pos = [], 0
    for mark, a, b in buckets:
        stop = len(bwd.calls) if mark == 'end' else bwd.marks[mark]
        assert stop >= pos, f'{name}: marks out of order at {mark!r}'
        if stop > pos:
            bwd.run_range(pos, stop)
            pos = stop
        torch.cuda.synchronize()
        snaps.append((mark, a, b, g[a:b].clone()))
    if pos < len(bwd.calls):
        bwd.run_range(pos, len(bwd.calls))
    torch.cuda.synchronize()
    owner = {}
    for n, (off, k) in st['slices'].items():
        for mark, a, b, _ in snaps:
            if a <= off < b:
                owner[n] = mark
    bad = []
    for mark, a, b, snap in snaps:
        if not torch.equal(snap, g[a:b]):
            # name the parameters that moved
            for n, (off, k) in st['slices'].items():
                if a <= off < b and not torch.equal(snap[off - a:off - a + k], g[off:off + k]):
                    bad.append((mark, n))
    assert not bad, f'{name}: gradients written AFTER the mark that declares them final: {bad[:12]} ({len(bad)} tensors)'
    # the test is vacuous if the claimed slices never received a gradient
    for mark, _ in groups:
        tot = sum(float(s.abs().sum()) for mk, a, b, s in snaps if mk == mark)
        assert tot > 0.0, f'{name}: group {mark!r} holds no gradient at all'
