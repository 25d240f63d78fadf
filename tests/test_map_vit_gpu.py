"""GPU parity of MAP-ViT (timm VisionTransformer trunk + the reference's MAPHead; the composition is builder-defined, see
oracle/map_vit_oracle.py) through the C ABI against the oracle: patch embedding, class token / position embedding, global
attention blocks (flash-style MFMA in bf16, plain fp32 form in the parity mode), feature taps, MultiScale, MAP head, fused
multi_group_loss, every gradient.  fp32 mode: logits / loss 1e-3, gradients 2e-2; bf16 mode reported (6e-2 / 2e-2)."""
import pytest
import torch

from _gradcheck import assert_grads_close, BF16_REL, BF16_COS

pytestmark = pytest.mark.gpu

V7 = dict(img_size=64, patch_size=16, embed_dim=64, depth=3, vit_heads=1, num_classes=40, last_dim=64, n_groups=2, n_tokens=2,
          gram_group=8, bp_dim=64, ca_dim=64, num_heads=4)
V7B = dict(img_size=96, patch_size=16, embed_dim=128, depth=6, vit_heads=2, num_classes=40, last_dim=64, n_groups=3, n_tokens=3,
           gram_group=4, bp_dim=48, ca_dim=64, num_heads=4)


def _oracle():
    from oracle import map_vit_oracle as O
    return O


def build(cfg, mode, drop_path=0.0):
    import imagenet_models_amd as A
    O = _oracle()
    m = A.MAP_ViT(img_size=cfg['img_size'], patch_size=cfg['patch_size'], embed_dim=cfg['embed_dim'], depth=cfg['depth'],
                  num_heads=cfg['vit_heads'], num_classes=cfg['num_classes'], drop_path_rate=drop_path, last_dim=cfg['last_dim'],
                  n_groups=cfg['n_groups'], n_tokens=cfg['n_tokens'], gram_group=cfg['gram_group'], bp_dim=cfg['bp_dim'],
                  ca_dim=cfg['ca_dim'], ca_heads=cfg['num_heads'], head_drop=0.0, head_attn_drop=0.0, math_mode=mode)
    sd = O.fill_state(cfg)
    m.load_state_dict(sd)
    return m.cuda(), sd


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize('over', [V7, V7B])
@pytest.mark.parametrize('mode,tol', [('fp32', 1e-3), ('bf16', 6e-2)])
def test_eval_logits(over, mode, tol):
    import imagenet_models_amd as A
    O = _oracle()
    cfg = O.make_cfg(**over)
    m, sd = build(cfg, mode)
    m.eval()
    x = O.gen_input(2, seed=0, size=cfg['img_size'])
    with torch.no_grad():
        outs = m(x.cuda())
        ref = O.forward(sd, x, cfg, training=False)
    err = max(rel(a, b) for a, b in zip(outs, ref))
    print(f'[map_vit {over["embed_dim"]} {mode}] eval logits rel err {err:.3e}')
    assert len(outs) == cfg['n_groups'] and err < tol
    if mode == 'fp32':
        _, idx = A.heads_mean_topk(outs, 5)
        want = (sum(ref) / len(ref)).topk(5, 1, True, True)[1]
        assert torch.equal(idx.cpu(), want)


@pytest.mark.parametrize('over', [V7, V7B])
@pytest.mark.parametrize('mode,tols,dp', [('fp32', (1e-3, 1e-3, 2e-2), 0.0), ('fp32', (1e-3, 1e-3, 2e-2), 0.3), ('bf16', (6e-2, 2e-2, 1.0), 0.0)])
def test_train_step(over, mode, tols, dp):
    import imagenet_models_amd as A
    O = _oracle()
    cfg = O.make_cfg(**over)
    cfg['drop_path_rate'] = dp
    B = 4
    m, sd = build(cfg, mode, dp)
    m.train()
    x = O.gen_input(B, seed=1, size=cfg['img_size'])
    target = torch.randint(0, cfg['num_classes'], (B,), generator=torch.Generator().manual_seed(5))
    masks = None
    if dp > 0:
        eng = m.engine(B, True)
        g = torch.Generator().manual_seed(5)
        masks = {}
        for site in eng.dp_scale:
            keep = 1 - eng.dp_rates[site]
            masks[site] = (torch.rand(B, generator=g) < keep).float() / keep
        eng.set_drop_path_masks(masks)
        eng.fixed_masks = True
    m.zero_grad()
    outs = m(x.cuda())
    loss = A.map_loss(outs, target.cuda(), -0.8)
    loss.backward()
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, dec_lam=-0.8, dp_masks=masks)
    e_out = max(max(rel(a, b) for a, b in zip(o, oo)) for o, oo in zip(outs, oouts))
    e_loss = abs(float(loss.detach()) - float(oloss)) / abs(float(oloss))
    from oracle import ga_convnext_oracle as GO
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    errs = GO.grad_errors(grads, ograds)
    if mode == 'bf16':
        gmax = max(float(g_.abs().max()) for g_ in ograds.values())
        errs = {n: e for n, e in errs.items() if float(ograds[n].abs().max()) >= 1e-4 * gmax}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    new_sd = m.state_dict()
    e_bn = max(rel(new_sd[n], ostats[n].float()) for n in ostats if not n.endswith('num_batches_tracked'))
    print(f'[map_vit {over["embed_dim"]} {mode} dp={dp}] logits {e_out:.2e} loss {e_loss:.2e} bn {e_bn:.2e} worst grads {worst}')
    assert e_out < tols[0] and e_loss < tols[1]
    if mode == 'bf16':      # whole-tensor gates (tests/_gradcheck.py): norm-relative error and direction of every gradient
        assert_grads_close(grads, ograds, BF16_REL, BF16_COS, 'bf16 train step')
    else:
        assert worst[0][1] < tols[2], worst
    assert e_bn < max(tols[0], 2e-3)


def test_registry_and_param_layout():
    import imagenet_models_amd as A
    O = _oracle()
    for name in ('map_vit_base_patch16_384', 'map_vit_small_patch16_224'):
        m = A.create_model(name)
        shapes = O.state_shapes(O.make_cfg(name))
        sd = m.state_dict()
        assert list(sd.keys()) == list(shapes.keys()) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert abs(sum(p.numel() for p in A.create_model('map_vit_base_patch16_384').parameters()) / 1e6 - 106.7) < 0.1
