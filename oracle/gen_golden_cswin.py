"""Generate tests/golden/cswin_*.npz from the REAL reference classes of /root/reference/GA/ga_cswin.py
(build container only; never runs on the GPU box).

Run:  python oracle/gen_golden_cswin.py

Imports the reference against oracle/timm_stub (timm is not installed; einops is), loads the deterministic
name-hashed weights of oracle.ga_cswin_oracle.fill_state into GA_CSWinTransformer / CSWinBlock / LePEAttention,
runs them, and stores outputs only (inputs come from closed-form generators).  While generating it checks the
restatement against the reference and prints the deviations.

The reference registers no GA-CSWin factory: the full-size "tiny" configuration is the survey's
candidate ("config unpinned"); what these fixtures pin is the ARITHMETIC of the classes.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'timm_stub'))
sys.path.insert(0, '/root/reference/GA')
sys.path.insert(0, os.path.dirname(HERE))

import ga_cswin as ref  # noqa: E402  (the reference)
from oracle import ga_cswin_oracle as O  # noqa: E402
from oracle.gen_golden import grad_stats, ref_loss, rel  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

# narrow whole-model config: head dims 8 / 8 / 8 / 16 (trunk), 16 (stage5), 32 (gram layer: 8 groups x 24 channels must stay
# 8-aligned for the grouped contraction GEMM), 8 (class attention)
V6 = dict(embed_dim=16, depth=(1, 1, 6, 1), split_size=(1, 2, 7, 7, 7), num_heads=(2, 4, 8, 16, 16),
          dims=(16, 32, 64, 256), naggre=2, gram_dim=192, num_classes=40, stage5_mlp_groups=4)
# same with the Bottleneck stage5 (ga_cswin.py:540-542)
V6B = dict(V6, stage5='bottleneck')


def build_ref(cfg):
    assert cfg['gram_heads'] == 6 and cfg['ga_heads'] == 8 and cfg['ga_expansion'] == 4 and cfg['gram_groups'] == 8, \
        'hard-coded in the reference (ga_cswin.py:560,568,585,283)'
    m = ref.GA_CSWinTransformer(img_size=cfg['img_size'], num_classes=cfg['num_classes'], embed_dim=cfg['embed_dim'],
                                depth=list(cfg['depth']), split_size=list(cfg['split_size']),
                                num_heads=list(cfg['num_heads']), mlp_ratio=cfg['mlp_ratio'], qkv_bias=cfg['qkv_bias'],
                                drop_path_rate=0.0, dims=list(cfg['dims']), stage3_naggre=cfg['naggre'],
                                ga_mlp_groups=cfg['ga_mlp_groups'], ga_layer_mlp_groups=cfg['ga_layer_mlp_groups'],
                                branches=cfg['branches'], gram_dim=cfg['gram_dim'], stage5=cfg['stage5'],
                                stage5_mlp_groups=cfg['stage5_mlp_groups'])
    sd = O.fill_state(cfg)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict key order differs from the reference:\n' + '\n'.join(
        f'{a} | {b}' for a, b in zip(ref_sd.keys(), sd.keys()) if a != b)
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (k, tuple(ref_sd[k].shape), tuple(sd[k].shape))
    m.load_state_dict(sd)
    return m, sd


def do_eval(tag, cfg, batch, nlog):
    m, sd = build_ref(cfg)
    m.eval()
    x = O.gen_input(batch, seed=0)
    with torch.no_grad():
        outs = m(x)
        mine = O.forward(sd, x, cfg, training=False)
    err = max(rel(a, b) for a, b in zip(mine, outs))
    print(f'[{tag}] eval: oracle vs reference max rel err = {err:.3e}')
    assert err < 1e-4
    s = sum(o.float() for o in outs)
    np.savez_compressed(os.path.join(OUT, f'{tag}_eval.npz'), cfg=json.dumps(cfg), batch=batch,
                        param_count=sum(p.numel() for p in m.parameters()), n_state=len(sd),
                        logits=torch.stack(outs)[:, :, :nlog].numpy(), top5=s.topk(5, 1, True, True)[1].numpy())


def do_train(tag, cfg, batch, lam=-0.8):
    m, sd = build_ref(cfg)
    m.train()
    x = O.gen_input(batch, seed=1)
    tg = torch.Generator().manual_seed(99)
    target = torch.randint(0, cfg['num_classes'], (batch,), generator=tg)
    outs = m(x)
    loss = ref_loss(outs, target, lam)
    loss.backward()
    grads = {n: p.grad.detach() for n, p in m.named_parameters()}
    new_sd = m.state_dict()
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, lam=lam)
    e_out = max(rel(a, b.detach()) for a, b in zip(oouts, outs))
    e_loss = abs(float(oloss) - float(loss.detach())) / abs(float(loss.detach()))
    e_g = max(O.grad_errors(ograds, grads).values())
    e_bn = max(rel(ostats[n].float(), new_sd[n].float()) for n in ostats)
    print(f'[{tag}] train B={batch}: oracle vs reference rel err: logits {e_out:.2e} loss {e_loss:.2e} '
          f'grads {e_g:.2e} bn-stats {e_bn:.2e}')
    assert max(e_out, e_loss, e_bn) < 1e-4 and e_g < 1e-2
    names, norm, ssum, head = grad_stats(grads)
    bn_names = [n for n in new_sd if n.endswith('running_mean') or n.endswith('running_var')]
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b{batch}.npz'), cfg=json.dumps(cfg), batch=batch, lam=lam,
                        target=target.numpy(), loss=float(loss),
                        logits=torch.stack([o.detach() for o in outs])[:, :, :40].numpy(),
                        grad_names=np.array(names), grad_norm=norm, grad_sum=ssum, grad_head=head,
                        bn_names=np.array(bn_names),
                        bn_head=np.stack([new_sd[n].reshape(-1)[:8].numpy() for n in bn_names]))


def module_input(shape, seed):
    g = torch.Generator().manual_seed(4321 + seed)
    return torch.randn(*shape, generator=g)


def do_modules():
    """V1-style per-module vectors: LePEAttention (three stripe kinds) and CSWinBlock (plain / grouped MLP) fwd + grads"""
    out = {}
    # LePEAttention on (B, L, C) q/k/v, reso 14: idx 0 (14 x 7 stripes), idx 1 (7 x 14), idx -1 (whole map, reso 7)
    for name, reso, idx, split, dim, heads in (('lepe_v', 14, 0, 7, 32, 2), ('lepe_h', 14, 1, 7, 32, 2),
                                               ('lepe_full', 7, -1, 7, 32, 4), ('lepe_s1', 28, 0, 1, 16, 1),
                                               ('lepe_s2h', 28, 1, 2, 32, 2)):
        att = ref.LePEAttention(dim, resolution=reso, idx=idx, split_size=split, num_heads=heads, dim_out=dim)
        rs = np.random.RandomState(17 + len(name))
        w = torch.tensor(rs.standard_normal((dim, 1, 3, 3)) / 3.0, dtype=torch.float32)
        bb = torch.tensor(rs.uniform(-0.1, 0.1, (dim,)), dtype=torch.float32)
        att.get_v.weight.data.copy_(w)
        att.get_v.bias.data.copy_(bb)
        qkv = module_input((3, 2, reso * reso, dim), seed=len(name)).requires_grad_(True)
        y = att(qkv)
        gy = module_input(tuple(y.shape), seed=100 + len(name))
        y.backward(gy)
        # oracle check
        q2 = qkv.detach().clone().requires_grad_(True)
        w2, b2 = w.clone().requires_grad_(True), bb.clone().requires_grad_(True)
        y2 = O.lepe_attention(q2[0], q2[1], q2[2], w2, b2, reso, idx, split, heads)
        y2.backward(gy)
        e = max(rel(y2.detach(), y.detach()), rel(q2.grad, qkv.grad), rel(w2.grad, att.get_v.weight.grad),
                rel(b2.grad, att.get_v.bias.grad))
        print(f'[modules] {name}: oracle vs reference max rel err {e:.2e}')
        assert e < 1e-4
        out.update({f'{name}.cfg': np.array([reso, idx, split, dim, heads]), f'{name}.w': w.numpy(), f'{name}.b': bb.numpy(),
                    f'{name}.y': y.detach().numpy(), f'{name}.dqkv': qkv.grad.numpy(),
                    f'{name}.dw': att.get_v.weight.grad.numpy(), f'{name}.db': att.get_v.bias.grad.numpy()})
    # CSWinBlock: 2-branch with plain MLP; 2-branch with grouped MLP; last-stage single branch
    for name, dim, reso, heads, split, last, mg in (('blk2', 32, 14, 4, 7, False, 1), ('blk2g', 32, 14, 4, 7, False, 4),
                                                    ('blk1', 32, 7, 4, 7, True, 1)):
        blk = ref.CSWinBlock(dim=dim, reso=reso, num_heads=heads, split_size=split, qkv_bias=True, last_stage=last,
                             mlp_groups=mg)
        shapes = O.OrderedDict()
        O._cswin_block_shapes('', dim, O.branch_num(reso, split, last), True, mg, 4.0, shapes)
        assert list(blk.state_dict().keys()) == list(shapes.keys())
        sd = O.OrderedDict()
        for k, shp in shapes.items():
            rs = np.random.RandomState((O.zlib.crc32((name + k).encode())) & 0x7FFFFFFF)
            if len(shp) >= 2:
                v = rs.standard_normal(shp) / np.sqrt(np.prod(shp[1:]))
            elif k.endswith('weight'):
                v = rs.uniform(0.8, 1.2, shp)
            else:
                v = rs.uniform(-0.1, 0.1, shp)
            sd[k] = torch.tensor(v, dtype=torch.float32)
        blk.load_state_dict(sd)
        x = module_input((2, reso * reso, dim), seed=7 + len(name)).requires_grad_(True)
        y = blk(x)
        gy = module_input(tuple(y.shape), seed=200 + len(name))
        y.backward(gy)
        leaf = O.OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
        x2 = x.detach().clone().requires_grad_(True)
        y2 = O.cswin_block(leaf, '', x2, reso, split, heads, last, mg)
        y2.backward(gy)
        pg = dict(blk.named_parameters())
        e = max([rel(y2.detach(), y.detach()), rel(x2.grad, x.grad)] + [rel(leaf[k].grad, pg[k].grad) for k in leaf])
        print(f'[modules] {name}: oracle vs reference max rel err {e:.2e}')
        assert e < 1e-4
        out.update({f'{name}.cfg': np.array([dim, reso, heads, split, int(last), mg]), f'{name}.y': y.detach().numpy(),
                    f'{name}.dx': x.grad.numpy()})
        for k in pg:
            out[f'{name}.g.{k}'] = pg[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'cswin_modules.npz'), **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    do_modules()
    v6 = O.make_cfg(**V6)
    do_eval('cswin_v6', v6, 2, 40)
    do_train('cswin_v6', v6, 4)
    v6b = O.make_cfg(**V6B)
    do_eval('cswin_v6b', v6b, 2, 40)
    do_train('cswin_v6b', v6b, 4)
    tiny = O.make_cfg('ga_CSWin_64_12211_tiny_224')
    do_eval('cswin_tiny', tiny, 2, 16)
    do_train('cswin_tiny', tiny, 4)
    print('golden vectors written to', OUT)
