"""Generate tests/golden/pit_*.npz from the REAL /root/reference/MAP/models/map_pit.py (build container only; timm's Block comes
from oracle/timm_stub).  While generating, the oracle restatement (oracle/map_pit_oracle.py) is checked against the reference.
nn.Dropout layers of the MAP head are set to p = 0 (their masks are not reproducible); DropPath rate 0."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'timm_stub'))
sys.path.insert(0, '/root/reference/MAP/models')
sys.path.insert(0, os.path.dirname(HERE))

import map_pit as ref  # noqa: E402  (the reference)
from oracle import map_pit_oracle as O  # noqa: E402
from oracle import map_oracle as MO  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
V8 = dict(image_size=64, patch_size=16, stride=8, base_dims=(48, 48, 48), depth=(1, 2, 1), heads=(1, 2, 4), num_classes=40, last_dim=64,
          n_groups=2, n_tokens=4, gram_group=8)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def build_ref(cfg):
    m = ref.PoolingTransformer(image_size=cfg['image_size'], patch_size=cfg['patch_size'], stride=cfg['stride'],
                               base_dims=list(cfg['base_dims']), depth=list(cfg['depth']), heads=list(cfg['heads']), mlp_ratio=4,
                               num_classes=cfg['num_classes'], pool_type='map', last_dim=cfg['last_dim'], n_groups=cfg['n_groups'],
                               n_tokens=cfg['n_tokens'], gram_group=cfg['gram_group'])
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    sd = O.fill_state(cfg)
    rsd = m.state_dict()
    assert list(rsd.keys()) == list(sd.keys()), [k for k in rsd if k not in sd][:5] + [k for k in sd if k not in rsd][:5]
    for k in sd:
        assert tuple(rsd[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    return m, sd


def do(tag, cfg, batch_eval=2, batch_train=4, train=True):
    m, sd = build_ref(cfg)
    m.eval()
    x = O.gen_input(batch_eval, seed=0, size=cfg['image_size'])
    with torch.no_grad():
        outs = m(x)
        mine = O.forward(sd, x, cfg, training=False)
    err = max(rel(a, b) for a, b in zip(mine, outs))
    print(f'[{tag}] eval: oracle vs reference max rel err = {err:.3e}')
    assert err < 1e-4
    s = sum(o.float() for o in outs) / len(outs)
    np.savez_compressed(os.path.join(OUT, f'{tag}_eval.npz'), cfg=json.dumps(cfg), batch=batch_eval,
                        param_count=sum(p.numel() for p in m.parameters()), logits=torch.stack(outs)[:, :, :40].numpy(),
                        top5=s.topk(5, 1, True, True)[1].numpy())
    if not train:
        return
    m.train()
    x = O.gen_input(batch_train, seed=1, size=cfg['image_size'])
    target = torch.randint(0, cfg['num_classes'], (batch_train,), generator=torch.Generator().manual_seed(99))
    outs = m(x)
    loss = MO.multi_group_loss(outs, target, -0.8)
    loss.backward()
    grads = {n: p.grad.detach() for n, p in m.named_parameters()}
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, dec_lam=-0.8)
    from oracle import ga_convnext_oracle as GO
    e_out = max(max(rel(a, b.detach()) for a, b in zip(o, oo)) for o, oo in zip(oouts, outs))
    e_loss = abs(float(oloss) - float(loss.detach())) / abs(float(loss.detach()))
    e_g = max(GO.grad_errors(ograds, grads).values())
    print(f'[{tag}] train B={batch_train}: oracle vs reference rel err: logits {e_out:.2e} loss {e_loss:.2e} grads {e_g:.2e}')
    assert max(e_out, e_loss) < 1e-4 and e_g < 1e-2
    names = list(grads.keys())
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b{batch_train}.npz'), cfg=json.dumps(cfg), batch=batch_train, dec_lam=-0.8,
                        target=target.numpy(), loss=float(loss), org=torch.stack([o[0].detach() for o in outs])[:, :, :40].numpy(),
                        avg=torch.stack([o[1].detach() for o in outs])[:, :, :40].numpy(), grad_names=np.array(names),
                        grad_norm=np.array([float(grads[n].double().norm()) for n in names]))


if __name__ == '__main__':
    torch.manual_seed(0)
    do('pit_v8', O.make_cfg(**V8))
    do('pit_s', O.make_cfg('map_pit_s'), train=False)
    print('golden vectors written to', OUT)
