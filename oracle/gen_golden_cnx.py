"""Generate tests/golden/cnx_*.npz from the REAL /root/reference/MAP/models/map_convnext.py ConvNeXt with global_pool='avg'
(build container only).  The oracle restatement (oracle/convnext_oracle.py) is checked against the reference while generating."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'timm_stub'))
sys.path.insert(0, '/root/reference/MAP/models')
sys.path.insert(0, os.path.dirname(HERE))

import map_convnext as ref  # noqa: E402  (the reference)
from oracle import convnext_oracle as O  # noqa: E402
from oracle import ga_convnext_oracle as GO  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
V9 = dict(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), num_classes=40)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def build_ref(cfg):
    m = ref.ConvNeXt(num_classes=cfg['num_classes'], depths=list(cfg['depths']), dims=list(cfg['dims']), drop_path_rate=0.0, global_pool='avg')
    sd = O.fill_state(cfg)
    rsd = m.state_dict()
    assert list(rsd.keys()) == list(sd.keys()), [(a, b) for a, b in zip(rsd, sd) if a != b][:5]
    for k in sd:
        assert tuple(rsd[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    return m, sd


def do(tag, cfg, train=True):
    m, sd = build_ref(cfg)
    m.eval()
    x = O.gen_input(2, seed=0)
    with torch.no_grad():
        out = m(x)
        mine = O.forward(sd, x, cfg)
    e = rel(mine, out)
    print(f'[{tag}] eval: oracle vs reference {e:.3e}')
    assert e < 1e-4
    np.savez_compressed(os.path.join(OUT, f'{tag}_eval.npz'), cfg=json.dumps({k: cfg[k] for k in ('dims', 'depths', 'num_classes')}), batch=2,
                        param_count=sum(p.numel() for p in m.parameters()), logits=out[:, :40].numpy(), top5=out.topk(5, 1, True, True)[1].numpy())
    if not train:
        return
    m.train()
    x = O.gen_input(4, seed=1)
    target = torch.randint(0, cfg['num_classes'], (4,), generator=torch.Generator().manual_seed(99))
    out = m(x)
    loss = F.cross_entropy(out, target)
    loss.backward()
    grads = {n: p.grad.detach() for n, p in m.named_parameters()}
    oloss, oout, ograds = O.train_step_grads(sd, x, target, cfg)
    e_out, e_loss = rel(oout, out.detach()), abs(float(oloss) - float(loss)) / abs(float(loss))
    e_g = max(GO.grad_errors(ograds, grads).values())
    print(f'[{tag}] train: oracle vs reference logits {e_out:.2e} loss {e_loss:.2e} grads {e_g:.2e}')
    assert max(e_out, e_loss) < 1e-4 and e_g < 1e-2
    names = list(grads.keys())
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b4.npz'), cfg=json.dumps({k: cfg[k] for k in ('dims', 'depths', 'num_classes')}), batch=4,
                        target=target.numpy(), loss=float(loss), logits=out.detach()[:, :40].numpy(), grad_names=np.array(names),
                        grad_norm=np.array([float(grads[n].double().norm()) for n in names]))


if __name__ == '__main__':
    torch.manual_seed(0)
    do('cnx_v9', O.make_cfg(**V9))
    do('cnx_tiny', O.make_cfg('convnext_tiny'), train=False)
    print('written to', OUT)
