import os, sys, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import imagenet_models_amd as A
from oracle import map_oracle as O
from oracle import ga_convnext_oracle as GO
for gd in (96, 128):
    cfg = O.make_cfg(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), last_dim=64, n_groups=2, n_tokens=2, gram_group=8, bp_dim=64, ca_dim=64,
                     num_heads=8, num_classes=40, gram_dim=gd)
    m = A.MAP_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], last_dim=64, n_groups=2, n_tokens=2, gram_group=8, bp_dim=64,
                       ca_dim=64, num_heads=8, head_drop=0.0, head_attn_drop=0.0, math_mode='fp32', gram_dim=gd)
    sd = O.fill_state(cfg); m.load_state_dict(sd); m = m.cuda().train()
    B = 4
    x = O.gen_input(B, seed=1); target = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(99))
    for rep in range(2):
        m.zero_grad()
        outs = m(x.cuda()); loss = A.map_loss(outs, target.cuda(), -0.8); loss.backward(); torch.cuda.synchronize()
        grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        oloss, oouts, ograds, _ = O.train_step_grads(sd, x, target, cfg, dec_lam=-0.8)
        errs = GO.grad_errors(grads, ograds)
        bad = [(n, round(e, 4)) for n, e in sorted(errs.items(), key=lambda kv: -kv[1]) if e > 2e-3]
        print('gd', gd, 'rep', rep, 'loss', float(loss), float(oloss), 'bad', bad[:8])
        if bad:
            n = bad[0][0]; d = (grads[n] - ograds[n]).abs().reshape(grads[n].shape[0], -1)
            print('   ', n, tuple(grads[n].shape), 'rows with err', (d.max(1)[0] > 1e-3 * float(ograds[n].abs().max())).nonzero().flatten().tolist()[:40])
