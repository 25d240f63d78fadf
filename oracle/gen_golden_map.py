"""Generate tests/golden/map_*.npz from the REAL reference classes of /root/reference/MAP/models/{map,map_convnext}.py
(build container only; never runs on the GPU box).

Run:  python oracle/gen_golden_map.py

map.py imports with torch alone; map_convnext.py needs oracle/timm_stub (create_model, trunc_normal_, DropPath,
register_model).  Every nn.Dropout inside the head is set to p = 0 before running (CABlock hard-codes 0.05, map.py:149;
dropout masks are not reproducible across implementations), everything else is the reference as it is.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'timm_stub'))
sys.path.insert(0, '/root/reference/MAP/models')
sys.path.insert(0, os.path.dirname(HERE))

import map_convnext as ref  # noqa: E402  (the reference; it imports `map` from the same directory)
from oracle import map_oracle as O  # noqa: E402
from oracle.gen_golden import grad_stats, rel  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

# narrow whole-model configs: 2 gram tokens (+1 self-distill) like map_convnext_tiny / 3 (+1) like map_convnext_small
V5 = dict(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), last_dim=64, n_groups=2, n_tokens=2, gram_group=8, bp_dim=64, ca_dim=64,
          num_heads=8, num_classes=40)
V5S = dict(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), last_dim=64, n_groups=3, n_tokens=3, gram_group=4, bp_dim=48, ca_dim=64,
           num_heads=8, num_classes=40)


def build_ref(cfg):
    m = ref.ConvNeXt(num_classes=cfg['num_classes'], depths=list(cfg['depths']), dims=list(cfg['dims']), drop_path_rate=0.0,
                     global_pool='mmcap', last_dim=cfg['last_dim'], n_groups=cfg['n_groups'], n_tokens=cfg['n_tokens'],
                     gram_group=cfg['gram_group'], bp_dim=cfg['bp_dim'], bp_groups=cfg['bp_groups'], ca_dim=cfg['ca_dim'],
                     num_heads=cfg['num_heads'])
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    sd = O.fill_state(cfg)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict key order differs from the reference:\n' + '\n'.join(
        f'{a} | {b}' for a, b in zip(ref_sd.keys(), sd.keys()) if a != b)
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (k, tuple(ref_sd[k].shape), tuple(sd[k].shape))
        if O.is_index_buffer(k):
            assert torch.equal(ref_sd[k], sd[k]), k
    m.load_state_dict(sd)
    return m, sd


def ref_multi_group_loss(outputs, target, dec_lam):
    """verbatim arithmetic of MAP/train.py:792-839 for distill_tokens == 0 and loss_fn = CrossEntropyLoss"""
    import torch.nn.functional as F
    loss = 0
    y_hat_aggre = 0
    for output in outputs:
        y_hat, y_mean_hat = output
        y_hat_aggre += y_hat
        adv_loss = F.kl_div(F.log_softmax(y_mean_hat, dim=1), F.log_softmax(y_hat, dim=1).detach(),
                            reduction='sum', log_target=True) / y_hat.numel()
        loss += F.cross_entropy(y_hat, target) + adv_loss
    if len(outputs) > 1:
        for output in outputs:
            y_hat, y_mean_hat = output
            loss += F.kl_div(F.log_softmax(y_hat, dim=1), F.log_softmax((y_hat_aggre.detach() / len(outputs)), dim=1),
                             reduction='mean', log_target=True) * dec_lam
    return loss


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
    s = sum(outs) / len(outs)
    np.savez_compressed(os.path.join(OUT, f'{tag}_eval.npz'), cfg=json.dumps(cfg), batch=batch,
                        param_count=sum(p.numel() for p in m.parameters()), n_state=len(sd),
                        logits=torch.stack(outs)[:, :, :nlog].numpy(), top5=s.topk(5, 1, True, True)[1].numpy())


def do_train(tag, cfg, batch, dec_lam=-0.8):
    m, sd = build_ref(cfg)
    m.train()
    x = O.gen_input(batch, seed=1)
    tg = torch.Generator().manual_seed(99)
    target = torch.randint(0, cfg['num_classes'], (batch,), generator=tg)
    outs = m(x)
    loss = ref_multi_group_loss(outs, target, dec_lam)
    loss.backward()
    grads = {n: p.grad.detach() for n, p in m.named_parameters()}
    new_sd = m.state_dict()
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, dec_lam=dec_lam)
    e_out = max(max(rel(a, b.detach()) for a, b in zip(oo, ro)) for oo, ro in zip(oouts, outs))
    e_loss = abs(float(oloss) - float(loss.detach())) / abs(float(loss.detach()))
    e_g = max(O.grad_errors(ograds, grads).values())
    e_bn = max(rel(ostats[n].float(), new_sd[n].float()) for n in ostats)
    print(f'[{tag}] train B={batch}: oracle vs reference rel err: logits {e_out:.2e} loss {e_loss:.2e} '
          f'grads {e_g:.2e} bn-stats {e_bn:.2e}')
    assert max(e_out, e_loss, e_bn) < 1e-4 and e_g < 1e-2
    names, norm, ssum, head = grad_stats(grads)
    bn_names = [n for n in new_sd if n.endswith('running_mean') or n.endswith('running_var')]
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b{batch}.npz'), cfg=json.dumps(cfg), batch=batch, dec_lam=dec_lam,
                        target=target.numpy(), loss=float(loss),
                        org=torch.stack([o[0].detach() for o in outs])[:, :, :40].numpy(),
                        avg=torch.stack([o[1].detach() for o in outs])[:, :, :40].numpy(),
                        grad_names=np.array(names), grad_norm=norm, grad_sum=ssum, grad_head=head,
                        bn_names=np.array(bn_names),
                        bn_head=np.stack([new_sd[n].reshape(-1)[:8].numpy() for n in bn_names]))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    v5 = O.make_cfg(**V5)
    do_eval('map_v5', v5, 2, 40)
    do_train('map_v5', v5, 4)
    v5s = O.make_cfg(**V5S)
    do_eval('map_v5s', v5s, 2, 40)
    do_train('map_v5s', v5s, 4)
    tiny = O.make_cfg('map_convnext_tiny')
    do_eval('map_tiny', tiny, 2, 16)
    do_train('map_tiny', tiny, 4)
    small = O.make_cfg('map_convnext_small')
    shapes = O.state_shapes(small)
    print('map_convnext_small params', sum(int(np.prod(s)) for k, s in shapes.items() if O.is_param(k)))
    print('golden vectors written to', OUT)
