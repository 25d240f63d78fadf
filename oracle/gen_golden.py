"""Generate tests/golden/*.npz from the REAL reference classes (build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

Imports /root/reference/GA/ga_convnext.py against oracle/timm_stub (timm is not installed),
loads the deterministic name-hashed weights of oracle.ga_convnext_oracle.fill_state, runs the
reference model, and stores ONLY input-independent metadata + outputs (inputs are regenerated
from the closed-form generator gen_input).  While generating, it also checks the oracle
restatement against the reference and prints the max deviations.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'timm_stub'))
sys.path.insert(0, '/root/reference/GA')
sys.path.insert(0, os.path.dirname(HERE))

import ga_convnext as ref  # noqa: E402  (the reference)
from oracle import ga_convnext_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

V2 = dict(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)


def build_ref(cfg):
    m = ref.GA_ConvNeXt(num_classes=cfg['num_classes'], depths=cfg['depths'], dims=cfg['dims'],
                        gram_embedding_gropus=cfg['gram_groups'], dim_embed=cfg['dim_embed'],
                        stage3_naggre=cfg['naggre'], gram_dim=cfg['gram_dim'], drop_path_rate=0.0)
    sd = O.fill_state(cfg)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict key order differs from the reference'
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    return m, sd


def ref_loss(outputs, target, lam):
    # verbatim formula of GA/train.py:735-745, applied to the reference model's outputs
    output = 0
    loss = 0
    for out in outputs:
        loss = loss + F.cross_entropy(out, target)
        output = output + out.data
    for out in outputs:
        loss = loss + F.kl_div(F.log_softmax(out + 0), F.log_softmax((output.detach() / len(outputs)) + 0),
                               reduction='mean', log_target=True) * lam
    return loss


def grad_stats(grads):
    names = list(grads.keys())
    norm = np.array([float(grads[n].double().norm()) for n in names])
    ssum = np.array([float(grads[n].double().sum()) for n in names])
    head = np.zeros((len(names), 16), dtype=np.float32)
    for i, n in enumerate(names):
        f = grads[n].reshape(-1)[:16]
        head[i, :f.numel()] = f.numpy()
    return names, norm, ssum, head


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


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
    np.savez_compressed(os.path.join(OUT, f'{tag}_eval.npz'),
                        cfg=json.dumps(cfg), batch=batch,
                        param_count=sum(p.numel() for p in m.parameters()),
                        n_state=len(sd),
                        logits=torch.stack(outs)[:, :, :nlog].numpy(),
                        top5=s.topk(5, 1, True, True)[1].numpy())


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
    # oracle check
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, lam=lam)
    e_out = max(rel(a, b.detach()) for a, b in zip(oouts, outs))
    e_loss = abs(float(oloss) - float(loss.detach())) / abs(float(loss.detach()))
    e_g = max(O.grad_errors(ograds, grads).values())
    e_bn = max(rel(ostats[n].float(), new_sd[n].float()) for n in ostats)
    print(f'[{tag}] train B={batch}: oracle vs reference rel err: logits {e_out:.2e} loss {e_loss:.2e} '
          f'grads {e_g:.2e} bn-stats {e_bn:.2e}')
    # fp32 round-off in the reference's OWN backward reaches ~4e-3 at B=128 (measured against a float64 run of the
    # restatement: reference 4.2e-3, restatement 9.5e-4), so gradients are gated at 1e-2, logits/loss at 1e-4.
    assert max(e_out, e_loss, e_bn) < 1e-4 and e_g < 1e-2
    names, norm, ssum, head = grad_stats(grads)
    bn_names = [n for n in new_sd if n.endswith('running_mean') or n.endswith('running_var')]
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b{batch}.npz'),
                        cfg=json.dumps(cfg), batch=batch, lam=lam, target=target.numpy(),
                        loss=float(loss), logits=torch.stack([o.detach() for o in outs])[:, :, :40].numpy(),
                        grad_names=np.array(names), grad_norm=norm, grad_sum=ssum, grad_head=head,
                        bn_names=np.array(bn_names),
                        bn_head=np.stack([new_sd[n].reshape(-1)[:8].numpy() for n in bn_names]))


def do_train_fp64(tag, cfg, batch, lam=-0.8):
    """float64 ground truth of one train step from the oracle restatement (same weights / input / target as do_train):
    per-tensor gradient norms, sums and 16-value heads, logits and loss.  The fp32-mode library gradients are gated at
    5e-3 against THESE (the fp32 reference's own backward round-off is up to 4e-3, see do_train)."""
    sd = O.fill_state(cfg, dtype=torch.float64)
    x = O.gen_input(batch, seed=1).double()
    tg = torch.Generator().manual_seed(99)
    target = torch.randint(0, cfg['num_classes'], (batch,), generator=tg)
    loss, outs, grads, _ = O.train_step_grads(sd, x, target, cfg, lam=lam)
    assert all(g.dtype == torch.float64 for g in grads.values()) and outs[0].dtype == torch.float64
    names = list(grads.keys())
    norm = np.array([float(grads[n].norm()) for n in names])
    ssum = np.array([float(grads[n].sum()) for n in names])
    amax = np.array([float(grads[n].abs().max()) for n in names])
    head = np.zeros((len(names), 16), dtype=np.float64)
    for i, n in enumerate(names):
        f = grads[n].reshape(-1)[:16]
        head[i, :f.numel()] = f.numpy()
    print(f'[{tag}] fp64 oracle train B={batch}: loss {float(loss):.9f}')
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b{batch}_fp64.npz'), cfg=json.dumps(cfg), batch=batch, lam=lam,
                        target=target.numpy(), loss=float(loss), logits=torch.stack(outs)[:, :, :40].numpy(),
                        grad_names=np.array(names), grad_norm=norm, grad_sum=ssum, grad_absmax=amax, grad_head=head)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    if '--odd-widths-only' in sys.argv:      # the 688 / 976 variants (a14), added in round 2
        t688 = O.make_cfg('ga_convnext_tiny_688')
        do_eval('t688', t688, 2, 16)
        do_train('t688', t688, 4)
        do_eval('b976', O.make_cfg('ga_convnext_base_976'), 2, 16)
        sys.exit(0)
    if '--fp64-only' in sys.argv:      # the float64 ground-truth fixtures need only the oracle
        do_train_fp64('v2', O.make_cfg(**V2), 4)
        do_train_fp64('t768', O.make_cfg('ga_convnext_tiny_768'), 4)
        sys.exit(0)
    v2 = O.make_cfg(**V2)
    do_eval('v2', v2, 2, 40)
    do_train('v2', v2, 4)
    do_train('v2', v2, 128)
    t768 = O.make_cfg('ga_convnext_tiny_768')
    do_eval('t768', t768, 2, 16)
    do_train('t768', t768, 4)
    do_train_fp64('v2', v2, 4)
    do_train_fp64('t768', t768, 4)
    b1024 = O.make_cfg('ga_convnext_base_1024')
    do_eval('b1024', b1024, 2, 16)
    t688 = O.make_cfg('ga_convnext_tiny_688')
    do_eval('t688', t688, 2, 16)
    do_train('t688', t688, 4)
    do_eval('b976', O.make_cfg('ga_convnext_base_976'), 2, 16)
    print('golden vectors written to', OUT)
