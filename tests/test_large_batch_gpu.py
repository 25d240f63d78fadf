"""GPU: the benchmark-size kernel forms of every family inside a model (BASELINE.json configs[2] and [4] and the MAP backbones).

test_model_gpu.py checks GA-ConvNeXt at B = 256; this file does the same for the other families, because their small-batch
parity tests (B = 4) never reach the large-M kernel selections -- the 3-slot ring / ping-pong / LDS-DMA GEMM forms, the wide
weight-gradient form, the MFMA stripe- and global-attention kernels at full token counts, the fused MLP bodies:

  * fp32 math mode at a batch the CPU oracle still finishes in well under a minute (64 images; 2 for ViT-B/16 @ 384) against
    the oracle: logits / loss 1e-3, gradients 3e-2 under oracle.grad_errors, top-5 of the summed / averaged heads bit-exact;
  * bf16 mode at the BENCH batch (256; 128 for ViT-B/16 @ 384) against this library's own fp32 mode on the same inputs: logits
    6e-2, loss 2e-2, every gradient tensor within the whole-tensor gates of tests/_gradcheck.py.

map_vit_base_patch16_384 is a builder-defined composition: its parity is against oracle/map_vit_oracle.py only --
"parity unpinned" by reference fixtures; its block and head are the reference's (pinned through map_pit_s / map_convnext)."""
import os

import pytest
import torch

from _gradcheck import assert_grads_close, BF16_REL, BF16_COS

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


FAMILIES = {
    # name: (oracle module, loss kind, fp32-vs-oracle batch, bf16-vs-fp32 batch)
    'ga_CSWin_64_12211_tiny_224': ('ga_cswin_oracle', 'ga', 64, 256),
    'map_convnext_tiny': ('map_oracle', 'map', 64, 256),
    'map_pit_s': ('map_pit_oracle', 'map', 64, 256),
    'map_vit_base_patch16_384': ('map_vit_oracle', 'map', 2, 128),
}


def _oracle(mod):
    import importlib
    return importlib.import_module('oracle.' + mod)


def _model(name, mode, O):
    import imagenet_models_amd as A
    cfg = O.make_cfg(name)
    sd = O.fill_state(cfg)
    # the MAP head's nn.Dropout(0.05) sites (map.py:149) are random in train mode: off for parity (tests/test_map_model_gpu.py
    # covers them with injected masks)
    kw = {} if name.startswith('ga_') else dict(head_drop=0.0, head_attn_drop=0.0)
    m = A.create_model(name, math_mode=mode, **kw)
    m.load_state_dict(sd)
    return m.cuda().train(), sd, cfg


def _flat(outs):
    """GA: list of logits; MAP (train): list of [org, avg] pairs"""
    flat = []
    for o in outs:
        flat.extend(o if isinstance(o, (list, tuple)) else [o])
    return flat


def _step(m, x, target, kind):
    import imagenet_models_amd as A
    m.zero_grad()
    outs = m(x.cuda())
    loss = A.ga_loss(outs, target.cuda(), -0.8) if kind == 'ga' else A.map_loss(outs, target.cuda(), -0.8)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    return [o.detach().float().cpu() for o in _flat(outs)], float(loss.detach()), grads


def _inputs(O, cfg, B, seed):
    size = cfg.get('img_size', cfg.get('image_size', 224))
    try:
        x = O.gen_input(B, seed=seed, size=size)
    except TypeError:
        x = O.gen_input(B, seed=seed)
    target = torch.randint(0, cfg['num_classes'], (B,), generator=torch.Generator().manual_seed(seed))
    return x, target


@pytest.mark.parametrize('name', list(FAMILIES))
def test_fp32_mode_vs_oracle_at_a_large_batch(name):
    import imagenet_models_amd as A
    from oracle import ga_convnext_oracle as GO
    mod, kind, B, _ = FAMILIES[name]
    O = _oracle(mod)
    m, sd, cfg = _model(name, 'fp32', O)
    x, target = _inputs(O, cfg, B, 7)
    outs, loss, grads = _step(m, x, target, kind)
    # eval-mode top-5 of the same model on the first 8 images, against the oracle's eval forward
    m.eval()
    with torch.no_grad():
        ev = m(x[:8].cuda()) if B >= 8 else m(x.cuda())
    nb = min(B, 8)
    sd_eval = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}     # (the train step moved the BatchNorm running statistics)
    del m
    torch.cuda.empty_cache()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    step_kw = dict(lam=-0.8) if kind == 'ga' else dict(dec_lam=-0.8)
    res = O.train_step_grads(sd, x, target, cfg, **step_kw)
    oloss, oouts, ograds = res[0], _flat(res[1]), res[2]
    e_out = max(rel(a, b) for a, b in zip(outs, oouts))
    e_loss = abs(loss - float(oloss)) / abs(float(oloss))
    errs = GO.grad_errors(grads, ograds)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f'[{name} fp32 B={B}] logits {e_out:.2e} loss {e_loss:.2e} worst grads {worst}')
    assert e_out < 1e-3 and e_loss < 1e-3
    assert worst[0][1] < 3e-2, worst
    with torch.no_grad():
        oev = O.forward(sd_eval, x[:nb], cfg, training=False)
    if kind == 'ga':
        _, idx = A.heads_topk(ev, 5)
        want = sum(o.float() for o in oev).topk(5, 1, True, True)[1]
    else:
        _, idx = A.heads_mean_topk(ev, 5)
        want = (sum(o.float() for o in oev) / len(oev)).topk(5, 1, True, True)[1]
    assert torch.equal(idx.cpu(), want), f'{name}: top-5 of the combined heads differs from the oracle'


@pytest.mark.parametrize('name', list(FAMILIES))
def test_bf16_mode_vs_own_fp32_mode_at_the_bench_batch(name):
    mod, kind, _, B = FAMILIES[name]
    O = _oracle(mod)
    m, sd, cfg = _model(name, 'fp32', O)
    x, target = _inputs(O, cfg, B, 8)
    o32, l32, g32 = _step(m, x, target, kind)
    del m
    torch.cuda.empty_cache()
    m, _, _ = _model(name, 'bf16', O)
    o16, l16, g16 = _step(m, x, target, kind)
    e_out = max(rel(a, b) for a, b in zip(o16, o32))
    e_loss = abs(l16 - l32) / abs(l32)
    print(f'[{name} bf16 vs fp32 mode, B={B}] logits {e_out:.2e} loss {e_loss:.2e}')
    assert all(torch.isfinite(g).all() for g in g16.values())
    assert e_out < 6e-2 and e_loss < 2e-2
    assert_grads_close(g16, g32, BF16_REL, BF16_COS, f'{name} bf16 vs fp32 mode B={B}')
