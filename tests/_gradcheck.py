"""Gradient gates of the bf16-mode parity tests (test infrastructure).

The fp32-mode tests gate every parameter gradient by max|a - b| / max|b| (oracle.grad_errors).  In the bf16 throughput mode
that measure is dominated by single elements (one rounded activation near a ReLU / GELU knee) and had to be set so loose
(0.35 .. 1.0) that a sign flip of a whole tensor would pass.  The bf16 gates therefore use two whole-tensor measures that a
wrong kernel cannot satisfy by accident:

    rel  = ||g - g*||_2 / ||g*||_2       (norm-relative error)
    cos  = <g, g*> / (||g|| ||g*||)      (direction)

over every tensor whose reference gradient is not analytically zero (max|g*| >= floor * global max; the excluded ones --
biases feeding a train-mode BatchNorm -- hold round-off noise only).  A sign flip gives rel = 2, cos = -1; a missing
contribution of 20 % gives rel = 0.2."""
import os

import torch

# The bf16-mode gate of every model test.  Calibrated on MI355X (round 3, every family, GA_GRADCHECK_REPORT=1): the worst tensors of
# the throughput mode sit at rel 0.13 .. 0.25, cos 0.969 .. 0.993 (small per-channel parameters -- depthwise biases, LayerNorm /
# BatchNorm affine gradients -- that sum bf16-rounded activations over 10^4 .. 10^6 rows; the weight matrices are at rel <= 0.1).
# rel <= 0.3 and cos >= 0.95 leave no room for a sign error (rel 2, cos -1), a dropped term of a third of the gradient or a
# wrong scale by more than 30 %.
BF16_REL, BF16_COS = 0.3, 0.95
REPORT_ONLY = os.environ.get('GA_GRADCHECK_REPORT', '') == '1'    # calibration runs: print, do not assert


def norm_errors(got, ref, floor=1e-4):
    """{name: (rel, cos)} for every tensor above the floor"""
    gmax = max(float(g.abs().max()) for g in ref.values())
    out = {}
    for n, b in ref.items():
        b = torch.as_tensor(b)
        if float(b.abs().max()) < floor * gmax:
            continue
        a = got[n].detach().double().flatten().cpu()
        b = b.detach().double().flatten().cpu()
        nb = float(b.norm())
        rel = float((a - b).norm()) / nb
        cos = float(torch.dot(a, b)) / (float(a.norm()) * nb + 1e-300)
        out[n] = (rel, cos)
    return out


def assert_grads_close(got, ref, rel_max, cos_min, what='', floor=1e-4, allow=()):
    """every tensor: rel <= rel_max and cos >= cos_min.  `allow`: {name substring: (rel_max, cos_min)} for tensors with a
    documented reason to be looser.  Returns (worst rel, worst cos) for the test's report line."""
    errs = norm_errors(got, ref, floor)
    assert errs, f'{what}: no gradient above the floor'
    bad = []
    for n, (rel, cos) in errs.items():
        rm, cm = rel_max, cos_min
        for key, lim in dict(allow).items():
            if key in n:
                rm, cm = lim
        if not (rel <= rm and cos >= cm):
            bad.append((n, round(rel, 4), round(cos, 5)))
    wr = max(errs.items(), key=lambda kv: kv[1][0])
    wc = min(errs.items(), key=lambda kv: kv[1][1])
    print(f'[{what}] gradients: {len(errs)} tensors, worst rel {wr[1][0]:.3e} ({wr[0]}), worst cos {wc[1][1]:.5f} ({wc[0]})')
    if REPORT_ONLY:
        if bad:
            print(f'[{what}] WOULD FAIL rel <= {rel_max}, cos >= {cos_min}: {sorted(bad, key=lambda t: -t[1])[:12]}')
        return wr[1][0], wc[1][1]
    assert not bad, f'{what}: {len(bad)} gradient tensors outside rel <= {rel_max}, cos >= {cos_min}: {sorted(bad, key=lambda t: -t[1])[:8]}'
    return wr[1][0], wc[1][1]
