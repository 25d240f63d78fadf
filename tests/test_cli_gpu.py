"""GPU: the timm-style command-line surface (train.py / validate.py, GA/train.py + MAP/train.py + MAP/validate.py flag names)
for the three model families on synthetic data, and the N > 1 launch of train.py (2 ranks sharing the box's GPU over gloo:
bucketed all-reduce, per-forward BatchNorm-buffer broadcast, epoch-end distribute_bn)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None, timeout=900):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return r.stdout + r.stderr


@pytest.mark.parametrize('model,extra', [('ga_convnext_tiny_768', ['--GA_lam', '-0.8', '--opt', 'lamb', '--bce-loss']),
                                         ('ga_CSWin_64_12211_tiny_224', ['--GA_lam', '-0.8', '--opt', 'adamw', '--lr', '1e-3', '--mixup', '0', '--cutmix', '0']),
                                         ('ga_convnext_tiny_768', ['--GA_lam', '-0.8', '--clip-mode', 'agc', '--clip-grad', '0.02', '--mixup-off-epoch', '1']),
                                         ('map_convnext_tiny', ['--dec-lam', '-0.8', '--opt', 'adamw', '--lr', '1e-3']),
                                         ('map_vit_small_patch16_224', ['--dec-lam', '-0.8', '--opt', 'adamw', '--lr', '1e-3']),
                                         ('map_pit_s', ['--dec-lam', '-0.8', '--opt', 'adamw', '--lr', '1e-3']),
                                         ('convnext_tiny', ['--opt', 'adamw', '--lr', '1e-3', '--smoothing', '0.1'])])
def test_train_cli_runs_every_family(model, extra):
    out = _run([sys.executable, 'train.py', '--synthetic', '--model', model, '-b', '8', '--epochs', '1', '--steps-per-epoch', '3',
                '--drop-path', '0.1', '--log-interval', '1', '--clip-grad', '5.0'] + extra)
    assert '*** epoch 0: train loss' in out and 'nan' not in out.lower()


def test_validate_cli_writes_results(tmp_path):
    res = os.path.join(tmp_path, 'r.json')
    out = _run([sys.executable, 'validate.py', '--synthetic', '--model', 'map_convnext_tiny', '-b', '8', '--batches', '2',
                '--results-file', res])
    assert 'Acc@1' in out
    r = json.load(open(res))
    assert r['model'] == 'map_convnext_tiny' and r['param_count'] == 47.83


def test_train_cli_two_ranks_gloo():
    env = dict(os.environ, GA_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', '29577', 'train.py', '--synthetic', '--model', 'map_convnext_tiny', '-b', '4', '--epochs', '1',
                '--steps-per-epoch', '2', '--dec-lam', '-0.8', '--opt', 'adamw', '--lr', '1e-3', '--log-interval', '1'], env=env)
    assert '*** epoch 0: train loss' in out
