"""GPU: the RCCL side of the boundary on the one GPU the test box has.

  * imagenet_models_amd.NativeComm (ga_comm_* / ga_allreduce_bucket / ga_reduce_scatter_bucket / ga_allgather_bucket /
    ga_comm_broadcast, include/gaext.h) with ONE rank: every collective is a real RCCL call on a real communicator and must
    be the identity (x scale; bf16 wire: the bf16 rounding of the input);
  * TrainStep's segmented backward + bucketed reduction (the path N > 1 takes, GA/train.py:514) forced on with one rank, the
    buckets going through the native communicator on its side stream: the step must equal the plain one-stream step;
  * the same through torch.distributed's `nccl` backend (= RCCL) with world_size 1, in a child process (tests/nccl_ws1_check.py).
Reference: NativeDDP's reducer averages gradients over ranks; with one rank the reduced gradient is the local one."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_comm_single_rank_collectives():
    import imagenet_models_amd as A
    c = A.NativeComm(wire='fp32')
    assert (c.rank, c.world) == (0, 1)
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn((1 << 20) + 3, device='cuda', generator=g)
    main = torch.cuda.current_stream()
    y = x.clone()
    c.after(main)
    c.allreduce(y)
    c.join(main)
    assert torch.equal(y, x)
    c.after(main)
    c.allreduce(y, scale=0.25)
    c.join(main)
    assert torch.equal(y, x * 0.25)
    shard, full = torch.empty_like(x), torch.empty_like(x)
    c.after(main)
    c.reduce_scatter(x, shard, scale=2.0)
    c.allgather(shard, full)
    c.broadcast(full, 0)
    c.join(main)
    assert torch.equal(full, x * 2.0)
    cb = A.NativeComm(wire='bf16')
    z = x[:(1 << 20)].clone()                   # (16-byte aligned slice)
    cb.after(main)
    cb.allreduce(z)
    cb.join(main)
    assert torch.equal(z, x[:(1 << 20)].bfloat16().float())
    torch.cuda.synchronize()
    c.close()
    cb.close()


def _small_model():
    import imagenet_models_amd as A
    from oracle import ga_convnext_oracle as O
    cfg = O.make_cfg(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)
    sd = O.fill_state(cfg)
    m = A.GA_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], gram_embedding_gropus=cfg['gram_groups'],
                      dim_embed=cfg['dim_embed'], stage3_naggre=cfg['naggre'], gram_dim=cfg['gram_dim'], math_mode='fp32')
    m.load_state_dict(sd)
    return m.cuda().train(), O


@pytest.mark.parametrize('wire', ['fp32', 'bf16'])
def test_trainstep_buckets_through_native_comm_equal_plain_step(wire):
    import imagenet_models_amd as A
    B = 8
    res = {}
    for tag in ('plain', 'native'):
        m, O = _small_model()
        opt = A.create_optimizer_v2(m, opt='sgd', lr=1e-2, momentum=0.9, weight_decay=0.05)
        x = O.gen_input(B, seed=2).cuda()
        y = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).cuda()
        if tag == 'native':
            comm = A.NativeComm(wire=wire)
            step = A.TrainStep(m, opt, B, lam=-0.8, comm=comm, force_buckets=True, bucket_elems=50_000, nan_guard=True)
            assert len(step.buckets) >= 4
        else:
            step = A.TrainStep(m, opt, B, lam=-0.8, overlap_optimizer=False)
        p0 = m.flat_state()['params'].clone()
        loss = step(x, y)
        torch.cuda.synchronize()
        res[tag] = (float(loss), m.flat_state()['params'].clone())
        slices = m.flat_state()['slices']
        if tag == 'native':
            assert abs(float(step.last_loss_sum) - float(loss)) <= (1e-6 if wire == 'fp32' else 1e-2) * abs(float(loss))
            comm.close()
    assert abs(res['plain'][0] - res['native'][0]) < 1e-6 * abs(res['plain'][0])
    _assert_same_update(res['plain'][1], res['native'][1], p0, slices, 1e-3 if wire == 'fp32' else 2e-2)


def _assert_same_update(pa, pb, p0, slices, tol):
    """SGD: the parameter change is lr x gradient, so comparing the two steps' updates slice by slice compares the reduced
    gradients.  Per slice: |difference| <= tol x (slice max) + 1e-4 x (global max) -- the criterion of tests/ddp_check.py: slices
    whose true gradient is zero (biases in front of a train-mode BatchNorm) hold only atomics / cancellation noise, which differs
    between any two backward passes; a slice reduced before it was complete, or twice, is off by its own magnitude."""
    ua, ub = pa - p0, pb - p0
    umax = float(ua.abs().max())
    for n, (off, k) in slices.items():
        d = float((ua[off:off + k] - ub[off:off + k]).abs().max())
        ref = float(ua[off:off + k].abs().max())
        assert d <= tol * ref + 1e-4 * umax, f'{n}: updates differ by {d:.3e} (slice max {ref:.3e}, global max {umax:.3e})'


def test_sync_batchnorm_single_rank_is_the_identity_and_runs_collectives():
    """--sync-bn (GA/train.py:449-455): with convert_sync_batchnorm(comm) every train-mode BatchNorm all-reduces its statistics
    sums (forward: sum / sum of squares; backward: the two column sums entering dx) through RCCL inside the launch plans.  With
    one rank the collectives are the identity: the step must equal the plain step, and the plans must really contain them."""
    import imagenet_models_amd as A
    B = 8
    res = {}
    for tag in ('plain', 'sync'):
        m, O = _small_model()
        if tag == 'sync':
            comm = A.NativeComm()
            m.convert_sync_batchnorm(comm)
        opt = A.create_optimizer_v2(m, opt='sgd', lr=1e-2, momentum=0.9, weight_decay=0.05)
        x = O.gen_input(B, seed=2).cuda()
        y = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).cuda()
        step = A.TrainStep(m, opt, B, lam=-0.8)
        p0 = m.flat_state()['params'].clone()
        loss = step(x, y)
        torch.cuda.synchronize()
        res[tag] = (float(loss), m.flat_state()['params'].clone(), m.flat_state()['buffers'].clone())
        slices = m.flat_state()['slices']
        if tag == 'sync':
            nf = sum(1 for fn, _, lab in step.eng.fwd.calls if 'sync.' in str(lab))
            nb = sum(1 for fn, _, lab in step.eng.bwd.calls if 'sync.' in str(lab))
            assert nf >= 2 * 14 and nb >= 2 * 14, (nf, nb)      # 14 BatchNorm layers in GA-ConvNeXt, two sums each way
    assert abs(res['plain'][0] - res['sync'][0]) < 1e-6 * abs(res['plain'][0])
    _assert_same_update(res['plain'][1], res['sync'][1], p0, slices, 1e-3)
    assert float((res['plain'][2] - res['sync'][2]).abs().max()) <= 1e-5 * float(res['plain'][2].abs().max())   # running statistics
    comm.close()


def test_trainstep_buckets_through_torch_nccl_world1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'nccl_ws1_check.py')], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'NCCL_WS1_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize('opt_name', ['adamw', 'sgd'])
def test_optimizer_update_by_slices_is_bit_identical_to_the_whole_step(opt_name):
    """step_begin / step_range(lo, hi)* / step_end over an exact partition of the flat buffers == step() + zero_grad(),
    bit for bit (same kernels on the same values), for two consecutive steps (bias corrections / first-step momentum)."""
    import imagenet_models_amd as A
    from imagenet_models_amd.trainer import make_buckets
    res = {}
    for tag in ('whole', 'slices'):
        m, O = _small_model()
        opt = A.create_optimizer_v2(m, opt=opt_name, lr=1e-2, momentum=0.9, weight_decay=0.05)
        st = m.flat_state()
        buckets = make_buckets(st, m.grad_groups(), 50_000)
        assert len(buckets) >= 4 and sum(b - a for _, a, b in buckets) == st['total']
        for it in range(2):
            st['grads'].copy_(torch.randn(st['total'], generator=torch.Generator().manual_seed(it)).cuda())
            if tag == 'whole':
                opt.step()
                opt.zero_grad()
            else:
                opt.step_begin()
                for _, a, b in buckets:
                    opt.step_range(a, b)
                opt.step_end()
            torch.cuda.synchronize()
            assert float(st['grads'].abs().max()) == 0.0
        res[tag] = [st['params'].clone()] + ([opt.m.clone(), opt.v.clone()] if opt_name == 'adamw' else [opt.buf.clone()])
        assert opt.steps == 2
    for a, b in zip(res['whole'], res['slices']):
        assert torch.equal(a, b)


@pytest.mark.parametrize('transport', ['none', 'native'])
def test_optimizer_update_behind_each_bucket_equals_the_update_after_backward(transport):
    """TrainStep(overlap_optimizer=True): the optimizer update of a bucket (and the zeroing of its gradients) is
    issued on a side stream as soon as backward has completed the bucket -- alone, and behind the bucket's RCCL all-reduce
    (world size 1: the identity).  SGD, one step, the per-slice criterion of _assert_same_update (two backward passes never agree
    bit for bit: atomics); a second step then runs on the zeroed gradients."""
    import imagenet_models_amd as A
    B = 8
    res = {}
    for tag in ('after', 'behind'):
        m, O = _small_model()
        opt = A.create_optimizer_v2(m, opt='sgd', lr=1e-2, momentum=0.9, weight_decay=0.05)
        comm = A.NativeComm() if (transport == 'native' and tag == 'behind') else None
        step = A.TrainStep(m, opt, B, lam=-0.8, overlap_optimizer=(tag == 'behind'), comm=comm, force_buckets=comm is not None,
                           bucket_elems=50_000)
        assert step.overlap_opt == (tag == 'behind')
        if tag == 'behind':
            assert len(step.buckets) >= 4
        x = O.gen_input(B, seed=2).cuda()
        y = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).cuda()
        p0 = m.flat_state()['params'].clone()
        loss = float(step(x, y))
        torch.cuda.synchronize()
        p1 = m.flat_state()['params'].clone()
        assert float(m.flat_state()['grads'].abs().max()) == 0.0           # zeroed (bucket by bucket on the overlapped path)
        loss2 = float(step(x, y))
        torch.cuda.synchronize()
        res[tag] = (loss, p1, loss2)
        slices = m.flat_state()['slices']
        if comm is not None:
            comm.close()
    assert abs(res['after'][0] - res['behind'][0]) <= 1e-6 * abs(res['after'][0])
    _assert_same_update(res['after'][1], res['behind'][1], p0, slices, 1e-3)
    assert abs(res['after'][2] - res['behind'][2]) <= 1e-3 * abs(res['after'][2])
