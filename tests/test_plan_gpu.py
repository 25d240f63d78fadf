"""ops.Plan stream semantics on the GPU: parallel regions (lanes > 0), the asynchronous lane (-1) with join_async /
async_mark, lane_signal / lane_wait.  Every check is a dependency that a missing wait would break: the buffers are large
enough (256 MB) that a consumer launched early would read unfinished data."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(n, *names):
    return {k: torch.zeros(n, device='cuda') for k in names}


def test_plan_lanes_async_and_signals():
    from imagenet_models_amd import ops
    n = 1 << 26
    x = torch.randn(n, device='cuda')
    b = _mk(n, 't', 'u1', 'u2', 'out', 'w', 'side', 'fin', 'a1', 'a2')
    p = ops.Plan(name='lanes')
    p.axpy_f32(b['t'], x, 1.0, n)                      # lane 0: t = x
    p.lane = 1
    p.axpy_f32(b['u1'], b['t'], 2.0, n)                # region: u1 = 5 t, u2 = 7 t, a2 = a1 = t (signal / wait)
    p.axpy_f32(b['u1'], b['t'], 3.0, n)
    p.axpy_f32(b['a1'], b['t'], 1.0, n)
    ev = p.lane_signal(1)
    p.lane = 2
    p.axpy_f32(b['u2'], b['t'], 7.0, n)
    p.lane_wait(2, ev)
    p.axpy_f32(b['a2'], b['a1'], 1.0, n)
    p.lane = 0
    p.axpy_f32(b['out'], b['u1'], 1.0, n)              # join: out = 12 t
    p.axpy_f32(b['out'], b['u2'], 1.0, n)
    p.lane = ops.ASYNC_LANE
    p.axpy_f32(b['w'], b['out'], 1.0, n)               # asynchronous lane: sees the finished `out`
    p.lane = 0
    p.async_mark('w')
    p.axpy_f32(b['side'], b['t'], 1.0, n)              # lane 0 goes on meanwhile
    p.join_async('w')
    p.axpy_f32(b['fin'], b['w'], 1.0, n)               # after the named point: fin = w
    p.lane = ops.ASYNC_LANE
    p.axpy_f32(b['fin'], b['side'], 1.0, n)            # (needs both lane-0 launches above) fin += t
    p.lane = 0
    p.join_async()
    p.axpy_f32(b['fin'], b['a2'], 1.0, n)              # fin = 12 t + t + t
    for rep in range(3):
        for v in b.values():
            v.zero_()
        p.run()
        torch.cuda.synchronize()
        assert torch.allclose(b['out'], 12.0 * x, rtol=1e-5, atol=1e-6), rep
        assert torch.equal(b['a2'], x), rep
        assert torch.allclose(b['fin'], 14.0 * x, rtol=1e-5, atol=1e-6), rep
    # the same plan replayed call by call on one stream (bench.py's instrumented pass, GAEXT_SYNC_DEBUG) gives the same
    for v in b.values():
        v.zero_()
    s = torch.cuda.current_stream().cuda_stream
    for fn, args, _ in p.calls:
        assert fn(*args, s) == 0
    torch.cuda.synchronize()
    assert torch.allclose(b['fin'], 14.0 * x, rtol=1e-5, atol=1e-6)


def test_plan_run_range_joins_at_range_end():
    from imagenet_models_amd import ops
    n = 1 << 26
    x = torch.randn(n, device='cuda')
    b = _mk(n, 'g', 'h')
    p = ops.Plan(name='ranges')
    p.lane = ops.ASYNC_LANE
    p.axpy_f32(b['g'], x, 3.0, n)
    p.lane = 0
    p.mark('cut')
    p.axpy_f32(b['h'], b['g'], 1.0, n)
    p.run_range(0, p.marks['cut'])                     # ends with a join: the next range sees g
    p.run_range(p.marks['cut'], len(p))
    torch.cuda.synchronize()
    assert torch.equal(b['h'], 3.0 * x)
