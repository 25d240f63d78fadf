#!/usr/bin/env python3
"""Headline benchmark: images/sec of the GA-ConvNeXt-T (ga_convnext_tiny_768) bf16 TRAINING step on synthetic
3x224x224 at batch 256 per MI355X (BASELINE.json configs[1]); one process per GPU, weak scaling.

A step = forward + GA loss (CE + lambda*KL, lambda=-0.8) + backward + gradient all-reduce (N>1) + fused AdamW, all
on the libgaext HIP kernels.  Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel family of the step (by summed device time), timed per launch with HIP
                  events on the launch stream in a separate instrumented step; achieved = its algorithmic FLOPs
                  (2*M*N*K per GEMM launch) / its device time; `step_frac` = whole-step img/s * 34.52 GFLOP / peak.
  cpu_baseline -- the oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores
                  on a bounded sample (batch 8 train steps), N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = 'ga_convnext_tiny_768'
GFLOP_PER_IMG = 34.52          # fwd+bwd, SURVEY.md section 8(d) / BASELINE.md section 2
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (the headline config is 256)')
    ap.add_argument('--model', default=MODEL)
    ap.add_argument('--opt', default='adamw')
    ap.add_argument('--math', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-times', action='store_true')
    ap.add_argument('--kernel-table', default='', help='write the per-kernel-family table (json) here')
    return ap.parse_args()


def kernel_family(label):
    for key, fam in (('.wg', 'gemm_tn(wgrad)'), ('wgrad', 'gemm_tn(wgrad)'), ('gram.', 'gemm/gram'),
                     ('.dww', 'dwconv7_wgrad'), ('.dwd', 'dwconv7'), ('.dw', 'dwconv7'),
                     ('.lnb', 'layernorm_bwd'), ('ln1b', 'layernorm_bwd'), ('ln2b', 'layernorm_bwd'), ('.ln', 'layernorm_fwd'),
                     ('prep.', 'weight_prep'), ('unf', 'weight_unfold'), ('bn', 'batchnorm'), ('agg.', 'aggregate'),
                     ('se.', 'squeeze_excite'), ('attn', 'class_attn'), ('zero', 'memset'), ('.dp', 'rowscale'),
                     ('ga_rowscale', 'rowscale'), ('loss', 'loss'), ('adamw', 'optimizer'), ('sgd', 'optimizer'),
                     ('lamb', 'optimizer')):
        if key in label:
            return fam
    return 'other'


def time_plan_calls(plan, fams, per_call=None):
    """run the plan once with a HIP event pair around EVERY launch (on the launch stream); accumulate per family"""
    from imagenet_models_amd import _lib as L
    s = torch.cuda.current_stream().cuda_stream
    evs = []
    for fn, args, label in plan.calls:
        if getattr(fn, '__module__', '') == 'imagenet_models_amd.ops':   # join / mark pseudo calls of the lanes: not launches
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(fn(*args, s), label)
        e1.record()
        evs.append((label, fn, args, e0, e1))
    torch.cuda.synchronize()
    for label, fn, args, e0, e1 in evs:
        ms = e0.elapsed_time(e1)
        d = getattr(args[0], '_obj', None) if args else None
        flops = 0.0
        if isinstance(d, L.WgradDesc):
            fam, flops = 'gemm_tn(wgrad)', 2.0 * d.M * d.N * d.K * d.batch
        elif isinstance(d, L.GemmDesc):
            fam, flops = 'gemm_nt', 2.0 * d.M * d.N * d.K * d.batch
        else:
            fam = kernel_family(label)
        if per_call is not None:
            shape = f'M{d.M} N{d.N} K{d.K} b{d.batch}' if d is not None and hasattr(d, 'M') else ''
            per_call.append(dict(label=f'{plan.name}:{label}', fam=fam, ms=round(ms, 4), shape=shape,
                                 tflops=round(flops / ms / 1e9, 1) if flops and ms > 0 else None))
        f = fams.setdefault(fam, dict(ms=0.0, flops=0.0, launches=0))
        f['ms'] += ms
        f['flops'] += flops
        f['launches'] += 1


def cpu_baseline(budget_s=10.0):
    """oracle (CPU restatement of the reference path) train step, fp32, batch 8, AdamW -- a reported baseline only"""
    from oracle import ga_convnext_oracle as O
    # the GPU box gives one GPU's share of the host (16 CPUs); more threads than that only oversubscribes
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))
    cfg = O.make_cfg(MODEL)
    sd = O.fill_state(cfg)
    B = 8
    g = torch.Generator().manual_seed(42)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (B,), generator=g)
    m, v = {}, {}
    times = []
    t_all = time.time()
    step = 0
    while True:
        t0 = time.time()
        loss, outs, grads, stats = O.train_step_grads(sd, x, y, cfg, lam=-0.8)
        params = {n: sd[n] for n in grads}
        newp, m, v = O.adamw_step(params, grads, m, v, step + 1, 1e-3, (0.9, 0.999), 1e-8, 0.05)
        sd.update(newp)
        sd.update({k: t for k, t in stats.items()})
        dt = time.time() - t0
        step += 1
        if step > 1:
            times.append(dt)
        if (time.time() - t_all > budget_s and len(times) >= 1) or len(times) >= 20:
            break
    best = min(times)
    return dict(value=round(B / best, 2), unit='images/sec', cores=torch.get_num_threads(), kind='port',
                sample=f'{len(times)} timed train steps (fwd + GA loss + bwd + AdamW) of {MODEL} at batch {B}, fp32, best step '
                       f'{best:.3f}s, oracle/ga_convnext_oracle.py on the host CPU')


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # one process per GPU over RCCL; GA_DIST_BACKEND=gloo + fewer GPUs than ranks is a rehearsal mode only
        # (ranks share a device, the all-reduce goes through the host): it exercises the N > 1 code, not its speed
        backend = os.environ.get('GA_DIST_BACKEND', 'nccl')
        local = local % max(1, torch.cuda.device_count()) if backend != 'nccl' else local
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', init_method='env://', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, init_method='env://')
    else:
        torch.cuda.set_device(0)
    assert a.gpus == world, f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1'

    import imagenet_models_amd as A
    torch.manual_seed(42 + rank)
    model = A.create_model(a.model, drop_path_rate=0.2, math_mode=a.math).cuda()
    model.train()
    if world > 1:  # DDP-style start: rank 0's parameters and buffers everywhere (GA/train.py:514)
        dist.broadcast(model.flat_state()['params'], 0)
        for b in model.buffers():
            dist.broadcast(b, 0)
    opt = A.create_optimizer_v2(model, opt=a.opt, lr=1e-3, weight_decay=0.05, momentum=0.9)
    step = A.TrainStep(model, opt, a.batch, lam=-0.8, loss='ce')
    g = torch.Generator().manual_seed(42 + rank)
    x = torch.randn(a.batch, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, model.num_classes, (a.batch,), generator=g).cuda()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = step(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(x, y)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / a.steps * 1e3
    value = a.batch * world * a.steps / elapsed
    loss_val = float(loss)

    out = dict(metric='images/sec (whole node) GA-ConvNeXt-T 3x224x224 bf16 training step', value=round(value, 1),
               unit='images/sec', n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_per_step, 3),
               higher_is_better=True, scaling='weak', vs_baseline=None, dtype=a.math, data='synthetic',
               config=dict(workload=f'{a.model} train step (fwd + GA loss lam=-0.8 + bwd + all-reduce + {a.opt}), '
                                    f'batch {a.batch}/GPU, drop_path 0.2, synthetic 3x224x224',
                           global_batch=a.batch * world, parallelism=f'dp{world}'),
               loss=round(loss_val, 4))

    if rank == 0 and not a.no_kernel_times:
        eng = step.eng
        fams, calls = {}, []
        eng.set_input(x)
        time_plan_calls(eng.prep, fams, calls)
        time_plan_calls(eng.fwd, fams, calls)
        time_plan_calls(eng.loss_plan, fams, calls)
        time_plan_calls(eng.bwd, fams, calls)
        time_plan_calls(opt.plan, fams, calls)
        opt.zero_grad()
        total_ms = sum(f['ms'] for f in fams.values())
        dom = max(fams.items(), key=lambda kv: kv[1]['ms'])
        name, f = dom
        launches = max(f['launches'], 1)
        achieved = f['flops'] / (f['ms'] * 1e-3) / 1e12 if f['ms'] > 0 else 0.0
        per_gpu = value / world
        # HBM bytes per launch of the family from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE over one step of THIS
        # workload, tools/pmc_step_traffic.py): a PMC run cannot be nested inside the timed bench
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, 'profiles', 'r01_pmc_step_traffic.json')
        if name == 'gemm_nt' and a.model == MODEL and a.batch == 256 and a.math == 'bf16' and os.path.exists(tj):
            with open(tj) as fh:
                g_ = json.load(fh)['gemm_nt']
            traffic = round((g_['fetch_mb'] + g_['write_mb']) * 1e6 / max(g_['launches'], 1))
            traffic_src = f"profiles/r01_pmc_step_traffic.json: {g_['launches']} gemm_nt dispatches per step, bytes per dispatch"
        out['roofline'] = dict(bound='mfma', kernel=name, achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS,
                               unit='TFLOP/s', frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=traffic,
                               traffic_source=traffic_src,
                               launches_per_step=f['launches'], avg_launch_us=round(f['ms'] * 1e3 / launches, 2),
                               algorithmic_gflop_per_launch=round(f['flops'] / launches / 1e9, 3),
                               share_of_step_device_time=round(f['ms'] / total_ms, 3),
                               step_achieved=round(per_gpu * GFLOP_PER_IMG / 1e3, 1),
                               step_frac=round(per_gpu * GFLOP_PER_IMG / 1e3 / PEAK_BF16_TFLOPS, 4))
        table = {k: dict(ms=round(v['ms'], 3), launches=v['launches'],
                         tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) if v['ms'] > 0 and v['flops'] else None)
                 for k, v in sorted(fams.items(), key=lambda kv: -kv[1]['ms'])}
        out['kernel_families_ms'] = {k: v['ms'] for k, v in table.items()}
        if a.kernel_table:
            with open(a.kernel_table, 'w') as fh:
                json.dump(dict(total_ms=total_ms, families=table,
                               top_calls=sorted(calls, key=lambda c: -c["ms"])), fh, indent=1)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
