#!/usr/bin/env python3
"""Headline benchmark: images/sec of the bf16 TRAINING step on synthetic 3x224x224 at batch 256 per MI355X, one process
per GPU, weak scaling.  Default workload = BASELINE.json configs[1]: GA-ConvNeXt-T (`ga_convnext_tiny_768`);
`--model ga_CSWin_64_12211_tiny_224` = configs[2], `--model ga_convnext_base_1024` = configs[3]'s model.

A step = forward + GA loss (CE + lambda*KL, lambda=-0.8) + backward + bucketed gradient all-reduce (N>1, RCCL) + fused
AdamW, all on the libgaext HIP kernels.  Prints ONE JSON line (rank 0).

`python bench.py --gpus N` without a torch.distributed launcher starts the N one-GPU worker processes itself (fresh
children, before this process touches a GPU) and prints rank 0's line; under `python -m torch.distributed.run` it is one
of the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Extra objects:
  roofline     -- the dominant kernel family of the step (by summed device time), timed per launch with HIP events on the
                  launch stream in a separate instrumented step; achieved = its algorithmic FLOPs (2*M*N*K per GEMM launch)
                  / its device time; `step_frac` = whole-step img/s * GFLOP/img / peak.  `measured_peaks` = this box's own
                  numbers (libgaext bf16 GEMM 8192^3, a streaming copy) beside the vendor figures used for `peak`.
  cpu_baseline -- the oracle (CPU restatement of the reference path, kind "port") timed on this box's host cores on a
                  bounded sample (batch 8 train steps), N=1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = 'ga_convnext_tiny_768'
# fwd+bwd GFLOP per image (2 FLOP/MAC over conv + linear + bmm, backward = 2x forward): SURVEY.md section 8(d) / BASELINE.md 2
GFLOP_PER_IMG = {'ga_convnext_tiny_768': 34.52, 'ga_convnext_tiny': 34.52, 'ga_convnext_tiny_688': 32.73,
                 'ga_convnext_small_768': 60.76, 'ga_convnext_small': 60.76, 'ga_convnext_small_688': 58.87,
                 'ga_convnext_base_1024': 105.54, 'ga_convnext_base': 105.54, 'ga_convnext_base_976': 104.00,
                 'ga_CSWin_64_12211_tiny_224': 36.5, 'map_convnext_tiny': 30.37, 'map_convnext_small': 55.85,
                 # MAP-ViT (builder-defined, configs[4]): 3 x forward, forward = 2 FLOP/MAC over patch embedding, qkv / proj / MLP linears and
                 # the two attention products of every block (N = (img/16)^2 + 1 tokens) + the MAP head on (img/32)^2 tokens
                 'map_vit_base_patch16_384': 336.1, 'map_vit_base_patch16_224': 106.6, 'map_vit_small_patch16_224': 28.6,
                 # map_pit_s (map_pit.py:224-251): trunk 2.85 GMAC (conv_embedding 0.08, stages 0.67 / 1.30 / 0.79) + MAP head 0.28 GMAC
                 'map_pit_s': 18.8,
                 # plain ConvNeXt (global_pool='avg' branch of map_convnext.py): 4.47 / 8.70 GMAC forward
                 'convnext_tiny': 26.8, 'convnext_small': 52.2}
LABEL = {'ga_convnext_tiny_768': 'GA-ConvNeXt-T', 'ga_convnext_small_768': 'GA-ConvNeXt-S', 'ga_convnext_base_1024': 'GA-ConvNeXt-B',
         'ga_CSWin_64_12211_tiny_224': 'GA-CSWin-T (candidate config, SURVEY F3)', 'map_convnext_tiny': 'MAP-ConvNeXt-T',
         'map_convnext_small': 'MAP-ConvNeXt-S', 'map_vit_base_patch16_384': 'MAP-ViT-B/16 @ 384 (builder-defined composition)',
         'map_vit_base_patch16_224': 'MAP-ViT-B/16 @ 224', 'map_vit_small_patch16_224': 'MAP-ViT-S/16 @ 224', 'map_pit_s': 'MAP-PiT-S', 'convnext_tiny': 'ConvNeXt-T (plain head)', 'convnext_small': 'ConvNeXt-S (plain head)'}
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (the headline config is 256)')
    ap.add_argument('--model', default=MODEL)
    ap.add_argument('--opt', default='adamw')
    ap.add_argument('--math', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-times', action='store_true')
    ap.add_argument('--no-measured-peaks', action='store_true')
    ap.add_argument('--no-ddp-bb', action='store_true', help='no per-forward BatchNorm buffer broadcast (GA/train.py:283)')
    ap.add_argument('--kernel-table', default='', help='write the per-kernel-family table (json) here')
    ap.add_argument('--overlap-optimizer', action='store_true', help='optimizer update behind each gradient bucket instead of '
                    'after the whole backward (A/B: slower on one GPU)')
    ap.add_argument('--comm', default='torch', choices=['torch', 'native', 'native-bf16'],
                    help="gradient exchange for N > 1: torch.distributed's nccl (= RCCL) backend, or the library's own RCCL entry points "
                         '(ga_allreduce_bucket; native-bf16: bf16 wire)')
    ap.add_argument('--force-buckets', action='store_true', help='N = 1: still run the segmented backward + bucket reductions')
    return ap.parse_args()


def kernel_family(label):
    for key, fam in (('stem.conv+ln', 'stem_conv_ln'), ('.pad', 'pad_copy'), ('unpad', 'pad_copy'), ('.wg', 'gemm_tn(wgrad)'), ('wgrad', 'gemm_tn(wgrad)'), ('gram.', 'gemm/gram'),
                     ('.dww', 'dwconv7_wgrad'), ('.dwd', 'dwconv7'), ('.dw', 'dwconv7'),
                     ('attnb', 'stripe_attn_bwd' if 'stage' in label or 'gram_layer' in label else 'class_attn_bwd'), ('.attn', 'stripe_attn_fwd' if 'stage' in label or 'gram_layer' in label else 'class_attn'),
                     ('.lnb', 'layernorm_bwd'), ('ln0b', 'layernorm_bwd'), ('ln1b', 'layernorm_bwd'), ('ln2b', 'layernorm_bwd'),
                     ('.ln', 'layernorm_fwd'), ('prep.', 'weight_prep'), ('unf', 'weight_unfold'), ('bn', 'batchnorm'),
                     ('agg.', 'aggregate'), ('se.', 'squeeze_excite'), ('attn', 'class_attn'), ('zero', 'memset'), ('.dp', 'rowscale'),
                     ('ga_rowscale', 'rowscale'), ('loss', 'loss'), ('adamw', 'optimizer'), ('sgd', 'optimizer'),
                     ('lamb', 'optimizer'), ('pack', 'input_pack')):
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
        elif isinstance(d, L.AttnDesc):       # global attention: QK^T and PV forward; S, dP, dQ, dK, dV backward (2.5x)
            bwd = 'attnb' in label
            fam = 'global_attn_bwd' if bwd else 'global_attn_fwd'
            flops = 4.0 * d.B * d.H * d.N * d.N * d.hd * (2.5 if bwd else 1.0)
        elif isinstance(d, L.MlpDesc):        # fc1 + fc2
            fam, flops = 'mlp_fused_fwd', 4.0 * d.M * d.C * d.H
        elif isinstance(d, L.MlpBwdDesc):     # re-computed fc1 + dgrad2 + dgrad1
            fam, flops = 'mlp_fused_bwd', 6.0 * d.M * d.C * d.H
        else:
            fam = kernel_family(label)
        if per_call is not None:
            shape = (f'M{d.M} N{d.N} K{d.K} b{d.batch}' if hasattr(d, 'K') else f'M{d.M} C{d.C} H{d.H}') if d is not None and hasattr(d, 'M') else (f'B{d.B} N{d.N} H{d.H} hd{d.hd}' if isinstance(d, L.AttnDesc) else '')
            per_call.append(dict(label=f'{plan.name}:{label}', fam=fam, ms=round(ms, 4), shape=shape,
                                 tflops=round(flops / ms / 1e9, 1) if flops and ms > 0 else None))
        f = fams.setdefault(fam, dict(ms=0.0, flops=0.0, launches=0))
        f['ms'] += ms
        f['flops'] += flops
        f['launches'] += 1


def measured_peaks():
    """this box's own ceilings: libgaext's bf16 GEMM on 8192^3 random operands and a streaming pass of
    ga_rowscale over 1 GiB (read + write), both timed with HIP events over 10 launches after 3 warm-ups"""
    from imagenet_models_amd import ops
    n = 8192
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randn(n, n, device='cuda', generator=g).bfloat16()
    b = torch.randn(n, n, device='cuda', generator=g).bfloat16()
    c = torch.empty(n, n, device='cuda', dtype=torch.bfloat16)
    p = ops.Plan(name='peak.gemm')
    p.gemm(a, b, c, n, n, n, ops.GA_BF16)
    x = torch.randn(1 << 29, device='cuda', generator=g).bfloat16()
    y = torch.empty_like(x)
    s = torch.ones(1, device='cuda')
    q = ops.Plan(name='peak.copy')
    q.rowscale(x, s, y, x.numel(), x.numel(), ops.GA_BF16)
    out = {}
    for name, plan, work in (('mfma_bf16_tflops_gemm_8192', p, 2.0 * n ** 3 / 1e12), ('hbm_gbs_stream_1gib', q, 2.0 * x.numel() * 2 / 1e9)):
        for _ in range(3):
            plan.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(work * 10 / (e0.elapsed_time(e1) * 1e-3), 1)
    return out


def cpu_baseline(model_name, budget_s=10.0):
    """oracle (CPU restatement of the reference path) train step, fp32, batch 8, AdamW -- a reported baseline only"""
    step_kw = dict(lam=-0.8)
    if 'CSWin' in model_name:
        from oracle import ga_cswin_oracle as O
        from oracle.ga_convnext_oracle import adamw_step
        mod = 'oracle/ga_cswin_oracle.py'
    elif model_name.startswith('map_vit'):
        from oracle import map_vit_oracle as O
        from oracle.ga_convnext_oracle import adamw_step
        mod = 'oracle/map_vit_oracle.py'
        step_kw = dict(dec_lam=-0.8)
    elif model_name.startswith('convnext'):
        from oracle import convnext_oracle as O
        from oracle.ga_convnext_oracle import adamw_step
        mod = 'oracle/convnext_oracle.py'
        step_kw = dict()
    elif model_name.startswith('map_pit'):
        from oracle import map_pit_oracle as O
        from oracle.ga_convnext_oracle import adamw_step
        mod = 'oracle/map_pit_oracle.py'
        step_kw = dict(dec_lam=-0.8)
    elif model_name.startswith('map_'):
        from oracle import map_oracle as O
        from oracle.ga_convnext_oracle import adamw_step
        mod = 'oracle/map_oracle.py'
        step_kw = dict(dec_lam=-0.8)
    else:
        from oracle import ga_convnext_oracle as O
        adamw_step = O.adamw_step
        mod = 'oracle/ga_convnext_oracle.py'
    # the GPU box gives one GPU's share of the host (16 CPUs); more threads than that only oversubscribes
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))
    cfg = O.make_cfg(model_name)
    sd = O.fill_state(cfg)
    B = 8
    g = torch.Generator().manual_seed(42)
    img = cfg.get('img_size', cfg.get('image_size', 224))
    x = torch.randn(B, 3, img, img, generator=g)
    y = torch.randint(0, 1000, (B,), generator=g)
    m, v = {}, {}
    times = []
    t_all = time.time()
    step = 0
    while True:
        t0 = time.time()
        res = O.train_step_grads(sd, x, y, cfg, **step_kw)
        loss, outs, grads, stats = res if len(res) == 4 else (res[0], res[1], res[2], {})
        params = {n: sd[n] for n in grads}
        newp, m, v = adamw_step(params, grads, m, v, step + 1, 1e-3, (0.9, 0.999), 1e-8, 0.05)
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
                sample=f'{len(times)} timed train steps (fwd + GA loss + bwd + AdamW) of {model_name} at batch {B}, fp32, best '
                       f'step {best:.3f}s, {mod} on the host CPU')


def spawn_workers(a):
    """`bench.py --gpus N` outside a launcher: N fresh one-GPU children (this process makes no GPU call); rank 0's JSON line
    is passed through.  GA_DIST_BACKEND=gloo + fewer devices than ranks = rehearsal (ranks share a device)."""
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_workers(a)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert a.gpus == world, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    backend = 'none'
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

    import imagenet_models_amd as A
    torch.manual_seed(42 + rank)
    model = A.create_model(a.model, drop_path_rate=0.2, math_mode=a.math).cuda()
    model.train()
    if world > 1:  # DDP-style start: rank 0's parameters and buffers everywhere (GA/train.py:514)
        dist.broadcast(model.flat_state()['params'], 0)
        dist.broadcast(model.flat_state()['buffers'], 0)
    opt = A.create_optimizer_v2(model, opt=a.opt, lr=1e-3, weight_decay=0.05, momentum=0.9)
    comm = None
    if a.comm != 'torch' and (world > 1 or a.force_buckets):
        comm = A.NativeComm(wire='bf16' if a.comm == 'native-bf16' else 'fp32')
    step = A.TrainStep(model, opt, a.batch, lam=-0.8, loss='ce', broadcast_buffers=not a.no_ddp_bb, comm=comm,
                       force_buckets=a.force_buckets, overlap_optimizer=a.overlap_optimizer)
    g = torch.Generator().manual_seed(42 + rank)
    img = getattr(model, 'cfg', {}).get('img_size', 224)
    x = torch.randn(a.batch, 3, img, img, generator=g).cuda()
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
    gflop = GFLOP_PER_IMG.get(a.model)
    label = LABEL.get(a.model, a.model)

    out = dict(metric=f'images/sec (whole node) {label} 3x{img}x{img} {a.math} training step', value=round(value, 1),
               unit='images/sec', n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_per_step, 3),
               higher_is_better=True, scaling='weak', vs_baseline=None, dtype=a.math, data='synthetic',
               config=dict(workload=f'{a.model} train step (fwd + GA loss lam=-0.8 + bwd + all-reduce + {a.opt}), '
                                    f'batch {a.batch}/GPU, drop_path 0.2, synthetic 3x{img}x{img}',
                           global_batch=a.batch * world, parallelism=f'dp{world}', dist_backend=backend, grad_exchange=a.comm,
                           allreduce_buckets=len(step.buckets), gflop_per_img=gflop),
               loss=round(loss_val, 4), library=A._lib.config_string())

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
        for tj in ('r03_pmc_step_traffic.json', 'r02_pmc_step_traffic.json', 'r01_pmc_step_traffic.json'):
            tp = os.path.join(ROOT, 'profiles', tj)
            if name == 'gemm_nt' and a.model == MODEL and a.batch == 256 and a.math == 'bf16' and os.path.exists(tp):
                with open(tp) as fh:
                    g_ = json.load(fh)['gemm_nt']
                traffic = round((g_['fetch_mb'] + g_['write_mb']) * 1e6 / max(g_['launches'], 1))
                traffic_src = f"profiles/{tj}: {g_['launches']} gemm_nt dispatches per step, bytes per dispatch"
                break
        out['roofline'] = dict(bound='mfma', kernel=name, achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS,
                               unit='TFLOP/s', frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=traffic,
                               traffic_source=traffic_src,
                               launches_per_step=f['launches'], avg_launch_us=round(f['ms'] * 1e3 / launches, 2),
                               algorithmic_gflop_per_launch=round(f['flops'] / launches / 1e9, 3),
                               share_of_step_device_time=round(f['ms'] / total_ms, 3))
        if gflop:
            out['roofline'].update(step_achieved=round(per_gpu * gflop / 1e3, 1),
                                   step_frac=round(per_gpu * gflop / 1e3 / PEAK_BF16_TFLOPS, 4))
        if world == 1 and not a.no_measured_peaks:
            mp_ = measured_peaks()
            out['roofline']['measured_peaks'] = mp_
            out['roofline']['frac_of_measured_gemm'] = round(achieved / mp_['mfma_bf16_tflops_gemm_8192'], 4)
            if gflop:
                out['roofline']['step_frac_of_measured_gemm'] = round(per_gpu * gflop / 1e3 / mp_['mfma_bf16_tflops_gemm_8192'], 4)
        table = {k: dict(ms=round(v['ms'], 3), launches=v['launches'],
                         tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) if v['ms'] > 0 and v['flops'] else None)
                 for k, v in sorted(fams.items(), key=lambda kv: -kv[1]['ms'])}
        out['kernel_families_ms'] = {k: v['ms'] for k, v in table.items()}
        if a.kernel_table:
            with open(a.kernel_table, 'w') as fh:
                json.dump(dict(total_ms=total_ms, families=table,
                               top_calls=sorted(calls, key=lambda c: -c["ms"])), fh, indent=1)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(a.model)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
