"""Run by tests/test_ddp_gpu.py under torch.distributed.run (2 ranks, gloo, ranks may share one GPU): the bucketed,
backward-overlapped gradient all-reduce of TrainStep must give the gradients that a plain "local backward -> ONE
all-reduce(mean) of the whole flat gradient buffer" gives, and leave every rank with identical parameters
(N-process gradient = mean of the per-shard gradients)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
    dist.init_process_group(os.environ.get('GA_DIST_BACKEND', 'gloo'), init_method='env://')
    import imagenet_models_amd as A
    from oracle import ga_convnext_oracle as O
    family = os.environ.get('GA_DDP_FAMILY', '')            # '' = the narrow GA-ConvNeXt; else a registered model name (full size, B = 2)
    if not family:
        cfg = O.make_cfg(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)
        sd = O.fill_state(cfg)
        B, NC, img, kind = 8, 40, 224, 0
    else:
        torch.manual_seed(7)                                 # the same initial state on every rank
        fkw = dict(head_drop=0.0, head_attn_drop=0.0) if family.startswith('map_') else {}   # (dropout masks: random per model instance)
        m0 = A.create_model(family, math_mode='fp32', **fkw)
        with torch.no_grad():                                # (layer-scale parameters start at 1e-6: any non-trivial values do)
            for p_ in m0.parameters():
                if p_.dim() <= 1 and float(p_.abs().max()) < 1e-3:
                    p_.fill_(0.1)
        sd = {k: v.clone() for k, v in m0.state_dict().items()}
        B, NC, img = 2, m0.num_classes, getattr(m0, 'cfg', {}).get('img_size', 224)
        kind = 0                                              # TrainStep / forward_loss pick the family's loss (GA / MAP) themselves
        del m0
    g = torch.Generator().manual_seed(100 + rank)          # every rank sees different data
    x = torch.randn(B, 3, img, img, generator=g).cuda()
    y = torch.randint(0, NC, (B,), generator=g).cuda()

    def make():
        if family:
            m = A.create_model(family, math_mode='fp32', **fkw)
        else:
            m = A.GA_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], gram_embedding_gropus=cfg['gram_groups'],
                              dim_embed=cfg['dim_embed'], stage3_naggre=cfg['naggre'], gram_dim=cfg['gram_dim'], math_mode='fp32')
        m.load_state_dict(sd)
        m = m.cuda().train()
        return m, A.create_optimizer_v2(m, opt='adamw', lr=1e-2, weight_decay=0.05)

    # (1) the product path, gradients only: the loop of TrainStep.__call__ -- backward in segments, each finished slice of
    # the flat gradient buffer all-reduced asynchronously -- without the optimizer
    m1, o1 = make()
    step = A.TrainStep(m1, o1, B, lam=-0.8)
    assert step.world == world and len(step.buckets) >= 2
    covered = sorted((a, b) for _, a, b in step.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == m1.flat_state()['total']
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1)), covered   # exact partition
    eng = step.eng
    eng.forward_loss(x, y, -0.8, 0, 0.0, 1.0 / world)
    works, pos = [], 0
    for mark, a, b in step.buckets:
        stop = len(eng.bwd.calls) if mark == 'end' else eng.bwd.marks[mark]
        if stop > pos:
            eng.bwd.run_range(pos, stop)
            pos = stop
        works.append(dist.all_reduce(step.flat_g[a:b], async_op=True))
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    g_ddp = step.flat_g.clone()

    # (2) the reference: whole local backward, then ONE all-reduce of the whole buffer
    m2, _ = make()
    e2 = m2.engine(B, True)
    e2.forward_loss(x, y, -0.8, 0, 0.0, 1.0 / world)
    e2.bwd.run()
    g_ref = m2.flat_state()['grads']
    dist.all_reduce(g_ref)
    torch.cuda.synchronize()
    err = 0.0
    gmax = float(g_ref.abs().max())
    for n, (off, k) in m1.flat_state()['slices'].items():
        d = float((g_ddp[off:off + k] - g_ref[off:off + k]).abs().max())
        ref = float(g_ref[off:off + k].abs().max())
        # fp32 atomics: two runs of the same backward differ by ~1e-5 of the values summed (slices whose true gradient is
        # zero -- biases in front of a train-mode BatchNorm -- hold only that noise); a slice reduced before it was
        # complete, or twice, is off by its own magnitude
        assert d <= 1e-3 * ref + 1e-4 * gmax, f'rank {rank}: {n}: bucketed differs by {d:.3e}, slice max {ref:.3e}'
        err = max(err, d / (ref + 0.1 * gmax))

    # (3) a real step (bucketed reduction + AdamW) leaves every rank with the same parameters
    step.opt.zero_grad()
    step(x, y)
    p_ddp = m1.flat_state()['params']
    chk = torch.stack([p_ddp.double().sum(), p_ddp.double().abs().sum()]).cpu()
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(gathered[0], t) for t in gathered), gathered
    # (4) the same step with the optimizer update (and the gradient zeroing) issued behind each bucket's all-reduce
    # (TrainStep(overlap_optimizer=True)): every rank again ends with the same parameters, the gradients are zero, and the update is
    # the one of (3) up to the atomics noise of a second backward pass (AdamW: compared where the gradient is not noise)
    m3, o3 = make()
    step3 = A.TrainStep(m3, o3, B, lam=-0.8, overlap_optimizer=True)
    assert step3.overlap_opt and step3.world == world
    step3(x, y)
    torch.cuda.synchronize()
    p3 = m3.flat_state()['params']
    assert float(m3.flat_state()['grads'].abs().max()) == 0.0
    chk3 = torch.stack([p3.double().sum(), p3.double().abs().sum()]).cpu()
    gathered = [torch.zeros_like(chk3) for _ in range(world)]
    dist.all_gather(gathered, chk3)
    assert all(torch.equal(gathered[0], t) for t in gathered), gathered
    big = g_ref.abs() > 1e-3 * gmax                      # elements whose gradient is well above the noise: same AdamW step
    dpar = float((p3 - p_ddp)[big].abs().max())
    assert dpar <= 2e-3 * 1e-2 + 1e-6, dpar              # lr 1e-2: the two updates agree to a fraction of one step
    if rank == 0:
        print(f'DDP_CHECK_OK {family or "ga_convnext(narrow)"} max rel grad diff {err:.2e} buckets {[(m, b - a) for m, a, b in step.buckets]}', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
