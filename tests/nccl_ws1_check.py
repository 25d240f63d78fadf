"""Run by tests/test_comm_gpu.py as a child process: torch.distributed backend `nccl` (= RCCL on ROCm) with world_size 1 on the
test box's one GPU.  TrainStep is forced down the segmented-backward + bucketed all-reduce branch (the N > 1 path,
GA/train.py:514), so RCCL's stream really synchronises with the plan lanes; the step must equal the plain one.  Then the same
with the library's own communicator created while the process group exists (the id travels through broadcast_object_list)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29577', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    import imagenet_models_amd as A
    from oracle import ga_convnext_oracle as O
    cfg = O.make_cfg(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)
    sd = O.fill_state(cfg)
    B = 8
    x = O.gen_input(B, seed=2).cuda()
    y = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).cuda()
    out = {}
    for tag in ('plain', 'nccl', 'native'):
        m = A.GA_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], gram_embedding_gropus=cfg['gram_groups'],
                          dim_embed=cfg['dim_embed'], stage3_naggre=cfg['naggre'], gram_dim=cfg['gram_dim'], math_mode='fp32')
        m.load_state_dict(sd)
        m = m.cuda().train()
        opt = A.create_optimizer_v2(m, opt='sgd', lr=1e-2, momentum=0.9, weight_decay=0.05)
        kw = {}
        if tag != 'plain':
            kw = dict(force_buckets=True, bucket_elems=50_000, nan_guard=True)
        if tag == 'native':
            kw['comm'] = A.NativeComm()
        step = A.TrainStep(m, opt, B, lam=-0.8, **kw)
        if tag != 'plain':
            assert len(step.buckets) >= 4 and step.world == 1
        p0 = m.flat_state()['params'].clone()
        loss = step(x, y)
        torch.cuda.synchronize()
        out[tag] = (float(loss), m.flat_state()['params'].clone())
        slices = m.flat_state()['slices']
        if tag != 'plain':
            assert abs(float(step.last_loss_sum) - float(loss)) <= 1e-6 * abs(float(loss))
    from test_comm_gpu import _assert_same_update
    for tag in ('nccl', 'native'):
        assert abs(out[tag][0] - out['plain'][0]) < 1e-5 * abs(out['plain'][0]), (tag, out[tag][0], out['plain'][0])
        _assert_same_update(out['plain'][1], out[tag][1], p0, slices, 1e-3)
    print('NCCL_WS1_OK', {k: v[0] for k, v in out.items()}, flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
