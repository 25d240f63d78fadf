import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from imagenet_models_amd import ops
dt = ops.GA_BF16
def timeit(plan, iters=20):
    for _ in range(3): plan.run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]
for (M, N, K) in [(50176, 256, 1024), (50176, 1024, 256), (50176, 768, 256), (50176, 256, 256), (12544, 512, 2048), (12544, 2048, 512), (200704, 128, 512), (200704, 512, 128)]:
    a = (torch.randn(M, K, device='cuda') * .5).bfloat16(); b = (torch.randn(N, K, device='cuda') * .5).bfloat16(); c = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    r = (torch.randn(M, N, device='cuda')).bfloat16(); bias = torch.randn(N, device='cuda')
    res = []
    for pp in ('0', None):
        if pp is None: os.environ.pop('GAEXT_NT_PP', None)
        else: os.environ['GAEXT_NT_PP'] = pp
        p = ops.Plan(); p.gemm(a, b, c, M, N, K, dt, bias=bias)
        t1 = timeit(p)
        p = ops.Plan(); p.gemm(a, b, c, M, N, K, dt, bias=bias, R=r, ldr=N)
        t2 = timeit(p)
        res.append((t1, t2))
    fl = 2.0 * M * N * K / 1e9
    print(f'M{M} N{N} K{K}: plain off {res[0][0]:.3f} ms ({fl/res[0][0]:.0f} TF) default {res[1][0]:.3f} ({fl/res[1][0]:.0f}) | fc2 off {res[0][1]:.3f} default {res[1][1]:.3f}', flush=True)
