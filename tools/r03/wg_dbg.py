import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops
torch.manual_seed(0)
for dt, gdt in ((torch.float32, ops.GA_F32), (torch.bfloat16, ops.GA_BF16)):
    for (R, N, K) in [(12, 64, 32), (12, 128, 32), (12, 64, 48), (12, 128, 48), (12, 64, 64), (12, 64, 40), (12, 64, 24)]:
        Y = torch.randn(R, N, device='cuda').to(dt); X = torch.randn(R, K, device='cuda').to(dt)
        big = torch.full((N * K + 4096,), 7.0, device='cuda')
        out = big[2048:2048 + N * K].view(N, K); out.zero_()
        db = torch.zeros(N, device='cuda')
        P = ops.Plan(eager=True)
        P.wgrad(Y, X, out, R, N, K, gdt, dbias=db)
        torch.cuda.synchronize()
        ref = Y.float().t() @ X.float()
        err = float((out - ref).abs().max() / ref.abs().max())
        lo, hi = big[:2048], big[2048 + N * K:]
        print(dt, (R, N, K), 'err', f'{err:.2e}', 'guard lo', int((lo != 7.0).sum()), 'hi', int((hi != 7.0).sum()))
