import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops
torch.manual_seed(0)
for (Bn, H, W, Cc) in [(1, 14, 14, 96), (1, 14, 56, 32), (1, 14, 28, 64)]:
    x = (torch.rand(Bn, H, W, Cc) + 1.0)
    X = x.cuda().bfloat16()
    w = torch.zeros(Cc, 1, 7, 7); w[:, 0, 3, 3] = 1.0
    w49 = w.reshape(Cc, 49).t().contiguous().cuda()
    Y = torch.full_like(X, 777.0)
    P = ops.Plan(eager=True)
    P.dwconv7_fwd(X, w49, torch.zeros(Cc).cuda(), Y, Bn, H, W, Cc, ops.GA_BF16)
    torch.cuda.synchronize()
    bad = (Y != X)[0]          # [H][W][C]
    rows = sorted(set(bad.nonzero()[:, 0].tolist()))
    b1 = bad[1].cpu()          # one bad row: [W][C]
    offs = sorted(set(((c * Cc + ch) * 2) for c, ch in b1.nonzero().tolist()))
    print((Bn, H, W, Cc), 'bad rows', rows, 'n bad in row1', len(offs))
    print('  byte offsets (row 1):', [(o, hex(o & 0x3ff)) for o in offs[:24]])
    vals = Y[0, 1][b1.cuda()].float().unique().tolist()[:8]
    print('  bad values', vals)
    rule = sorted(o for o in range(0, W * Cc * 2, 2) if (o & 0x300) == 0x300 and (o & 0xc) == 4)
    print('  rule match:', rule == offs, len(rule))
