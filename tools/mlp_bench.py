"""fused MLP bodies vs the ga_gemm pairs they replace, at the GA-ConvNeXt-T stage shapes (development aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagenet_models_amd as A
from imagenet_models_amd import ops
from imagenet_models_amd.ops import ACT_GELU


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dt = torch.bfloat16
    gd = ops.GA_BF16
    for C, M in ((96, 256 * 56 * 56), (192, 256 * 28 * 28)):
        H = 4 * C
        X = torch.randn(M, C, device='cuda').to(dt); R = torch.randn(M, C, device='cuda').to(dt)
        DY = torch.randn(M, C, device='cuda').to(dt)
        W1 = (torch.randn(H, C, device='cuda') * C ** -0.5).to(dt); W2 = (torch.randn(C, H, device='cuda') * H ** -0.5).to(dt)
        W1T, W2T = W1.T.contiguous(), W2.T.contiguous()
        b1 = torch.randn(H, device='cuda'); b2 = torch.randn(C, device='cuda')
        a = torch.empty(M, H, device='cuda', dtype=dt); g = torch.empty_like(a); dh = torch.empty_like(a)
        Y = torch.empty(M, C, device='cuda', dtype=dt); DX = torch.empty_like(Y)
        rs = torch.ones(256, device='cuda')
        p = ops.Plan(eager=True)
        def unfused_fwd():
            p.gemm(X, W1, a, M, H, C, gd, bias=b1, act=ACT_GELU, C2=g, c2_mode=2)
            p.gemm(a, W2, Y, M, C, H, gd, bias=b2, rowscale=rs, rows_per_scale=M // 256, R=R, ldr=C)
        def fused_fwd():
            p.mlp_fwd(X, W1, b1, W2, b2, Y, M, C, gd, R=R, rowscale=rs, rows_per_scale=M // 256)
        def unfused_bwd():
            p.gemm(DY, W2T, dh, M, H, C, gd, H=g, ldh=H, h_is_deriv=True)
            p.gemm(dh, W1T, DX, M, C, H, gd)
        def fused_bwd():
            p.mlp_bwd(X, DY, W1, b1, W2T, W1T, a, dh, DX, M, C, gd)
        gf = 2 * 2 * M * C * H / 1e9
        for name, fn, k in (('fwd unfused', unfused_fwd, 1), ('fwd fused', fused_fwd, 1), ('bwd unfused', unfused_bwd, 1), ('bwd fused', fused_bwd, 1.5)):
            ms = timeit(fn)
            print(f'C={C} M={M} {name}: {ms:.3f} ms  ({gf * k / ms:.0f} TFLOP/s)', flush=True)


if __name__ == '__main__':
    main()
