#!/usr/bin/env python3
"""Validation script (timm-style surface of MAP/validate.py / GA/train.py:validate): sums the head logits in fp32,
top-1/top-5 with bit-exact lowest-index tie-break, on the libgaext HIP kernels.  Synthetic data only.

  python validate.py --synthetic --model ga_convnext_tiny_768 -b 256 --batches 10 [--checkpoint x.pth.tar] [--results-file r.json]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

parser = argparse.ArgumentParser()
parser.add_argument('data', nargs='?', default='')
parser.add_argument('--model', '-m', default='ga_convnext_tiny_768')
parser.add_argument('-b', '--batch-size', type=int, default=256)
parser.add_argument('--batches', type=int, default=10)
parser.add_argument('--num-classes', type=int, default=None)
parser.add_argument('--checkpoint', default='')
parser.add_argument('--fp32', action='store_true')
parser.add_argument('--synthetic', action='store_true')
parser.add_argument('--results-file', default='')


def main():
    args = parser.parse_args()
    if not args.synthetic:
        raise SystemExit('validate.py: only --synthetic data is shipped')
    import imagenet_models_amd as A
    model = A.create_model(args.model, num_classes=args.num_classes, checkpoint_path=args.checkpoint,
                           math_mode='fp32' if args.fp32 else 'bf16').cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(0)
    n = c1 = c5 = 0
    with torch.no_grad():
        img = getattr(model, 'cfg', {}).get('img_size', 224)
        model(torch.randn(args.batch_size, 3, img, img, device='cuda', generator=g))  # warm-up (MAP/validate.py:240)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.batches):
            x = torch.randn(args.batch_size, 3, img, img, device='cuda', generator=g)
            y = torch.randint(0, model.num_classes, (args.batch_size,), device='cuda', generator=g)
            _, idx = A.heads_topk(model(x), 5)
            a1, a5 = A.accuracy_from_topk(idx, y, (1, 5))
            c1 += float(a1) * x.size(0) / 100
            c5 += float(a5) * x.size(0) / 100
            n += x.size(0)
        torch.cuda.synchronize()
        dt = time.time() - t0
    res = dict(model=args.model, top1=round(100 * c1 / n, 4), top5=round(100 * c5 / n, 4),
               param_count=round(sum(p.numel() for p in model.parameters()) / 1e6, 2), img_per_sec=round(n / dt, 1))
    print(f' * Acc@1 {res["top1"]:.3f} Acc@5 {res["top5"]:.3f}  ({res["img_per_sec"]:.1f} img/s)')
    print('--result\n' + json.dumps(res, indent=4))
    if args.results_file:
        json.dump(res, open(args.results_file, 'w'), indent=4)


if __name__ == '__main__':
    main()
