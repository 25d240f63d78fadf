#!/bin/bash
# GA-CSWin's grouped gram_contraction as one dense block-diagonal product: parity + A/B
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_small_kernels_gpu.py tests/test_cswin_model_gpu.py tests/test_grad_marks_gpu.py -m gpu -x -q > gpurun_out/r03/t_gc.log 2>&1 || { tail -30 gpurun_out/r03/t_gc.log; exit 1; }
tail -2 gpurun_out/r03/t_gc.log
for v in 1 0 1 0; do
GAEXT_GC_DENSE=$v python bench.py --model ga_CSWin_64_12211_tiny_224 --no-cpu-baseline --no-measured-peaks --no-kernel-times > gpurun_out/r03/gc_$v.log 2>&1 || { tail -20 gpurun_out/r03/gc_$v.log; exit 1; }
echo "gc_dense=$v $(tail -1 gpurun_out/r03/gc_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done
