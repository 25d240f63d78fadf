#!/bin/bash
# direct first conv of the GA-CSWin stem: parity, model parity, A/B
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_cswin_kernels_gpu.py -m gpu -x -q -k "stem_first or stride2 or conv3" > gpurun_out/r03/t_c0.log 2>&1 || { tail -30 gpurun_out/r03/t_c0.log; exit 1; }
tail -2 gpurun_out/r03/t_c0.log
timeout -k 10 900 python -m pytest tests/test_cswin_model_gpu.py -m gpu -x -q > gpurun_out/r03/t_c0b.log 2>&1 || { tail -30 gpurun_out/r03/t_c0b.log; exit 1; }
tail -2 gpurun_out/r03/t_c0b.log
for v in 1 0 1; do
GAEXT_CONV0_DIRECT=$v python bench.py --model ga_CSWin_64_12211_tiny_224 --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r03/kt_c0w_$v.json > gpurun_out/r03/c0w_$v.log 2>&1 || { tail -20 gpurun_out/r03/c0w_$v.log; exit 1; }
echo "conv0_direct=$v $(tail -1 gpurun_out/r03/c0w_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done
