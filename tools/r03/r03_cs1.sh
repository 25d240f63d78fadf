#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_cswin_model_gpu.py tests/test_cswin_kernels_gpu.py tests/test_fp64_truth_gpu.py -m gpu -x -q > gpurun_out/r03/t_cs.log 2>&1 || { tail -40 gpurun_out/r03/t_cs.log; exit 1; }
tail -2 gpurun_out/r03/t_cs.log
timeout -k 10 600 python -m pytest tests/test_large_batch_gpu.py tests/test_grad_marks_gpu.py -m gpu -x -q -k "CSWin" > gpurun_out/r03/t_cs2.log 2>&1 || { tail -40 gpurun_out/r03/t_cs2.log; exit 1; }
tail -2 gpurun_out/r03/t_cs2.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times --model ga_CSWin_64_12211_tiny_224"
for v in 2 1 3 2 1; do
GAEXT_FWD_SPLIT=$v python bench.py $B > gpurun_out/r03/bench_cs_s$v.log 2>&1 || { tail -20 gpurun_out/r03/bench_cs_s$v.log; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r03/bench_cs_s$v.log').read().strip().splitlines()[-1]); print('split$v', d['ms_per_step'], d['value'])
PY
done
