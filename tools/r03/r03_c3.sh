#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv3" > gpurun_out/r03/t_c3.log 2>&1 || { tail -40 gpurun_out/r03/t_c3.log; exit 1; }
tail -2 gpurun_out/r03/t_c3.log
timeout -k 10 120 python tools/r03/c3_bench.py || exit 1
timeout -k 10 120 python tools/r03/c3w_bench.py || exit 1
timeout -k 10 900 python -m pytest tests/test_cswin_model_gpu.py -m gpu -x -q > gpurun_out/r03/t_cs.log 2>&1 || { tail -40 gpurun_out/r03/t_cs.log; exit 1; }
tail -2 gpurun_out/r03/t_cs.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times --model ga_CSWin_64_12211_tiny_224"
for v in 1 0 1 0; do
GAEXT_CONV3_DIRECT=$v python bench.py $B > gpurun_out/r03/bench_cs_c$v.log 2>&1 || { tail -20 gpurun_out/r03/bench_cs_c$v.log; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r03/bench_cs_c$v.log').read().strip().splitlines()[-1]); print('direct$v', d['ms_per_step'], d['value'])
PY
done
