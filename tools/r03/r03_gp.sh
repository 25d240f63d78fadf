#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_map_kernels_gpu.py -m gpu -x -q -k "gram" > gpurun_out/r03/t_gp.log 2>&1 || { tail -40 gpurun_out/r03/t_gp.log; exit 1; }
tail -2 gpurun_out/r03/t_gp.log
timeout -k 10 900 python -m pytest tests/test_map_model_gpu.py tests/test_cswin_model_gpu.py tests/test_cswin_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "not trajectory" > gpurun_out/r03/t_gp2.log 2>&1 || { tail -40 gpurun_out/r03/t_gp2.log; exit 1; }
tail -2 gpurun_out/r03/t_gp2.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
for m in map_convnext_tiny ga_CSWin_64_12211_tiny_224; do for v in 1 0 1; do
GAEXT_GRAM_LDS=$v python bench.py $B --model $m > gpurun_out/r03/gp_${m}_$v.log 2>&1 || { tail -20 gpurun_out/r03/gp_${m}_$v.log; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r03/gp_${m}_$v.log').read().strip().splitlines()[-1]); print('$m lds=$v', d['ms_per_step'], d['value'])
PY
done; done
