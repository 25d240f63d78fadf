#!/bin/bash
# vectorised loads of the LDS-staged gram-pack backward: parity + kernel times on the MAP models
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_map_kernels_gpu.py tests/test_kernels_gpu.py tests/test_map_model_gpu.py tests/test_map_pit_gpu.py -m gpu -x -q -k "gram or map" > gpurun_out/r03/t_gp2.log 2>&1 || { tail -30 gpurun_out/r03/t_gp2.log; exit 1; }
tail -2 gpurun_out/r03/t_gp2.log
for m in map_convnext_tiny map_pit_s; do
python bench.py --model $m --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r03/kt_gp2_$m.json > gpurun_out/r03/gp2_$m.log 2>&1 || { tail -20 gpurun_out/r03/gp2_$m.log; exit 1; }
echo "$m $(tail -1 gpurun_out/r03/gp2_$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
python - <<PY
import json
d=json.load(open('gpurun_out/r03/kt_gp2_$m.json'))
for c in d['top_calls']:
    if 'pack' in c['label']: print('   ', c['label'], c['ms'])
PY
done
