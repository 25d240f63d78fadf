#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_map_pit_gpu.py tests/test_map_vit_gpu.py -m gpu -x -q > gpurun_out/r03/t_vit.log 2>&1 || { tail -40 gpurun_out/r03/t_vit.log; exit 1; }
tail -2 gpurun_out/r03/t_vit.log
timeout -k 10 600 python -m pytest tests/test_large_batch_gpu.py tests/test_grad_marks_gpu.py -m gpu -x -q -k "pit or vit" > gpurun_out/r03/t_vit2.log 2>&1 || { tail -40 gpurun_out/r03/t_vit2.log; exit 1; }
tail -2 gpurun_out/r03/t_vit2.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
for v in 2 1 2 1; do
GAEXT_FWD_SPLIT=$v python bench.py $B --model map_pit_s > gpurun_out/r03/bench_pit_s$v.log 2>&1 || { tail -20 gpurun_out/r03/bench_pit_s$v.log; exit 1; }
GAEXT_FWD_SPLIT=$v python bench.py $B --model map_vit_base_patch16_384 --batch 128 --steps 30 --warmup 8 > gpurun_out/r03/bench_vit_s$v.log 2>&1 || { tail -20 gpurun_out/r03/bench_vit_s$v.log; exit 1; }
python - <<PY
import json
for m in ('pit','vit'):
    d=json.loads(open('gpurun_out/r03/bench_%s_s$v.log' % m).read().strip().splitlines()[-1]); print(m, 'split$v', d['ms_per_step'], d['value'])
PY
done
