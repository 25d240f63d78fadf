#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_comm_gpu.py tests/test_grad_marks_gpu.py -m gpu -x -q -s > gpurun_out/r03/t_comm.log 2>&1 || { tail -60 gpurun_out/r03/t_comm.log; exit 1; }
tail -4 gpurun_out/r03/t_comm.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
python bench.py $B --force-buckets --comm native > gpurun_out/r03/bench_fb_native.log 2>&1 || { tail -20 gpurun_out/r03/bench_fb_native.log; exit 1; }
python bench.py $B > gpurun_out/r03/bench_plain9.log 2>&1 || exit 1
python - <<'PY'
import json
for f in ('bench_fb_native','bench_plain9'):
    d=json.loads(open(f'gpurun_out/r03/{f}.log').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['config'].get('allreduce_buckets'), d['config'].get('grad_exchange'))
PY
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_cswin_model_gpu.py tests/test_map_model_gpu.py tests/test_map_pit_gpu.py tests/test_map_vit_gpu.py tests/test_convnext_gpu.py -m gpu -q -x -k "bf16" > gpurun_out/r03/t_bf16.log 2>&1 || { tail -30 gpurun_out/r03/t_bf16.log; exit 1; }
tail -3 gpurun_out/r03/t_bf16.log
