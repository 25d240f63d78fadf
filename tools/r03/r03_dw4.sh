#!/bin/bash
# register-sliding depthwise form inside the models: parity, then the step time with / without it (same box)
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_convnext_gpu.py tests/test_map_model_gpu.py -m gpu -x -q > gpurun_out/r03/t_dwm.log 2>&1 || { tail -40 gpurun_out/r03/t_dwm.log; exit 1; }
tail -2 gpurun_out/r03/t_dwm.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
python bench.py $B > gpurun_out/r03/bench_dw1.log 2>&1 || { tail -20 gpurun_out/r03/bench_dw1.log; exit 1; }
GAEXT_DW_RS=0 python bench.py $B > gpurun_out/r03/bench_dw0.log 2>&1 || exit 1
python bench.py $B > gpurun_out/r03/bench_dw1b.log 2>&1 || exit 1
python - <<'PY'
import json
for f in ('bench_dw1','bench_dw0','bench_dw1b'):
    d=json.loads(open(f'gpurun_out/r03/{f}.log').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d.get('library'))
PY
