#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_comm_gpu.py tests/test_grad_marks_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py tests/test_cli_gpu.py -m gpu -x -q > gpurun_out/r03/t_opt.log 2>&1 || { tail -40 gpurun_out/r03/t_opt.log; exit 1; }
tail -2 gpurun_out/r03/t_opt.log
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
python bench.py $B > gpurun_out/r03/bench_o1.log 2>&1 || { tail -20 gpurun_out/r03/bench_o1.log; exit 1; }
python bench.py $B --no-overlap-optimizer > gpurun_out/r03/bench_o0.log 2>&1 || exit 1
python bench.py $B > gpurun_out/r03/bench_o1b.log 2>&1 || exit 1
python - <<'PY'
import json
for f in ('bench_o1','bench_o0','bench_o1b'):
    d=json.loads(open(f'gpurun_out/r03/{f}.log').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d.get('library'))
PY
