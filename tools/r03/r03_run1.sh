#!/bin/bash
# round-3 run 1: full GPU test suite on the refactored knob layer + baseline bench lines (same box) with and without the forward half-batch chains
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/t_full1.log 2>&1 || { tail -30 gpurun_out/r03/t_full1.log; exit 1; }
tail -3 gpurun_out/r03/t_full1.log
B="--no-cpu-baseline --no-measured-peaks"
python bench.py $B --kernel-table gpurun_out/r03/kt_base.json > gpurun_out/r03/bench_base.log 2>&1 || exit 1
GAEXT_FWD_SPLIT=1 python bench.py $B --kernel-table gpurun_out/r03/kt_split1.json > gpurun_out/r03/bench_split1.log 2>&1 || exit 1
python bench.py $B --no-kernel-times > gpurun_out/r03/bench_base2.log 2>&1 || exit 1
python - <<'PY'
import json
for f in ('bench_base','bench_split1','bench_base2'):
    d=json.loads(open(f'gpurun_out/r03/{f}.log').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d.get('library'))
PY
