#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
B="--no-cpu-baseline --no-measured-peaks"
python bench.py $B --kernel-table gpurun_out/r03/kt_r3.json > gpurun_out/r03/bench_r3.log 2>&1 || { tail -20 gpurun_out/r03/bench_r3.log; exit 1; }
GAEXT_NT_R3=0 python bench.py $B --no-kernel-times > gpurun_out/r03/bench_r3off.log 2>&1 || exit 1
GAEXT_FWD_SPLIT=1 python bench.py $B --kernel-table gpurun_out/r03/kt_r3_split1.json > gpurun_out/r03/bench_r3_split1.log 2>&1 || exit 1
python bench.py $B --no-kernel-times > gpurun_out/r03/bench_r3b.log 2>&1 || exit 1
for i in 1 2; do
EW_WHAT=ln GB_ITERS=30 python tools/ew_bench.py > gpurun_out/r03/ew_lnA$i.log 2>&1 || exit 1
GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_b.so EW_WHAT=ln GB_ITERS=30 python tools/ew_bench.py > gpurun_out/r03/ew_lnB$i.log 2>&1 || exit 1
done
python - <<'PY'
import json
for f in ('bench_r3','bench_r3off','bench_r3_split1','bench_r3b'):
    d=json.loads(open(f'gpurun_out/r03/{f}.log').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d.get('library'))
PY
paste gpurun_out/r03/ew_lnA1.log gpurun_out/r03/ew_lnB1.log gpurun_out/r03/ew_lnA2.log gpurun_out/r03/ew_lnB2.log | awk -F'\t' '{printf "%s |", substr($1,1,52); for(i=2;i<=NF;i++){split($i,a," "); n=split($i,b," "); printf " %s", b[n-3]" "b[n-2]} print ""}'
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "t768 or v2_train" > gpurun_out/r03/t_model.log 2>&1 || { tail -30 gpurun_out/r03/t_model.log; exit 1; }
tail -3 gpurun_out/r03/t_model.log
