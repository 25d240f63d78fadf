#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times --model ga_CSWin_64_12211_tiny_224"
run() { tag=$1; shift; env "$@" python bench.py $B > gpurun_out/r03/cs_$tag.log 2>&1 || { tail -5 gpurun_out/r03/cs_$tag.log; return 1; }; python - <<PY
import json
d=json.loads(open('gpurun_out/r03/cs_$tag.log').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d.get('library'))
PY
}
run def1 X=1 && run wgs96 GAEXT_TN2_WGS=96 && run wgs64 GAEXT_TN2_WGS=64 && run wgs128 GAEXT_TN2_WGS=128 && run wgs256 GAEXT_TN2_WGS=256 && run def2 X=1 && run partmin GAEXT_TN2_PART_MIN=131072 && run notn2 GAEXT_TN2=0 && run heads2 GAEXT_HEAD_STREAMS=2 && run def3 X=1
