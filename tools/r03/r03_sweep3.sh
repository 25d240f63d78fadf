#!/bin/bash
# wgrad workgroup budget and other library knobs re-measured in the step (same box, default interleaved)
set -o pipefail
mkdir -p gpurun_out/r03
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
run() { tag=$1; shift; env "$@" python bench.py $B > gpurun_out/r03/sw_$tag.log 2>&1 || { tail -5 gpurun_out/r03/sw_$tag.log; return 1; }; python - <<PY
import json
d=json.loads(open('gpurun_out/r03/sw_$tag.log').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d.get('library'))
PY
}
run def1 X=1 && run wgs256 GAEXT_TN2_WGS=256 && run wgs128 GAEXT_TN2_WGS=128 && run wgs224 GAEXT_TN2_WGS=224 && run def2 X=1 && run heads3 GAEXT_HEAD_STREAMS=3 && run heads8 GAEXT_HEAD_STREAMS=8 && run nopar GAEXT_PAR_BRANCH=0 && run def3 X=1 && run dwmfma GAEXT_DW_RS=0 GAEXT_DW_MFMA=2 && run ntpp0 GAEXT_NT_PP=0 && run def4 X=1
