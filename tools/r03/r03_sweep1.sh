#!/bin/bash
# host-side scheduling switches re-measured with the faster depthwise kernels (same box, interleaved with the default)
set -o pipefail
mkdir -p gpurun_out/r03
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
run() { tag=$1; shift; env "$@" python bench.py $B > gpurun_out/r03/sw_$tag.log 2>&1 || { tail -5 gpurun_out/r03/sw_$tag.log; return 1; }; python - <<PY
import json
d=json.loads(open('gpurun_out/r03/sw_$tag.log').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d.get('library'))
PY
}
run def1 X=1 && run split1 GAEXT_FWD_SPLIT=1 && run split3 GAEXT_FWD_SPLIT=3 && run def2 X=1 && run mlp192 GA_FUSED_MLP=96,192 && run nomlp GA_FUSED_MLP=0 && run rows56 GAEXT_DW_RS_ROWS=56 && run rows14 GAEXT_DW_RS_ROWS=14 && run def3 X=1 && run skew0 GAEXT_FWD_SKEW=0 && run skew1 GAEXT_FWD_SKEW=1 && run skew2 GAEXT_FWD_SKEW=2 && run dww2 GAEXT_DWW_RS=2 && run def4 X=1
