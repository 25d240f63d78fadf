#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
B="--no-cpu-baseline --no-measured-peaks --no-kernel-times"
run() { tag=$1; shift; env "$@" python bench.py $B > gpurun_out/r03/sw_$tag.log 2>&1 || { tail -5 gpurun_out/r03/sw_$tag.log; return 1; }; python - <<PY
import json
d=json.loads(open('gpurun_out/r03/sw_$tag.log').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d.get('library'))
PY
}
for i in 1 2 3 4; do run def$i X=1 && run nomlp$i GA_FUSED_MLP=0 || exit 1; done
run nomlp_s1 GA_FUSED_MLP=0 GAEXT_FWD_SPLIT=1 && run nomlp_s3 GA_FUSED_MLP=0 GAEXT_FWD_SPLIT=3
