#!/bin/bash
# TN2 row-split sweep on GA-CSWin (partial-tile traffic of the 256 x 256-output weight gradients)
set -o pipefail
mkdir -p gpurun_out/r03
for v in 0 48 96 0 64 128; do
GAEXT_TN2_WGS=$v python bench.py --model ga_CSWin_64_12211_tiny_224 --no-cpu-baseline --no-measured-peaks --no-kernel-times > gpurun_out/r03/tn2_$v.log 2>&1 || { tail -20 gpurun_out/r03/tn2_$v.log; exit 1; }
echo "tn2_wgs=$v $(tail -1 gpurun_out/r03/tn2_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done
