#!/bin/bash
# workgroup count of the LayerNorm backward with parameter gradients (persistent form)
set -o pipefail
mkdir -p gpurun_out/r03
for v in 1024 512 256 1024 512; do
echo "== LN_BWD_WGS=$v"
GAEXT_LN_BWD_WGS=$v EW_WHAT=ln python tools/ew_bench.py 2>&1 | grep "affine"
done
