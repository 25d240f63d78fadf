#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv" > gpurun_out/r03/t_dw.log 2>&1 || { tail -40 gpurun_out/r03/t_dw.log; exit 1; }
tail -2 gpurun_out/r03/t_dw.log
for r in 0 14 56; do
echo "== DWW_RS=1 rows=$r"
GAEXT_DWW_RS_ROWS=$r EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "bwd-weight" || exit 1
done
echo "== DWW_RS=0"
GAEXT_DWW_RS=0 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "bwd-weight" || exit 1
