#!/bin/bash
# round 3: register-sliding depthwise form: parity, then A/B against the LDS-staged forms (same box)
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv" > gpurun_out/r03/t_dw.log 2>&1 || { tail -40 gpurun_out/r03/t_dw.log; exit 1; }
tail -2 gpurun_out/r03/t_dw.log
for r in 0 7 14 28 56; do
echo "== DW_RS=1 rows=$r"
GAEXT_DW_RS_ROWS=$r EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "fwd\|bwd-data" || exit 1
done
echo "== DW_RS=0"
GAEXT_DW_RS=0 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "dwconv" || exit 1
