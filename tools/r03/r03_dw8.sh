#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv" > gpurun_out/r03/t_dw.log 2>&1 || { tail -40 gpurun_out/r03/t_dw.log; exit 1; }
tail -2 gpurun_out/r03/t_dw.log
for v in 2 1 0; do echo "== DWW_RS=$v"; GAEXT_DWW_RS=$v EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "bwd-weight" || exit 1; done
