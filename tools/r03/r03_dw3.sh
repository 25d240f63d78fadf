#!/bin/bash
# parity of the register-sliding depthwise kernel, then its timing variants (results wrong): what each phase costs
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv" > gpurun_out/r03/t_dw.log 2>&1 || { tail -40 gpurun_out/r03/t_dw.log; exit 1; }
tail -2 gpurun_out/r03/t_dw.log
for r in 0 14 56; do
echo "== release rows=$r"
GAEXT_DW_RS_ROWS=$r EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "fwd\|bwd-data" || exit 1
done
for v in 2 3 7; do
echo "== variant $v"
GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_dw$v.so GB_STAGES=0 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "fwd\|bwd-data" || exit 1
done
