#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_dbg.so timeout -k 10 400 python tools/r3_probe.py > gpurun_out/r03/r3_probe.log 2>&1 || { tail -20 gpurun_out/r03/r3_probe.log; exit 1; }
cat gpurun_out/r03/r3_probe.log
EW_WHAT=ln timeout -k 10 300 python tools/ew_bench.py > gpurun_out/r03/ew_ln.log 2>&1 || { tail -20 gpurun_out/r03/ew_ln.log; exit 1; }
cat gpurun_out/r03/ew_ln.log
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm or rowscale or ln_" > gpurun_out/r03/t_ln.log 2>&1 || { tail -30 gpurun_out/r03/t_ln.log; exit 1; }
tail -3 gpurun_out/r03/t_ln.log
