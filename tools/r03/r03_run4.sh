#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ring3 or (lds_dma_form and r3)" > gpurun_out/r03/t_r3.log 2>&1 || { tail -40 gpurun_out/r03/t_r3.log; exit 1; }
tail -2 gpurun_out/r03/t_r3.log
GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_dbg.so timeout -k 10 400 python tools/r3_probe.py > gpurun_out/r03/r3_probe2.log 2>&1 || { tail -20 gpurun_out/r03/r3_probe2.log; exit 1; }
cat gpurun_out/r03/r3_probe2.log
R3_EXTRA=1 timeout -k 10 500 python tools/r3_ab.py gpurun_out/r03/r3_ab2.json > gpurun_out/r03/r3_ab2.log 2>&1 || { tail -20 gpurun_out/r03/r3_ab2.log; exit 1; }
cat gpurun_out/r03/r3_ab2.log
