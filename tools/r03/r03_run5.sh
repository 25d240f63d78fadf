#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ring3 or (lds_dma_form and r3)" > gpurun_out/r03/t_r3.log 2>&1 || { tail -40 gpurun_out/r03/t_r3.log; exit 1; }
tail -2 gpurun_out/r03/t_r3.log
R3_EXTRA=0 R3_ARMS=0,-1,40,80,240 R3_ROUNDS=5 timeout -k 10 500 python tools/r3_ab.py gpurun_out/r03/r3_ab3.json > gpurun_out/r03/r3_ab3.log 2>&1 || { tail -20 gpurun_out/r03/r3_ab3.log; exit 1; }
cat gpurun_out/r03/r3_ab3.log
