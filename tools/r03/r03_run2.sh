#!/bin/bash
# round-3 run 2: parity of the 3-slot ring NT form, then its A/B against the default dispatch
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ring3 or (lds_dma_form and r3)" > gpurun_out/r03/t_r3.log 2>&1 || { tail -40 gpurun_out/r03/t_r3.log; exit 1; }
tail -3 gpurun_out/r03/t_r3.log
timeout -k 10 500 python tools/r3_ab.py gpurun_out/r03/r3_ab.json > gpurun_out/r03/r3_ab.log 2>&1 || { tail -20 gpurun_out/r03/r3_ab.log; exit 1; }
cat gpurun_out/r03/r3_ab.log
