#!/bin/bash
# odd-width variants: parity tests, then the bench line + kernel table of ga_convnext_tiny_688
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_small_kernels_gpu.py tests/test_model_gpu.py tests/test_all_entrypoints_gpu.py -m gpu -x -q -k "688 or 976 or odd or pad_groups" > gpurun_out/r03/t_688.log 2>&1 || { tail -30 gpurun_out/r03/t_688.log; exit 1; }
tail -3 gpurun_out/r03/t_688.log
python bench.py --model ga_convnext_tiny_688 --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r03/kt_688e.json > gpurun_out/r03/bench_688e.log 2>&1 || { tail -20 gpurun_out/r03/bench_688e.log; exit 1; }
tail -1 gpurun_out/r03/bench_688e.log | cut -c1-300
