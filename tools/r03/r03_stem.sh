#!/bin/bash
# fused stem conv + LayerNorm: parity, model parity, A/B of the headline
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "stem" > gpurun_out/r03/t_stem.log 2>&1 || { tail -30 gpurun_out/r03/t_stem.log; exit 1; }
tail -2 gpurun_out/r03/t_stem.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_convnext_gpu.py tests/test_map_model_gpu.py -m gpu -x -q > gpurun_out/r03/t_stemb.log 2>&1 || { tail -30 gpurun_out/r03/t_stemb.log; exit 1; }
tail -2 gpurun_out/r03/t_stemb.log
for v in 1 0 1 0; do
GAEXT_STEM_FUSED=$v python bench.py --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r03/kt_stem_$v.json > gpurun_out/r03/stem_$v.log 2>&1 || { tail -20 gpurun_out/r03/stem_$v.log; exit 1; }
echo "stem_fused=$v $(tail -1 gpurun_out/r03/stem_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done
