#!/bin/bash
# slab walk of the streaming kernels: parity of the element-wise / norm kernels, micro-benchmark, headline
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "not dwconv and not gemm" > gpurun_out/r03/t_slab.log 2>&1 || { tail -30 gpurun_out/r03/t_slab.log; exit 1; }
tail -2 gpurun_out/r03/t_slab.log
EW_WHAT=ln,bn python tools/ew_bench.py > gpurun_out/r03/ew_slab.txt 2>&1 || { tail gpurun_out/r03/ew_slab.txt; exit 1; }
echo "== slab walk"; cat gpurun_out/r03/ew_slab.txt
if [ -f imagenet-models_amd/csrc/libgaext_base.so ]; then
GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_base.so EW_WHAT=ln,bn python tools/ew_bench.py > gpurun_out/r03/ew_stride.txt 2>&1
echo "== grid-stride walk (previous build)"; cat gpurun_out/r03/ew_stride.txt
fi
for v in 1 0 1 0; do
L=""; if [ $v = 0 ]; then L=$PWD/imagenet-models_amd/csrc/libgaext_base.so; fi
GAEXT_LIB=$L python bench.py --no-cpu-baseline --kernel-table gpurun_out/r03/kt_slab_$v.json > gpurun_out/r03/slab_$v.log 2>&1 || { tail -20 gpurun_out/r03/slab_$v.log; exit 1; }
echo "slab=$v $(tail -1 gpurun_out/r03/slab_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['measured_peaks'])")"
done
