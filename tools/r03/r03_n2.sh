#!/bin/bash
# NEIGH2 on the ring form: kernel parity, CSWin model parity, A/B of the GA-CSWin step
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_cswin_kernels_gpu.py -m gpu -x -q -k "stride2" > gpurun_out/r03/t_n2.log 2>&1 || { tail -30 gpurun_out/r03/t_n2.log; exit 1; }
tail -2 gpurun_out/r03/t_n2.log
timeout -k 10 900 python -m pytest tests/test_cswin_model_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "not dwconv" > gpurun_out/r03/t_n2b.log 2>&1 || { tail -30 gpurun_out/r03/t_n2b.log; exit 1; }
tail -2 gpurun_out/r03/t_n2b.log
for v in 1 0 1; do
GAEXT_NT_R3_CONV3S2=$v python bench.py --model ga_CSWin_64_12211_tiny_224 --no-cpu-baseline --no-measured-peaks --no-kernel-times > gpurun_out/r03/n2_$v.log 2>&1 || { tail -20 gpurun_out/r03/n2_$v.log; exit 1; }
echo "conv3s2=$v $(tail -1 gpurun_out/r03/n2_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done
python bench.py --no-cpu-baseline --no-measured-peaks --no-kernel-times > gpurun_out/r03/n2_head.log 2>&1 || exit 1
echo "headline $(tail -1 gpurun_out/r03/n2_head.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
