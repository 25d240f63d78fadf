#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/r03/t_kern.log 2>&1 || { tail -40 gpurun_out/r03/t_kern.log; exit 1; }
tail -2 gpurun_out/r03/t_kern.log
B="--no-cpu-baseline --no-measured-peaks"
python bench.py $B --kernel-table gpurun_out/r03/kt_r3b.json > gpurun_out/r03/bench_r3c.log 2>&1 || { tail -20 gpurun_out/r03/bench_r3c.log; exit 1; }
for m in ga_CSWin_64_12211_tiny_224 map_convnext_tiny map_pit_s convnext_tiny; do
python bench.py --model $m $B --kernel-table gpurun_out/r03/kt_$m.json > gpurun_out/r03/bench_$m.log 2>&1 || { tail -20 gpurun_out/r03/bench_$m.log; exit 1; }
GAEXT_NT_R3=0 python bench.py --model $m $B --no-kernel-times > gpurun_out/r03/bench_${m}_r3off.log 2>&1 || exit 1
done
python bench.py --model map_vit_base_patch16_384 --batch 128 --steps 30 --warmup 8 $B --kernel-table gpurun_out/r03/kt_mapvit.json > gpurun_out/r03/bench_mapvit.log 2>&1 || exit 1
GAEXT_NT_R3=0 python bench.py --model map_vit_base_patch16_384 --batch 128 --steps 30 --warmup 8 $B --no-kernel-times > gpurun_out/r03/bench_mapvit_r3off.log 2>&1 || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_*.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'], d.get('library'))
    except Exception as e: print(f, 'ERR', e)
PY
