#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-kernel-times --no-measured-peaks"
rocprofv3 --kernel-trace -d gpurun_out/r03/trace -o bench -- python3 bench.py --steps 6 --warmup 3 $B > gpurun_out/r03/trace.log 2>&1 || { tail -20 gpurun_out/r03/trace.log; exit 1; }
DB=$(ls gpurun_out/r03/trace/*/*.db | head -1)
python3 tools/step_timeline.py $DB gpurun_out/r03/timeline.json
ls -la $DB
GA_GRADCHECK_REPORT=1 python -m pytest tests/test_model_gpu.py tests/test_cswin_model_gpu.py tests/test_map_model_gpu.py tests/test_map_pit_gpu.py tests/test_map_vit_gpu.py tests/test_convnext_gpu.py -m gpu -q -s -k "bf16" > gpurun_out/r03/t_bf16_report.log 2>&1; echo "bf16 report rc=$?"
grep -h "gradients:\|WOULD FAIL" gpurun_out/r03/t_bf16_report.log | cut -c1-400
tail -3 gpurun_out/r03/t_bf16_report.log
