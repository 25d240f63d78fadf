#!/bin/bash
set -o pipefail
O=gpurun_out/r03/cs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-kernel-times --no-measured-peaks"
rocprofv3 --kernel-trace -d $O/trace -o bench -- python3 bench.py --model ga_CSWin_64_12211_tiny_224 --steps 8 --warmup 4 $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 tools/step_timeline.py $O/trace/bench_results.db $O/timeline.json > $O/timeline.txt || exit 1
python3 tools/kernel_stats_from_db.py $O/trace/bench_results.db $O/stats.csv 8 || exit 1
rm -rf $O/trace
head -40 $O/timeline.txt
