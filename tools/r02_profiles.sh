#!/bin/bash
# round-2 profile collection on the GPU box (run from the repo root through gpurun): kernel-trace stats of the bench command
# (default lanes and serial), per-kernel tables of the three model families, PMC traffic of one step.
set -o pipefail
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-kernel-times --no-measured-peaks"
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/trace -o bench -- python3 bench.py --steps 20 --warmup 5 $B > gpurun_out/r02/trace.log 2>&1 || exit 1
echo trace done
export GAEXT_ASYNC_WGRAD=0 GAEXT_HEAD_STREAMS=1 GAEXT_FWD_SPLIT=1 GAEXT_PAR_BRANCH=0
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/trace_serial -o bench -- python3 bench.py --steps 20 --warmup 5 $B > gpurun_out/r02/trace_serial.log 2>&1 || exit 1
unset GAEXT_ASYNC_WGRAD GAEXT_HEAD_STREAMS GAEXT_FWD_SPLIT GAEXT_PAR_BRANCH
echo serial done
python3 bench.py --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r02/ktable_convnext.json > gpurun_out/r02/bench_convnext.log 2>&1 || exit 1
python3 bench.py --model ga_CSWin_64_12211_tiny_224 --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r02/ktable_cswin.json > gpurun_out/r02/bench_cswin.log 2>&1 || exit 1
python3 bench.py --model map_convnext_tiny --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r02/ktable_map.json > gpurun_out/r02/bench_map.log 2>&1 || exit 1
python3 bench.py --model map_vit_base_patch16_384 --batch 128 --steps 30 --warmup 8 --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r02/ktable_mapvit.json > gpurun_out/r02/bench_mapvit.log 2>&1 || exit 1
python3 bench.py --model map_pit_s --no-cpu-baseline --no-measured-peaks --kernel-table gpurun_out/r02/ktable_mappit.json > gpurun_out/r02/bench_mappit.log 2>&1 || exit 1
echo tables done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02/pmcF -o f -- python3 bench.py --steps 2 --warmup 1 $B > gpurun_out/r02/pmcF.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02/pmcW -o w -- python3 bench.py --steps 2 --warmup 1 $B > gpurun_out/r02/pmcW.log 2>&1 || exit 1
echo pmc done
ls -R gpurun_out/r02 | head -50
