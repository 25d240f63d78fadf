#!/bin/bash
# round-3 profile collection on the GPU box (run from the repo root through gpurun): kernel-trace of the bench command (default lanes and
# serial), kernel tables of the families, PMC traffic of one step (two passes), MFMA-busy pass over the GEMM micro-benchmark.
set -o pipefail
O=gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-kernel-times --no-measured-peaks"
rocprofv3 --kernel-trace -d $O/trace -o bench -- python3 bench.py --steps 20 --warmup 5 $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
echo trace done
export GAEXT_ASYNC_WGRAD=0 GAEXT_HEAD_STREAMS=1 GAEXT_FWD_SPLIT=1 GAEXT_PAR_BRANCH=0
rocprofv3 --kernel-trace -d $O/trace_serial -o bench -- python3 bench.py --steps 20 --warmup 5 $B > $O/trace_serial.log 2>&1 || exit 1
unset GAEXT_ASYNC_WGRAD GAEXT_HEAD_STREAMS GAEXT_FWD_SPLIT GAEXT_PAR_BRANCH
echo serial done
python3 tools/kernel_stats_from_db.py $O/trace/bench_results.db $O/bench_kernel_stats.csv 25 || exit 1
python3 tools/kernel_stats_from_db.py $O/trace_serial/bench_results.db $O/bench_kernel_stats_serial.csv 25 || exit 1
python3 tools/step_timeline.py $O/trace/bench_results.db $O/timeline.json > $O/timeline.txt || exit 1
python3 bench.py --no-cpu-baseline --kernel-table $O/ktable_convnext.json > $O/bench_convnext.log 2>&1 || exit 1
python3 bench.py --model ga_CSWin_64_12211_tiny_224 --no-cpu-baseline --no-measured-peaks --kernel-table $O/ktable_cswin.json > $O/bench_cswin.log 2>&1 || exit 1
python3 bench.py --model map_convnext_tiny --no-cpu-baseline --no-measured-peaks --kernel-table $O/ktable_map.json > $O/bench_map.log 2>&1 || exit 1
python3 bench.py --model map_vit_base_patch16_384 --batch 128 --steps 30 --warmup 8 --no-cpu-baseline --no-measured-peaks --kernel-table $O/ktable_mapvit.json > $O/bench_mapvit.log 2>&1 || exit 1
python3 bench.py --model map_pit_s --no-cpu-baseline --no-measured-peaks --kernel-table $O/ktable_mappit.json > $O/bench_mappit.log 2>&1 || exit 1
python3 bench.py --model ga_convnext_tiny_688 --no-cpu-baseline --no-measured-peaks --kernel-table $O/ktable_688.json > $O/bench_688.log 2>&1 || exit 1
python3 bench.py --model convnext_tiny --no-cpu-baseline --no-measured-peaks --kernel-table $O/ktable_cnx.json > $O/bench_cnx.log 2>&1 || exit 1
echo tables done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcF -o f -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmcF.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcW -o w -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmcW.log 2>&1 || exit 1
python3 tools/pmc_step_traffic.py $O/pmcF/f_results.db $O/pmcW/w_results.db $O/pmc_step_traffic || exit 1
echo pmc traffic done
R3_EXTRA=1 R3_ROUNDS=3 R3_ITERS=3 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcM -o m -- python3 tools/r3_ab.py > $O/pmcM.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/pmcM/m_results.db gemm_nt > $O/pmc_mfma_summary.txt || exit 1
rm -rf $O/trace $O/trace_serial $O/pmcF $O/pmcW $O/pmcM
echo all done; ls $O
