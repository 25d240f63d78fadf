#!/bin/bash
# PMC passes over the stage-0 depthwise launches (register-sliding form): where do the cycles go
set -o pipefail
O=gpurun_out/r03/dwpmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export EW_WHAT=dw GB_STAGES=0 GB_ITERS=4 GAEXT_DW_RS_ROWS=${ROWS:-56}
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/avail_sq.txt
run() { n=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/p$n -o p -- python3 tools/ew_bench.py > $O/p$n.log 2>&1 || { tail -5 $O/p$n.log; return 1; }; python3 tools/pmc_summary.py $O/p$n/p_results.db dwconv7_rs > $O/p$n.txt; cat $O/p$n.txt; rm -rf $O/p$n; }
run 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE &&
run 2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA &&
run 3 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU &&
run 4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM &&
run 5 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH &&
run 6 FETCH_SIZE WRITE_SIZE
