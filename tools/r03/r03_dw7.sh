#!/bin/bash
# evidence file for profiles/: the depthwise forms side by side (same box), the timing variants, PMC passes of the register-sliding kernel
set -o pipefail
O=gpurun_out/r03/dwev
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{
echo "# depthwise 7x7, bf16, B = 256 (tools/ew_bench.py; GB/s = tensors read / written once; TF/s = 2 x 49 MAC per output element)"
echo "== register-sliding forms (default; backward-weight forced on every map with DWW_RS=2)"
GAEXT_DWW_RS=2 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep dwconv
echo "== default dispatch (backward-weight: register-sliding on 56x56 only)"
EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep dwconv
echo "== LDS-staged forms of rounds 1-2 (DW_RS=0 DWW_RS=0)"
GAEXT_DW_RS=0 GAEXT_DWW_RS=0 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep dwconv
for r in 14 28 56; do echo "== register-sliding forward / backward-data, rows per segment $r"; GAEXT_DW_RS_ROWS=$r GB_STAGES=0 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "fwd\|bwd-data"; done
for v in 2 3 7; do
echo "== timing variant DW_DBG=$v (results wrong; 1 = no DMA, 2 = no stores, 4 = no ring reads), stage 0"
GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_dw$v.so GB_STAGES=0 EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "fwd\|bwd-data"
done
} > $O/dwconv_ab.txt 2>&1 || { tail -5 $O/dwconv_ab.txt; exit 1; }
export EW_WHAT=dw GB_STAGES=0 GB_ITERS=4
run() { n=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/p$n -o p -- python3 tools/ew_bench.py > $O/p$n.log 2>&1 || { tail -5 $O/p$n.log; return 1; }; python3 tools/pmc_summary.py $O/p$n/p_results.db dwconv7 >> $O/dwconv_pmc.txt; rm -rf $O/p$n; }
echo "# PMC passes over the stage-0 depthwise launches (256 x 56 x 56 x 96, bf16): rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/ew_bench.py" > $O/dwconv_pmc.txt
run 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE && run 2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && run 3 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU && run 4 SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA && run 5 FETCH_SIZE && run 6 WRITE_SIZE
cat $O/dwconv_ab.txt | head -50
