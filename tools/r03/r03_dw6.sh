#!/bin/bash
# half-batch (forward chains) depthwise launches: rows per segment
set -o pipefail
for r in 0 7 14 28 56; do
echo "== B=128 rows=$r"
GB_BATCH=128 GAEXT_DW_RS_ROWS=$r EW_WHAT=dw timeout -k 10 200 python tools/ew_bench.py 2>&1 | grep "fwd" || exit 1
done
