#!/bin/bash
# the full GPU suite as the driver runs it
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t_full.log 2>&1; rc=$?
tail -5 gpurun_out/r03/t_full.log
exit $rc
