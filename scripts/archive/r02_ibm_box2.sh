#!/bin/bash
# forced box beside the window launch: priority of the window's stream, forcing as one workgroup or as a launch chain
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/r02_ibm_box_bench2.log
for t in "bg_priority=0" "bg_priority=1" "bg_priority=1,ibm_step_chain=1"; do
  echo "# LBM_TUNE=$t" | tee -a $O/r02_ibm_box_bench2.log
  LBM_IBM_SIZES=2048,16384 LBM_TUNE=$t timeout -k 10 300 python scripts/model_bench.py ibm 2>/dev/null | cut -c1-170 | tee -a $O/r02_ibm_box_bench2.log
done
LBM_IBM_SIZES=16384 LBM_TUNE=bg_priority=1 bash scripts/r02_ibm_trace.sh > $O/r02_ibm_trace_16384.txt 2>&1
tail -45 $O/r02_ibm_trace_16384.txt | cut -c1-140
