#!/bin/bash
# kernel trace of the immersed-boundary bench (what sits on the critical path of a step)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_ibm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ibm -- python3 $R/scripts/model_bench.py ibm > $O/prof_ibm.log 2>&1 || { tail -5 $O/prof_ibm.log; exit 14; }
cd $R
python scripts/prof_summary.py $O/prof_ibm | head -14 | cut -c1-200
