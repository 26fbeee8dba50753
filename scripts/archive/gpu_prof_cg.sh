#!/bin/bash
# kernel trace of the two-phase bench (inner / frame launches of the tile kernel)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_cg
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cg -- python3 $R/scripts/model_bench.py cg > $O/prof_cg.log 2>&1 || { tail -5 $O/prof_cg.log; exit 14; }
cd $R
python scripts/prof_summary.py $O/prof_cg | head -10 | cut -c1-220
