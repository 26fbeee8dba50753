#!/bin/bash
# rocprofv3 evidence for the secondary configs (KBC, two-phase, IBM): kernel trace + stats, then the
# HBM counters each in its own pass (never combined with a trace).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/mprof_kt $O/mprof_fetch $O/mprof_write
timeout -k 10 400 python3 $R/scripts/model_bench.py kbc cg ibm 2>&1 | grep -v amdgpu.ids | tee $O/model_bench.log || exit 12
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mprof_kt -- python3 $R/scripts/model_bench.py kbc cg ibm > $O/mprof_kt.log 2>&1 || exit 13
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/mprof_fetch -- python3 $R/scripts/model_bench.py cg > $O/mprof_fetch.log 2>&1 || exit 14
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/mprof_write -- python3 $R/scripts/model_bench.py cg > $O/mprof_write.log 2>&1 || exit 15
cd $R
python scripts/prof_summary.py $O/mprof_kt $O/mprof_fetch $O/mprof_write | tee $O/mprof_summary.txt
