#!/bin/bash
# tests + headline bench + rocprofv3 (kernel trace, then PMC passes each on its own)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest_gpu.log
rc=${PIPESTATUS[0]}; echo "pytest rc=$rc"
if [ "$rc" != "0" ] && [ "$rc" != "1" ]; then exit 11; fi
timeout -k 10 600 python bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_n1.json || exit 12
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 $R/bench.py --no-cpu-baseline > $O/prof_kt.log 2>&1 || exit 13
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_fetch.log 2>&1 || exit 14
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_write.log 2>&1 || exit 15
cd $R
python scripts/prof_summary.py $O/prof_kt $O/prof_fetch $O/prof_write | tee $O/prof_summary.txt
find $O/prof_kt -name "*stats*.csv" | head -3
