#!/bin/bash
# FETCH_SIZE calibration on known-bytes kernels (scripts/calib/fetch_calib.hip); output -> gpurun_out/r03_fetch_calib.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/calib_a $O/calib_b
$R/scripts/calib/fetch_calib > $O/r03_fetch_calib_model.txt 2>&1 || exit 11
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_a -- $R/scripts/calib/fetch_calib > /dev/null 2>&1 || exit 12
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/calib_b -- $R/scripts/calib/fetch_calib > /dev/null 2>&1 || echo "second pass failed"
cd $R
{ cat $O/r03_fetch_calib_model.txt; python scripts/prof_summary.py $O/calib_a $O/calib_b; } > $O/r03_fetch_calib.txt
cat $O/r03_fetch_calib.txt
