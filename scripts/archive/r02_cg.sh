#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_cg.py tests/test_gpu_fullsize.py -m gpu -x -q ${CG_TEST_K:+-k "$CG_TEST_K"} > $O/r02_cg_tests.log 2>&1; rc=$?; tail -12 $O/r02_cg_tests.log
[ "$rc" = "0" ] || exit 11
LBM_CG_STRIP2=${LBM_CG_STRIP2:-4,2,1} LBM_CG_ROWS2=${LBM_CG_ROWS2:-32,64,128} timeout -k 10 500 python scripts/model_bench.py cg 2>/dev/null | tee $O/r02_cg_bench.log
