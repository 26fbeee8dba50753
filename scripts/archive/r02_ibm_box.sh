#!/bin/bash
# forced box (rows AND columns cut) for the immersed-boundary blocks: tests, then config 5 on one block per variant
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ibm.py tests/test_gpu_fullsize.py tests/test_gpu_physics.py -m gpu -x -q -k "not 5000 and not taylor and not laplace" > $O/r02_ibm_box_tests.log 2>&1; rc=$?; tail -5 $O/r02_ibm_box_tests.log
[ "$rc" = "0" ] || exit 11
: > $O/r02_ibm_box_bench.log
for t in "ibm_box=0" "ibm_box=1,ibm_box_overlap=0" "ibm_box=1,ibm_box_overlap=1,cu_mask=0" "ibm_box=1,ibm_box_overlap=1"; do
  echo "# LBM_TUNE=$t" | tee -a $O/r02_ibm_box_bench.log
  LBM_TUNE=$t timeout -k 10 300 python scripts/model_bench.py ibm 2>/dev/null | tee -a $O/r02_ibm_box_bench.log
done
