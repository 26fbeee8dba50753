#!/bin/bash
# forced-box chain as one launch of a few workgroups: tests, single-block and slab timings with and without
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ibm.py tests/test_gpu_drivers.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r02_ibm_chain_tests.log 2>&1; rc=$?; tail -5 $O/r02_ibm_chain_tests.log
[ "$rc" = "0" ] || exit 11
: > $O/r02_ibm_chain_bench.log
for t in "ibm_chain_kernel=0" "ibm_chain_kernel=1" "ibm_chain_wgs=8" "ibm_chain_wgs=32"; do
  echo "# LBM_TUNE=$t" | tee -a $O/r02_ibm_chain_bench.log
  LBM_TUNE=$t timeout -k 10 300 python scripts/model_bench.py ibm 2>/dev/null | cut -c1-160 | tee -a $O/r02_ibm_chain_bench.log
done
B=lattice-boltzmann-method_amd/drivers/bin
timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 --check 1 2>&1 | tee $O/r02_cyl_emulated8_chain.json | cut -c1-900 || exit 14
