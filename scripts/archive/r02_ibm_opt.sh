#!/bin/bash
# forcing kernel with lane-major tables + the source term as its own launch: tests, micro-timings, config 5 over slabs and on one block
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ibm.py tests/test_gpu_drivers.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r02_ibm_opt_tests.log 2>&1; rc=$?; tail -5 $O/r02_ibm_opt_tests.log
[ "$rc" = "0" ] || exit 11
timeout -k 10 200 python scripts/r02_ibm_force.py > $O/r02_ibm_force.txt 2>&1 || exit 12
timeout -k 10 200 python scripts/r02_ibm_force2.py >> $O/r02_ibm_force.txt 2>&1 || exit 13
grep -v amdgpu.ids $O/r02_ibm_force.txt
B=lattice-boltzmann-method_amd/drivers/bin
timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 --check 1 2>&1 | tee $O/r02_cyl_emulated8_b.json || exit 14
LBM_TUNE=ibm_step_opt=0,ibm_step_split=0 timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 2>&1 | tee $O/r02_cyl_emulated8_b_r1kernel.json || exit 15
timeout -k 10 300 python scripts/model_bench.py ibm 2>/dev/null | tee $O/r02_ibm_model_bench.log
