#!/bin/bash
# forced box in the slab engine: tests, then config 5 over 8 emulated slabs (bitwise check) and per-variant timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ibm.py tests/test_gpu_drivers.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r02_ibm_box3_tests.log 2>&1; rc=$?; tail -5 $O/r02_ibm_box3_tests.log
[ "$rc" = "0" ] || exit 11
B=lattice-boltzmann-method_amd/drivers/bin
timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 --check 1 2>&1 | tee $O/r02_cyl_emulated8_box.json || exit 14
LBM_TUNE=ibm_box=0 timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 2>&1 | tee $O/r02_cyl_emulated8_band.json || exit 15
LBM_TUNE=ibm_step_chain=1 timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 2>&1 | tee $O/r02_cyl_emulated8_box_chain.json || exit 16
LBM_TUNE=bgk_fast_delta=1 timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 2>&1 | tee $O/r02_cyl_emulated8_box_fastdelta.json || exit 17
timeout -k 10 300 $B/slab_ring_cylinder --spawn 1 --rows 16384 --cols 4096 --steps 50 --warmup 10 --check 1 --id-file /tmp/cyl_id 2>&1 | grep driver | tee $O/r02_cyl_one_slab_box.json
