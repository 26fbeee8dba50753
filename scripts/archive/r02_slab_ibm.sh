#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ibm.py tests/test_gpu_drivers.py -m gpu -x -q -k "blocks_over_slabs or cylinder" > $O/r02_slab_ibm_tests.log 2>&1; rc=$?; tail -25 $O/r02_slab_ibm_tests.log
[ "$rc" = "0" ] || exit 11
B=lattice-boltzmann-method_amd/drivers/bin
# BASELINE config 5 layout on one GPU: 8 slabs of 2048 x 4096, cylinder d = 300 at rows / 4 = the seam 1|2
timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 25 --warmup 5 --check 1 2>&1 | tee $O/r02_cyl_emulated8.json
# the same slab size as a real one-rank ring (owner, no neighbours) and the whole config as one slab
timeout -k 10 300 $B/slab_ring_cylinder --spawn 1 --rows 16384 --cols 4096 --steps 50 --warmup 10 --check 1 --id-file /tmp/cyl_id 2>&1 | tee $O/r02_cyl_one_slab.json
