#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
B=lattice-boltzmann-method_amd/drivers/bin
timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 --check 1 2>&1 | tee $O/r02_cyl_emulated8.json
timeout -k 10 300 $B/slab_ring_cylinder --spawn 1 --rows 16384 --cols 4096 --steps 50 --warmup 10 --check 1 --id-file /tmp/cyl_id 2>&1 | grep driver | tee $O/r02_cyl_one_slab.json
LBM_TUNE=bgk_fast_delta=1 timeout -k 10 600 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 2>&1 | tee $O/r02_cyl_emulated8_fastdelta.json
