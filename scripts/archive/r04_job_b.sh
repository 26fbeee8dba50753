#!/bin/bash
# round 4, second evidence job: the experiments build through the kernel-equality tests, the new driver test, the N-rank
# rehearsal on one GPU (6 ranks: the pool allows at most 6 processes on a card), then the profiling passes
mkdir -p gpurun_out/r04
LBM_HIP_LIB=$PWD/lattice-boltzmann-method_amd/lib_exp/liblbm_hip.so timeout -k 10 400 python -m pytest tests/test_gpu_cg.py tests/test_gpu_bgk.py tests/test_gpu_ibm.py -x -q -m gpu > gpurun_out/r04/exp_build_tests.log 2>&1; tail -3 gpurun_out/r04/exp_build_tests.log
timeout -k 10 300 python -m pytest tests/test_gpu_drivers.py -x -q -m gpu -k "ring_driver or cylinder_driver or emulated" > gpurun_out/r04/driver_tests.log 2>&1; tail -3 gpurun_out/r04/driver_tests.log
timeout -k 10 500 python bench.py --gpus 6 --steps 20 --warmup 5 --share-gpu > gpurun_out/r04/bench_share6.json 2> gpurun_out/r04/bench_share6.err; echo "share6 rc=$?"; tail -c 1200 gpurun_out/r04/bench_share6.json; tail -5 gpurun_out/r04/bench_share6.err
scripts/r04_profile.sh
