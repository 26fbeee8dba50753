#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_ibm.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r04/ibm_tests.log 2>&1; tail -3 gpurun_out/r04/ibm_tests.log
scripts/r04_ibm_reserve_sweep.sh
scripts/r04_kbc_ab.sh
scripts/r04_headline_ab.sh
