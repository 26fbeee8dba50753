#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bgk.py tests/test_gpu_kbc.py tests/test_gpu_pressure_blocks.py tests/test_gpu_ibm.py -m gpu -x -q -k "walls or wall or pressure or blocks or graph" > $O/r02_walls_tests.log 2>&1; rc=$?; tail -5 $O/r02_walls_tests.log
[ "$rc" = "0" ] || exit 11
LBM_WALL_DEPTH=5 LBM_WALL_SPLIT=2,1,2,1 timeout -k 10 400 python scripts/model_bench.py walls 2>/dev/null | cut -c1-170 | tee $O/r02_walls_bench.log
for sp in 2 1; do LBM_TUNE=sw_split=$sp timeout -k 10 400 python scripts/model_bench.py pressure ibm 2>/dev/null | cut -c1-200 | sed "s/^/sw_split=$sp /" | tee -a $O/r02_walls_bench.log; done
