#!/bin/bash
# round 2, first GPU pass: the whole -m gpu suite, then the driver's bench command and the ring rehearsal
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu.log 2>&1
rc=$?; tail -15 $O/r02_pytest_gpu.log; echo "pytest rc=$rc"
if [ "$rc" != "0" ] && [ "$rc" != "1" ]; then exit 11; fi
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r02_bench_driver.err | tee $O/r02_bench_driver.json || exit 12
timeout -k 10 300 python bench.py --no-cpu-baseline 2> $O/r02_bench_default.err | tee $O/r02_bench_default.json || exit 13
timeout -k 10 300 python bench.py --no-cpu-baseline --force-halo --steps 20 --warmup 5 2> $O/r02_bench_halo.err | tee $O/r02_bench_halo.json || exit 14
