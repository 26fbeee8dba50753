#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_pressure_blocks.py tests/test_gpu_presets.py tests/test_gpu_kbc.py tests/test_gpu_bgk.py -m gpu -x -q > $O/r02_pressure_tests.log 2>&1; rc=$?; tail -12 $O/r02_pressure_tests.log
[ "$rc" = "0" ] || exit 11
timeout -k 10 300 python scripts/model_bench.py pressure 2>/dev/null | tee $O/r02_pressure_bench.log
for hg in 1024 256 64; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --force-halo --steps 20 --warmup 5 --tune halo_grid=$hg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('halo_grid $hg steps',d['steps'],'value',d['value'],'batch',d['timing']['batch_ms'],'phases',d['ring_phases'])" | tee -a $O/r02_ring_probe2.txt
done
