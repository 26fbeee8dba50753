#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_bgk.py -m gpu -x -q -k "sliding_window_temporal" > $O/r02_pf2_tests.log 2>&1; tail -3 $O/r02_pf2_tests.log
for t in 0 1 0 1; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 100 --warmup 10 --tune sw_pf2=$t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('sw_pf2=$t value',d['value'],'kernel_ms',d['roofline']['kernel_ms'])" | tee -a $O/r02_pf2.txt
done
