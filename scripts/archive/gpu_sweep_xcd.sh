#!/bin/bash
# XCD-grouped strip order of the headline kernel, re-measured now that the kernel sits at 91 % of the copy ceiling
R=${GRAFT_REPO_ROOT:-$PWD}
for g in 0 2 4 8 16 0; do
  timeout -k 10 200 python $R/bench.py --tune sw_xcd=$g --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('sw_xcd=$g', round(d['value'],1), d['ms_per_step'], d['roofline']['kernel_ms'])" || echo "sw_xcd=$g failed"
done
