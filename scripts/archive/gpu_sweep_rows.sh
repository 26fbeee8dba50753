#!/bin/bash
# rows-per-wave sweep of the headline sliding-window kernel (tail quantisation of the launch)
R=${GRAFT_REPO_ROOT:-$PWD}
for rows in 48 56 64 72 80 88 96 103 104 112 128 160; do
  timeout -k 10 120 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --tune sw_rows=$rows 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print($rows, round(d['value'],1), d['ms_per_step'])" || exit 9
done
