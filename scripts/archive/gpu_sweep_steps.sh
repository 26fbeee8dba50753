#!/bin/bash
# how long must the timed region be for a steady-state figure? (clock ramp vs. power throttling)
R=${GRAFT_REPO_ROOT:-$PWD}
for sw in "20 3" "100 10" "500 10" "500 100" "2000 100" "5000 100" "100 10"; do
  set -- $sw
  timeout -k 10 200 python $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print($1, $2, round(d['value'],1), d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 9
done
