#!/bin/bash
# steps per launch x waves per block of the headline kernel (reassociated BGK)
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "5 2" "6 2" "4 2" "5 4" "5 1"; do
  set -- $cfg
  timeout -k 10 200 python $R/bench.py --xn $1 --tune sw_waves=$2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('xn=$1 waves=$2', round(d['value'],1), d['ms_per_step'])" || echo "xn=$1 waves=$2 failed"
done
