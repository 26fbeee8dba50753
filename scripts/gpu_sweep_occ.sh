#!/bin/bash
# steps per launch with 64-byte aligned strips
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "3 2" "4 2" "5 2" "6 2"; do
  set -- $cfg
  timeout -k 10 200 python $R/bench.py --xn $1 --tune sw_waves=$2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('xn=$1 waves=$2', round(d['value'],1), d['ms_per_step'])" || echo "xn=$1 waves=$2 failed"
done
LBM_KBC_SIZE=4096,8192 LBM_WALL_DEPTH=4,5,3 timeout -k 10 400 python $R/scripts/model_bench.py kbc walls 2>/dev/null | cut -c1-150
