#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
for rows in -1 171 241 304 -1 152 205; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 100 --warmup 10 --sw-rows $rows 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('sw_rows $rows value',d['value'],'kernel_ms',d['roofline']['kernel_ms'])" | tee -a $O/r02_rows.txt
done
