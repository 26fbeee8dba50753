#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
for rows in -1 108 -1 108; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 100 --warmup 10 --sw-rows $rows 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('sw_rows $rows value',d['value'],'kernel_ms',d['roofline']['kernel_ms'], 'ref', d['reference_order']['value'])" | tee -a $O/r02_rows2.txt
done
timeout -k 10 400 python scripts/model_bench.py kbc walls pressure 2>/dev/null | cut -c1-200 | tee $O/r02_models_after_rows.log
