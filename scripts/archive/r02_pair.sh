#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_bgk.py -m gpu -x -q -k "paired_strip" > $O/r02_pair_tests.log 2>&1; rc=$?; tail -8 $O/r02_pair_tests.log
[ "$rc" = "0" ] || exit 11
for t in "sw_pair=0" "sw_pair=2" "sw_pair=4" "sw_pair=0" "sw_pair=2" "sw_pair=4"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 100 --warmup 10 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$t value',d['value'],'kernel_ms',d['roofline']['kernel_ms'], 'ref_order', d.get('reference_order',{}).get('value'))" | tee -a $O/r02_pair.txt
done
