#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('no ring value',d['value'])" | tee -a $O/r02_edge.txt
for e in 32 16 8 5 32 8; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --force-halo --steps 20 --warmup 5 --edge-rows $e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('edge_rows $e value',d['value'],'batch',d['timing']['batch_ms']['median'],'phases',d['ring_phases'][0])" | tee -a $O/r02_edge.txt
done
