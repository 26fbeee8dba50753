#!/bin/bash
# ring rehearsal at N = 1 (self send/recv every launch): batch of 20 vs 200 steps, then a kernel trace
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 200 python -m pytest tests/test_gpu_bgk.py -m gpu -x -q -k "ring or mixed or ghost_rows" > $O/r02_ring_tests.log 2>&1; tail -3 $O/r02_ring_tests.log
for st in 20 200; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --force-halo --steps $st --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('steps',d['steps'],'value',d['value'],'batch',d['timing']['batch_ms'],'phases',d['ring_phases'])" | tee -a $O/r02_ring_probe.txt
done
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_ring
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ring -- python3 $R/bench.py --no-cpu-baseline --force-halo --steps 20 --warmup 5 --repeats 3 --min-warm-s 0.05 > $O/prof_ring.log 2>&1 || { tail -5 $O/prof_ring.log; exit 14; }
cd $R
python scripts/timeline.py $O/prof_ring 400 > $O/r02_ring_timeline.txt; tail -5 $O/r02_ring_timeline.txt
