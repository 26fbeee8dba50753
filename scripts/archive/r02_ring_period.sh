#!/bin/bash
# ring with one exchange per `period` launches: tests, then N = 1 self-ring bench beside the plain launch, same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_bgk.py tests/test_gpu_ibm.py tests/test_gpu_drivers.py -m gpu -x -q -k "ring or slab" > $O/r02_ring_period_tests.log 2>&1; rc=$?; tail -5 $O/r02_ring_period_tests.log
[ "$rc" = "0" ] || exit 11
: > $O/r02_ring_period.log
timeout -k 10 200 python bench.py --no-pmc --no-cpu-baseline >> $O/r02_ring_period.log 2>&1 || exit 12
for p in 1 2 3; do
  timeout -k 10 200 python bench.py --no-pmc --no-cpu-baseline --force-halo --ring-period $p >> $O/r02_ring_period.log 2>&1 || exit 13
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_ring_period.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["config"]["halo"], d["timing"]["batch_ms"], d.get("ring_phases"))
PY
