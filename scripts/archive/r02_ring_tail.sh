#!/bin/bash
# where does a launch-step with an exchange lose its 0.1 ms?  issue order of edge / interior launches, interior rounds planned for fewer slots
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
: > $O/r02_ring_tail.log
run() { echo "# $*" >> $O/r02_ring_tail.log; timeout -k 10 200 python bench.py --no-pmc --no-cpu-baseline "$@" >> $O/r02_ring_tail.log 2>&1 || exit 13; }
run
for p in 1 2; do
  for f in 0 1; do
    for s in 100 85 71 60; do
      run --force-halo --ring-period $p --tune ring_interior_first=$f --tune ring_interior_slots=$s
    done
  done
done
run
python - <<'PY'
import json
for l in open("gpurun_out/r02_ring_tail.log"):
    if l.startswith("#"): print(l.strip(), end="  ")
    elif l.startswith("{"):
        d = json.loads(l); print(d["value"], d["timing"]["batch_ms"]["median"])
PY
