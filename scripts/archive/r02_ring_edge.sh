#!/bin/bash
# self-ring bench, exchange period 2: rows per end computed ahead of the exchange (edge kernel: 2 x E rows in 2 x 147 waves)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
: > $O/r02_ring_edge.log
run() { echo "# $*" >> $O/r02_ring_edge.log; timeout -k 10 200 python bench.py --no-pmc --no-cpu-baseline "$@" >> $O/r02_ring_edge.log 2>&1 || exit 13; }
run
for e in 32 10 16 24 32 10 16 48; do run --force-halo --edge-rows $e; done
run
python - <<'PY'
import json
for l in open("gpurun_out/r02_ring_edge.log"):
    if l.startswith("#"): print(l.strip(), end="  ")
    elif l.startswith("{"):
        d = json.loads(l); print(d["value"], d["timing"]["batch_ms"]["median"], (d.get("ring_phases") or [{}])[0].get("edge_rows_ms"))
PY
