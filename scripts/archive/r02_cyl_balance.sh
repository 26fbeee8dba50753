#!/bin/bash
# config 5 over 8 emulated slabs: uniform heights vs heights that give the slabs carrying the forced band fewer far rows
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
B=lattice-boltzmann-method_amd/drivers/bin
: > $O/r02_cyl_balance.txt
run() { echo "# $*" >> $O/r02_cyl_balance.txt; timeout -k 10 500 $B/slab_ring_cylinder --emulate 8 --cols 4096 --steps 50 --warmup 10 "$@" >> $O/r02_cyl_balance.txt 2>&1 || exit 12; }
run --rows 2048 --check 1
run --slab-rows 2813,1283,1283,2201,2201,2201,2201,2201 --check 1
run --slab-rows 3072,1024,1024,2253,2253,2253,2253,2252
run --slab-rows 1800,1800,1000,2357,2357,2357,2357,2356 --check 1
run --slab-rows 1850,1850,900,2357,2357,2357,2357,2356
python - <<'PY'
import json
for l in open("gpurun_out/r02_cyl_balance.txt"):
    if l.startswith("#"): print(l.strip())
    elif l.startswith("{"):
        d = json.loads(l)
        print("  ", d["slowest_slab_ms_per_block"], round(d["chain_mlups_at_the_slowest_slabs_pace"] / 1e3, 1), "k MLUPS at the slowest slab's pace;",
              [(p["rows"], p["owner"], p["ms_per_block"]) for p in d["per_slab"]], d.get("check", ""))
PY
