#!/bin/bash
# config 5 over 8 emulated slabs, per-slab block times by HIP events in a running chain (all slabs of a block enqueued back to back)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
B=lattice-boltzmann-method_amd/drivers/bin
: > $O/r02_cyl_events.txt
for t in "ibm_box=1" "ibm_box=0" "ibm_chain_kernel=1" "bgk_fast_delta=1"; do
  echo "# LBM_TUNE=$t" >> $O/r02_cyl_events.txt
  LBM_TUNE=$t timeout -k 10 400 $B/slab_ring_cylinder --emulate 8 --rows 2048 --cols 4096 --steps 50 --warmup 10 $( [ "$t" = "ibm_box=1" ] && echo --check 1 ) >> $O/r02_cyl_events.txt 2>&1 || exit 12
done
python - <<'PY'
import json
for l in open("gpurun_out/r02_cyl_events.txt"):
    if l.startswith("#"): print(l.strip())
    elif l.startswith("{"):
        d = json.loads(l); print("  ", [(p["slab"], p["ms_per_block"], round(p["mlups"] / 1e3, 1)) for p in d["per_slab"]], d.get("check", ""))
PY
