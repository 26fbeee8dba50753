#!/bin/bash
# kernel timeline of the owner slab's block in the balanced 8-slab layout (emulated chain)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_cyl
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_cyl -- $R/lattice-boltzmann-method_amd/drivers/bin/slab_ring_cylinder --emulate 8 --cols 4096 --steps 20 --warmup 5 --slab-rows 1850,1850,900,2357,2357,2357,2357,2356 > $O/r02_cyl_trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_cyl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
for f in glob.glob("gpurun_out/prof_cyl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "memcpy")))
rows.sort()
# last occurrence of k_ibm_step: walk back to the start of that owner block (previous k_box_copy pair)
idx = [i for i, r in enumerate(rows) if "k_ibm_step" in r[3]]
last = idx[-1]
start = max(0, last - 40)
t0 = rows[start][0]
out = [f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {q:5s} {n}" for s, e, q, n in rows[start:last + 25]]
open("gpurun_out/r02_cyl_trace_owner_block.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
