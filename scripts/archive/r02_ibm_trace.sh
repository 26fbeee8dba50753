#!/bin/bash
# kernel trace of the immersed-boundary block (2048 x 4096): the chain of the forced box beside the window launch
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_ibm
LBM_IBM_SIZES=${LBM_IBM_SIZES:-2048} LBM_TUNE=${LBM_TUNE:-ibm_box=1} timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_ibm --output-format csv -- python $R/scripts/model_bench.py ibm > $O/r02_ibm_trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ibm/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 60 kernels: 2-3 blocks
t0 = int(rows[-60]["Start_Timestamp"])
out = []
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.append(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:90]}")
open("gpurun_out/r02_ibm_trace_tail.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
