#!/bin/bash
# kernel trace of the immersed-boundary bench (what sits on the critical path of a step)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_ibm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ibm -- python3 $R/scripts/model_bench.py ibm > $O/prof_ibm.log 2>&1 || { tail -5 $O/prof_ibm.log; exit 14; }
cd $R
python scripts/prof_summary.py $O/prof_ibm | head -14 | cut -c1-200
python - <<'PY'
import csv,glob,os
p=sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_ibm/**/*kernel_trace.csv",recursive=True))[-1]
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][5:60]) for r in csv.DictReader(open(p)))
i0=[i for i,e in enumerate(ev) if 'k_ibm_step' in e[2]][400]
t0=ev[i0-2][0]
for s,e,n in ev[i0-2:i0+26]: print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {n}")
PY
