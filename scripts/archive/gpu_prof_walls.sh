#!/bin/bash
# kernel trace of the wall-bounded BGK bench (plain interior launch + wall-carrying frame launches)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_walls
LBM_WALL_DEPTH=${LBM_WALL_DEPTH:-5} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_walls -- python3 $R/scripts/model_bench.py walls > $O/prof_walls.log 2>&1 || { tail -5 $O/prof_walls.log; exit 14; }
cd $R
python scripts/prof_summary.py $O/prof_walls > $O/prof_walls_summary.txt; head -12 $O/prof_walls_summary.txt | cut -c1-200
python - <<'PY'
import csv,glob,os
p=sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_walls/**/*kernel_trace.csv",recursive=True))[-1]
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][5:70]) for r in csv.DictReader(open(p)))
i0=[i for i,e in enumerate(ev) if 'k_stream_collide_sw' in e[2]][600]
t0=ev[i0][0]
for s,e,n in ev[i0:i0+12]: print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {n}")
PY
