cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/cgpmc
LBM_CG_TILES=0,1,2 LBM_CG_XCD=0,1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/cgpmc -- python3 $R/scripts/model_bench.py cg > $R/gpurun_out/cgpmc.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
p=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/cgpmc/**/*counter_collection.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(p)) if "k_cg_fused" in r["Kernel_Name"]]
# group consecutive dispatches by kernel name order
seq=[]
for r in rows:
    key=r["Kernel_Name"][:60]
    if not seq or seq[-1][0]!=key: seq.append([key,[]])
    seq[-1][1].append(float(r["Counter_Value"]))
for k,v in seq: print(k, len(v), sum(v)/len(v)*2*1024/16.777216e6, "B/node read")
PY
