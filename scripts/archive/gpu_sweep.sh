#!/bin/bash
# bench tuning sweep: each line of $1 (or stdin) = extra bench args
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/${2:-sweep}.log
: > $out
while IFS= read -r line; do
  [ -z "$line" ] && continue
  echo "== $line" >> $out
  timeout -k 10 200 python bench.py --warmup 6 --no-cpu-baseline $line 2>&1 | grep -v amdgpu.ids >> $out || { echo "FAILED: $line" >> $out; exit 12; }
done < "${1:-/dev/stdin}"
python - "$out" <<'PY'
import json, sys
tag = None
for ln in open(sys.argv[1]):
    if ln.startswith("== "): tag = ln[3:].strip()
    elif ln.startswith("{"):
        d = json.loads(ln); print(f"{d['value']:9.1f} MLUPS  {d['roofline']['achieved']:7.1f} GB/s  frac {d['roofline']['frac']:.4f}   {tag}")
PY
