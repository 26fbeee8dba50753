#!/bin/bash
# shader clock / power while a two-phase kernel runs: rocm-smi polled beside `bench.py --secondary-only --secondary cg`
mkdir -p gpurun_out/probe
for t in ${TUNES:-41 0}; do
  ( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/probe/smi_$t.txt &
  W=$!
  timeout -k 10 200 python bench.py --secondary-only --secondary cg --no-pmc --secondary-steps 3000 --tune cg_strip2=$t 2>&1 | grep -o '"value": [0-9.]*' | head -1
  wait $W
  echo "cg_strip2=$t:"; sort gpurun_out/probe/smi_$t.txt | uniq -c | sort -rn | head -6
done
