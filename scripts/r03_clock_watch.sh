#!/bin/bash
# shader clock / power while a kernel runs: rocm-smi polled beside bench.py.  WORK = "cg:41 cg:0 kbc headline" ...
mkdir -p gpurun_out/probe
for w in ${WORK:-cg:41 cg:0}; do
  name=${w%%:*}; t=${w#*:}
  case $name in
    cg) args="--secondary-only --secondary cg --no-pmc --secondary-steps 3000 --tune cg_strip2=$t";;
    kbc) args="--secondary-only --secondary kbc --no-pmc --secondary-steps 6000";;
    ibm) args="--secondary-only --secondary ibm --no-pmc --secondary-steps 3000";;
    headline) args="--no-secondary --no-pmc --no-cpu-baseline --steps 6000 --warmup 100 --repeats 2";;
  esac
  ( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/probe/smi_${name}_$t.txt &
  W=$!
  v=$(timeout -k 10 200 python bench.py $args 2>&1 | grep -o '"value": [0-9.]*' | head -1)
  wait $W
  echo "$w $v; busiest samples (sclk, W):"; grep -v "S: \|(5[0-9][0-9]Mhz\|^$" gpurun_out/probe/smi_${name}_$t.txt | sort | uniq -c | sort -rn | head -5
done
