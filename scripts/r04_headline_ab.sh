#!/bin/bash
# the headline kernel of the round-2 build (commit 74d84b8, ab/r2) against this build, alternating on ONE
# box; each line = value (MLUPS) of `bench.py --gpus 1 --steps 20 --warmup 5` without PMC / secondary / CPU passes.
out=gpurun_out/r04/headline_ab.txt
mkdir -p gpurun_out/r04
: > $out
val() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['value'], d['roofline'].get('kernel_ms'), (d['roofline'].get('power') or {}).get('sclk_mhz'), (d['roofline'].get('power') or {}).get('package_w'))" $1; }
for i in 1 2 3 4; do
  timeout -k 10 200 python ab/r2/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /tmp/ab_r2.json 2>/tmp/ab_r2.err || { echo "r2 run failed" >> $out; tail -3 /tmp/ab_r2.err >> $out; exit 1; }
  echo "r2   $(val /tmp/ab_r2.json)" >> $out
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-secondary > /tmp/ab_r4.json 2>/tmp/ab_r4.err || { echo "r4 run failed" >> $out; tail -3 /tmp/ab_r4.err >> $out; exit 1; }
  echo "r4   $(val /tmp/ab_r4.json)" >> $out
done
cat $out
