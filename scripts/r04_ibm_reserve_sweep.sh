#!/bin/bash
# config 5, one block (16384 x 4096, 5-step blocks): wave slots the far-row window leaves free for the forced-box chain
# ("ibm_reserve_waves"; 0 = the round-3 launch, every slot taken), default collision and reference order
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ibm_reserve.txt
: > $out
for rv in 0 64 0 32 128 256 64; do
  timeout -k 10 200 python bench.py --secondary-only --secondary ibm --no-pmc --no-power --tune ibm_reserve_waves=$rv > /tmp/ibm_rv.json 2> /tmp/ibm_rv.err || { echo "run failed ($rv)" >> $out; tail -3 /tmp/ibm_rv.err >> $out; continue; }
  python3 -c "
import json,sys
d=json.load(open('/tmp/ibm_rv.json'))['secondary'][0]
print('ibm_reserve_waves', sys.argv[1], 'default form', d['value'], 'MLUPS', d['roofline']['kernel_ms'], 'ms per 5-step block; reference order', d.get('opt_in',{}).get('value'))" $rv >> $out
done
cat $out
