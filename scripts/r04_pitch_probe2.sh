#!/bin/bash
# Does a power-of-two row stride cost the BGK headline and the KBC window anything?  Same kernels at neighbouring widths.
mkdir -p gpurun_out/r04
out=gpurun_out/r04/pitch_probe_bgk_kbc.txt
: > $out
for rep in 1 2; do
for cols in 8192 8256 8128 8192; do
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cols $cols --no-pmc --no-cpu-baseline --no-secondary --no-power > /tmp/pp.json 2>/tmp/pp.err || { echo "bgk $cols failed" >> $out; continue; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('/tmp/pp.json') if l.startswith('{')][-1])
print('bgk cols', sys.argv[1], d['value'], 'MLUPS', d['roofline']['kernel_ms'], 'ms per 5-step launch', 'ref-order', d.get('reference_order',{}).get('value'))" $cols >> $out
done
for cols in 4096 4160 4032; do
  LBM_BENCH_KBC_COLS=$cols timeout -k 10 200 python bench.py --secondary-only --secondary kbc --no-pmc --no-power > /tmp/pp.json 2>/tmp/pp.err || { echo "kbc $cols failed" >> $out; continue; }
  python3 -c "
import json,sys
d=json.load(open('/tmp/pp.json'))['secondary'][0]
print('kbc cols', sys.argv[1], d['value'], 'MLUPS')" $cols >> $out
done
done
cat $out
