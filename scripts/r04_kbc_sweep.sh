#!/bin/bash
# KBC 4096^2 (config 3) after the strip-width change: the defaults against their neighbours (bench.py --secondary-only, no PMC)
out=gpurun_out/r04/kbc_sweep.txt
mkdir -p gpurun_out/r04
: > $out
sec() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['secondary'][0]; r=d['roofline']; print(d['value'], r.get('kernel_ms'), (r.get('power') or {}).get('sclk_mhz'), (r.get('power') or {}).get('package_w'))" $1; }
for t in "" "row_pad=0" "kbc_depth=2" "sw_ldsring=0" "sw_rows=64" "sw_rows=256" "sw_xcd=2" "sw_xcd=4" "" ; do
  args=""; for kv in $t; do args="$args --tune $kv"; done
  timeout -k 10 200 python bench.py --secondary-only --secondary kbc --no-pmc $args > /tmp/k.json 2>/tmp/k.err || { echo "[$t] failed" >> $out; tail -3 /tmp/k.err >> $out; continue; }
  echo "[$t] $(sec /tmp/k.json)" >> $out
done
cat $out
