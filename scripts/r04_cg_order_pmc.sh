#!/bin/bash
# config 4: HBM-side read bytes (FETCH_SIZE x 2) and MLUPS of the shipped 16 x 64 tile kernel under different XCD tile orders
# ("cg_big_xcd": 0 none, 2 pairs of column neighbours, 100 PR + PC patches), PMC passes of bench.py, one box
mkdir -p gpurun_out/r04
out=gpurun_out/r04/cg_order_pmc.txt
: > $out
for x in 0 2 1601 402 802 1602 804; do
  timeout -k 10 200 python bench.py --secondary-only --secondary cg --no-power --tune cg_big_xcd=$x > /tmp/cg_o.json 2> /tmp/cg_o.err || { echo "run failed ($x)" >> $out; continue; }
  python3 -c "
import json,sys
d=json.load(open('/tmp/cg_o.json'))['secondary'][0]; r=d['roofline']
k=[v for n,v in r['pmc']['per_kernel_KiB_per_group'].items() if 'k_cg_tile_mn' in n and 'false>' in n]
print('cg_big_xcd', sys.argv[1], 'MLUPS', d['value'], 'inner kernel read GB', round(2*1024*k[0]['fetch_raw']/k[0]['launches']/1e9,3) if k else None, 'all reads GB', round(r['pmc']['fetch_bytes']/1e9,3), 'writes GB', round(r['pmc']['write_bytes']/1e9,3), 'traffic/algorithmic', r['traffic_over_algorithmic'], 'TB/s', r['achieved'])" $x >> $out
done
cat $out
