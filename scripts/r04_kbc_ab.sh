#!/bin/bash
# KBC 4096^2 (config 3): the build before the scalar-base addressing of the LDS-ring window (ab/pre_saddr) against this
# build, alternating on one box, same bench.py, the library chosen by LBM_HIP_LIB.
out=gpurun_out/r04/kbc_ab.txt
mkdir -p gpurun_out/r04
: > $out
val() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['secondary'][0]; print(d['value'], d['roofline'].get('kernel_ms'))" $1; }
for i in 1 2 3 4; do
  LBM_HIP_LIB=$PWD/ab/pre_saddr/lattice-boltzmann-method_amd/lib/liblbm_hip.so timeout -k 10 200 python bench.py --secondary-only --secondary kbc --no-pmc --no-power > /tmp/ab_a.json 2>/tmp/ab_a.err || { echo "pre run failed" >> $out; tail -3 /tmp/ab_a.err >> $out; exit 1; }
  echo "pre_saddr $(val /tmp/ab_a.json)" >> $out
  timeout -k 10 200 python bench.py --secondary-only --secondary kbc --no-pmc --no-power > /tmp/ab_b.json 2>/tmp/ab_b.err || { echo "new run failed" >> $out; tail -3 /tmp/ab_b.err >> $out; exit 1; }
  echo "saddr     $(val /tmp/ab_b.json)" >> $out
done
cat $out
