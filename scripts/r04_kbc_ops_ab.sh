#!/bin/bash
# KBC 4096^2: the 235-operation collision (ab/kbc_235ops_w56: this tree with round 3's KbcFastModel::collide) against the trimmed
# one of this build (196 operations at first, 190 since; label "ops196"), BOTH with 56-column strips and 4 steps per launch, alternating on one box.  MLUPS, ms per launch, sclk MHz, package W.
out=gpurun_out/r04/kbc_ops_ab.txt
mkdir -p gpurun_out/r04
: > $out
sec() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['secondary'][0]; r=d['roofline']; print(d['value'], r.get('kernel_ms'), (r.get('power') or {}).get('sclk_mhz'), (r.get('power') or {}).get('package_w'))" $1; }
for i in 1 2 3; do
  for which in ops235 ops196; do
    if [ $which = ops235 ]; then export LBM_HIP_LIB=$PWD/ab/kbc_235ops_w56/lattice-boltzmann-method_amd/lib/liblbm_hip.so; else unset LBM_HIP_LIB; fi
    timeout -k 10 200 python bench.py --secondary-only --secondary kbc --no-pmc > /tmp/k.json 2>/tmp/k.err || { echo "$which failed" >> $out; tail -3 /tmp/k.err >> $out; continue; }
    echo "$which $(sec /tmp/k.json)" >> $out
  done
done
cat $out
