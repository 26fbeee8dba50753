#!/bin/bash
# The sliding-window kernels with the prefetched row taken over AHEAD of the stores (this build) against the build before
# (ab/kbc_235: scripts/r04_ab_setup.sh <commit before> kbc_235), alternating on one box, same bench.py, the library chosen by
# LBM_HIP_LIB: headline (8192^2 BGK, 5 steps per launch), KBC 4096^2 (3 steps), cylinder 16384 x 4096 (5 steps).
out=gpurun_out/r04/store_wait_ab.txt
mkdir -p gpurun_out/r04
: > $out
old=$PWD/ab/${AB_OLD:-kbc_235}/lattice-boltzmann-method_amd/lib/liblbm_hip.so
head() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']; print(d['value'], r.get('kernel_ms'), (r.get('power') or {}).get('sclk_mhz'), (r.get('power') or {}).get('package_w'))" $1; }
sec() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['secondary'][0]; r=d['roofline']; print(d['value'], r.get('kernel_ms'), (r.get('power') or {}).get('sclk_mhz'), (r.get('power') or {}).get('package_w'))" $1; }
for i in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export LBM_HIP_LIB=$old; else unset LBM_HIP_LIB; fi
    timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-secondary > /tmp/ab.json 2>/tmp/ab.err || { echo "$which headline failed" >> $out; tail -3 /tmp/ab.err >> $out; exit 1; }
    echo "headline $which $(head /tmp/ab.json)" >> $out
    for s in kbc ibm; do
      timeout -k 10 200 python bench.py --secondary-only --secondary $s --no-pmc > /tmp/ab.json 2>/tmp/ab.err || { echo "$which $s failed" >> $out; tail -3 /tmp/ab.err >> $out; exit 1; }
      echo "$s $which $(sec /tmp/ab.json)" >> $out
    done
  done
done
cat $out
