#!/bin/bash
# KBC 4096^2 (config 3): the 235-operation collision of rounds 3 / early 4 (ab/kbc_235, built by scripts/r04_ab_setup.sh HEAD kbc_235
# before the change) against the 196-operation one of this build, alternating on one box, same bench.py, the library chosen by
# LBM_HIP_LIB; then a sweep of the window's tunings on the new build.
out=gpurun_out/r04/kbc_trim_ab.txt
mkdir -p gpurun_out/r04
: > $out
val() { python3 -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['secondary'][0]; r=d['roofline']; print(d['value'], r.get('kernel_ms'), (r.get('power') or {}).get('sclk_mhz'), (r.get('power') or {}).get('package_w'))" $1; }
run() { # label, lib ('' = this build), extra args
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export LBM_HIP_LIB=$PWD/$lib; else unset LBM_HIP_LIB; fi
  timeout -k 10 200 python bench.py --secondary-only --secondary kbc --no-pmc "$@" > /tmp/ab.json 2>/tmp/ab.err || { echo "$label failed" >> $out; tail -3 /tmp/ab.err >> $out; return 1; }
  echo "$label $(val /tmp/ab.json)" >> $out
}
old=ab/kbc_235/lattice-boltzmann-method_amd/lib/liblbm_hip.so
for i in 1 2 3; do
  run "ops235" $old || exit 1
  run "ops196" "" || exit 1
done
for t in sw_rows=64 sw_rows=128 sw_rows=256 sw_rows=512 kbc_depth=2 sw_ldsring=0 "sw_ldsring=0 --tune kbc_depth=4"; do
  run "ops196 $t" "" --tune $t || exit 1
done
run "ops235 kbc_depth=2" $old --tune kbc_depth=2
cat $out
