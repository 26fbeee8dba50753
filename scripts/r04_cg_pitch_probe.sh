#!/bin/bash
# Is the rate loss of the large XCD patches a power-of-two row stride effect?  Config 4's kernel at 2048 columns (16 KiB rows:
# vertical neighbours of a patch are 256 KiB apart) against 2112 and 1984 columns, per patch order.
mkdir -p gpurun_out/r04
out=gpurun_out/r04/cg_pitch_probe.txt
: > $out
for cols in 2048 2112 1984; do
  echo "## cols $cols" >> $out
  LBM_BENCH_CG_COLS=$cols timeout -k 10 300 python scripts/r04_cg_big_sweep.py 2 2,402,802,804,1604 2>/dev/null | grep "^{" >> $out
done
cat $out
