#!/bin/bash
# A/B on one box: level-D row stored at the end of its own iteration (liblbm_hip_late.so, -DLBM_SW_STORE_LATE=1) against
# stored by the next iteration behind level 1 (default build); headline + KBC, alternating
L=$PWD/lattice-boltzmann-method_amd/lib
for rep in 1 2 3; do for v in late new; do
  if [ $v = late ]; then export LBM_HIP_LIB=$L/liblbm_hip_late.so; else unset LBM_HIP_LIB; fi
  h=$(timeout -k 10 300 python bench.py --no-secondary --no-pmc --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1)
  k=$(timeout -k 10 300 python bench.py --secondary-only --secondary kbc --no-pmc 2>&1 | grep -o '"value": [0-9.]*' | head -1)
  i=$(timeout -k 10 300 python bench.py --secondary-only --secondary ibm --no-pmc 2>&1 | grep -o '"value": [0-9.]*' | head -1)
  echo "$v headline $h kbc $k ibm $i"
done; done
