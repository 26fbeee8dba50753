#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for ch in default 1 2 4 8; do
  if [ "$ch" = "default" ]; then unset NCCL_MAX_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch; fi
  echo "NCCL_MAX_NCHANNELS=$ch"
  timeout -k 10 200 python scripts/r02_ring_vs_plain.py 2>&1 | grep MLUPS | tail -3
done 2>&1 | tee gpurun_out/r02_rccl_env.txt
