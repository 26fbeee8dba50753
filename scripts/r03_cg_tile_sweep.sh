mkdir -p gpurun_out/probe
: > gpurun_out/probe/cg_sweep.log
for t in 4 5 6; do for x in 0 1 2 4 8; do
  echo "== cg_tile=$t cg_xcd=$x" >> gpurun_out/probe/cg_sweep.log
  timeout -k 10 120 python bench.py --secondary-only --secondary cg --no-pmc --tune cg_tile=$t --tune cg_xcd=$x >> gpurun_out/probe/cg_sweep.log 2>&1 || exit 1
done; done
