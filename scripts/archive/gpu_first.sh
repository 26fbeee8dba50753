#!/bin/bash
# First-contact GPU run: smoke -> parity tests -> bench tuning sweep. Output under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log || exit 10
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log
rc=${PIPESTATUS[0]}
echo "pytest rc=$rc"
if [ "$rc" != "0" ] && [ "$rc" != "1" ]; then exit 11; fi
: > gpurun_out/sweep.log
for tune in "variant=2" "variant=1" "variant=2 nt=1" "variant=2 nt=2" "variant=2 nt=3" "variant=2 grid_cap=2048" "variant=2 grid_cap=4096" "variant=2 grid_cap=8192" "variant=1 grid_cap=4096" "variant=1 nt=3"; do
  args=""
  for kv in $tune; do args="$args --tune $kv"; done
  echo "== $tune" | tee -a gpurun_out/sweep.log
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $args 2>&1 | tee -a gpurun_out/sweep.log || exit 12
done
