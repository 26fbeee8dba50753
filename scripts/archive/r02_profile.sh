#!/bin/bash
# round 2 evidence: full GPU suite, the driver's bench command, rocprofv3 kernel trace + PMC passes of the
# same command (each counter group in its own pass), secondary model numbers
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r02_pytest_gpu.log 2>&1; rc=$?; tail -5 $O/r02_pytest_gpu.log; echo "pytest rc=$rc"
[ "$rc" = "0" ] || exit 11
fi
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r02_bench_driver.err | tee $O/r02_bench_n1.json || exit 12
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_sq
BARGS="--steps 100 --warmup 10 --repeats 3 --min-warm-s 0.05 --no-cpu-baseline --no-pmc"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 $R/bench.py $BARGS > $O/prof_kt.log 2>&1 || exit 13
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 10 --warmup 5 --repeats 2 --min-warm-s 0 --no-cpu-baseline --no-pmc > $O/prof_fetch.log 2>&1 || exit 14
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 10 --warmup 5 --repeats 2 --min-warm-s 0 --no-cpu-baseline --no-pmc > $O/prof_write.log 2>&1 || exit 15
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_sq -- python3 $R/bench.py --steps 10 --warmup 5 --repeats 2 --min-warm-s 0 --no-cpu-baseline --no-pmc > $O/prof_sq.log 2>&1 || exit 16
cd $R
grep '"metric"' $O/prof_kt.log | tail -1 > $O/r02_bench_under_rocprof.json
python scripts/prof_summary.py $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_sq > $O/r02_prof_full.txt
grep -E "^#|calls|k_stream_collide_sw" $O/r02_prof_full.txt | cut -c1-170 | tee $O/r02_bgk_fast_sw5_rocprof_summary.txt
find $O/prof_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_bgk_fast_sw5_kernel_stats.csv
