#!/bin/bash
# round 4 evidence: the driver's bench command (headline + secondary configs 3 / 4 / 5 with their live PMC rooflines), then
# rocprofv3 kernel traces and PMC passes (each counter group in its own pass, program directly after `--`) of the headline
# and of every secondary workload.  Summaries land in gpurun_out/r04_*; copy what is judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/r04_bench_driver.err > $O/r04_bench_n1.json || exit 12
tail -c 400 $O/r04_bench_n1.json; echo
export TMPDIR=/tmp
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
# ---- headline ----
rm -rf $O/p_h_kt $O/p_h_f $O/p_h_w $O/p_h_sq
HARGS="--steps 100 --warmup 10 --repeats 3 --min-warm-s 0.05 --no-cpu-baseline --no-pmc --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_h_kt -- python3 $R/bench.py $HARGS > $O/p_h_kt.log 2>&1 || exit 13
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_h_f -- python3 $R/bench.py --pmc-child headline > /dev/null 2>&1 || exit 14
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_h_w -- python3 $R/bench.py --pmc-child headline > /dev/null 2>&1 || exit 15
timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/p_h_sq -- python3 $R/bench.py --pmc-child headline > /dev/null 2>&1 || exit 16
grep '"metric"' $O/p_h_kt.log | tail -1 > $O/r04_bench_under_rocprof.json
python3 $R/scripts/prof_summary.py $O/p_h_kt $O/p_h_f $O/p_h_w $O/p_h_sq | grep -E "^#|calls|k_stream_collide_sw" | cut -c1-190 > $O/r04_bgk_fast_sw5_rocprof_summary.txt
find $O/p_h_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_bgk_fast_sw5_kernel_stats.csv
# ---- secondary workloads ----
for W in kbc cg ibm; do
  rm -rf $O/p_${W}_kt $O/p_${W}_f $O/p_${W}_w $O/p_${W}_sq
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${W}_kt -- python3 $R/bench.py --secondary-only --secondary $W --no-pmc > $O/p_${W}_kt.log 2>&1 || exit 21
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_${W}_f -- python3 $R/bench.py --pmc-child $W > /dev/null 2>&1 || exit 22
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_${W}_w -- python3 $R/bench.py --pmc-child $W > /dev/null 2>&1 || exit 23
  timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/p_${W}_sq -- python3 $R/bench.py --pmc-child $W > /dev/null 2>&1 || exit 24
  { echo "## bench.py --secondary-only --secondary $W under rocprofv3 --kernel-trace (its own JSON line, then per-kernel durations);"
    echo "## then FETCH_SIZE / WRITE_SIZE / SQ passes over 'bench.py --pmc-child $W' (set-up, marker, launch groups, marker: averages over ALL dispatches of a kernel)"
    grep '"secondary"' $O/p_${W}_kt.log | tail -1 | cut -c1-1500
    python3 $R/scripts/prof_summary.py $O/p_${W}_kt $O/p_${W}_f $O/p_${W}_w $O/p_${W}_sq | grep -E "^#|calls|lbm::" | grep -v "k_layout\|k_equilibrium\|k_lbm_marker" | cut -c1-190
  } > $O/r04_${W}_rocprof_summary.txt
done
ls -la $O/r04_*summary.txt
