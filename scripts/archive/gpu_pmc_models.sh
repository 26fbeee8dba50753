#!/bin/bash
# the model kernels (KBC sliding window, two-phase tile kernel, wall-carrying BGK window) under PMC:
# one SQ pass (stall breakdown), then FETCH_SIZE and WRITE_SIZE each in its own pass (together they
# exceed what one pass can collect on gfx950)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_msq $O/prof_mfetch $O/prof_mwrite
if [ "${SKIP_SQ:-0}" != "1" ]; then
LBM_WALL_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_msq -- python3 $R/scripts/model_bench.py kbc cg walls > $O/prof_msq.log 2>&1 || { tail -5 $O/prof_msq.log; exit 14; }
fi
LBM_WALL_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_mfetch -- python3 $R/scripts/model_bench.py kbc cg walls > $O/prof_mfetch.log 2>&1 || { tail -5 $O/prof_mfetch.log; exit 15; }
LBM_WALL_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_mwrite -- python3 $R/scripts/model_bench.py kbc cg walls > $O/prof_mwrite.log 2>&1 || { tail -5 $O/prof_mwrite.log; exit 16; }
cd $R
python scripts/prof_summary.py $O/prof_msq $O/prof_mfetch $O/prof_mwrite > $O/prof_models_counters_full.txt
grep -E "counters|k_stream_collide_sw|k_cg_fused|calls" $O/prof_models_counters_full.txt | cut -c1-170 > $O/prof_models_counters.txt
cat $O/prof_models_counters.txt
