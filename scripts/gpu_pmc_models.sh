#!/bin/bash
# wave-level stall breakdown of the model kernels (KBC sliding window, two-phase tile kernel,
# wall-carrying BGK window): one SQ pass + one TCC (HBM traffic) pass over scripts/model_bench.py
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_msq $O/prof_mtcc
LBM_WALL_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_msq -- python3 $R/scripts/model_bench.py kbc cg walls > $O/prof_msq.log 2>&1 || { tail -5 $O/prof_msq.log; exit 14; }
LBM_WALL_DEPTH=4 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/prof_mtcc -- python3 $R/scripts/model_bench.py kbc cg walls > $O/prof_mtcc.log 2>&1 || { tail -5 $O/prof_mtcc.log; exit 15; }
cd $R
python scripts/prof_summary.py $O/prof_msq $O/prof_mtcc | grep -E "counters|k_stream_collide_sw|k_cg_fused|calls" | cut -c1-170 | tee $O/prof_models_counters.txt
