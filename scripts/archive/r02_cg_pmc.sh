#!/bin/bash
# the two-phase kernels under PMC: one SQ pass, one LDS/extra pass, FETCH and WRITE each on its own
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_cgsq $O/prof_cgsq2 $O/prof_cgfetch $O/prof_cgwrite
export LBM_CG_STRIP2=${LBM_CG_STRIP2:-4} LBM_CG_ROWS2=${LBM_CG_ROWS2:-0}
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_cgsq -- python3 $R/scripts/model_bench.py cg > $O/prof_cgsq.log 2>&1 || { tail -5 $O/prof_cgsq.log; exit 14; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/prof_cgsq2 -- python3 $R/scripts/model_bench.py cg > $O/prof_cgsq2.log 2>&1 || { tail -5 $O/prof_cgsq2.log; exit 15; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_cgfetch -- python3 $R/scripts/model_bench.py cg > $O/prof_cgfetch.log 2>&1 || { tail -5 $O/prof_cgfetch.log; exit 16; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_cgwrite -- python3 $R/scripts/model_bench.py cg > $O/prof_cgwrite.log 2>&1 || { tail -5 $O/prof_cgwrite.log; exit 17; }
cd $R
python scripts/prof_summary.py $O/prof_cgsq $O/prof_cgsq2 $O/prof_cgfetch $O/prof_cgwrite | grep -E "counters|k_cg_|calls" | cut -c1-150 > $O/r02_cg_counters.txt
cat $O/r02_cg_counters.txt
