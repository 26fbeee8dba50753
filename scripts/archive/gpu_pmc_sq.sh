#!/bin/bash
# wave-level stall breakdown of the headline kernel (one PMC pass, SQ block only)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof_sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_sq -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_sq.log 2>&1 || { tail -5 $O/prof_sq.log; exit 14; }
cd $R
python scripts/prof_summary.py $O/prof_sq | grep -E "counters|k_stream_collide_sw|calls" | cut -c1-150 | tee $O/prof_sq_summary.txt
