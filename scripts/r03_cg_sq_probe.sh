#!/bin/bash
# where the waves of a two-phase kernel spend their cycles: three SQ passes over `bench.py --pmc-child cg` (EXTRA = tuning)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/sq
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS"
B="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM"
n=0
for P in "$A" "$B" "$C"; do case " ${PASSES:-1 2 3} " in *" $((n+1)) "*) ;; *) n=$((n+1)); continue;; esac
  n=$((n+1)); rm -rf $O/p$n
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/p$n -- python3 $R/bench.py --pmc-child cg ${EXTRA:-} > $O/p$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/p$n.log; }
done
python3 $R/scripts/prof_summary.py $O/p1 $O/p2 $O/p3 | grep -E "^#|calls|${KERNEL:-k_cg_fused<16, 32, 4, false, 1>}" | cut -c1-150 > $O/sq${TAG:-}_summary.txt
find $O/p? -name "*.csv" -size +2M -delete
