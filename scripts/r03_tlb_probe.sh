#!/bin/bash
# address-translation / texture-path counters of the two-phase step (tile kernel) beside the BGK headline kernel:
# is the two-phase step, whose every wave touches 36 + 18 planes 134 MB apart, bound by the UTCL1 / UTCL2 path?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/tlb
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE"
P2="TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TA_ADDR_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum"
P4="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
P5="TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_LFIFO_FULL_sum TCP_GATE_EN2_sum"
for W in ${WORK:-cg headline}; do
  n=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    n=$((n+1)); case " ${PASSES:-1 2 3 4 5} " in *" $n "*) ;; *) continue;; esac; rm -rf $O/${W}_$n
    timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $O/${W}_$n -- python3 $R/bench.py --pmc-child $W ${EXTRA:-} > $O/${W}_$n.log 2>&1 || { echo "pass $n of $W failed"; tail -3 $O/${W}_$n.log; }
  done
  python3 $R/scripts/prof_summary.py $(ls -d $O/${W}_? ) | grep -E "^#|calls|k_cg_fused<16, 32, 4, false, 1>|k_cg_walk|k_cg_strip|k_stream_collide_sw<" | cut -c1-170 > $O/${W}${TAG:-}_summary.txt
  find $O/${W}_? -name "*.csv" -size +2M -delete
done

