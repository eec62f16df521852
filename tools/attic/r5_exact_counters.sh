#!/bin/bash
# Round 5: counters of exact_scan_kernel (16384 x 768, 2048 waves) in separate rocprofv3 --pmc passes (sums over the chip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; rm -rf $O; mkdir -p $O
run() {
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/$n -o p -- python tools/r5_exact_one.py > $O/$n.log 2>&1
  echo "## $*" >> $O/counters.txt
  python tools/rocpd_summary.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "exact_s|kernel " >> $O/counters.txt
  rm -rf $O/$n
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run c SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run e GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
cat $O/counters.txt
