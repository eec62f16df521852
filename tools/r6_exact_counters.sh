#!/bin/bash
# Round 6: counters of the SHIPPED exact_scan_kernel<0> (a row's eight lanes read 128 contiguous bytes per load) at
# 10 000 and 16 384 rows of 768 floats, in separate rocprofv3 --pmc passes (sums over the chip per launch).
# -> $O/exact_scan_counters.txt   (O defaults to gpurun_out/r6x)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=${O:-gpurun_out/r6x}; mkdir -p $O; rm -f $O/exact_scan_counters.txt
run() {
  rows=$1; n=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc_$n -o p -- python tools/r5_exact_one.py --rows $rows > $O/pmc_$n.log 2>&1
  echo "## $rows x 768: $*" >> $O/exact_scan_counters.txt
  python tools/rocpd_summary.py $(ls $O/pmc_$n/*.db $O/pmc_$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "exact_s|exact_p|kernel " >> $O/exact_scan_counters.txt
  rm -rf $O/pmc_$n $O/pmc_$n.log
}
for rows in 10000 16384; do
  run $rows a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run $rows b SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
  run $rows c SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
  run $rows d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
  run $rows e GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
done
cat $O/exact_scan_counters.txt
