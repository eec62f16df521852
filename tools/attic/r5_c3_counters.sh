#!/bin/bash
# Round 5: counters of THIS round's fp16 key kernel (batch_score_f16pp_kernel, filtered pass) at C3 -- cosine <2,false,4>
# and L2 <0,false,4> on the bench's L2 corpus (norms U(0.5, 2)) -- in separate rocprofv3 --pmc passes (sums over the
# chip), and the kernel stats of both 1024-query calls.  Output: gpurun_out/r5c3/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3; rm -rf $O; mkdir -p $O
run() { # metric name counters...
  m=$1; n=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/$n -o p -- python bench.py --batch 1024 --metric $m --steps 2 --warmup 1 --no-cpu-baseline > $O/$n.log 2>&1
  echo "## $m: $*" >> $O/counters.txt
  python tools/rocpd_summary.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "batch_score_f16pp_kernel<[02], false" | grep -v "avg_us" >> $O/counters.txt
  rm -rf $O/$n
}
for m in cosine l2; do
  run $m a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
  run $m b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE
  run $m c SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
  run $m e GRBM_GUI_ACTIVE
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$m -o c -- python bench.py --batch 1024 --metric $m --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_$m.log 2>&1
  python tools/rocpd_summary.py $(ls $O/prof_$m/*.db $O/prof_$m/*/*.db 2>/dev/null | head -1) > $O/c3_${m}_kernel_stats.txt 2>&1
  rm -rf $O/prof_$m
done
cat $O/counters.txt
head -12 $O/c3_l2_kernel_stats.txt
