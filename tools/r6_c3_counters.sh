#!/bin/bash
# Round 6: counters of the fp16 key kernel's filtered pass (batch_score_f16pp_kernel<.., false, 4>) at C3's shape, 1024-query
# calls on 1 M x 768: cosine, and L2 on the bench's L2 corpus (norms U(0.5, 2)) with the fp16 plane grouped by norm (shipped)
# and in row order (bench.py --plane-in-row-order) -- separate rocprofv3 --pmc passes, sums over the chip per launch.
# Output: gpurun_out/r6c3/counters.txt  (-> profiles/r06_c3_f16pp_counters.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c3; rm -rf $O; mkdir -p $O
{
  echo "# Round 6: PMC counters of the fp16 key kernel (tsh::batch_score_f16pp_kernel, filtered pass), 1 M x 768, 1024-query calls: cosine <2,false,4>;"
  echo "# L2 <0,false,4> on the bench's L2 corpus (norms U(0.5, 2)) with the plane grouped by norm inside blocks of 8192 rows (shipped) and in row"
  echo "# order (--plane-in-row-order: what round 5 measured); separate rocprofv3 --pmc passes, sums over the chip per launch (tools/r6_c3_counters.sh;"
  echo "# columns of the first line of a pass: calls, avg / min / max us, share of the GPU time)."
} > $O/counters.txt
run() { # label metric extra-flag name counters...
  l=$1; m=$2; x=$3; n=$4; shift 4
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/$n -o p -- python bench.py --batch 1024 --metric $m $x --steps 2 --warmup 1 --no-cpu-baseline > $O/$n.log 2>&1
  echo "## $l: $*" >> $O/counters.txt
  python tools/rocpd_summary.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "batch_score_f16pp_kernel<[02], false" | grep -v "avg_us" >> $O/counters.txt
  rm -rf $O/$n $O/$n.log
}
for cfg in "cosine|cosine|" "l2 grouped|l2|" "l2 row order|l2|--plane-in-row-order"; do
  IFS='|' read l m x <<< "$cfg"
  run "$l" $m "$x" a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
  run "$l" $m "$x" c SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
  run "$l" $m "$x" e GRBM_GUI_ACTIVE
done
cat $O/counters.txt
