#!/bin/bash
# stall / busy counters of the fp16 key kernel (tsh_batch_f16.hip.h), 1 M x 768 cosine, 1024-query batches; separate
# rocprofv3 --pmc passes, values summed over the chip.  TSH_F16_DBG selects a probe variant (see tools/attic/f16_probe.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/f16pmc${TSH_F16_DBG:+_dbg$TSH_F16_DBG}; rm -rf $O; mkdir -p $O
run() { # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $O/$n -o p -- python bench.py --batch 1024 --metric cosine --steps 2 --warmup 1 --no-cpu-baseline > $O/$n.log 2>&1
  echo "## $*" >> $O/summary.txt
  python tools/rocpd_summary.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "batch_score_f16_kernel<2, false" | grep -v "avg_us" >> $O/summary.txt
  rm -rf $O/$n
}
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run d SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run e GRBM_GUI_ACTIVE
run f FETCH_SIZE
cat $O/summary.txt
