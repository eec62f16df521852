cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3pmc; mkdir -p $O
run() { # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $O/$n -o p -- python bench.py --batch 1024 --metric cosine --steps 2 --warmup 1 --no-cpu-baseline > $O/$n.log 2>&1
  echo "## $*" >> $O/summary.txt
  python tools/rocpd_summary.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "batch_score.*(false)" | grep -v "avg_us" >> $O/summary.txt
}
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run d SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
cat $O/summary.txt; tail -3 $O/a.log
