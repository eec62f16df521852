#!/bin/bash
# Kernel durations of the batch tail (sample select, final select, re-rank) from a kernel trace of a short C3 run.
# usage: [NQ=1024] tools/rr_ablate.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/tail_${NQ:-1024}; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o c -- python bench.py --batch ${NQ:-1024} --metric cosine --steps 4 --warmup 2 --no-cpu-baseline > $O/log 2>&1
python tools/rocpd_summary.py $O/c_results.db | grep -E 'kernel  |rerank_batch|final_select|sample_select|batch_score' | cut -c1-40,73-
rm -rf $O/c_results.db
