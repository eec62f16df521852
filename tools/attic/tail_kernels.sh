#!/bin/bash
# Kernel durations and a stretch of the kernel timeline of a batched C3 call.
# usage: [NQ=1024] tools/attic/tail_kernels.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/tail_${NQ:-1024}; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o c -- python bench.py --batch ${NQ:-1024} --metric cosine --steps 4 --warmup 2 --no-cpu-baseline > $O/log 2>&1
python tools/rocpd_summary.py $O/c_results.db | grep -E 'kernel  |rerank_batch|final_select|sample_select|batch_score|half_rows' | cut -c1-40,73-
python tools/trace_timeline.py $O/c_results.db 14 ${LINES:-12}
rm -rf $O/c_results.db
