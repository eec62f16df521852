#!/bin/bash
# Config C5 with the per-query masked scan (batching off), keep 1 %: queries/s and a stretch of the kernel timeline.
# usage: [KEEP=0.01] tools/attic/c5_probe.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
K=${KEEP:-0.01}
timeout 300 python bench.py --mask-keep $K --steps 300 --warmup 50 --no-side --no-cpu-baseline --recall-queries 0 > $O/c5.json 2> $O/c5.err
python -c "import json; j=json.load(open('$O/c5.json')); print('keep $K:', round(j['value']), 'q/s', round(j['ms_per_step']*1e3,1), 'us per query; scan kernel', round(j['roofline']['kernel_us'],1), 'us')"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o c -- python bench.py --mask-keep $K --steps 100 --warmup 20 --no-side --no-cpu-baseline --recall-queries 0 > $O/prof.log 2>&1
python tools/trace_timeline.py $O/prof/c_results.db 300 ${LINES:-18}
python tools/rocpd_summary.py $O/prof/c_results.db | head -6 | cut -c1-50,73-
rm -rf $O/prof
