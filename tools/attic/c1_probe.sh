#!/bin/bash
# Config C1 (10 k x 128, k = 10), one query at a time: latency percentiles and the kernel timeline of a few calls.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c1; rm -rf $O; mkdir -p $O
timeout 300 python bench.py --rows 10000 --dim 128 --k 10 --steps 300 --warmup 50 --inflight 1 --no-side --no-cpu-baseline --recall-queries 0 > $O/c1.json 2> $O/c1.err
python -c "import json; j=json.load(open('$O/c1.json')); print('latency', j.get('latency_ms'), 'value', round(j['value']))"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o c -- python bench.py --rows 10000 --dim 128 --k 10 --steps 100 --warmup 20 --inflight 1 --no-side --no-cpu-baseline --recall-queries 0 > $O/prof.log 2>&1
python tools/trace_timeline.py $O/prof/c_results.db 150 12
python tools/rocpd_summary.py $O/prof/c_results.db | head -8 | cut -c1-50,73-
rm -rf $O/prof
