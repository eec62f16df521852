#!/bin/bash
# Round 3: the batched path's tail: parity tests, then the kernel durations, a stretch of the timeline, the host
# timeline of a call and the bench line (one caller and two).
# usage: [TESTS=0] [NQ=1024] [M=cosine] tools/attic/r3_tail.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3tail; rm -rf $O; mkdir -p $O
if [ "${TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_bands.py tests/test_gpu_irregular.py tests/test_gpu_fuzz.py "tests/test_gpu_full_size.py::test_c3_batch_of_1024_full_size" -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
fi
for g in 0; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p$g -o c -- python bench.py --batch ${NQ:-1024} --metric ${M:-cosine} --steps 6 --warmup 2 --no-cpu-baseline > $O/prof$g.log 2>&1
  python tools/rocpd_summary.py $O/p$g/c_results.db | grep -E 'kernel  |rerank_|final_select|sample_select|batch_score|half_rows' | cut -c1-40,73-
  python tools/trace_timeline.py $O/p$g/c_results.db 14 ${LINES:-9}
  rm -rf $O/p$g
  for i in 1 2; do
    TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch ${NQ:-1024} --metric ${M:-cosine} --steps 10 --warmup 2 --no-cpu-baseline 2> $O/trace$g.$i.err > $O/bench$g.$i.json
    python3 -c "
import json; j=json.load(open('$O/bench$g.$i.json')); print('run $i:', round(j['value']), 'q/s', round(j['ms_per_step'],4), 'ms; key passes', round(j['roofline']['kernel_us'],1), 'us frac', round(j['roofline']['frac'],4), 'two callers', j.get('two_callers'))"
  done
  grep "tsh batch\] nq" $O/trace$g.2.err | sed -n 4,8p
done
