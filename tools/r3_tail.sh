#!/bin/bash
# Round 3: the batched path's tail (re-rank shapes A/B): parity tests, kernel durations, host timeline of a call.
# usage: [GENS="0 1"] tools/r3_tail.sh     (TSH_RERANK_GEN=1: the four-wave workgroup shape; else the wave-autonomous one)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3tail; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_bands.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for g in ${GENS:-0 1}; do
  export TSH_RERANK_GEN=$g
  echo "== TSH_RERANK_GEN=$g"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p$g -o c -- python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/prof$g.log 2>&1
  python tools/rocpd_summary.py $O/p$g/c_results.db | grep -E 'kernel  |rerank_batch|final_select|sample_select|batch_score|half_rows' | cut -c1-40,73-
  python tools/trace_timeline.py $O/p$g/c_results.db 14 ${LINES:-14}
  rm -rf $O/p$g
  for i in 1 2; do
    TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --no-cpu-baseline 2> $O/trace$g.$i.err > $O/bench$g.$i.json
    python3 -c "
import json; j=json.load(open('$O/bench$g.$i.json')); print('gen $g run $i:', round(j['value']), 'q/s', round(j['ms_per_step'],4), 'ms; key passes', round(j['roofline']['kernel_us'],1), 'us frac', round(j['roofline']['frac'],4), 'exact', j.get('ids_and_distances_bit_exact'))"
  done
  grep "tsh batch\]" $O/trace$g.2.err | tail -6
done
