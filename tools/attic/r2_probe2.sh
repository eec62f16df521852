#!/bin/bash
# round 2, probe 2: fused batch tail -- tests, host timeline and kernel timeline of a C3 batch
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p2
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_irregular.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/c3_trace.json 2> $O/c3_trace.err
timeout 300 python bench.py --batch 1024 --metric cosine --steps 20 --warmup 3 > $O/c3.json 2> $O/c3.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c -- python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3.log 2>&1
python tools/rocpd_summary.py $O/prof_c3/c_results.db > $O/c3_kernel_stats.txt 2>&1
python tools/trace_timeline.py $O/prof_c3/c_results.db 20 30 > $O/c3_timeline.txt 2>&1
rm -rf $O/prof_c3
for nq in 16 128 256; do TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch $nq --metric cosine --steps 10 --warmup 3 --no-cpu-baseline 2> $O/c3_$nq.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nq $nq', round(j['value']), round(j['ms_per_step'],3))"; done
grep "nq=1024" $O/c3_trace.err | tail -3; grep "chunk" $O/c3_trace.err | tail -8
python -c "import json; j=json.load(open('$O/c3.json')); print(j['value'], j['ms_per_step'], j.get('ids_and_distances_bit_exact'), j.get('checked_queries'))"
cat $O/c3_timeline.txt | head -24
