#!/bin/bash
# round 2, probe 4: fused single-dispatch path -- tests, C1 latency with kernel trace, C5 rates
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p4
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/attic/c1_latency.py 2>&1 | tail -1
python tools/attic/c1_latency.py 2>&1 | tail -1 | sed 's/^/fused: /'
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c -- python tools/attic/c1_latency.py > $O/c1.log 2>&1
python tools/rocpd_summary.py $O/prof_c1/c_results.db 2>&1 | head -6
rm -rf $O/prof_c1
for keep in 0.01 0.1 0.5; do
  timeout 300 python bench.py --mask-keep $keep --steps 2000 --warmup 100 --no-cpu-baseline --recall-queries 0 --no-side 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C5 keep $keep', round(j['value']), 'q/s, scan', round(j['roofline']['kernel_us'],1), 'us')"
  timeout 300 python bench.py --mask-keep $keep --steps 2000 --warmup 100 --no-cpu-baseline --recall-queries 0 --no-side 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('   three kernels', round(j['value']), 'q/s')"
done
for r in 125000 30000; do
  timeout 300 python bench.py --rows $r --steps 4000 --warmup 300 --no-cpu-baseline --recall-queries 0 --no-side 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('rows $r', round(j['value']), 'q/s', j['latency_ms_one_at_a_time']['p50'])"
  timeout 300 python bench.py --rows $r --steps 4000 --warmup 300 --no-cpu-baseline --recall-queries 0 --no-side 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('   three kernels', round(j['value']), 'q/s', j['latency_ms_one_at_a_time']['p50'])"
done
