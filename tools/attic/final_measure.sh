#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun); results under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
rm -rf $O && mkdir -p $O
# 1. headline (config C2), with the CPU baseline and the 1000-query recall check
timeout 900 python bench.py 2>$O/bench_default.err > $O/bench_default.json
# 2. the same command under rocprofv3 (kernel trace + stats); shorter, no CPU leg
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_default -o d -- python bench.py --steps 1500 --warmup 200 --no-cpu-baseline --recall-queries 0 > $O/prof_default.log 2>&1
python tools/rocpd_summary.py $O/prof_default/d_results.db > $O/bench_default_kernel_stats.txt 2>&1
rm -rf $O/prof_default
# 3. config C3: library default (auto = fp16 keys for cosine), then each kernel forced
timeout 600 python bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_f16.json
timeout 600 python bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --batch-kernel 1 2>>$O/c3.err > $O/bench_c3_bf16x3.json
timeout 600 python bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --batch-kernel 0 2>>$O/c3.err > $O/bench_c3_f32mfma.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c -- python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3.log 2>&1
python tools/rocpd_summary.py $O/prof_c3/c_results.db > $O/bench_c3_kernel_stats.txt 2>&1
rm -rf $O/prof_c3
# 4. side configs
{
  echo "per-rank load of N = 2 / 4 / 8 (rows per GPU of the C2 corpus), one GPU, no exchange:"
  for r in 500000 250000 125000; do
    timeout 300 python bench.py --rows $r --steps 6000 --warmup 500 --no-cpu-baseline --recall-queries 200 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('  rows $r: %d queries/s, scan in pipeline %.1f us (alone %.1f us), recall %s, bit-exact %s' % (j['value'], r['kernel_us'], r['kernel_us_back_to_back_alone'], j.get('recall_at_k'), j.get('ids_and_distances_bit_exact')))"
  done
  echo "the N>1 code path (process group, all-gather, merge) with one rank, 125000 rows:"
  timeout 300 python bench.py --rows 125000 --steps 6000 --warmup 500 --no-cpu-baseline --recall-queries 0 --force-sharded 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  %d queries/s (groups of %s)' % (j['value'], j['config'].get('queries_per_call')))"
  echo "config C4 row width, one shard of 8 (1.25M x 1536, IP, k=100):"
  timeout 600 python bench.py --rows 1250000 --dim 1536 --metric ip --steps 1500 --warmup 100 --no-cpu-baseline --recall-queries 100 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('  %d queries/s, scan %.1f us = %.1f%% of HBM peak, recall %s, bit-exact %s' % (j['value'], r['kernel_us'], 100*r['frac'], j.get('recall_at_k'), j.get('ids_and_distances_bit_exact')))"
  echo "config C1 (10k x 128, L2, k=10):"
  timeout 300 python bench.py --rows 10000 --dim 128 --k 10 --steps 20000 --warmup 2000 --recall-queries 1000 --cpu-seconds 3 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); l=j['latency_ms_one_at_a_time']; print('  %d queries/s pipelined, one at a time p50 %.0f us p99 %.0f us, recall %s over %s queries, bit-exact %s, oracle 1 thread %.1f queries/s' % (j['value'], l['p50']*1e3, l['p99']*1e3, j.get('recall_at_k'), j.get('recall_queries'), j.get('ids_and_distances_bit_exact'), j['cpu_baseline']['value']))"
} > $O/side_configs.txt 2>&1
cat $O/side_configs.txt
python -c "import json; j=json.load(open('$O/bench_default.json')); print({k: j[k] for k in ('value','ms_per_step','recall_at_k','ids_and_distances_bit_exact','cpu_baseline') if k in j}); print(j['roofline'])"
python -c "import json; [print(f, round(json.load(open('$O/'+f))['value'])) for f in ('bench_c3_f16.json','bench_c3_bf16x3.json','bench_c3_f32mfma.json')]"
