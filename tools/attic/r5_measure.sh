#!/bin/bash
# Round-5 measurement set (run on the GPU box through gpurun); results under gpurun_out/r5final/, copied into profiles/r05_*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5final
rm -rf $O && mkdir -p $O
# 1. the driver's exact command, plain (the line the round is judged on) ...
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo "wall seconds of the driver's command: $SECONDS" > $O/bench_driver_args.time
# ... and the same command under rocprofv3 (kernel trace + stats): the scan kernel's average must agree with roofline.kernel_us
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_drv -o d -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/prof_drv.json 2> $O/prof_drv.err
python tools/rocpd_summary.py $O/prof_drv/d_results.db > $O/bench_driver_args_kernel_stats.txt 2>&1
rm -rf $O/prof_drv
# 2. HBM traffic of the scan kernel: PMC passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/pmc_$c.log 2>&1
  echo "## $c" >> $O/bench_pmc_fetch_write.txt
  python tools/rocpd_summary.py $(ls $O/pmc_$c/*.db $O/pmc_$c/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "scan_kernel" | grep -v avg_us >> $O/bench_pmc_fetch_write.txt
  rm -rf $O/pmc_$c
done
# 3. C3 (cosine), its L2 / IP twins, the other key kernels
for m in cosine l2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_$m -o c -- python3 bench.py --batch 1024 --metric $m --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3_$m.log 2>&1
  python tools/rocpd_summary.py $O/prof_c3_$m/c_results.db > $O/c3_${m}_kernel_stats.txt 2>&1
  rm -rf $O/prof_c3_$m
done
for kk in 3 1 0; do timeout 600 python3 bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --batch-kernel $kk 2>>$O/c3.err > $O/bench_c3_k$kk.json; done
timeout 600 python3 bench.py --batch 1024 --metric l2 --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_l2.json
timeout 600 python3 bench.py --batch 1024 --metric l2 --unit-rows --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_l2_unit_rows.json
timeout 600 python3 bench.py --batch 1024 --metric ip --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_ip.json
# 4. selective masks: the list scan's kernel stats at keep 1 %
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c -- python3 bench.py --mask-keep 0.01 --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/prof_c5.log 2>&1
python tools/rocpd_summary.py $O/prof_c5/c_results.db > $O/c5_keep1_kernel_stats.txt 2>&1
rm -rf $O/prof_c5
# 5. one rank's share of the headline at N = 8 through tsh_search_sharded (real RCCL, a world of one): the library call
#    without bench.py around it, the two ways to launch the exchange, three group schedules
{
  for rep in 1 2 3; do for ah in 0 1; do for g in 0 20 10; do echo -n "exchange_ahead=$ah group=$g (0 = the library's 10 + 5 + 5): "; timeout 200 python tools/attic/r5_s8_probe.py --group $g --ahead $ah --calls 60 2>&1 | grep "per call"; done; done; done
  timeout 200 python tools/attic/r5_s8_probe.py --calls 60 2>&1 | grep "_us"
} > $O/shard_of_8_probe.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_s8 -- python tools/attic/r5_s8_probe.py --calls 12 > $O/trace_s8.log 2>&1
python tools/attic/r5_s8_trace.py $O/trace_s8 110 > $O/shard_of_8_timeline.txt 2>&1
rm -rf $O/trace_s8
# 6. row widths whose scan variants spill or were never measured: dense and masked (keep 80 %)
{
  echo "500 k rows, dense:"; DIMS="1400 1500 3072 4096" ROWS=500000 bash tools/attic/dims_probe.sh
  echo "500 k rows, Bernoulli mask keeping 80 %:"
  for d in 1400 3072 4096; do
    timeout 300 python bench.py --dim $d --rows 500000 --mask-keep 0.8 --steps 200 --warmup 20 --no-side --no-cpu-baseline --recall-queries 8 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('d=$d keep 80 %', round(j['value'],1),'q/s scan', round(r['kernel_us'],1),'us', round(r['achieved']),'GB/s (useful bytes) frac', round(r['frac'],3), 'recall', j.get('recall_at_k'), 'exact', j.get('ids_and_distances_bit_exact'))"
  done
} > $O/row_widths.txt 2>&1
# 7. the driver's N > 1 command rehearsed over the RCCL branch (ranks share the one GPU; not a scaling figure)
for N in 2 8; do
  timeout -k 10 900 python3 bench.py --gpus $N --fake-rccl --steps 20 --warmup 5 --cpu-seconds 4 --c4-rows-per-rank 150000 > $O/rehearsal_fake_rccl_n$N.json 2> $O/rehearsal_fake_rccl_n$N.err
  echo "N=$N rc=$?" >> $O/rehearsal.txt
done
cat $O/bench_driver_args.time; head -8 $O/bench_driver_args_kernel_stats.txt; cat $O/bench_pmc_fetch_write.txt; head -9 $O/c3_cosine_kernel_stats.txt; head -9 $O/c3_l2_kernel_stats.txt; head -6 $O/c5_keep1_kernel_stats.txt; cat $O/shard_of_8_probe.txt; cat $O/row_widths.txt; cat $O/rehearsal.txt
python3 -c "
import json
j=json.load(open('$O/bench_driver_args.json')); r=j['roofline']
print('driver args', round(j['value'],1), 'q/s', j['ms_per_step'], 'frac', round(r['frac'],4), 'scan us', round(r['kernel_us'],2), 'recall', j.get('recall_at_k'), j.get('recall_queries'), j.get('ids_and_distances_bit_exact'), 'cpu', j.get('cpu_baseline',{}).get('value'), 'lat', j.get('latency_ms_one_at_a_time'))
s=j.get('side',{})
print('  side seconds', s.get('seconds'))
s8=s.get('shard_of_8',{}); print('  shard_of_8', {k:s8.get(k) for k in ('us_per_query','upper_bound_speedup','group_sweep_us_per_query','ids_and_distances_bit_exact','queries_per_exchange','error')})
print('  C1', s['C1'].get('latency_us'), s['C1'].get('value'))
c3=s['C3']; print('  C3', c3.get('value'), c3.get('ms_per_step'), c3.get('ms_per_step_p99'), c3.get('ms_per_step_max'), c3.get('roofline',{}).get('frac'), c3.get('checked_queries'), c3.get('ids_and_distances_bit_exact'), c3.get('two_callers',{}).get('value'), c3.get('smaller_calls'))
for k,e in s['C5'].items():
    if isinstance(e,dict): print('  C5', k, round(e['value']), e['roofline']['kernel'], round(e['roofline']['frac'],3), round(e['roofline']['kernel_us'],1), e.get('library_default_path',{}).get('value'), e.get('checked_queries'), e.get('ids_and_distances_bit_exact'))
for n in ('c3_k3','c3_k1','c3_k0','c3_l2','c3_l2_unit_rows','c3_ip'):
    try:
        j=json.load(open('$O/bench_%s.json'%n)); print(n, round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_us'], j.get('ids_and_distances_bit_exact'), j['counters'])
    except Exception as e: print(n, 'no line', e)
for N in (2, 8):
    try:
        j=json.load(open('$O/rehearsal_fake_rccl_n%d.json'%N)); print('rehearsal N=%d'%N, round(j['value'],1), j['ms_per_step'], j.get('recall_at_k'), j.get('ids_and_distances_bit_exact'), j['config']['sharding'], j['config'].get('queries_per_call'))
    except Exception as e: print('rehearsal N=%d'%N, 'no line', e)
"
# 8. minutes of the fuzz probe on the final build
timeout -k 10 $((60*${FUZZ_MIN:-5}+120)) python tests/probes/long_fuzz.py ${FUZZ_MIN:-5} > $O/long_fuzz.txt 2>&1; tail -3 $O/long_fuzz.txt
