#!/bin/bash
# Round-4 measurement set (run on the GPU box through gpurun); results under gpurun_out/r4final/, copied into profiles/r04_*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4final
rm -rf $O && mkdir -p $O
# 1. the driver's exact command, plain (the line the round is judged on) ...
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2> $O/bench_driver_args.time
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
# 3. C3 (cosine) and its L2 twin: kernel stats of the 1024-query batch, the key kernel per variant
for m in cosine l2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_$m -o c -- python3 bench.py --batch 1024 --metric $m --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3_$m.log 2>&1
  python tools/rocpd_summary.py $O/prof_c3_$m/c_results.db > $O/c3_${m}_kernel_stats.txt 2>&1
  rm -rf $O/prof_c3_$m
done
for kk in 3 1 0; do timeout 600 python3 bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --batch-kernel $kk 2>>$O/c3.err > $O/bench_c3_k$kk.json; done
timeout 600 python3 bench.py --batch 1024 --metric l2 --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_l2.json
timeout 600 python3 bench.py --batch 1024 --metric ip --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_ip.json
# 4. selective masks: the list scan's kernel stats + traffic at keep 1 %
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c -- python3 bench.py --mask-keep 0.01 --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/prof_c5.log 2>&1
python tools/rocpd_summary.py $O/prof_c5/c_results.db > $O/c5_keep1_kernel_stats.txt 2>&1
rm -rf $O/prof_c5
cat $O/bench_driver_args.time; head -8 $O/bench_driver_args_kernel_stats.txt; cat $O/bench_pmc_fetch_write.txt; head -9 $O/c3_cosine_kernel_stats.txt; head -9 $O/c3_l2_kernel_stats.txt; head -6 $O/c5_keep1_kernel_stats.txt
python3 -c "
import json
j=json.load(open('$O/bench_driver_args.json')); r=j['roofline']
print('driver args', round(j['value'],1), 'q/s', j['ms_per_step'], 'frac', round(r['frac'],4), 'scan us', round(r['kernel_us'],2), 'recall', j.get('recall_at_k'), j.get('recall_queries'), j.get('ids_and_distances_bit_exact'), 'cpu', j.get('cpu_baseline',{}).get('value'), 'lat', j.get('latency_ms_one_at_a_time'))
s=j.get('side',{})
print('  side seconds', s.get('seconds'))
print('  C1', s['C1'].get('latency_us'), s['C1'].get('value'))
c3=s['C3']; print('  C3', c3.get('value'), c3.get('ms_per_step'), c3.get('ms_per_step_p99'), c3.get('ms_per_step_max'), c3.get('roofline',{}).get('frac'), c3.get('checked_queries'), c3.get('ids_and_distances_bit_exact'), c3.get('two_callers',{}).get('value'), c3.get('smaller_calls'))
for k,e in s['C5'].items():
    if isinstance(e,dict): print('  C5', k, round(e['value']), round(e['roofline']['frac'],3), round(e['roofline']['kernel_us'],1), e.get('library_default_path',{}).get('value'), e.get('checked_queries'), e.get('ids_and_distances_bit_exact'))
for n in ('c3_k3','c3_k1','c3_k0','c3_l2','c3_ip'):
    try:
        j=json.load(open('$O/bench_%s.json'%n)); print(n, round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_us'], j.get('ids_and_distances_bit_exact'), j['counters'])
    except Exception as e: print(n, 'no line', e)
"
# 5. minutes of the fuzz probe on the final build
timeout -k 10 $((60*${FUZZ_MIN:-5}+120)) python tests/probes/long_fuzz.py ${FUZZ_MIN:-5} > $O/long_fuzz.txt 2>&1; tail -3 $O/long_fuzz.txt
