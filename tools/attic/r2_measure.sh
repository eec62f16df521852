#!/bin/bash
# Round-2 measurement set (run on the GPU box through gpurun); results under gpurun_out/r2final/, copied into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2final
rm -rf $O && mkdir -p $O
# 1. the driver's exact command, plain (the line the round is judged on) ...
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2> $O/bench_driver_args.time
# ... and the same command under rocprofv3 (kernel trace + stats): the scan kernel's average must agree with roofline.kernel_us
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_drv -o d -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/prof_drv.json 2> $O/prof_drv.err
python tools/rocpd_summary.py $O/prof_drv/d_results.db > $O/bench_driver_args_kernel_stats.txt 2>&1
rm -rf $O/prof_drv
# 2. default arguments (1000 steps x 3 regions)
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
# 3. HBM traffic of the scan kernel: PMC passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/pmc_$c.log 2>&1
  echo "## $c" >> $O/bench_pmc_fetch_write.txt
  python tools/rocpd_summary.py $(ls $O/pmc_$c/*.db $O/pmc_$c/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "scan_kernel" | grep -v avg_us >> $O/bench_pmc_fetch_write.txt
  rm -rf $O/pmc_$c
done
# 4. C3 kernel stats
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c -- python3 bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3.log 2>&1
python tools/rocpd_summary.py $O/prof_c3/c_results.db > $O/c3_kernel_stats.txt 2>&1
rm -rf $O/prof_c3
for kk in 3 1 0; do timeout 600 python3 bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --batch-kernel $kk 2>>$O/c3.err > $O/bench_c3_k$kk.json; done
cat $O/bench_driver_args.time; tail -c 300 $O/bench_driver_args.json; echo; head -8 $O/bench_driver_args_kernel_stats.txt; cat $O/bench_pmc_fetch_write.txt; head -9 $O/c3_kernel_stats.txt
python3 -c "
import json
for f in ('bench_driver_args','bench_default'):
    j=json.load(open('$O/'+f+'.json')); r=j['roofline']
    print(f, round(j['value'],1), 'q/s', j['ms_per_step'], 'frac', round(r['frac'],4), 'scan us', round(r['kernel_us'],2), 'recall', j.get('recall_at_k'), j.get('ids_and_distances_bit_exact'), 'cpu', j.get('cpu_baseline',{}).get('value'))
    s=j.get('side',{})
    if s:
        print('  C1', s['C1'].get('latency_us'), s['C1'].get('value'))
        print('  C3', s['C3'].get('value'), s['C3'].get('ms_per_step'), s['C3'].get('roofline',{}).get('frac'), s['C3'].get('f32_mfma_variant',{}).get('roofline',{}).get('frac'), s['C3'].get('smaller_calls'))
        for k in ('keep_1%','keep_10%','keep_50%'):
            e=s['C5'][k]; print('  C5', k, round(e['value']), e['roofline']['frac'], e.get('library_default_path',{}).get('value'), e.get('ids_and_distances_bit_exact'))
        print('  side seconds', s.get('seconds'))
for kk in (3,1,0):
    j=json.load(open('$O/bench_c3_k%d.json'%kk)); print('C3 kernel',kk, round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_us'], j.get('ids_and_distances_bit_exact'))
"
