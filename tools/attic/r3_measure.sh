#!/bin/bash
# Round-3 measurement set (run on the GPU box through gpurun); results under gpurun_out/r3final/, copied into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3final
rm -rf $O && mkdir -p $O
# 1. the driver's exact command, plain (the line the round is judged on) ...
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err ) 2> $O/bench_driver_args.time
# ... and the same command under rocprofv3 (kernel trace + stats): the scan kernel's average must agree with roofline.kernel_us
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_drv -o d -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/prof_drv.json 2> $O/prof_drv.err
python tools/rocpd_summary.py $O/prof_drv/d_results.db > $O/bench_driver_args_kernel_stats.txt 2>&1
rm -rf $O/prof_drv
# 2. HBM traffic of the scan kernel: PMC passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/pmc_$c.log 2>&1
  echo "## $c" >> $O/bench_pmc_fetch_write.txt
  python tools/rocpd_summary.py $(ls $O/pmc_$c/*.db $O/pmc_$c/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "scan_kernel" | grep -v avg_us >> $O/bench_pmc_fetch_write.txt
  rm -rf $O/pmc_$c
done
# 3. C3 kernel stats (the ping-pong key kernel) and its counters
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c -- python3 bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3.log 2>&1
python tools/rocpd_summary.py $O/prof_c3/c_results.db > $O/c3_kernel_stats.txt 2>&1
rm -rf $O/prof_c3
run_pmc() { n=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace -d $O/$n -o p -- python bench.py --batch 1024 --metric cosine --steps 2 --warmup 1 --no-cpu-baseline > $O/$n.log 2>&1
  echo "## $*" >> $O/c3_f16pp_counters.txt
  python tools/rocpd_summary.py $(ls $O/$n/*.db $O/$n/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "batch_score_f16pp_kernel<2, false" | grep -v "avg_us" >> $O/c3_f16pp_counters.txt
  rm -rf $O/$n
}
run_pmc a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pmc d SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run_pmc e GRBM_GUI_ACTIVE
run_pmc f FETCH_SIZE
for kk in 3 1 0; do timeout 600 python3 bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 --batch-kernel $kk 2>>$O/c3.err > $O/bench_c3_k$kk.json; done
cat $O/bench_driver_args.time; head -8 $O/bench_driver_args_kernel_stats.txt; cat $O/bench_pmc_fetch_write.txt; head -9 $O/c3_kernel_stats.txt; cat $O/c3_f16pp_counters.txt
python3 -c "
import json
j=json.load(open('$O/bench_driver_args.json')); r=j['roofline']
print('driver args', round(j['value'],1), 'q/s', j['ms_per_step'], 'frac', round(r['frac'],4), 'scan us', round(r['kernel_us'],2), 'recall', j.get('recall_at_k'), j.get('ids_and_distances_bit_exact'), 'cpu', j.get('cpu_baseline',{}).get('value'), 'lat', j.get('latency_ms_one_at_a_time'))
s=j.get('side',{})
print('  C1', s['C1'].get('latency_us'), s['C1'].get('value'), s['C1'].get('reference_ann_restated'))
print('  C3', s['C3'].get('value'), s['C3'].get('ms_per_step'), s['C3'].get('roofline',{}).get('frac'), s['C3'].get('smaller_calls'))
for k in ('keep_1%','keep_10%','keep_50%'):
    e=s['C5'][k]; print('  C5', k, round(e['value']), e['roofline']['frac'], e.get('library_default_path',{}).get('value'), e.get('ids_and_distances_bit_exact'))
for kk in (3,1,0):
    j=json.load(open('$O/bench_c3_k%d.json'%kk)); print('C3 kernel',kk, round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_us'], j.get('ids_and_distances_bit_exact'))
"
# 4. ten minutes of the fuzz probe on the final build
timeout -k 10 $((60*${FUZZ_MIN:-5}+120)) python tests/probes/long_fuzz.py ${FUZZ_MIN:-5} > $O/long_fuzz.txt 2>&1; tail -3 $O/long_fuzz.txt
