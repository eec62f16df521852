#!/bin/bash
# round 2, probe 1: kernel stats of the driver's bench command, host timeline of a C3 batch, C1 kernel timeline
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p1
rm -rf $O && mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_drv -o d -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/prof_drv.json 2> $O/prof_drv.err
python tools/rocpd_summary.py $O/prof_drv/d_results.db > $O/bench_driver_kernel_stats.txt 2>&1
rm -rf $O/prof_drv
TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/c3_trace.json 2> $O/c3_trace.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c -- python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3.log 2>&1
python tools/rocpd_summary.py $O/prof_c3/c_results.db > $O/c3_kernel_stats.txt 2>&1
python tools/trace_timeline.py $O/prof_c3/c_results.db 30 40 > $O/c3_timeline.txt 2>&1
rm -rf $O/prof_c3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c -- python tools/attic/c1_latency.py > $O/c1.log 2>&1
python tools/rocpd_summary.py $O/prof_c1/c_results.db > $O/c1_kernel_stats.txt 2>&1
rm -rf $O/prof_c1
TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 16 --metric cosine --steps 10 --warmup 3 --no-cpu-baseline > $O/c3_16.json 2> $O/c3_16.err
tail -3 $O/c1.log; tail -30 $O/c3_trace.err
