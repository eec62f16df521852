#!/bin/bash
# Round 5, second measurement set (after the exact path of short searches): the driver's command plain and under
# rocprofv3 (kernel stats by launch grid), C5 keep 1 % under rocprofv3, the exact path's A/B probe and kernel stats.
# Results under gpurun_out/r5final2/, copied into profiles/r05_*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5final2
rm -rf $O && mkdir -p $O
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo "wall seconds of the driver's command: $SECONDS" > $O/bench_driver_args.time
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_drv -o d -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/prof_drv.json 2> $O/prof_drv.err
python tools/rocpd_summary.py $O/prof_drv/d_results.db --by-grid > $O/bench_driver_args_kernel_stats.txt 2>&1
rm -rf $O/prof_drv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c -- python3 bench.py --mask-keep 0.01 --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/prof_c5.log 2>&1
python tools/rocpd_summary.py $O/prof_c5/c_results.db > $O/c5_keep1_kernel_stats.txt 2>&1
rm -rf $O/prof_c5
timeout 900 python tools/attic/r5_exact_probe.py --rounds 3 2>&1 | grep -v amdgpu.ids > $O/exact_probe.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_x -o p -- python tools/attic/r5_exact_probe.py --rounds 1 > $O/prof_x.log 2>&1
python tools/rocpd_summary.py $O/prof_x/p_results.db --by-grid 2>&1 | head -26 > $O/exact_kernel_stats.txt
rm -rf $O/prof_x
# the other metrics' E1 (16384 x 768, one query at a time)
for m in 1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_m$m -o p -- python tools/r5_exact_one.py --metric $m > $O/prof_m$m.log 2>&1
  python tools/rocpd_summary.py $O/prof_m$m/p_results.db 2>&1 | grep -E "exact_s" >> $O/exact_kernel_stats.txt
  rm -rf $O/prof_m$m
done
V=$(ls tostore_amd/csrc/_build/v*/libtostore_hip_v*.so 2>/dev/null | head -1)
if [ -n "$V" ]; then
  TSH_LIB_PATH=$V TSH_X2_TRACE=1 timeout 600 python tools/attic/r5_exact_probe.py --rounds 1 2> $O/x2.err > /dev/null
  grep "\[x2\]" $O/x2.err | awk '{k+=$3; s+=$5; e+=$7; r+=$9; f+=$12; a+=$17; c+=$19; n++} END {printf "exact_select_kernel phases over %d launches (probe build, 100 MHz stamps): keys %.2f, select %.2f (the round\047s adds %.2f, its scan %.2f), entries %.2f us; %.2f histogram rounds, %.1f ranked\n", n, k/n, s/n, a/n, c/n, e/n, r/n, f/n}' >> $O/exact_kernel_stats.txt
  rm -f $O/x2.err
fi
cat $O/bench_driver_args.time; python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5final2/bench_driver_args.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_us"))
s = d.get("side", {})
for kx, v in s.get("C5", {}).items():
    if isinstance(v, dict) and "value" in v:
        print("C5", kx, round(v["value"]), v["roofline"]["kernel"], round(v["roofline"]["frac"], 3), v.get("library_default_path", {}).get("value"))
print("C1", {k: v for k, v in s.get("C1", {}).items() if k in ("value", "p50_ms", "p99_ms", "one_at_a_time")})
print("s8", s.get("shard_of_8", {}).get("us_per_query"), s.get("shard_of_8", {}).get("upper_bound_speedup"))
print("C3", s.get("C3", {}).get("value"))
PY
head -30 $O/exact_kernel_stats.txt; cat $O/exact_probe.txt | tail -34; head -8 $O/c5_keep1_kernel_stats.txt
