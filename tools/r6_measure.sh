#!/bin/bash
# Round-6 measurement set (run on the GPU box through gpurun); results under gpurun_out/r6final/, copied into
# profiles/r06_* by tools/r6_collect.py (which also regenerates profiles/kernel_us.json and profiles/pmc_traffic.json
# from them).  STEPS="1 2 3 ..." selects parts (default: all).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6final
mkdir -p $O
STEPS=${STEPS:-"1 2 3 4 5 6 7 8 9 10 11"}
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has 1; then  # the driver's exact command, plain (the line the round is judged on)
  SECONDS=0
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
  echo "wall seconds of the driver's command: $SECONDS" > $O/bench_driver_args.time
fi
if has 2; then  # ... and under rocprofv3 (kernel trace + stats, by launch grid): the scan kernel's average must agree with roofline.kernel_us
  timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/prof_drv -o d -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args_under_rocprofv3.json 2> $O/prof_drv.err
  python tools/rocpd_summary.py $O/prof_drv/d_results.db --by-grid > $O/bench_driver_args_kernel_stats.txt 2>&1
  rm -rf $O/prof_drv
fi
if has 3; then  # HBM traffic of the scan kernel: PMC passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass)
  rm -f $O/bench_pmc_fetch_write.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/pmc_$c.log 2>&1
    echo "## $c" >> $O/bench_pmc_fetch_write.txt
    python tools/rocpd_summary.py $(ls $O/pmc_$c/*.db $O/pmc_$c/*/*.db 2>/dev/null | head -1) 2>&1 | grep -E "scan_kernel" | grep -v avg_us >> $O/bench_pmc_fetch_write.txt
    rm -rf $O/pmc_$c $O/pmc_$c.log
  done
fi
if has 4; then  # C3 (cosine), its L2 / IP twins: kernel stats and bench lines
  for m in cosine l2; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3_$m -o c -- python3 bench.py --batch 1024 --metric $m --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_c3_$m.log 2>&1
    python tools/rocpd_summary.py $O/prof_c3_$m/c_results.db > $O/c3_${m}_kernel_stats.txt 2>&1
    rm -rf $O/prof_c3_$m $O/prof_c3_$m.log
  done
  timeout 600 python3 bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_cosine.json
  timeout 600 python3 bench.py --batch 1024 --metric l2 --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_l2.json
  timeout 600 python3 bench.py --batch 1024 --metric l2 --unit-rows --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_l2_unit_rows.json
  timeout 600 python3 bench.py --batch 1024 --metric ip --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_ip.json
  # row norms a factor 32 apart (U(0.1, 3.2)): the automatic key-kernel choice keeps fp16 (round 6)
  timeout 600 python3 bench.py --batch 1024 --metric l2 --norm-range 0.1,3.2 --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_l2_wide_norms.json
  timeout 600 python3 bench.py --batch 1024 --metric ip --norm-range 0.1,3.2 --steps 10 --warmup 2 2>>$O/c3.err > $O/bench_c3_ip_wide_norms.json
fi
if has 5; then  # selective masks: the kernels of a keep-1 % mask inside real searches (the source of profiles/kernel_us.json)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c -- python3 bench.py --mask-keep 0.01 --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 0 --no-side > $O/prof_c5.log 2>&1
  python tools/rocpd_summary.py $O/prof_c5/c_results.db > $O/c5_keep1_kernel_stats.txt 2>&1
  rm -rf $O/prof_c5 $O/prof_c5.log
fi
if has 6; then  # one masked query at a time: pointer | handle, wide pick | one-workgroup select (same box, alternating) + the GPU's share
  {
    echo "# tools/r6_lone_probe.py: 1 M x 768 behind Bernoulli masks, L2, k = 100; C1's shape; the mask as a pointer (sliced, counted, listed"
    echo "# on the host per call) and as a handle (tsh_mask_create); behind the exact scan the wide pick (shipped) or round 5's select"
    timeout 900 python tools/r6_lone_probe.py --rounds 3 2>&1 | grep -v amdgpu.ids
    echo "# the GPU's share of a lone search, rocprofv3 --kernel-trace of the same script (tools/r6_lone_trace.py): E1, gap, E2' / E2, span"
  } > $O/mask_handles.txt
  timeout 600 rocprofv3 --kernel-trace -d $O/prof_lone -o p -- python tools/r6_lone_probe.py --rounds 1 > $O/prof_lone.log 2>&1
  python tools/r6_lone_trace.py $O/prof_lone/p_results.db >> $O/mask_handles.txt
  python tools/rocpd_summary.py --by-grid $O/prof_lone/p_results.db 2>&1 | grep -i "exact\|kernel " | head -14 >> $O/mask_handles.txt
  rm -rf $O/prof_lone $O/prof_lone.log
fi
if has 7; then O=$O bash tools/r6_exact_counters.sh > /dev/null; fi
if has 8; then  # the driver's N > 1 command rehearsed over the RCCL branch (ranks share the one GPU; not a scaling figure)
  for N in 2 8; do
    timeout -k 10 900 python3 bench.py --gpus $N --fake-rccl --steps 20 --warmup 5 --cpu-seconds 4 --c4-rows-per-rank 150000 > $O/rehearsal_fake_rccl_n$N.json 2> $O/rehearsal_fake_rccl_n$N.err
    echo "N=$N rc=$?" >> $O/rehearsal.txt
    # ... and the single-process deployment: one handle over N devices (here sharing the one GPU), no collective
    timeout -k 10 900 python3 bench.py --gpus $N --in-process --shards-share-gpu --steps 20 --warmup 5 --cpu-seconds 4 > $O/rehearsal_in_process_n$N.json 2> $O/rehearsal_in_process_n$N.err
    echo "in-process N=$N rc=$?" >> $O/rehearsal.txt
  done
fi
if has 9; then timeout -k 10 $((60*${FUZZ_MIN:-5}+120)) python tests/probes/long_fuzz.py ${FUZZ_MIN:-5} > $O/long_fuzz.txt 2>&1; tail -3 $O/long_fuzz.txt; fi
if has 10; then  # the fp16 plane grouped by norm (default) against the plane in row order, same box, alternating; both corpora
  O=$O/ab_tmp bash tools/r6_sorted_blocks_ab.sh > $O/grouped_plane_ab.txt 2>&1
  NORMS=0.1,3.2 O=$O/ab_tmp bash tools/r6_sorted_blocks_ab.sh > $O/grouped_plane_ab_wide_norms.txt 2>&1
  rm -rf $O/ab_tmp
fi
if has 11; then  # one query at a time through the C ABI alone (no Python in the loop): C1's shape, the headline's shard of 8
  gcc -std=c99 -O2 -I include tools/cabi_driver.c -o /tmp/cabi_driver -L tostore_amd -ltostore_hip -Wl,-rpath,$PWD/tostore_amd -lm
  {
    echo "# tools/cabi_driver.c [rows dim queries k]: a host with nothing but the C ABI -- one at a time / pipelined / batched, us per query"
    for shape in "10000 128 1000 10" "125000 768 500 100" "1000000 768 256 100"; do
      echo "## $shape"; timeout 600 /tmp/cabi_driver $shape 2>&1 | grep -v amdgpu.ids
    done
  } > $O/c_abi_latency.txt
fi
python3 - <<'PY'
import json, os
O = "gpurun_out/r6final"
def load(n):
    try:
        return json.load(open(os.path.join(O, n)))
    except Exception as e:
        return None
j = load("bench_driver_args.json")
if j:
    r = j["roofline"]
    print("driver args", round(j["value"], 1), "q/s", j["ms_per_step"], "frac", round(r["frac"], 4), "scan us", round(r["kernel_us"], 2), "recall", j.get("recall_at_k"), j.get("recall_queries"), j.get("ids_and_distances_bit_exact"), "cpu", j.get("cpu_baseline", {}).get("value"), "lat", j.get("latency_ms_one_at_a_time"))
    s = j.get("side", {})
    print("  side seconds", s.get("seconds"))
    for key in ("shard_of_8", "C4_shard_of_8"):
        x = s.get(key, {})
        print(" ", key, {k: x.get(k) for k in ("us_per_query", "upper_bound_speedup", "ids_and_distances_bit_exact", "checked_queries", "error")}, x.get("roofline", {}).get("frac"), (x.get("batch_1024") or {}).get("value"))
    print("  C1", s.get("C1", {}).get("latency_us"), s.get("C1", {}).get("value"))
    c3 = s.get("C3", {}); print("  C3", c3.get("value"), c3.get("ms_per_step"), c3.get("roofline", {}).get("frac"), c3.get("ids_and_distances_bit_exact"))
    for k, e in s.get("C5", {}).items():
        if isinstance(e, dict) and "value" in e:
            print("  C5", k, round(e["value"]), e["roofline"]["kernel"][:28], round(e["roofline"]["frac"], 3), "default", round(e.get("library_default_path", {}).get("value", 0)), "handle", round((e.get("mask_handle") or {}).get("value", 0)), (e.get("mask_handle") or {}).get("one_at_a_time_us"), e.get("ids_and_distances_bit_exact"))
for n in ("c3_cosine", "c3_l2", "c3_l2_unit_rows", "c3_ip", "c3_l2_wide_norms", "c3_ip_wide_norms"):
    j = load("bench_%s.json" % n)
    if j: print(n, round(j["value"]), j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel_us"], j.get("ids_and_distances_bit_exact"), j["counters"])
for N in (2, 8):
    j = load("rehearsal_fake_rccl_n%d.json" % N)
    if j: print("rehearsal N=%d" % N, round(j["value"], 1), j["ms_per_step"], j.get("recall_at_k"), j.get("ids_and_distances_bit_exact"), j["config"]["sharding"])
    j = load("rehearsal_in_process_n%d.json" % N)
    if j: print("in-process N=%d" % N, round(j["value"], 1), j["ms_per_step"], j.get("recall_at_k"), j.get("ids_and_distances_bit_exact"), j["config"]["sharding"][:60])
PY
for f in bench_driver_args.time bench_pmc_fetch_write.txt mask_handles.txt grouped_plane_ab.txt grouped_plane_ab_wide_norms.txt c_abi_latency.txt; do [ -f $O/$f ] && cat $O/$f; done
[ -f $O/bench_driver_args_kernel_stats.txt ] && head -14 $O/bench_driver_args_kernel_stats.txt
[ -f $O/c5_keep1_kernel_stats.txt ] && head -8 $O/c5_keep1_kernel_stats.txt
[ -f $O/exact_scan_counters.txt ] && head -60 $O/exact_scan_counters.txt
