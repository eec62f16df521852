#!/bin/bash
# Round 5, after the in-order select of small exact searches: tests of the paths it touches, the A/B probe, the driver's command
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5final4
rm -rf $O && mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_exact_scan.py tests/test_gpu_shard_stream.py tests/test_gpu_list_scan.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_concurrency.py tests/test_gpu_sharded.py tests/test_gpu_comm.py -x -q -m gpu 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
timeout 900 python tools/attic/r5_exact_probe.py --rounds 2 2>&1 | grep -v amdgpu.ids > $O/exact_probe.txt
grep "exact  " $O/exact_probe.txt
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo "wall seconds of the driver's command: $SECONDS" > $O/bench_driver_args.time
cat $O/bench_driver_args.time
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5final4/bench_driver_args.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_us"))
s = d.get("side", {})
for kx, v in s.get("C5", {}).items():
    if isinstance(v, dict) and "value" in v:
        print("C5", kx, round(v["value"]), round(v["roofline"]["frac"], 3), v.get("library_default_path", {}).get("value"))
print("C1", s["C1"]["value"], s["C1"]["latency_us"])
print("s8", s.get("shard_of_8", {}).get("us_per_query"), s.get("shard_of_8", {}).get("upper_bound_speedup"))
print("C3", s.get("C3", {}).get("value"))
PY
