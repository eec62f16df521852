#!/bin/bash
# round 6: the whole GPU suite (with durations), then the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1 || { tail -30 gpurun_out/r6_build.log; exit 1; }
tail -2 gpurun_out/r6_build.log
timeout 2400 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -45 | tee gpurun_out/r6_gpu_suite.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_args.json 2> gpurun_out/r6_bench_driver_args.err
echo "bench rc $?"; tail -c 600 gpurun_out/r6_bench_driver_args.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_bench_driver_args.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "exact", d.get("ids_and_distances_bit_exact"))
s=d["side"]
print("side seconds", s.get("seconds"))
for k in ("shard_of_8","C4_shard_of_8"):
    x=s.get(k,{})
    print(k, {kk:x.get(kk) for kk in ("error","us_per_query","upper_bound_speedup","ids_and_distances_bit_exact","checked_queries","floor_us_per_query")}, x.get("roofline",{}).get("frac"), x.get("batch_1024"))
for kk,v in s.get("C5",{}).items():
    if isinstance(v,dict) and "value" in v: print(kk, round(v["value"]), v["roofline"]["kernel"][:30], v.get("mask_handle"), v.get("ids_and_distances_bit_exact"))
print("C1", s.get("C1",{}).get("latency_us"), s.get("C1",{}).get("value"))
print("C3", s.get("C3",{}).get("value"), s.get("C3",{}).get("roofline",{}).get("frac"))
PY
