#!/bin/bash
# Round 3 checkpoint: the whole GPU suite, then the driver's exact bench command (plain and under rocprofv3).
O=gpurun_out/r3ck; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3ck/bench_driver_args.json"))
print({k:d[k] for k in ("value","ms_per_step","recall_at_k","ids_and_distances_bit_exact")}, d["roofline"]["frac"], d["latency_ms_one_at_a_time"])
s=d.get("side",{})
print("C1", s.get("C1",{}).get("latency_us"), s.get("C1",{}).get("value"))
c3=s.get("C3",{}); print("C3", c3.get("value"), c3.get("ms_per_step"), c3.get("ms_per_step_mean"), (c3.get("roofline") or {}).get("frac"))
for k,v in s.get("C5",{}).items():
    if isinstance(v,dict): print("C5",k,v.get("value"),(v.get("roofline") or {}).get("frac"),(v.get("library_default_path") or {}).get("value"))
PY
