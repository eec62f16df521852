#!/bin/bash
# Round 3: the batched path's sample size (1 / div of the rows) against the whole step, C3
for d in ${DIVS:-32 64 128 16}; do
  TSH_SAMPLE_DIV=$d TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 20 --warmup 3 --no-cpu-baseline > /tmp/sd_$d.json 2> /tmp/sd_$d.err
  grep "nq=1024" /tmp/sd_$d.err | tail -1 | sed "s/^/div=$d /"
  python - <<PY
import json
d=json.load(open("/tmp/sd_$d.json")); r=d.get("roofline") or {}
print("div=$d", {k:d.get(k) for k in ("value","ms_per_step","ids_and_distances_bit_exact")}, "kernel_us", r.get("kernel_us"))
PY
done
