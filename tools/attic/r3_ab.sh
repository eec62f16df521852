#!/bin/bash
# Round 3: C3 (1 M x 768 cosine, 1024-query batches) with the ping-pong key kernel (default) and with the second
# generation (TSH_F16_GEN=2: compiled into probe builds only, -DTSH_PROBES), alternating on one box.
# Output: gpurun_out/r3ab/
O=gpurun_out/r3ab; mkdir -p $O
for rep in 1 2; do for g in 3 2; do
  TSH_F16_GEN=$g timeout 300 python bench.py --batch 1024 --metric cosine --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline > $O/c3_gen${g}_$rep.json 2> $O/c3_gen${g}_$rep.err
  python - <<PY
import json
try:
    d=json.load(open("$O/c3_gen${g}_$rep.json")); r=d.get("roofline") or {}
    print("gen $g rep $rep:", {k:d.get(k) for k in ("value","ms_per_step","recall_at_k","ids_and_distances_bit_exact")}, "kernel_us", r.get("kernel_us"), "frac", r.get("frac"))
except Exception as e: print("gen $g: no line", e)
PY
done; done
