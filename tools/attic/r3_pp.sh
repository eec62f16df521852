#!/bin/bash
# Round 3: the ping-pong fp16 key kernel (tsh_batch_f16pp.hip.h) against the second generation: parity tests, then
# C3 (1 M x 768 cosine, 1024-query batches) with each.  Output: gpurun_out/r3pp/
O=gpurun_out/r3pp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_bands.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for g in 3 2; do
  TSH_F16_GEN=$g TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 6 --warmup 2 --no-cpu-baseline > $O/c3_gen$g.json 2> $O/c3_gen$g.err
  echo "gen $g rc=$?"; grep "nq=1024" $O/c3_gen$g.err | tail -3
  python - <<PY
import json
try:
    d=json.load(open("$O/c3_gen$g.json")); print({k:d.get(k) for k in ("value","ms_per_step","recall_at_k","ids_and_distances_bit_exact")}, d.get("roofline"))
except Exception as e: print("no line", e)
PY
done
