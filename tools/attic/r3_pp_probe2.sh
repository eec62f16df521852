#!/bin/bash
# Round 3: probe variants of the ping-pong fp16 key kernel (library built with -DTSH_PROBES): gemm time per variant
O=gpurun_out/r3pp; mkdir -p $O
for d in ${DBGS:-0 4 12 20}; do
  TSH_F16_DBG=$d TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 4 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "nq=1024" | tail -2 | sed "s/^/dbg=$d /"
done
