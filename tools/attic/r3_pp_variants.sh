#!/bin/bash
# Round 3: compile-time probe variants of the key kernel (tools/build_variants.py), gemm time of each
for v in ${VARS}; do
  lib=""; [ "$v" != "base" ] && lib="$PWD/tostore_amd/csrc/_build/var_$v.so"
  for d in ${DBGS:-4}; do
    TSH_LIB_PATH=$lib TSH_F16_DBG=$d TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 4 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "nq=1024" | tail -2 | sed "s/^/$v dbg=$d /"
  done
done
