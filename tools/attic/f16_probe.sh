#!/bin/bash
# which part of the fp16 key kernel takes the time: TSH_F16_DBG 1 = no loads in the loop, 2 = no multiply, 4 = no epilogue
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -3
for d in ${F16_DBGS:-0 4 5 6 13}; do
  TSH_F16_DBG=$d TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 4 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "nq=1024" | tail -2 | sed "s/^/dbg=$d /"
done
