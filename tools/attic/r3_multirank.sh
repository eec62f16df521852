#!/bin/bash
# Round 3: the N > 1 path rehearsed on a one-GPU box -- ranks share the GPU and exchange over the host transport
# (tsh_search_sharded, gloo underneath).  Throughput is bounded by the one GPU; what this shows is that the command
# the driver runs works at every N, what the host side costs per query (tiny shards) and whether the container's
# CPU quota throttles 8 ranks.  Output: gpurun_out/r3mr/
O=gpurun_out/r3mr; mkdir -p $O
for N in 2 4 8; do
  timeout 900 python bench.py --gpus $N --ranks-share-gpu --steps 200 --warmup 20 --cpu-seconds 4 > $O/share_1m_n$N.json 2> $O/share_1m_n$N.err
  echo "N=$N 1M rows rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/share_1m_n$N.json")); print({k:d.get(k) for k in ("value","ms_per_step","recall_at_k","ids_and_distances_bit_exact","host_cpu")}, d["config"]["sharding"])
except Exception as e: print("no line", e)
PY
done
# host-bound: 16 k rows per rank
for N in 1 8; do
  timeout 900 python bench.py --gpus $N --ranks-share-gpu --rows $((16384*8)) --steps 400 --warmup 40 --cpu-seconds 2 --no-side > $O/tiny_n$N.json 2> $O/tiny_n$N.err
  echo "N=$N tiny rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/tiny_n$N.json")); print({k:d.get(k) for k in ("value","ms_per_step","recall_at_k","host_cpu")})
except Exception as e: print("no line", e)
PY
done
