#!/bin/bash
# Round 4: the driver's N > 1 command rehearsed on a one-GPU box over the RCCL BRANCH of tsh_search_sharded:
# the ranks share the GPU and the library's dlopen is pointed at tests/fake_rccl (bench.py --fake-rccl).
# Throughput is bounded by the one GPU; what this shows is that the command works at every N, that the
# exchange_timeline of the line accounts for its ms_per_step, and what the C4-per-rank leg costs.
# Output: gpurun_out/r4mr/
O=gpurun_out/r4mr; mkdir -p $O
for N in ${NS:-2 8}; do
  timeout 1500 python bench.py --gpus $N --fake-rccl --steps ${STEPS:-20} --warmup 5 --cpu-seconds 4 > $O/fake_rccl_n$N.json 2> $O/fake_rccl_n$N.err
  echo "N=$N rc=$?"; tail -3 $O/fake_rccl_n$N.err; python - <<PY
import json
try:
    d=json.load(open("$O/fake_rccl_n$N.json"))
    print({k:d.get(k) for k in ("value","ms_per_step","recall_at_k","ids_and_distances_bit_exact")}, d["config"]["sharding"])
    t=d["exchange_timeline"]; print("timeline: call %.3f phases %.3f of step %.3f ms (ratio %.2f), accounted incl. closing fence:" % (t["call_ms_per_step"], t["phases_sum_ms_per_step"], t["ms_per_step"], t["phases_sum_over_ms_per_step"]), t.get("accounted_over_ms_per_step"))
    print("  max over ranks:", {k: round(v,4) for k,v in t["max_over_ranks"].items()})
    print("  host_cpu:", d.get("host_cpu",{}).get("cpus_busy"), [round(r["cpus_busy"],2) for r in d["host_cpu"]["per_rank"]])
    c=d["side"]["C4_per_rank"]; print("C4 per rank:", {k:c.get(k) for k in ("value","ms_per_step","recall_at_k","ids_and_distances_bit_exact","seconds","error")})
    print("  roofline", c["roofline"]["frac"], "batch_1024", c["batch_1024"]["value"], c["batch_1024"]["ms_per_call"])
    t=c["exchange_timeline"]; print("  timeline: call %.3f phases %.3f of step %.3f ms (ratio %.2f)" % (t["call_ms_per_step"], t["phases_sum_ms_per_step"], t["ms_per_step"], t["phases_sum_over_ms_per_step"]))
except Exception as e: print("no line", repr(e))
PY
done
