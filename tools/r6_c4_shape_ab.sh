O=gpurun_out/abc4; mkdir -p $O
for rep in 1 2; do
for x in "--plane-in-row-order" ""; do
  timeout 600 python3 bench.py --batch 1024 --metric ip --rows 1250000 --dim 1536 --steps 6 --warmup 2 --no-cpu-baseline $x 2>/dev/null > $O/r.json
  python3 - <<PY
import json
j=json.loads(open("$O/r.json").read().strip().splitlines()[-1])
print("IP 1.25M x 1536 '$x': %.0f q/s, %.3f ms, key passes %.1f us, frac %.3f, cand %.1f, exact %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_us"], j["roofline"]["frac"], j["counters"]["candidates_per_query"], j.get("ids_and_distances_bit_exact")))
PY
done
done
