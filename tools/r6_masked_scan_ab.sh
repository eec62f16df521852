#!/bin/bash
# masked tile-walking scans (C5 keep 10 / 50 %, one 10 % range): bench.py --mask-keep as the main line -- queries/s, the scan kernel's
# own duration and its fraction of the HBM peak in useful bytes.  TSH_LIB_PATH selects a variant library for an A/B.
O=${O:-gpurun_out/ms}; mkdir -p $O
for rep in 1 2; do
for lib in shipped ${VARIANT:-}; do
  [ "$lib" = shipped ] && unset TSH_LIB_PATH || export TSH_LIB_PATH=$PWD/$lib
  for kp in 0.1 0.5; do
    timeout 300 python3 bench.py --mask-keep $kp --steps 200 --warmup 20 --no-cpu-baseline --recall-queries 100 --no-side 2>/dev/null > $O/m.json
    python3 - <<PY
import json
j=json.loads(open("$O/m.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("$lib keep $kp: %.0f queries/s, %.2f us per query, kernel %.2f us, frac %.3f, exact %s" % (j["value"], j["ms_per_step"]*1e3, r["kernel_us"], r["frac"], j.get("ids_and_distances_bit_exact")))
PY
  done
done
done
