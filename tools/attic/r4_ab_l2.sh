#!/bin/bash
# Round 4: L2 1024-query batches (1 M x 768) on the third-generation fp16 key kernel (default since round 4: the per-row
# |v|^2 term rides in the accumulators' start values) against the second generation (probe variant, TSH_F16_GEN=2),
# and against cosine on the same box.  Alternating.  Output: gpurun_out/r4ab/
O=gpurun_out/r4ab; mkdir -p $O
python tools/build_variants.py gen2: > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
VAR=$(tail -1 $O/build.log)
one() {  # name metric env...
  local name=$1 metric=$2; shift 2
  env "$@" timeout 400 python bench.py --batch 1024 --metric $metric --batch-kernel 2 --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); r=d.get("roofline") or {}
    print("$name:", {k:d.get(k) for k in ("value","ms_per_step")}, "key passes us", round(r.get("kernel_us",0),1), "frac", round(r.get("frac",0),4), "cand/q", d["counters"]["candidates_per_query"])
except Exception as e: print("$name: no line", e)
PY
}
for rep in 1 2; do
  # the verdict's comparison: the same unit-norm rows under L2 (gen 3 / gen 2) and under cosine
  EXTRA=--unit-rows one l2unit_gen3_$rep l2 TSH_NOP=1
  EXTRA=--unit-rows one l2unit_gen2_$rep l2 TSH_LIB_PATH=$VAR TSH_F16_GEN=2
  one cos_gen3_$rep cosine TSH_NOP=1
  # the bench's default L2 corpus (row norms U(0.5, 2): a wider band, more survivors per tile)
  one l2_gen3_$rep l2 TSH_NOP=1
  one l2_gen2_$rep l2 TSH_LIB_PATH=$VAR TSH_F16_GEN=2
done
python - <<PY
import json,glob
out={}
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.load(open(f)); out[f.split("/")[-1][:-5]]={"value":d["value"],"ms_per_step":d["ms_per_step"],"key_passes_us":d["roofline"]["kernel_us"],"frac":d["roofline"]["frac"],"candidates_per_query":d["counters"]["candidates_per_query"]}
    except Exception as e: pass
json.dump(out,open("$O/summary.json","w"),indent=1)
PY
