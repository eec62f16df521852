#!/bin/bash
# A/B on one box: the C3 shape (1 M x 768, 1024-query calls) under L2 / inner product with the fp16 plane grouped by norm
# inside blocks of 8192 rows (the default, TSH_OPT_BATCH_GROUP) and in row order; cosine beside them (same MFMA work, no
# per-row term: what the other two are held against).  NORMS=lo,hi for another corpus (default the bench's 0.5,2).
O=${O:-gpurun_out/sb}; mkdir -p $O
NR=${NORMS:-0.5,2}
for rep in 1 2; do
for m in l2 ip; do
  timeout 300 python3 bench.py --batch 1024 --metric $m --norm-range $NR --steps 10 --warmup 2 --plane-in-row-order 2>$O/$m.a.err > $O/$m.row_order.$rep.json
  timeout 300 python3 bench.py --batch 1024 --metric $m --norm-range $NR --steps 10 --warmup 2 2>$O/$m.b.err > $O/$m.grouped.$rep.json
done
done
timeout 300 python3 bench.py --batch 1024 --metric cosine --steps 10 --warmup 2 2>/dev/null > $O/cos.json
python3 - <<PY
import json,glob
print("file  queries/s  ms_per_call  key_passes_us  candidates_per_query  bit_exact")
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(j["value"]), round(j["ms_per_step"],4), round(j["roofline"]["kernel_us"],1), round(j.get("counters",{}).get("candidates_per_query",0),1), j.get("ids_and_distances_bit_exact"))
    except Exception as e: print(f,"ERR",e)
PY
