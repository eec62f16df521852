#!/bin/bash
# Round 4: the finalising re-rank's shape -- four waves x 64 candidates, 32-float pieces (default) against round 3's two
# waves, 64-float pieces (TSH_RF_SHAPE=2) -- on the C3 call (cosine, ~130 candidates per query) and on the bench's L2
# corpus (~240 per query): kernel durations under rocprofv3 and the bench line.  Output: gpurun_out/r4rf/
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4rf; mkdir -p $O
for m in ${METRICS:-cosine l2}; do for sh in ${SHAPES:-4 2}; do
  TSH_RF_SHAPE=$sh timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_${m}_$sh -o c -- python3 bench.py --batch 1024 --metric $m --steps 8 --warmup 2 --no-cpu-baseline > $O/${m}_$sh.json 2> $O/${m}_$sh.err
  echo "== $m shape $sh: $(python3 -c "import json;d=json.loads(open('$O/${m}_$sh.json').read().strip().splitlines()[-1]);print(round(d['value']), d['ms_per_step'])" 2>/dev/null)"
  python tools/rocpd_summary.py $O/p_${m}_$sh/c_results.db 2>&1 | grep -E "rerank_final|final_select" | cut -c1-130
  rm -rf $O/p_${m}_$sh
done; done
