#!/bin/bash
# round 6: the hub rows' bound on | off, batched L2 / inner product / L2 on unit rows / L2 + IP on norms U(0.1, 3.2) -- same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r6h
one() {  # label, args...
  lab=$1; shift
  timeout 600 python3 bench.py --batch 1024 --steps 10 --warmup 2 "$@" 2>gpurun_out/r6h/err.txt > gpurun_out/r6h/$lab.json || { tail -5 gpurun_out/r6h/err.txt; return; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r6h/$lab.json')); print('%-22s' % '$lab', round(j['value']), 'q/s', round(j['ms_per_step'],3), 'ms  key passes', round(j['roofline']['kernel_us']), 'us frac', round(j['roofline']['frac'],3), 'exact', j.get('ids_and_distances_bit_exact'), 'cands', round(j['counters']['candidates_per_query'],1), 'fallbacks', j['counters']['fallback_searches'])"
}
for rep in 1 2; do
  one l2_hub_$rep --metric l2 --hub
  one l2_nohub_$rep --metric l2
  one ip_hub_$rep --metric ip --hub
  one ip_nohub_$rep --metric ip
done
one l2_unit_hub --metric l2 --unit-rows --hub
one l2_unit_nohub --metric l2 --unit-rows
one l2_wide_hub --metric l2 --norm-range 0.1,3.2 --hub
one l2_wide_nohub --metric l2 --norm-range 0.1,3.2
one ip_wide_hub --metric ip --norm-range 0.1,3.2 --hub
one ip_wide_nohub --metric ip --norm-range 0.1,3.2
one cosine --metric cosine
