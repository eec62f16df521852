#!/bin/bash
# Single-query scan at row widths that do not fill their last 1 KiB chunk (and two that do): queries/s, scan kernel time,
# fraction of the HBM peak.  usage: [DIMS="768 384 ..."] tools/attic/dims_probe.sh
cd "$GRAFT_REPO_ROOT" || exit 1
for d in ${DIMS:-768 384 1000 320 512 960 1280 200}; do
  timeout 300 python bench.py --dim $d --rows ${ROWS:-1000000} --steps 200 --warmup 20 --no-side --no-cpu-baseline --recall-queries 8 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('d=$d', round(j['value'],1),'q/s scan', round(r['kernel_us'],1),'us', round(r['achieved']),'GB/s frac', round(r['frac'],3), 'recall', j.get('recall_at_k'), 'exact', j.get('ids_and_distances_bit_exact'))"
done
