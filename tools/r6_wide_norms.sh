#!/bin/bash
# round 6: fp16 keys on a corpus whose norms spread over a factor 32 (U(0.1, 3.2)): forced fp16 / bf16x3 / the automatic choice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r6w
for m in l2 ip; do
  for kk in 2 1 3; do
    timeout 600 python3 bench.py --batch 1024 --metric $m --norm-range 0.1,3.2 --steps 8 --warmup 2 --batch-kernel $kk 2>gpurun_out/r6w/err.txt > gpurun_out/r6w/wide_${m}_k$kk.json
    python3 -c "
import json; j=json.load(open('gpurun_out/r6w/wide_${m}_k$kk.json')); print('$m kernel $kk ->', j['config']['batch_kernel'], round(j['value']), 'q/s', round(j['ms_per_step'],3), 'ms  frac', round(j['roofline']['frac'],3), 'key us', round(j['roofline']['kernel_us']), 'exact', j.get('ids_and_distances_bit_exact'), j['counters'])"
  done
done
