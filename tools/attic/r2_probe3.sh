#!/bin/bash
# round 2, probe 3: full GPU test suite + C3 numbers with the second-generation fp16 key kernel
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p3
rm -rf $O && mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/tests.txt
cat $O/tests.txt
timeout 300 python bench.py --batch 1024 --metric cosine --steps 20 --warmup 3 > $O/c3.json 2> $O/c3.err
python -c "import json; j=json.load(open('$O/c3.json')); print('C3', round(j['value']), j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_us'], j.get('ids_and_distances_bit_exact'), j.get('checked_queries'), j['counters'])"
for m in ip l2; do timeout 300 python bench.py --batch 1024 --metric $m --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', round(j['value']), j['batch_kernel'] if 'batch_kernel' in j else j['config'], j['roofline']['kernel_us'], j.get('ids_and_distances_bit_exact'))"; done
for nq in 16 64 128 256 512 2048; do timeout 300 python bench.py --batch $nq --metric cosine --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nq $nq', round(j['value']), round(j['ms_per_step'],3))"; done
