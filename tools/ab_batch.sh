# A/B of an environment switch on the C3 batch: prints queries/s and the main pass's microseconds, alternating
V=${1:-TSH_BATCH_F16_WAVES8}
one() { timeout 300 python bench.py --batch 1024 --metric ${M:-cosine} --steps 8 --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %s %.0f q/s, main pass %.0f us' % (sys.argv[1], d['value'], d['roofline']['kernel_us']))" "$1"; }
for i in 1 2 3; do export $V=1; one "$V=1"; unset $V; one "default "; done
