# A/B of an environment switch on the C3 batch: queries/s with one caller and with two, and the key passes'
# microseconds, alternating.  usage: tools/attic/ab_batch.sh VAR [value]   (VAR=value against VAR unset; value defaults to 1)
V=${1:-TSH_BATCH_PROVEN_TAU}; VAL=${2:-1}
one() { timeout 300 python bench.py --batch 1024 --metric ${M:-cosine} --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %s %.0f q/s (%.3f ms), two callers %.0f q/s, key passes %.0f us' % (sys.argv[1], d['value'], d['ms_per_step'], (d.get('two_callers') or {}).get('value', float('nan')), d['roofline']['kernel_us']))" "$1"; }
for i in 1 2 3; do export $V=$VAL; one "$V=$VAL"; unset $V; one "default "; done
