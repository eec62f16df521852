# config C5: 1 M x 768 with row masks, queries/s per mask kind and selectivity
for kind in bernoulli range; do for keep in 0.01 0.1 0.5; do
  timeout 200 python bench.py --mask-keep $keep --mask-kind $kind --steps 4000 --warmup 200 --no-cpu-baseline --recall-queries 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d.get('counters',{})
print('$kind keep $keep  %.0f queries/s  scan in-pipeline %.1f us  fallbacks %s' % (d['value'], r['kernel_us'], c.get('fallback_searches')))"
done; done
