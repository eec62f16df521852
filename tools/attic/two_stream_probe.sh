run() { timeout 200 python bench.py "$@" --no-cpu-baseline --recall-queries 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   %.0f queries/s' % d['value'])"; }
for mt in 0 400 1000 100000; do
  export TSH_TWO_STREAM_MIN_TILES=$mt
  echo "min_tiles=$mt"
  echo " bern 0.01"; run --mask-keep 0.01 --steps 4000 --warmup 200
  echo " bern 0.03"; run --mask-keep 0.03 --steps 4000 --warmup 200
  echo " C1 10k x 128 k10"; run --rows 10000 --dim 128 --k 10 --steps 6000 --warmup 300
  echo " 40k x 768"; run --rows 40000 --steps 6000 --warmup 300
  echo " 125k x 768"; run --rows 125000 --steps 4000 --warmup 300
done
