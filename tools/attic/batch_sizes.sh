# queries/s of the batched path for several batch sizes (1 M x 768, cosine, k = 100); TSH_BATCH_TILE=256 forces the big tile
one() { timeout 300 python bench.py --batch $1 --metric cosine --steps ${2:-10} --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  batch %5d %s: %.0f q/s, %.3f ms/batch, key passes %.0f us' % ($1, sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_us']))" "${TSH_BATCH_TILE:-auto}"; }
for b in 16 64 128; do unset TSH_BATCH_TILE; one $b 20; export TSH_BATCH_TILE=256; one $b 20; done
unset TSH_BATCH_TILE
for b in 256 512 2048 4096; do one $b 8; done
