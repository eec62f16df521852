O=gpurun_out/r3mr2; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","recall_at_k","host_cpu")})
except Exception as e: print("no line", e)
PY
}
run tiny8 --gpus 8 --ranks-share-gpu --rows 131072 --steps 400 --warmup 40 --cpu-seconds 2
run tiny2 --gpus 2 --ranks-share-gpu --rows 131072 --steps 400 --warmup 40 --cpu-seconds 2
run c2n2 --gpus 2 --ranks-share-gpu --steps 100 --warmup 10 --cpu-seconds 2
run tiny2torch --gpus 2 --ranks-share-gpu --rows 131072 --steps 400 --warmup 40 --cpu-seconds 2 --exchange torch
