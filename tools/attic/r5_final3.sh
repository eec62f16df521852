#!/bin/bash
# Round 5, last pass on the final build: the driver's command (the line profiles/ carries) and a fuzz soak
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5final3
rm -rf $O && mkdir -p $O
if [ -z "$FUZZ_ONLY" ]; then
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
echo "wall seconds of the driver's command: $SECONDS" > $O/bench_driver_args.time
cat $O/bench_driver_args.time; tail -c 600 $O/bench_driver_args.json
fi
timeout -k 10 $((60*${FUZZ_MIN:-12}+120)) python tests/probes/long_fuzz.py ${FUZZ_MIN:-12} > $O/long_fuzz.txt 2>&1; tail -3 $O/long_fuzz.txt
