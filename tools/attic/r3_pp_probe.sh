#!/bin/bash
# Round 3: phase timeline of the ping-pong fp16 key kernel (library built with -DTSH_PROBES).  Output: gpurun_out/r3pp/
O=gpurun_out/r3pp; mkdir -p $O
TSH_F16_DBG=${DBG:-32} TSH_TRACE_BATCH=1 timeout 300 python bench.py --batch 1024 --metric cosine --steps 2 --warmup 1 --no-cpu-baseline > $O/probe.json 2> $O/probe.err
echo "rc=$?"; grep "nq=1024" $O/probe.err | tail -2
grep "pp dbg" $O/probe.err | tail -244 > $O/probe_timeline.txt; wc -l $O/probe_timeline.txt
