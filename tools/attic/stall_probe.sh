#!/bin/bash
# Looks for one-off stalls in the batched path: traces every call and prints the calls whose GPU wait or whose
# whole library time is far above the usual 2.1-2.5 ms, plus the sum the library accounts for.
# usage: tools/attic/stall_probe.sh [extra bench.py options]
export TSH_TRACE_BATCH=1
timeout 600 python bench.py --batch 1024 --metric cosine --steps 20 --warmup 3 "$@" 2>&1 |
  awk '/tsh batch\] nq=/ { n++; w=$0; sub(/.*gpu wait /,"",w); sub(/ us.*/,"",w); if (w+0 > 4000) print "call", n, $0 }
       /tsh search\] nq=/ { m++; s=$0; sub(/.*shards /,"",s); sub(/ us.*/,"",s); f=$0; sub(/.*finalize /,"",f); sub(/ us.*/,"",f);
                            tot += s+f; if (s+f > 3500) print "search", m, $0 }
       /^\{"metric"/ { match($0, /"ms_per_step": [0-9.]+/); print substr($0, RSTART, RLENGTH) }
       END { print n, "batch calls traced;", m, "searches, library time", tot/1000, "ms" }'
