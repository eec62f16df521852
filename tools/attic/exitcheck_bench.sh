#!/bin/bash
# Does bench.py exit cleanly UNDER rocprofv3?  (Round 3: with torch in the process the tool is finalised before the
# library's atexit handler, and the handler's HIP calls then abort -- signal 6 -- after which the process ignores
# SIGTERM: always `timeout -k`.  Fixed by destroying the CU-masked streams with the last shard of a device.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/exitcheck_bench; rm -rf $O; mkdir -p $O
try() { n=$1; shift
  timeout -k 5 ${LIMIT:-240} rocprofv3 --kernel-trace -d $O/$n -o x -- python bench.py "$@" > $O/$n.json 2> $O/$n.err
  echo "$n rc=$? abrt=$(grep -c 'caught signal 6' $O/$n.err) segv=$(grep -c 'caught signal 11' $O/$n.err)"
  rm -rf $O/$n
}
try minimal --steps 20 --warmup 5 --no-side --recall-queries 0 --no-cpu-baseline --lat-queries 20
[ -n "$FULL" ] && try full --steps 20 --warmup 5
