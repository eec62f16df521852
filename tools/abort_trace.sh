#!/bin/bash
# native backtrace of a process that aborts on the GPU box: tools/abort_trace.sh <command ...>
gcc -O1 -g -fPIC -shared "$(dirname "$0")/micro/abort_trace.c" -o /tmp/abort_trace.so || exit 1
LD_PRELOAD=/tmp/abort_trace.so "$@"
