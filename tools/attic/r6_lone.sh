#!/bin/bash
# round 6: where a lone short search's time goes -- caller's view (p50) and the GPU's share (kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6l
export TMPDIR=/tmp
LIBV=${1:-}
[ -n "$LIBV" ] && export TSH_LIB_PATH=$LIBV
timeout 600 python tools/r6_lone_probe.py --rounds 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6l/lone_probe.txt
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/r6l/prof -o p -- python tools/r6_lone_probe.py --rounds 1 > gpurun_out/r6l/prof.log 2>&1
python tools/r6_lone_trace.py gpurun_out/r6l/prof/p_results.db | tee gpurun_out/r6l/lone_trace.txt
python tools/rocpd_summary.py --by-grid gpurun_out/r6l/prof/p_results.db 2>&1 | grep -i "exact\|kernel " | head -12 | tee gpurun_out/r6l/lone_kernels.txt
rm -rf gpurun_out/r6l/prof
