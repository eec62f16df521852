#!/bin/bash
# round 6: the wide pick (exact_pick_kernel) -- its tests, then the lone-query probe and trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6p
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_exact_scan.py tests/test_gpu_list_scan.py tests/test_gpu_parity.py tests/test_gpu_shard_stream.py tests/test_gpu_sharded.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_mask_handle.py tests/test_gpu_irregular.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r6p/tests.log
bash tools/attic/r6_lone.sh
