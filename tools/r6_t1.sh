#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bands.py tests/test_gpu_batch.py tests/test_gpu_comm_multirank.py::test_bench_in_process_multi_gpu -x -q -m gpu 2>&1 | tail -25
bash tools/r6_wide_norms.sh 2>&1 | grep "kernel 3"
