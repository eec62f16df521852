#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "hub or auto_kernel" 2>&1 | tail -15
