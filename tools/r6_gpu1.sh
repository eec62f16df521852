#!/bin/bash
# round 6, first GPU pass: the new tests, then the whole GPU suite, then the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6_build.log 2>&1 || { tail -30 gpurun_out/r6_build.log; exit 1; }
tail -3 gpurun_out/r6_build.log
timeout 1500 python -m pytest tests/test_gpu_mask_handle.py tests/test_gpu_ngh_dir.py tests/test_gpu_exact_scan.py tests/test_gpu_list_scan.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r6_new_tests.log
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu --durations=12 2>&1 | tail -40 | tee gpurun_out/r6_full_size.log
