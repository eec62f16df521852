# round 5: the exact path's tests and its A/B probe
mkdir -p gpurun_out/r5u
timeout 900 python -m pytest tests/test_gpu_exact_scan.py tests/test_gpu_shard_stream.py tests/test_gpu_list_scan.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5u/exact_tests.log
cat gpurun_out/r5u/exact_tests.log
timeout 900 python tools/r5_exact_probe.py --rounds 2 > gpurun_out/r5u/exact_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5u/exact_probe.txt
