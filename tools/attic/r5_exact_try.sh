# round 5: the exact path's tests, its A/B probe, its kernels under rocprofv3, the select kernel's phases (probe build)
mkdir -p gpurun_out/r5u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_exact_scan.py tests/test_gpu_shard_stream.py tests/test_gpu_list_scan.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5u/exact_tests.log
cat gpurun_out/r5u/exact_tests.log
timeout 900 python tools/attic/r5_exact_probe.py --rounds 2 > gpurun_out/r5u/exact_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r5u/exact_probe.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5u/prof -o p -- python tools/attic/r5_exact_probe.py --rounds 1 > gpurun_out/r5u/prof.log 2>&1
python tools/rocpd_summary.py gpurun_out/r5u/prof/p_results.db --by-grid 2>&1 | head -24 > gpurun_out/r5u/exact_kernel_stats.txt
cat gpurun_out/r5u/exact_kernel_stats.txt
rm -rf gpurun_out/r5u/prof
V=$(ls tostore_amd/csrc/_build/v*/libtostore_hip_v*.so 2>/dev/null | head -1)
if [ -n "$V" ]; then
  TSH_LIB_PATH=$V TSH_X2_TRACE=1 timeout 600 python tools/attic/r5_exact_probe.py --rounds 1 2> gpurun_out/r5u/x2.err > /dev/null
  grep "\[x2\]" gpurun_out/r5u/x2.err | awk '{k+=$3; s+=$5; e+=$7; r+=$9; f+=$12; n++} END {printf "exact_select_kernel phases over %d launches: keys %.2f, select %.2f, entries %.2f us; %.2f histogram rounds, %.1f ranked\n", n, k/n, s/n, e/n, r/n, f/n}' | tee gpurun_out/r5u/x2_phases.txt
  rm -f gpurun_out/r5u/x2.err
fi
