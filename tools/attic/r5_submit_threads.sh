# round 5 experiment: submitting threads of a multi-query call on the exact path (probe build, TSH_SUBMIT_THREADS)
mkdir -p gpurun_out/r5s; rm -f gpurun_out/r5s/threads.txt
V=$(ls tostore_amd/csrc/_build/v*/libtostore_hip_v*.so | head -1)
for rep in 1 2; do
  for t in 2 3 4; do
    echo "== TSH_SUBMIT_THREADS=$t" >> gpurun_out/r5s/threads.txt
    TSH_LIB_PATH=$V TSH_SUBMIT_THREADS=$t timeout 600 python tools/attic/r5_exact_probe.py --rounds 1 2>/dev/null | grep "exact  " | cut -c1-70 >> gpurun_out/r5s/threads.txt
  done
done
cat gpurun_out/r5s/threads.txt
