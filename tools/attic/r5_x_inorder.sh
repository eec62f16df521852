# round 5 experiment (its result is in job_enqueue: in order below ~8 k rows of 768 floats): the exact path's select on the scan
# stream, in order behind its scan, against the tail queues -- needs the switch back in a probe build (TSH_X_INORDER, removed)
mkdir -p gpurun_out/r5y
V=$(ls tostore_amd/csrc/_build/v*/libtostore_hip_v*.so | head -1)
for rep in 1 2; do
  for x in 0 1; do
    echo "== TSH_X_INORDER=$x" >> gpurun_out/r5y/inorder.txt
    TSH_LIB_PATH=$V TSH_X_INORDER=$x timeout 600 python tools/attic/r5_exact_probe.py --rounds 1 2>/dev/null | grep "exact  " >> gpurun_out/r5y/inorder.txt
  done
done
cat gpurun_out/r5y/inorder.txt
