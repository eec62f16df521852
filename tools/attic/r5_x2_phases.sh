# round 5: exact_select_kernel's phases on the probe builds (100 MHz stamps), one histogram copy against four
mkdir -p gpurun_out/r5x
for V in tostore_amd/csrc/_build/v*/libtostore_hip_v*.so; do
  for rep in 1 2; do
    TSH_LIB_PATH=$V TSH_X2_TRACE=1 timeout 600 python tools/attic/r5_exact_probe.py --rounds 1 2> gpurun_out/r5x/x2.err > /dev/null
    grep "\[x2\]" gpurun_out/r5x/x2.err | awk -v v=$V '{k+=$3; s+=$5; e+=$7; r+=$9; f+=$12; a+=$17; c+=$19; n++} END {printf "%s: %d launches: keys %.2f, select %.2f (adds %.2f, scan %.2f), entries %.2f us; %.2f rounds, %.1f ranked\n", v, n, k/n, s/n, a/n, c/n, e/n, r/n, f/n}' | tee -a gpurun_out/r5x/x2_phases.txt
  done
done
rm -f gpurun_out/r5x/x2.err
