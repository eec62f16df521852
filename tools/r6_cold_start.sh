#!/bin/bash
# Cold start A/B on one box: the shipped (pipelined) loader against the serial loader of rounds 1-5, built here as a variant
# library from the last commit that had it (git is not on the box: the variant is built BEFORE gpurun, see below).
#   here:  bash tools/r6_cold_start.sh build      -> tostore_amd/csrc/_build/var_oldcold.so (needs git + hipcc)
#   box:   gpurun -- 'bash tools/r6_cold_start.sh'
OLD_COMMIT=${OLD_COMMIT:-fa9fe68}
if [ "$1" = build ]; then
  set -e
  T=$(mktemp -d); mkdir -p $T/tostore_amd; cp -r tostore_amd/csrc $T/tostore_amd/; cp -r include $T/
  git show $OLD_COMMIT:tostore_amd/csrc/tsh_host_coldstart.inl.h > $T/tostore_amd/csrc/tsh_host_coldstart.inl.h
  (cd $T/tostore_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result -Wno-unused-function -c -o $OLDPWD/tostore_amd/csrc/_build/var_oldcold.o tsh_lib.hip)
  hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o tostore_amd/csrc/_build/var_oldcold.so tostore_amd/csrc/_build/tsh_scan_tu.o tostore_amd/csrc/_build/tsh_batch_tu.o tostore_amd/csrc/_build/var_oldcold.o
  rm -rf $T; ls -la tostore_amd/csrc/_build/var_oldcold.so; exit 0
fi
O=${O:-gpurun_out/cold}; mkdir -p $O
ROWS=${ROWS:-1000000}; DIM=${DIM:-768}
{
  echo "# tools/r6_cold_start.sh: tsh_index_open_ngh / tsh_index_open_ngh_shard on a $ROWS x $DIM index directory (16 KB pages, 16 MB partition"
  echo "# files, 0.1 % tombstones), files in the page cache; shipped = the pipelined loader (pread + CRC + copy on the host pool into pinned"
  echo "# memory, the append of batch i under the reads of batch i + 1), var_oldcold = the serial loader of rounds 1-5; alternating"
  for rep in 1 2; do
    timeout 900 python tests/probes/cold_start.py $ROWS $DIM 3 2>&1 | grep -v amdgpu.ids
    [ -f tostore_amd/csrc/_build/var_oldcold.so ] && TSH_LIB_PATH=$PWD/tostore_amd/csrc/_build/var_oldcold.so timeout 900 python tests/probes/cold_start.py $ROWS $DIM 3 2>&1 | grep -v amdgpu.ids
  done
} > $O/cold_start.txt
cat $O/cold_start.txt
rm -rf /tmp/tsh_cold_*
