#!/bin/bash
# Round 4: selective masks -- the tile-walking masked scan (TSH_LIST_DIV=0) against the list scan (forced: TSH_LIST_DIV=1
# = every mask below 100 %), per selectivity.  Output: gpurun_out/r4list/
O=gpurun_out/r4list; mkdir -p $O
for div in 0 1; do TSH_LIST_DIV=$div timeout 600 python tests/probes/list_scan_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/div_$div.txt; done
