#!/bin/bash
# compact per-kernel resource table: name vgpr sgpr spills scratch occupancy lds
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-unused-result -Rpass-analysis=kernel-resource-usage -o /tmp/kres.so "$@" 2>&1 | python3 -c '
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur={"name":m.group(1)};rows.append(cur);continue
    if cur is None: continue
    for k,pat in (("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("sgpr",r"TotalSGPRs: (\d+)"),("spill",r"VGPRs Spill: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)")):
        m=re.search(pat,l)
        if m: cur[k]=m.group(1)
import subprocess
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(.*","",n)
    print("%-72s v=%s s=%s spill=%s scratch=%s occ=%s lds=%s"%(n,r.get("vgpr"),r.get("sgpr"),r.get("spill"),r.get("scratch"),r.get("occ"),r.get("lds")))
'
