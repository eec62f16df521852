#!/usr/bin/env python
"""Copies the round-6 measurement set (tools/r6_measure.sh -> gpurun_out/r6final/) into profiles/r06_* and REGENERATES
the two small files bench.py reads from the committed summaries -- so that they cannot drift from their sources:
  profiles/kernel_us.json     C5 keep 1 %: the scan kernel's average inside real searches, from r06_c5_keep1_kernel_stats.txt
  profiles/pmc_traffic.json   HBM bytes per scan launch, from r06_bench_pmc_fetch_write.txt (2 x FETCH_SIZE KB + WRITE_SIZE KB:
                              the guide's gfx950 half-count correction)
python tools/r6_collect.py [source dir]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6final")
PROF = os.path.join(ROOT, "profiles")
copied = []
for f in sorted(os.listdir(SRC)):
    if f.endswith((".err", ".log", ".time")) or os.path.isdir(os.path.join(SRC, f)):
        continue
    shutil.copy(os.path.join(SRC, f), os.path.join(PROF, "r06_" + f))
    copied.append("r06_" + f)
print("copied", len(copied), "files")

# kernel_us.json from the C5 keep-1 % kernel stats
p = os.path.join(PROF, "r06_c5_keep1_kernel_stats.txt")
if os.path.exists(p):
    ent = None
    for ln in open(p):
        m = re.match(r"\s*(void )?(tsh::exact_scan_kernel<0>)\S*\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", ln)
        if m:
            ent = {"kernel": m.group(2), "avg_us": float(m.group(4)), "calls": int(m.group(3)), "round": 6,
                   "source": "profiles/r06_c5_keep1_kernel_stats.txt"}
            break
    if ent:
        out = {"_comment": "Average durations of kernels inside real searches under rocprofv3 --kernel-trace --stats, WRITTEN BY "
                           "tools/r6_collect.py from the committed summaries named in `source` (tools/r6_measure.sh step 5). bench.py "
                           "prints them beside its own HIP-event figures where a leg's roofline fraction has been disputed (side.C5): "
                           "NOT measured inside the bench run.",
               "C5.keep_1%": ent}
        json.dump(out, open(os.path.join(PROF, "kernel_us.json"), "w"), indent=1)
        print("kernel_us.json:", ent)
# pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes (the 1 M-row launches: the largest value per counter)
p = os.path.join(PROF, "r06_bench_pmc_fetch_write.txt")
if os.path.exists(p):
    vals = {}
    for ln in open(p):
        m = re.search(r"scan_kernel.*?\s(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", ln)
        if m:
            vals.setdefault(m.group(1), []).append(float(m.group(3)))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        f, w = max(vals["FETCH_SIZE"]), max(vals["WRITE_SIZE"])
        out = {"_comment": "HBM bytes per scan_kernel launch from rocprofv3 PMC passes of their own (FETCH_SIZE and WRITE_SIZE do not "
                           "fit one pass): 2*FETCH_SIZE*1024 (gfx950 half-count correction, MI355X_MICROARCH.md HBM section) + "
                           "WRITE_SIZE*1024.  WRITTEN BY tools/r6_collect.py from `source`.  bench.py copies the matching entry into "
                           "roofline.traffic and names the file in roofline.traffic_source: it is NOT measured inside the bench run.",
               "1000000x768": {"fetch_size_kb": f, "write_size_kb": w, "traffic_bytes": int(round((2 * f + w) * 1024)), "round": 6,
                               "source": "profiles/r06_bench_pmc_fetch_write.txt"}}
        json.dump(out, open(os.path.join(PROF, "pmc_traffic.json"), "w"), indent=1)
        print("pmc_traffic.json:", out["1000000x768"])
