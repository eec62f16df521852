"""Reads a rocprofv3 --kernel-trace CSV of tools/attic/r5_s8_probe.py and prints the kernels of the last few calls as a
timeline (start offset, duration, queue, name) plus the gaps between consecutive scan kernels."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
if not rows:
    sys.exit("no kernel trace rows under " + path)
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 140
tail = rows[-n_show:]
t0 = tail[0][0]
prev_scan_end = None
for s, e, q, name in tail:
    short = name.split("(")[0][-60:]
    gap = ""
    if "scan_kernel" in name or "scan_list" in name:
        if prev_scan_end is not None:
            gap = "  gap since previous scan's end %.1f us" % ((s - prev_scan_end) / 1e3)
        prev_scan_end = e
    print("%9.1f us  +%7.1f us  q%-4s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short, gap))
