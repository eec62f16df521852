#!/usr/bin/env python
"""Print a window of the kernel timeline from a rocprofv3 rocpd .db: start (us,
relative), duration, stream/queue, name -- to see what overlaps what."""
import sqlite3
import sys

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = sqlite3.connect(path)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "tid")
rows = c.execute(f"select start, end, {qcol}, name from kernels where name like '%tsh::%' order by start").fetchall()
rows = rows[skip:skip + count]
t0 = rows[0][0]
prev_end = {}
for st, en, q, name in rows:
    short = name.split("(")[0].replace("void ", "").replace("tsh::", "")[:28]
    print("%10.1f  +%8.1f us  q=%-4s %s" % ((st - t0) / 1e3, (en - st) / 1e3, q, short))
