"""Reads the rocpd .db of `rocprofv3 --kernel-trace -- python tools/r6_lone_probe.py` and prints, per launch grid of
exact_scan_kernel, what the GPU does for one short search: E1's duration, the gap to E2's start, E2's duration, and
the span from E1's start to E2's end -- medians over the pairs that ran ALONE (nothing else of the library within the
span: the one-at-a-time legs), and over the pairs inside pipelined calls."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
gcol = next((g for g in ("grid_size", "grid_x", "grid_size_x", "grid") if g in cols), None)
rows = c.execute(f"select start, end, {name}, {gcol} from kernels where {name} like '%tsh::%' order by start").fetchall()
by = defaultdict(lambda: {"alone": [], "piped": []})
for i, (s, e, nm, g) in enumerate(rows):
    if "exact_scan_kernel" not in nm:
        continue
    # its select: the next exact_select_kernel that starts after this scan ended
    j = i + 1
    while j < len(rows) and not (("exact_select_kernel" in rows[j][2] or "exact_pick_kernel" in rows[j][2]) and rows[j][0] >= e):
        j += 1
    if j >= len(rows):
        break
    s2, e2 = rows[j][0], rows[j][1]
    others = [r for r in rows[max(0, i - 4):j + 4] if r[0] < e2 and r[1] > s and r is not rows[i] and r is not rows[j]]
    by[g]["alone" if not others else "piped"].append((e - s, s2 - e, e2 - s2, e2 - s))


def med(v):
    v = sorted(v)
    return v[len(v) // 2] / 1e3 if v else float("nan")


print("%-10s %-6s %7s %9s %9s %9s %9s" % ("E1 grid", "", "pairs", "E1 us", "gap us", "E2 us", "span us"))
for g in sorted(by):
    for kind in ("alone", "piped"):
        p = by[g][kind]
        if p:
            print("%-10s %-6s %7d %9.2f %9.2f %9.2f %9.2f" % (g, kind, len(p), med([x[0] for x in p]), med([x[1] for x in p]),
                                                           med([x[2] for x in p]), med([x[3] for x in p])))
