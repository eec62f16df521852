#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd .db (kernel-trace / pmc) as text: per-kernel
calls, avg/min/max duration, share of GPU time; and per-kernel PMC averages.
Usage: rocpd_summary.py [--by-grid] results.db [...]
--by-grid: one line per (kernel, launch grid) -- a command that runs one kernel on corpora of several sizes (bench.py's
side legs scan 125 k-row and 10 k-row shards with the instantiation that scans the headline's 1 M rows) then shows each
size's average on its own line."""
import sqlite3
import sys


def main():
    by_grid = "--by-grid" in sys.argv[1:]
    for path in [a for a in sys.argv[1:] if a != "--by-grid"]:
        c = sqlite3.connect(path)
        print("==", path)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        name = "name" if "name" in cols else "kernel_name"
        if by_grid:
            gcol = next((g for g in ("grid_size", "grid_x", "grid_size_x", "grid") if g in cols), None)
            if gcol is None:
                print("(no grid column among", cols, ")")
            else:
                rows = c.execute(f"select {name}, {gcol}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                                 f"from kernels group by {name}, {gcol} order by 7 desc").fetchall()
                tot = sum(r[6] for r in rows) or 1
                print("%-64s %9s %7s %12s %12s %12s %7s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us", "time%"))
                for r in rows[:40]:
                    print("%-64s %9s %7d %12.2f %12.2f %12.2f %6.2f%%" % (r[0][:64], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                                          100.0 * r[6] / tot))
                continue
        rows = c.execute(f"select {name}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                         f"from kernels group by {name} order by 6 desc").fetchall()
        tot = sum(r[5] for r in rows) or 1
        print("%-72s %7s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "time%"))
        for r in rows:
            print("%-72s %7d %12.2f %12.2f %12.2f %6.2f%%" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                              100.0 * r[5] / tot))
        try:
            pm = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                           "group by kernel_name, counter_name order by 5 desc").fetchall()
            if pm:
                print("%-72s %-14s %7s %16s" % ("kernel", "counter", "calls", "avg_value"))
                for r in pm:
                    print("%-72s %-14s %7d %16.1f" % (r[0][:72], r[1], r[2], r[3]))
        except sqlite3.Error as e:
            print("(no counters:", e, ")")


if __name__ == "__main__":
    main()
