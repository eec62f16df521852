"""Round 5: short searches -- the exact path (tsh_exact.hip.h) against the f32 pre-filter on the same index, same box,
alternating: a 1 M x 768 corpus behind Bernoulli masks (keep 1 % / 0.2 % / 1.6 %) and config C1's shape (10 k x 128,
k = 10).  Per mode: microseconds per query of 64-query calls (pipelined single-query searches) and p50 / p99 of
searches one at a time.  python tools/attic/r5_exact_probe.py [--rows 1000000]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def timed(idx, qs, k, mask, calls, per):
    idx.search(qs[:per], k, None, mask)
    t0 = time.perf_counter()
    for c in range(calls):
        idx.search(qs[(c * per) % 512:(c * per) % 512 + per], k, None, mask)
    return (time.perf_counter() - t0) / (calls * per) * 1e6


def lat(idx, qs, k, mask, n=300):
    idx.search(qs[0], k, None, mask)
    t = []
    for i in range(n):
        t0 = time.perf_counter()
        idx.search(qs[i % 512], k, None, mask)
        t.append((time.perf_counter() - t0) * 1e6)
    t = np.sort(t)
    return float(t[len(t) // 2]), float(t[int(len(t) * 0.99)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import torch  # noqa: F401  (initialises its ROCm runtime first)

    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(1)
    d, k = 768, 100
    qs = rng.standard_normal((576, d)).astype(np.float32)
    with HipVectorIndex(d, 0, capacity_rows=a.rows) as idx:
        step = 100_000
        for lo in range(0, a.rows, step):
            idx.append(lo, rng.standard_normal((min(step, a.rows - lo), d)).astype(np.float32))
        idx.set_batch_min_nq(0)
        for keep in (0.01, 0.002, 0.016):
            mask = np.packbits(rng.random(a.rows) < keep, bitorder="little")
            for r in range(a.rounds):
                for mode, rows in (("exact", 16384), ("prefilter", 0)):
                    idx.set_exact_scan_rows(rows)
                    c0 = idx.counters()
                    us = timed(idx, qs, k, mask, 16, 64)
                    p50, p99 = lat(idx, qs, k, mask)
                    c1 = idx.counters()
                    print("keep %.1f %% %-9s: %6.1f us/query in 64-query calls; one at a time p50 %6.1f p99 %6.1f us  "
                          "(exact scans %d of %d)" % (keep * 100, mode, us, p50, p99, c1["exact_scans"] - c0["exact_scans"],
                                                      c1["scan_launches"] - c0["scan_launches"]), flush=True)
    d, k, n = 128, 10, 10_000
    qs = rng.standard_normal((576, d)).astype(np.float32)
    with HipVectorIndex(d, 0, capacity_rows=n) as idx:
        idx.append(0, rng.standard_normal((n, d)).astype(np.float32))
        idx.set_batch_min_nq(0)
        for r in range(a.rounds):
            for mode, rows in (("exact", 16384), ("prefilter", 0)):
                idx.set_exact_scan_rows(rows)
                us = timed(idx, qs, k, None, 16, 64)
                p50, p99 = lat(idx, qs, k, None, 1000)
                print("C1 10k x 128 %-9s: %6.1f us/query in 64-query calls; one at a time p50 %6.1f p99 %6.1f us" % (mode, us, p50, p99),
                      flush=True)
    # small shards of wider rows: where does the exact path stop paying?  (rows x 768, no mask)
    d, k = 768, 100
    qs = rng.standard_normal((576, d)).astype(np.float32)
    for n in (2000, 8000, 16384):
        with HipVectorIndex(d, 0, capacity_rows=n) as idx:
            idx.append(0, rng.standard_normal((n, d)).astype(np.float32))
            idx.set_batch_min_nq(0)
            for mode, rows in (("exact", 16384), ("prefilter", 0), ("exact", 16384), ("prefilter", 0)):
                idx.set_exact_scan_rows(rows)
                us = timed(idx, qs, k, None, 16, 64)
                p50, p99 = lat(idx, qs, k, None)
                print("%5d x 768 %-9s: %6.1f us/query in 64-query calls; one at a time p50 %6.1f p99 %6.1f us" % (n, mode, us, p50, p99),
                      flush=True)


if __name__ == "__main__":
    main()
