"""Round 6: one masked query at a time -- the mask as a pointer and as a handle (tsh_mask_create) -- on a 1 M x 768
corpus at keep 1 % / 0.2 %, and config C1's shape (10 k x 128, k = 10, no mask): p50 / p99 of the call as the caller
sees it, and microseconds per query of 64-query calls.  Under rocprofv3 --kernel-trace, tools/r6_lone_trace.py turns
the same run into the GPU's share of a lone query (E1, the gap, E2).  python tools/r6_lone_probe.py [--rounds 2]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def lat(idx, qs, k, mask, n=400):
    for i in range(20):
        idx.search(qs[i], k, None, mask)
    t = []
    for i in range(n):
        t0 = time.perf_counter()
        idx.search(qs[i % 512], k, None, mask)
        t.append((time.perf_counter() - t0) * 1e6)
    t = np.sort(t)
    return float(t[len(t) // 2]), float(t[int(len(t) * 0.99)])


def piped(idx, qs, k, mask, calls=16, per=64):
    idx.search(qs[:per], k, None, mask)
    t0 = time.perf_counter()
    for c in range(calls):
        idx.search(qs[(c * per) % 512:(c * per) % 512 + per], k, None, mask)
    return (time.perf_counter() - t0) / (calls * per) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    import gc

    import torch  # noqa: F401  (initialises its ROCm runtime first)

    from tostore_amd import HipVectorIndex

    gc.disable()
    rng = np.random.default_rng(1)
    d, k = 768, 100
    qs = rng.standard_normal((576, d)).astype(np.float32)
    with HipVectorIndex(d, 0, capacity_rows=a.rows) as idx:
        step = 100_000
        for lo in range(0, a.rows, step):
            idx.append(lo, rng.standard_normal((min(step, a.rows - lo), d)).astype(np.float32))
        idx.set_batch_min_nq(0)
        for keep in (0.01, 0.002):
            mask = np.packbits(rng.random(a.rows) < keep, bitorder="little")
            t0 = time.perf_counter()
            mh = idx.make_mask(mask)
            t_make = (time.perf_counter() - t0) * 1e6
            for r in range(a.rounds):
                for wide in (True, False):  # E2': the wide pick (shipped) | E2: round 5's one-workgroup select -- same box, alternating
                    idx.set_exact_select(wide)
                    for form, arg in (("pointer", mask), ("handle", mh)):
                        p50, p99 = lat(idx, qs, k, arg)
                        us = piped(idx, qs, k, arg)
                        print("keep %.1f %% %-7s %-6s: one at a time p50 %6.1f p99 %6.1f us; %6.1f us/query in 64-query calls%s"
                              % (keep * 100, form, "pick" if wide else "select", p50, p99, us,
                                 "  (handle made in %.0f us)" % t_make if form == "handle" else ""), flush=True)
            idx.set_exact_select(True)
            mh.close()
    d, k, n = 128, 10, 10_000
    qs = rng.standard_normal((576, d)).astype(np.float32)
    with HipVectorIndex(d, 0, capacity_rows=n) as idx:
        idx.append(0, rng.standard_normal((n, d)).astype(np.float32))
        idx.set_batch_min_nq(0)
        for r in range(a.rounds):
            for wide in (True, False):
                idx.set_exact_select(wide)
                p50, p99 = lat(idx, qs, k, None, 1000)
                print("C1 10k x 128 %-6s       : one at a time p50 %6.1f p99 %6.1f us; %6.1f us/query in 64-query calls"
                      % ("pick" if wide else "select", p50, p99, piped(idx, qs, k, None)), flush=True)


if __name__ == "__main__":
    main()
