"""Probe: masked single-query scans at several selectivities, tile walk (TSH_LIST_DIV=0) against the list scan.
Run once per setting of TSH_LIST_DIV (read once per process): prints kernel microseconds (tsh_bench_scan), the
useful-bytes fraction of the HBM peak, and queries/s of 64-query calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from tostore_amd import HipVectorIndex  # noqa: E402

n, d, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("DIM", 768)), 100
g = torch.Generator(device="cuda")
g.manual_seed(1)
idx = HipVectorIndex(d, 0, capacity_rows=n)
for r0 in range(0, n, 131072):
    m = min(131072, n - r0)
    x = torch.randn((m, d), generator=g, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    idx.append_device(r0, m, x.data_ptr())
idx.set_batch_min_nq(0)
rng = np.random.default_rng(2)
qs = rng.standard_normal((256, d)).astype(np.float32)
print("TSH_LIST_DIV =", os.environ.get("TSH_LIST_DIV", "(default)"))
for keep, kind in ((0.002, "b"), (0.005, "b"), (0.01, "b"), (0.02, "b"), (0.03, "b"), (0.04, "b"), (0.06, "b"), (0.10, "b"), (0.01, "r"), (0.03, "r")):
    bits = np.zeros(n, bool)
    if kind == "r":
        s0 = n // 3
        bits[s0:s0 + int(n * keep)] = True
    else:
        bits = rng.random(n) < keep
    mask = np.packbits(bits, bitorder="little")
    kept = int(bits.sum())
    us = idx.bench_scan(qs[0], 50, mask)
    idx.search(qs[:64], k, None, mask)
    el = 1e9
    for rep in range(3):  # (the best of three: the first pass after a new mask may still grow the contexts' buffers)
        t0 = time.perf_counter()
        for i in range(0, 256, 64):
            idx.search(qs[i:i + 64], k, None, mask)
        el = min(el, time.perf_counter() - t0)
    useful = kept * d * 4 + n / 8
    print("keep %.3f %s: kept %7d  scan %7.1f us  useful %.3f of HBM peak  %7.0f queries/s" % (keep, kind, kept, us, useful / (us * 1e-6) / 8e12, 256 / el))
