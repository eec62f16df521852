#!/usr/bin/env python
"""Where the batched (matrix-core) path starts to beat pipelined single-query scans: time per call for
nq = 1..128 queries with the batch path forced on / off.  python tools/attic/batch_crossover.py [rows] [dim] [metric]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tostore_amd import HipVectorIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.randn((n, d), generator=g, device="cuda"); rows /= rows.norm(dim=1, keepdim=True)
idx = HipVectorIndex(d, metric, capacity_rows=n)
torch.cuda.synchronize(); idx.append_device(0, n, rows.data_ptr())
qs = np.random.default_rng(2).standard_normal((256, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
k = 100
for nq in (1, 2, 3, 4, 6, 8, 16, 32, 64, 128):
    res = []
    for min_nq in (0, 1):
        idx.set_batch_min_nq(min_nq)
        idx.search(qs[:nq], k); idx.search(qs[:nq], k)
        t = time.perf_counter()
        for r in range(5):
            a = idx.search(qs[r * 7 % 64: r * 7 % 64 + nq], k)
        res.append((time.perf_counter() - t) / 5 * 1e3)
    print(f"rows {n} d {d} metric {metric} nq {nq:4d}: pipelined scans {res[0]:7.3f} ms   batched {res[1]:7.3f} ms", flush=True)
