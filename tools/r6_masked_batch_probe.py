#!/usr/bin/env python
"""64-query masked calls on the matrix-core path (what bench.py's side.C5.*.library_default_path times), the fp16 plane
grouped by norm and in row order, alternating on one index: 1 M x 768, L2, norms U(0.5, 2), k = 100, Bernoulli masks.
  python tools/r6_masked_batch_probe.py [keeps=0.01,0.1,0.5] [calls=30] [queries per call=64]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from tostore_amd import HipVectorIndex  # noqa: E402

keeps = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0.01,0.1,0.5").split(",")]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, d, k, nq = 1_000_000, 768, 100, int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(5)
idx = HipVectorIndex(d, 0, capacity_rows=n)
for c in range(0, n, 125_000):
    x = torch.randn((125_000, d), generator=g, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    x *= torch.rand((125_000, 1), generator=g, device=dev) * 1.5 + 0.5
    torch.cuda.synchronize()
    idx.append_device(c, 125_000, x.data_ptr())
    del x
qs = np.random.default_rng(6).standard_normal((nq, d)).astype(np.float32)
qs /= np.linalg.norm(qs, axis=1, keepdims=True)
idx.set_batch_min_nq(2)
rng = np.random.default_rng(7)
for keep in keeps:
    bits = np.packbits(rng.random(n) < keep, bitorder="little")
    ref = None
    for rep in range(2):
        for grouped in (True, False):
            idx.set_batch_group(grouped)
            with idx.make_mask(bits) as m:
                for form, mm in (("pointer", bits), ("handle", m)):
                    for _ in range(3):
                        out = idx.search(qs, k, None, mm)
                    t = time.perf_counter()
                    for _ in range(calls):
                        out = idx.search(qs, k, None, mm)
                    dt = (time.perf_counter() - t) / calls
                    if ref is None:
                        ref = out
                    same = all(np.array_equal(a, b) for a, b in zip(out, ref))
                    c = idx.counters()
                    print("keep %5.2f %% %-9s %-7s: %7.1f us per call = %6.1f k queries/s  same=%s fallbacks=%d kernel=%d"
                          % (keep * 100, "grouped" if grouped else "row order", form, dt * 1e6, nq / dt / 1e3, same, c["fallback_searches"], c["batch_kernel_last"]), flush=True)
idx.close()
