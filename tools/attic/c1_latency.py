#!/usr/bin/env python
"""One-at-a-time latency of config C1 (10 k x 128, L2, k = 10) through the C-ABI, with the host-side split:
time inside tsh_search vs the ctypes call around it.  Run under `rocprofv3 --kernel-trace` + tools/trace_timeline.py
to see the device side (three kernels and the gaps between them)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from tostore_amd import HipVectorIndex

n, d, k = 10_000, 128, 10
rng = np.random.default_rng(1)
rows = rng.standard_normal((n, d)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
qs = rng.standard_normal((2000, d)).astype(np.float32)
with HipVectorIndex(d, 0) as idx:
    idx.append(0, rows)
    for q in qs[:200]:
        idx.search(q, k)
    lat = []
    for q in qs:
        t = time.perf_counter()
        idx.search(q, k)
        lat.append(time.perf_counter() - t)
    lat = np.sort(np.array(lat)) * 1e6
    print(f"C1 one at a time: p50 {lat[len(lat)//2]:.1f} us  p10 {lat[len(lat)//10]:.1f}  p90 {lat[len(lat)*9//10]:.1f}  p99 {lat[int(len(lat)*.99)]:.1f}")
