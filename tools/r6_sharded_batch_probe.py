#!/usr/bin/env python
"""Where a 1024-query tsh_search_sharded call's time goes on one rank of C4's shape (1.25 M x 1536, inner product, k = 100; a world
of one over real RCCL): the communicator's timeline per call, and the same queries as one plain batched call on the shard.
  python tools/r6_sharded_batch_probe.py [rows=1250000] [dim=1536] [nq=1024]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from tostore_amd import HipVectorIndex  # noqa: E402
from tostore_amd.sharded import CommSearcher  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
k = 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(11)
idx = HipVectorIndex(d, 1, capacity_rows=n, shard_device=0, row_base=0)
step = 125_000
for c in range(0, n, step):
    m = min(step, n - c)
    x = torch.randn((m, d), generator=g, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    x *= torch.rand((m, 1), generator=g, device=dev) * 1.5 + 0.5
    torch.cuda.synchronize()
    idx.append_device(c, m, x.data_ptr())
    del x
qs = np.random.default_rng(12).standard_normal((nq, d)).astype(np.float32)
cs = CommSearcher(idx, 1, 0, CommSearcher.unique_id(), 0)
for _ in range(3):
    out = cs.search(qs, k)
cs.timeline(reset=True)
calls = 6
t = time.perf_counter()
for _ in range(calls):
    out = cs.search(qs, k)
dt = (time.perf_counter() - t) / calls
tl = cs.timeline()
print("tsh_search_sharded, %d queries: %.3f ms per call" % (nq, dt * 1e3))
for key, v in tl.items():
    if key.endswith("_us"):
        print("  %-18s %9.1f us per call" % (key, v / calls))
    else:
        print("  %-18s %s" % (key, v))
for grp in [int(x) for x in os.environ.get("GSWEEP", "").split(",") if x]:  # uniform groups of n (tsh_comm_set_group) instead of the schedule
    cs.set_group(grp)
    for _ in range(2):
        cs.search(qs, k)
    t = time.perf_counter()
    for _ in range(calls):
        o2 = cs.search(qs, k)
    print("  groups of %4d: %.3f ms per call; same answers: %s" % (grp, (time.perf_counter() - t) / calls * 1e3,
                                                              all(np.array_equal(a, b) for a, b in zip(out, o2))))
cs.set_group(0)
idx.set_batch_min_nq(2)
for _ in range(3):
    ref = idx.search(qs, k)
t = time.perf_counter()
for _ in range(calls):
    ref = idx.search(qs, k)
dt2 = (time.perf_counter() - t) / calls
print("tsh_search on the shard (device-finalised batch): %.3f ms per call; same answers: %s"
      % (dt2 * 1e3, all(np.array_equal(a, b) for a, b in zip(out, ref))))
cs.close()
idx.close()
