"""Probe: a batched search on a device filled with a ballast allocation (no pytest, no faulthandler)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
from tostore_amd import HipVectorIndex
leave = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(11)
n, d, k = 400_000, 256, 50
rows = rng.standard_normal((n, d)).astype(np.float32)
qs = rng.standard_normal((512, d)).astype(np.float32)
idx = HipVectorIndex(d, 0, capacity_rows=n)
idx.append(0, rows)
idx.set_batch_min_nq(0)
idx.search(qs[:64], k)
idx.set_batch_min_nq(2)
torch.cuda.synchronize()
free, _ = torch.cuda.mem_get_info()
print("free MB", free >> 20, flush=True)
b = torch.empty(max(0, free - (leave << 20)), dtype=torch.uint8, device="cuda")
print("free MB after ballast", torch.cuda.mem_get_info()[0] >> 20, flush=True)
got = idx.search(qs[:64], k)
print("counters", idx.counters(), flush=True)
