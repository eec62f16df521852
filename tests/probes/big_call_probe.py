import sys, time, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.init()
import oracle
from tostore_amd import HipVectorIndex
rng = np.random.default_rng(0)
n, d, nq, k = 200000, 128, 30000, 10
rows = rng.standard_normal((n, d)).astype(np.float32)
qs = rng.standard_normal((nq, d)).astype(np.float32)
with HipVectorIndex(d, 0) as idx:
    idx.append(0, rows)
    idx.search(qs[:256], k)
    t = time.perf_counter(); ids, dist, cnt = idx.search(qs, k); dt = time.perf_counter() - t
    print(f"{nq} queries in one call: {dt*1e3:.1f} ms = {nq/dt:.0f} q/s", idx.counters()["batch_launches"], idx.counters()["fallback_searches"])
    for i in rng.integers(0, nq, 40):
        e, ed = oracle.search_heap(rows, qs[i], 0, k)
        assert np.array_equal(ids[i], e) and np.array_equal(dist[i], ed), i
    print("ok")
