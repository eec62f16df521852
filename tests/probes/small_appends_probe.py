import sys, time, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.init()
from tostore_amd import HipVectorIndex
rng = np.random.default_rng(0)
d = 768
rows = rng.standard_normal((20000, d)).astype(np.float32)
for per in (1, 10, 100, 1000):
    with HipVectorIndex(d, 0) as idx:
        t = time.perf_counter()
        for s in range(0, 20000 if per >= 10 else 2000, per):
            idx.append(s, rows[s:s + per])
        n = idx.size
        dt = time.perf_counter() - t
        ids, dist, cnt = idx.search(rows[n - 1], 1)
        assert ids[0, 0] == n - 1
        print(f"appends of {per} rows: {dt / (n / per) * 1e6:.0f} us per call, {n / dt:.0f} rows/s")
