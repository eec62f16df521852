import sys, time, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.init()
from tostore_amd import HipVectorIndex
n, d = 1_000_000, 768
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.randn((n, d), generator=g, device="cuda")
idx = HipVectorIndex(d, 0, capacity_rows=n)
torch.cuda.synchronize(); idx.append_device(0, n, rows.data_ptr())
q = np.random.default_rng(2).standard_normal(d).astype(np.float32)
for k in (100, 1000, 1024, 1025, 2000, 5000, 20000):
    idx.search(q, k)
    t = time.perf_counter()
    for _ in range(3): ids, dist, cnt = idx.search(q, k)
    dt = (time.perf_counter() - t) / 3
    c = idx.counters()
    print(f"k={k}: {dt*1e3:.2f} ms, count {cnt[0]}, sorted {bool(np.all(np.diff(dist[0,:cnt[0]]) >= 0))}, fallbacks so far {c['fallback_searches']}", flush=True)
