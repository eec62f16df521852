import sys, time, threading, numpy as np, torch
sys.path.insert(0, ".")
from tostore_amd import HipVectorIndex
n, d, k = int(sys.argv[1]), 768, 100
torch.manual_seed(0)
rows = torch.randn(n, d, device="cuda"); rows /= rows.norm(dim=1, keepdim=True)
idx = HipVectorIndex(d, 0, capacity_rows=n)
torch.cuda.synchronize(); idx.append_device(0, n, rows.data_ptr()); idx.set_batch_min_nq(0)
qs = np.random.default_rng(1).standard_normal((4096, d)).astype(np.float32)
def work(t, T, reps):
    for r in range(reps):
        for g in range(t, 64, T):
            idx.search(qs[g * 64:(g + 1) * 64], k)
for T in (1, 2, 3, 4):
    work(0, 1, 1)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t, T, 2)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"rows {n} threads {T}: {2 * 4096 / dt:.0f} queries/s", flush=True)
