import sys, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.init()
import oracle
from tostore_amd import HipVectorIndex
rng = np.random.default_rng(0)
for d in (260, 516, 772, 1028, 1100, 1280, 1284, 1540, 1700, 1792, 1796, 2048, 2052, 2304, 2500, 2560, 3072, 3076, 3500, 3584, 4000, 4073, 4096):
    n = 2000
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((3, d)).astype(np.float32)
    for metric in (0, 1, 2):
        with HipVectorIndex(d, metric) as idx:
            idx.set_batch_min_nq(0)
            idx.append(0, rows)
            ok = True
            for q in qs:
                qq = oracle.normalize_f32(q) if metric == 2 else q
                ids, dist, cnt = idx.search(qq, 10)
                e, ed = oracle.search_heap(rows, qq, metric, 10)
                ok &= bool(np.array_equal(ids[0], e) and np.array_equal(dist[0], ed))
            print("d", d, "metric", metric, "OK" if ok else "MISMATCH", idx.counters()["fallback_searches"], flush=True)
