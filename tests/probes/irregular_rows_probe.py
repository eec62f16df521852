"""What a few IRREGULAR rows (NaN / inf / > 1e15 elements) cost a 1 M x 768 index: before the quarantine one such
row put the whole shard into safe mode (every search re-ranked every row, ~10 ms); now they are kept out of
the scan and re-ranked on the side.  Prints queries/s of pipelined single queries and of a 1024-query batch
for 0 / 1 / 10 / 100 / 1000 irregular rows, and checks the irregular rows really are among the answers when they
should be (IP: a +inf element with a positive query component gives distance -inf, the best there is)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.init()
from tostore_amd import HipVectorIndex
n, d, k = 1_000_000, 768, 100
metric = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator(device="cuda"); g.manual_seed(1)
rows = torch.randn((n, d), generator=g, device="cuda")
idx = HipVectorIndex(d, metric, capacity_rows=n)
torch.cuda.synchronize(); idx.append_device(0, n, rows.data_ptr())
rng = np.random.default_rng(2)
qs = rng.standard_normal((1024, d)).astype(np.float32)
if metric == 2:
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
done = 0
for total in (0, 1, 10, 100, 1000):
    while done < total:
        r = int(rng.integers(0, n))
        row = rng.standard_normal((1, d)).astype(np.float32)
        row[0, int(rng.integers(0, d))] = [np.nan, np.inf, 1e20][done % 3]
        idx.append(r, row)
        done += 1
    c = idx.counters()
    idx.set_batch_min_nq(0)
    idx.search(qs[:16], k)
    t = time.perf_counter()
    ids, dist, cnt = idx.search(qs[:256], k)
    single = 256 / (time.perf_counter() - t)
    t = time.perf_counter()
    for i in range(20): idx.search(qs[i], k)
    lat = (time.perf_counter() - t) / 20
    idx.set_batch_min_nq(1)
    idx.search(qs, k)
    t = time.perf_counter()
    idsb, distb, cntb = idx.search(qs, k)
    batch = 1024 / (time.perf_counter() - t)
    same = np.array_equal(ids, idsb[:256]) and np.array_equal(dist, distb[:256], equal_nan=True)
    c2 = idx.counters()
    print(f"metric {metric} irregular rows {total}: quarantined {c['quarantined_rows']} safe_mode {c['safe_mode']} | "
          f"pipelined {single:.0f} q/s, alone {lat*1e3:.3f} ms/query, batch of 1024 {batch:.0f} q/s | "
          f"single == batched {same}, fallbacks {c2['fallback_searches'] - c['fallback_searches']}, "
          f"min dist {np.nanmin(dist):.3g}", flush=True)
