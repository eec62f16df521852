"""Worker of tests/test_gpu_comm_multirank.py: N ranks share the one GPU of the test box and run the library's OWN
sharded search (tsh_search_sharded: groups, look-ahead scans, per-rank query slices, result all-gather, overflow
retry, error protocol) over the host transport (tsh_comm_create_host, gloo underneath) -- everything of the
N > 1 path except the RCCL call itself.  Every rank checks the full answer against the oracle on the whole corpus."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from tostore_amd import HipVectorIndex, _ffi  # noqa: E402
from tostore_amd.sharded import CommSearcher  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
n, d = int(sys.argv[1]), 64
rng = np.random.default_rng(0)  # same corpus on every rank
rows = rng.standard_normal((n, d)).astype(np.float32)
rows[n // 2 - 1] = rows[n // 2] = rows[3]  # ties across a shard boundary -> global id order
qs_all = rng.standard_normal((300, d)).astype(np.float32)
keep = np.packbits(rng.random(n) < 0.3, bitorder="little")
per = (n + world - 1) // world
lo, hi = min(n, rank * per), min(n, (rank + 1) * per)


def say(what, ok):
    # one write() per line: the ranks share the parent's pipe and print() would interleave words
    os.write(1, ("rank %d %s %s\n" % (rank, what, "ok" if ok else "MISMATCH")).encode())


def check(got, qs, metric, k, mask=None, thr=None):
    ids, dd, cnt = got
    ok = True
    for i in range(len(qs)):
        e, ed = oracle.search_exhaustive(rows, qs[i], metric, k, thr, mask)
        ok &= bool(cnt[i] == len(e) and np.array_equal(ids[i, :cnt[i]], e) and np.array_equal(dd[i, :cnt[i]], ed))
    return ok


for metric in (0, 2):
    qs = qs_all if metric == 0 else np.stack([oracle.normalize_f32(q) for q in qs_all])
    idx = HipVectorIndex(d, metric, capacity_rows=hi - lo, shard_device=0, row_base=lo)
    idx.append(lo, rows[lo:hi])
    cs = CommSearcher.over_torch(idx, device=0)
    idx.set_batch_min_nq(0)
    say("m%d one query" % metric, check(cs.search(qs[0], 10), qs[:1], metric, 10))
    say("m%d fewer queries than ranks" % metric, check(cs.search(qs[:2], 7), qs[:2], metric, 7))
    say("m%d 37 queries in groups" % metric, check(cs.search(qs[:37], 10), qs[:37], metric, 10))
    say("m%d masked + threshold" % metric,
        check(cs.search(qs[:9], 10, 1.2 if metric == 0 else 0.9, keep), qs[:9], metric, 10, keep, 1.2 if metric == 0 else 0.9))
    cs.set_group(5)  # 300 queries in 60 exchanges: the look-ahead pipeline at length
    say("m%d 300 queries, groups of 5" % metric, check(cs.search(qs, 3), qs, metric, 3))
    cs.set_group(0)
    idx.set_batch_min_nq(1)  # every rank's shard answers the call on the matrix cores
    say("m%d 300 queries batched" % metric, check(cs.search(qs, 10), qs, metric, 10))
    idx.set_batch_min_nq(0)
    # ---- a rank that fails locally stays in the collective: it gets its own error, the others TSH_E_PEER,
    # and the communicator keeps working
    try:
        cs.search(qs[:5], 10, shard=None if rank == 1 else ...)
        verdict = "no error"
    except _ffi.TshError as e:
        verdict = e.code
    say("m%d failing rank -> %s" % (metric, verdict), verdict == (_ffi.TSH_E_BAD_ARG if rank == 1 else _ffi.TSH_E_PEER))
    say("m%d usable after a failed call" % metric, check(cs.search(qs[:4], 10), qs[:4], metric, 10))
    cs.close()
    idx.close()

# ---- ties wider than a block on ONE rank: every rank must retry the group with the same larger entry count
same = np.tile(rows[:1], (n, 1))
idx = HipVectorIndex(d, 0, capacity_rows=hi - lo, shard_device=0, row_base=lo)
idx.append(lo, same[lo:hi] if rank == world - 1 else rows[lo:hi])  # only the last shard is all ties
idx.set_batch_min_nq(0)
idx.set_exact_scan_rows(0)  # (a shard this small would answer from its exact sums: k rows, nothing to retry)
cs = CommSearcher.over_torch(idx, device=0)
ref_rows = rows.copy()
lo_last = (world - 1) * per
ref_rows[lo_last:] = same[lo_last:]
ids, dd, cnt = cs.search(np.stack([rows[0], qs_all[1]]), 10)
e0, ed0 = oracle.search_exhaustive(ref_rows, rows[0], 0, 10)
e1, ed1 = oracle.search_exhaustive(ref_rows, qs_all[1], 0, 10)
say("overflow retry", bool(np.array_equal(ids[0], e0) and np.array_equal(dd[0], ed0) and np.array_equal(ids[1], e1)
                           and np.array_equal(dd[1], ed1)))
cs.close()
idx.close()
dist.barrier()
dist.destroy_process_group()
