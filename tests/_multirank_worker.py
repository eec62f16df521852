"""Worker of tests/test_gpu_multirank.py: N ranks share the one GPU of the test box, exchange over
gloo, and compare the sharded answer (tostore_amd.sharded) with the oracle on the whole corpus."""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tostore_amd import HipVectorIndex
from tostore_amd.sharded import ShardedSearcher

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
n, d, k = int(sys.argv[1]), 64, 10
rng = np.random.default_rng(0)
rows = rng.standard_normal((n, d)).astype(np.float32)
qs = rng.standard_normal((4, d)).astype(np.float32)
per = (n + world - 1) // world
lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
idx = HipVectorIndex(d, 0, capacity_rows=hi - lo, shard_device=0, row_base=lo)
idx.append(lo, rows[lo:hi])
idx.set_batch_min_nq(0)
s = ShardedSearcher(idx)
for mode in ("search", "search_many"):
    if mode == "search":
        ids, dd, cnt = s.search(qs, k)
    else:
        ids, dd, cnt = s.search_many(qs, k, group=2)
    ok = True
    for i in range(len(qs)):
        e, ed = oracle.search_exhaustive(rows, qs[i], 0, k)
        ok &= bool(np.array_equal(ids[i, :cnt[i]], e) and np.array_equal(dd[i, :cnt[i]], ed))
        if not ok and rank == 0 and i == 0:
            print("got", ids[0], "want", e, flush=True)
    # one write() per line: the ranks share the parent's pipe and print() would interleave words
    os.write(1, ("rank %d %s %s size %d lo %d\n" % (rank, mode, "ok" if ok else "MISMATCH", idx.size, lo)).encode())
dist.barrier()
dist.destroy_process_group()
