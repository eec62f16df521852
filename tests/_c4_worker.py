"""Worker of tests/test_gpu_full_size.py::test_c4_two_ranks_at_size_over_the_rccl_branch: one of two processes sharing
the test box's ONE GPU, each holding one REAL-SIZE row-range shard of BASELINE.json's C4 (1.25 M x 1536 f32 = 7.7 GB,
inner product, k = 100) and answering through the entry points a rank of the 8-GPU deployment uses -- tsh_comm_create +
tsh_search_sharded over the library's RCCL branch (tests/fake_rccl: real RCCL refuses two ranks on one device).  The
shard is generated on the device (torch, here only for that: the seed depends on the rank alone, so the parent test
regenerates the same rows for the oracle); the expected answers over both shards' 2.5 M rows come from the parent in
an .npz.  argv: refs.npz  id-file"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()  # (torch's ROCm runtime first: conftest.py hip_lib)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tostore_amd import HipVectorIndex, _ffi  # noqa: E402
from tostore_amd.sharded import CommSearcher  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
refs_file, id_file = sys.argv[1], sys.argv[2]
assert os.environ.get("TSH_RCCL_LIB"), "this worker is for the stand-in library only"
_ffi.enable_test_hooks()


def say(what, ok):
    os.write(1, ("rank %d %s %s\n" % (rank, what, "ok" if ok else "MISMATCH")).encode())


def c4_shard_rows(r, per, d, dev):
    """Shard r of the test's C4 corpus, on the device: unit directions scaled by U(0.5, 2), seeded by r alone (the same
    function lives in test_gpu_full_size.py)."""
    g = torch.Generator(device=dev)
    g.manual_seed(20260614 + 1000 * r)
    x = torch.empty((per, d), dtype=torch.float32, device=dev)
    for s in range(0, per, 65536):
        e = min(per, s + 65536)
        t = torch.randn((e - s, d), generator=g, device=dev)
        t /= t.norm(dim=1, keepdim=True)
        t *= torch.rand((e - s, 1), generator=g, device=dev) * 1.5 + 0.5
        x[s:e] = t
    torch.cuda.synchronize()
    return x


ref = np.load(refs_file)
qs, per, d, k = ref["queries"], int(ref["per"]), int(ref["dim"]), int(ref["k"])
dev = torch.device("cuda", 0)
x = c4_shard_rows(rank, per, d, dev)
idx = HipVectorIndex(d, 1, capacity_rows=per, shard_device=0, row_base=rank * per)
torch.cuda.synchronize()
idx.append_device(rank * per, per, x.data_ptr())
torch.cuda.synchronize()
del x
torch.cuda.empty_cache()

path = id_file
if rank == 0:
    uid = CommSearcher.unique_id()
    with open(path + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(path + ".tmp", path)
else:
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 300:
            raise RuntimeError("rank 0 never published the communicator id")
        time.sleep(0.01)
    uid = open(path, "rb").read()
cs = CommSearcher(idx, world, rank, uid, 0)


def same(got, nq):
    ids, dd, cnt = got
    return bool((cnt[:nq] == k).all() and np.array_equal(ids[:nq], ref["ids"][:nq]) and
                np.array_equal(dd[:nq].view(np.uint64), ref["dist"][:nq].view(np.uint64)))


nq = len(qs)
idx.set_batch_min_nq(0)  # every query scans this rank's 7.7 GB on its own; the library cuts the call into groups
got = cs.search(qs, k)
say("C4 two ranks: %d single-query scans per shard" % nq, same(got, nq))
t = cs.timeline(reset=True)
say("C4 two ranks: %d groups" % t["groups"], t["groups"] >= 2 and t["world"] == 2 and t["transport"] == "TSH_RCCL_LIB")
say("C4 two ranks: one query", same(cs.search(qs[:1], k), 1))
idx.set_batch_min_nq(2)  # the same call on every shard's matrix cores
got_b = cs.search(qs, k)
say("C4 two ranks: batched", same(got_b, nq) and idx.counters()["batch_launches"] >= 1)
say("C4 two ranks: no fallbacks", idx.counters()["fallback_searches"] == 0)
cs.close()
idx.close()
