"""Worker of tests/test_gpu_comm_rccl_multirank.py: one of W processes sharing the test box's ONE GPU, running the
library's sharded search over its RCCL branch -- tsh_comm_unique_id / tsh_comm_create / tsh_search_sharded with
ncclAllGather on device buffers, the pitched copy of this rank's query slice, the device-side result all-gather,
comm_agree's device path -- against tests/fake_rccl (TSH_RCCL_LIB), because the real library refuses two ranks on one
device.  No torch in here: the 128-byte id travels through a file, as a Dart host would ship it over its own channel.
Every rank checks the full answer against the oracle on the whole corpus.

argv: rows  id-file"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from tostore_amd import HipVectorIndex, _ffi  # noqa: E402
from tostore_amd.sharded import CommSearcher  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n, id_file, d = int(sys.argv[1]), sys.argv[2], 64
assert os.environ.get("TSH_RCCL_LIB"), "this worker is for the stand-in library only"
_ffi.enable_test_hooks()  # TSH_RCCL_LIB is obeyed only in a process that asked for the test hooks, before its first tsh_comm_* call
if os.environ.get("WORKER_EXCHANGE_AHEAD") == "1":  # (this script's own switch, not the library's: it makes the call)
    _ffi.check(_ffi.lib().tsh_index_set_option(None, _ffi.TSH_OPT_EXCHANGE_AHEAD, 1))


def say(what, ok):
    os.write(1, ("rank %d %s %s\n" % (rank, what, "ok" if ok else "MISMATCH")).encode())


def share_id(tag):
    """rank 0 makes the id and publishes it (write + rename: never seen half-written); the others pick it up"""
    path = "%s.%s" % (id_file, tag)
    if rank == 0:
        uid = CommSearcher.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise RuntimeError("rank 0 never published the communicator id")
        time.sleep(0.01)
    return open(path, "rb").read()


rng = np.random.default_rng(0)  # same corpus on every rank
rows = rng.standard_normal((n, d)).astype(np.float32)
rows[n // 2 - 1] = rows[n // 2] = rows[3]  # ties across a shard boundary -> global id order
qs_all = rng.standard_normal((300, d)).astype(np.float32)
keep = np.packbits(rng.random(n) < 0.3, bitorder="little")
per = (n + world - 1) // world
lo, hi = min(n, rank * per), min(n, (rank + 1) * per)


def check(got, qs, metric, k, mask=None, thr=None):
    ids, dd, cnt = got
    ok = True
    for i in range(len(qs)):
        e, ed = oracle.search_exhaustive(rows, qs[i], metric, k, thr, mask)
        ok &= bool(cnt[i] == len(e) and np.array_equal(ids[i, :cnt[i]], e) and np.array_equal(dd[i, :cnt[i]], ed))
    return ok


for metric in (0, 1, 2):
    qs = qs_all if metric != 2 else np.stack([oracle.normalize_f32(q) for q in qs_all])
    idx = HipVectorIndex(d, metric, capacity_rows=hi - lo, shard_device=0, row_base=lo)
    idx.append(lo, rows[lo:hi])
    cs = CommSearcher(idx, world, rank, share_id("m%d" % metric), 0)
    idx.set_batch_min_nq(0)
    say("m%d one query" % metric, check(cs.search(qs[0], 10), qs[:1], metric, 10))
    say("m%d fewer queries than ranks" % metric, check(cs.search(qs[:2], 7), qs[:2], metric, 7))
    say("m%d 37 queries in groups" % metric, check(cs.search(qs[:37], 10), qs[:37], metric, 10))
    thr = 1.2 if metric == 0 else (-0.5 if metric == 1 else 0.9)
    say("m%d masked + threshold" % metric, check(cs.search(qs[:9], 10, thr, keep), qs[:9], metric, 10, keep, thr))
    cs.set_group(5)  # 300 queries in 60 exchanges: the look-ahead pipeline at length; slices of 5 queries over W ranks
    say("m%d 300 queries, groups of 5" % metric, check(cs.search(qs, 3), qs, metric, 3))
    cs.set_group(7)  # a group size no world size here divides: ragged slices, empty slices on the last ranks
    say("m%d 100 queries, groups of 7" % metric, check(cs.search(qs[:100], 12), qs[:100], metric, 12))
    cs.set_group(0)
    idx.set_batch_min_nq(1)  # every rank's shard answers the call on the matrix cores
    say("m%d 300 queries batched" % metric, check(cs.search(qs, 10), qs, metric, 10))
    idx.set_batch_min_nq(0)
    # ---- the timeline of the calls so far: every phase was visited, and the calling thread's phases add up
    t = cs.timeline(reset=True)
    main = sum(t[p] for p in ("reserve_us", "pre_enqueue_us", "wait_scan_us", "exchange_wait_us", "merge_us", "result_gather_us",
                              "copy_out_us", "retry_scan_us"))
    say("m%d timeline (%d calls, %d groups, phases %.0f of %.0f us)" % (metric, t["calls"], t["groups"], main, t["call_us"]),
        t["calls"] == 7 and t["queries"] == 1 + 2 + 37 + 9 + 300 + 100 + 300 and t["groups"] >= 1 + 1 + 1 + 1 + 60 + 15 + 2
        and t["world"] == world and t["rank"] == rank and t["transport"] == "TSH_RCCL_LIB"
        and 0.9 * t["call_us"] <= main <= 1.001 * t["call_us"] and t["gather_us"] > 0 and t["scan_us"] > 0
        # (the stand-in's all-gather synchronises the stream inside the call: a pre-enqueued exchange's device time
        # then falls into pre_enqueue_us instead of exchange_wait_us)
        # (... and one exchange in four is timed and counted four times: a loose bound)
        and t["gather_us"] + t["slice_d2h_us"] <= (t["exchange_wait_us"] + t["pre_enqueue_us"]) * 4.2 + 200 * t["groups"])
    say("m%d timeline reset" % metric, cs.timeline()["calls"] == 0)
    # ---- a rank that fails locally stays in the collective: it gets its own error, the others TSH_E_PEER,
    # and the communicator keeps working
    try:
        cs.search(qs[:5], 10, shard=None if rank == 1 else ...)
        verdict = "no error"
    except _ffi.TshError as e:
        verdict = e.code
    say("m%d failing rank -> %s" % (metric, verdict), verdict == (_ffi.TSH_E_BAD_ARG if rank == 1 else _ffi.TSH_E_PEER))
    say("m%d usable after a failed call" % metric, check(cs.search(qs[:4], 10), qs[:4], metric, 10))
    # ... also when the call is cut into several groups: how it is cut must not depend on anything the failing rank
    # knows alone (its handle is NULL: no dimension, no rows)
    cs.set_group(7)
    try:
        cs.search(qs[:20], 10, shard=None if rank == world - 1 else ...)
        verdict = "no error"
    except _ffi.TshError as e:
        verdict = e.code
    cs.set_group(0)
    say("m%d failing rank, three groups -> %s" % (metric, verdict),
        verdict == (_ffi.TSH_E_BAD_ARG if rank == world - 1 else _ffi.TSH_E_PEER)
        and check(cs.search(qs[:20], 10), qs[:20], metric, 10))
    cs.close()
    idx.close()

# ---- ties wider than a block on ONE rank: every rank must retry the group with the same larger entry count
same = np.tile(rows[:1], (n, 1))
idx = HipVectorIndex(d, 0, capacity_rows=hi - lo, shard_device=0, row_base=lo)
idx.append(lo, same[lo:hi] if rank == world - 1 else rows[lo:hi])  # only the last shard is all ties
idx.set_batch_min_nq(0)
idx.set_exact_scan_rows(0)  # (a shard this small would answer from its exact sums: k rows, nothing to retry)
cs = CommSearcher(idx, world, rank, share_id("ties"), 0)
ref_rows = rows.copy()
lo_last = (world - 1) * per
ref_rows[lo_last:] = same[lo_last:]
ids, dd, cnt = cs.search(np.stack([rows[0], qs_all[1]]), 10)
e0, ed0 = oracle.search_exhaustive(ref_rows, rows[0], 0, 10)
e1, ed1 = oracle.search_exhaustive(ref_rows, qs_all[1], 0, 10)
say("overflow retry", bool(np.array_equal(ids[0], e0) and np.array_equal(dd[0], ed0) and np.array_equal(ids[1], e1)
                           and np.array_equal(dd[1], ed1)) and cs.timeline()["retries"] >= 1)
cs.close()

# ---- a failing all-gather (every rank's 3rd collective of a fresh communicator): TSH_E_RCCL on every rank, no hang
os.environ["TSH_FAKE_RCCL_FAIL_AT"] = "3"
cs = CommSearcher(idx, world, rank, share_id("fail"), 0)
del os.environ["TSH_FAKE_RCCL_FAIL_AT"]
try:
    cs.search(qs_all[:20], 5)
    verdict = "no error"
except _ffi.TshError as e:
    verdict = e.code
say("failing all-gather -> %s" % verdict, verdict == _ffi.TSH_E_RCCL)
cs.close()
idx.close()

# ---- shards big enough for the library's own schedule to cut a 20-query call into 10 + 5 + 5 (a scan of >= 30 us):
# every rank takes the same cut -- also the rank whose handle is NULL, which knows neither rows nor dimension
if os.environ.get("WORKER_BIG_SHARDS") == "1":
    nb = 800_000 * world
    big = np.random.default_rng(5).standard_normal((nb, d)).astype(np.float32)
    perb = nb // world
    idx = HipVectorIndex(d, 0, capacity_rows=perb, shard_device=0, row_base=rank * perb)
    idx.append(rank * perb, big[rank * perb:(rank + 1) * perb])
    idx.set_batch_min_nq(0)
    cs = CommSearcher(idx, world, rank, share_id("big"), 0)
    q20 = qs_all[:20]

    def check_big(got):
        ids, dd, cnt = got
        ok = True
        for i in (0, 9, 10, 14, 15, 19):  # a query of every group
            e, ed = oracle.search_exhaustive(big, q20[i], 0, 10)
            ok &= bool(cnt[i] == len(e) and np.array_equal(ids[i, :cnt[i]], e) and np.array_equal(dd[i, :cnt[i]], ed))
        return ok

    got = cs.search(q20, 10)
    t = cs.timeline(reset=True)
    say("big shards: 20 queries in %d groups" % t["groups"], t["groups"] == 3 and check_big(got))
    try:
        cs.search(q20, 10, shard=None if rank == world - 1 else ...)
        verdict = "no error"
    except _ffi.TshError as e:
        verdict = e.code
    say("big shards: failing rank, the library's own groups -> %s" % verdict,
        verdict == (_ffi.TSH_E_BAD_ARG if rank == world - 1 else _ffi.TSH_E_PEER))
    say("big shards: usable afterwards", check_big(cs.search(q20, 10)))
    cs.close()
    # ... and with ranks that send calls to their matrix cores (TSH_OPT_BATCH_MIN_NQ != 0, told to each other in the agreement) the
    # same 20 queries go as ONE group -- a group is a batched call per shard --, also when one rank has no handle to ask
    idx.set_batch_min_nq(1)
    cs = CommSearcher(idx, world, rank, share_id("bigb"), 0)
    b0 = idx.counters()["batch_launches"]
    got = cs.search(q20, 10)
    t = cs.timeline(reset=True)
    say("big shards, batching ranks: 20 queries in %d group(s)" % t["groups"],
        t["groups"] == 1 and check_big(got) and idx.counters()["batch_launches"] > b0)
    try:
        cs.search(q20, 10, shard=None if rank == world - 1 else ...)
        verdict = "no error"
    except _ffi.TshError as e:
        verdict = e.code
    say("big shards, batching ranks: failing rank -> %s" % verdict,
        verdict == (_ffi.TSH_E_BAD_ARG if rank == world - 1 else _ffi.TSH_E_PEER))
    say("big shards, batching ranks: usable afterwards", check_big(cs.search(q20, 10)))
    cs.close()
    idx.close()
