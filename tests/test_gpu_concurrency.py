"""GPU: writers arriving while a caller pipelines asynchronous searches (the handle was
share-locked from submit to wait, a waiting append closed the gate for new readers, and the pipelining caller
-- who must submit before it waits -- hung together with the writer and every later searcher)."""
import threading
import time
from collections import deque

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipelined_submit_with_concurrent_append_and_delete(hip_lib, oracle_mod):
    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(5)
    n0, d, k = 20000, 64, 10
    rows = rng.standard_normal((n0 + 64 * 40, d)).astype(np.float32)
    qs = rng.standard_normal((400, d)).astype(np.float32)
    stop = threading.Event()
    errors, busy, appended = [], [0], [n0]

    with HipVectorIndex(d, 0, capacity_rows=len(rows)) as idx:
        idx.append(0, rows[:n0])

        def pipeline():  # submit-ahead-of-wait, depth 4: the documented use of the asynchronous API
            try:
                pend = deque()
                for i in range(len(qs)):
                    while True:
                        if len(pend) == 4:
                            idx.wait(pend.popleft())
                        try:
                            pend.append(idx.submit(qs[i], k))
                            break
                        except _ffi.TshError as e:
                            if e.code != _ffi.TSH_E_BUSY:
                                raise
                            busy[0] += 1  # a writer is waiting: drain, then submit again
                            while pend:
                                idx.wait(pend.popleft())
                    if i % 7 == 0:  # a synchronous call (and a size query) from the ticket holder itself
                        idx.search(qs[i], k)
                        assert idx.size >= n0
                while pend:
                    idx.wait(pend.popleft())
            except Exception as e:  # noqa: BLE001
                errors.append(e)
            finally:
                stop.set()

        def writer():
            try:
                step = 0
                while not stop.is_set() and appended[0] + 64 <= len(rows):
                    idx.append(appended[0], rows[appended[0]:appended[0] + 64])
                    appended[0] += 64
                    if step % 3 == 0:
                        idx.set_deleted([step])
                    step += 1
                    time.sleep(0.0005)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        def other_searcher():
            try:
                while not stop.is_set():
                    idx.search(qs[:3], k)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        th = [threading.Thread(target=f, daemon=True) for f in (pipeline, writer, other_searcher)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        hung = [t for t in th if t.is_alive()]
        assert not hung, "pipelined submit + concurrent append deadlocked"
        assert not errors, errors
        assert appended[0] > n0, "the writer never got in"
        # the handle is intact: an exhaustive check against the oracle on the final rows
        n = idx.size
        dead = np.zeros(n, bool)
        dead[[s for s in range(0, (appended[0] - n0) // 64, 3)]] = True
        ids, dist, cnt = idx.search(qs[0], k)
        eids, edist = oracle_mod.search_exhaustive(rows[:n], qs[0], 0, k, keep=np.packbits(~dead, bitorder="little"))
        assert np.array_equal(ids[0, :cnt[0]], eids) and np.array_equal(dist[0, :cnt[0]], edist)


def test_submit_reports_busy_while_a_writer_waits(hip_lib):
    """With a ticket open and an append waiting for it, a further submit answers TSH_E_BUSY instead of queueing
    behind the writer; a synchronous search from the same caller still goes through; after the wait the writer runs."""
    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(6)
    d = 32
    rows = rng.standard_normal((5000, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with HipVectorIndex(d, 0, capacity_rows=6000) as idx:
        idx.append(0, rows[:4000])
        t1 = idx.submit(q, 5)
        done = threading.Event()

        def writer():
            idx.append(4000, rows[4000:])
            done.set()

        w = threading.Thread(target=writer, daemon=True)
        w.start()
        deadline = time.time() + 5
        got_busy = False
        while time.time() < deadline and not got_busy:
            try:
                t2 = idx.submit(q, 5)
                idx.wait(t2)  # the writer had not reached the lock yet
                time.sleep(0.001)
            except _ffi.TshError as e:
                assert e.code == _ffi.TSH_E_BUSY
                got_busy = True
        assert got_busy and not done.is_set()
        ids, _, cnt = idx.search(q, 5)  # same caller, synchronous: passes the waiting writer
        assert cnt[0] == 5 and not done.is_set()
        idx.wait(t1)
        assert done.wait(10), "the writer did not run after the ticket was waited"
        w.join()
        assert idx.size == 5000


def test_batched_searches_with_a_concurrent_writer(hip_lib, oracle_mod):
    """Two threads send 96-query L2 calls to the matrix cores while a third appends rows past the handle's capacity: the fp16 copy
    of the rows, its norm-grouped order (perm / psq, reallocated with the row store) and the blocks it is sorted in are kept
    current under the handle's lock, call by call.  While the writer runs every answer must be SOUND (ids of rows that
    exist, ascending exact distances, k of them); once it is done the answers must be the oracle's."""
    import threading
    import time

    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(77)
    n0, n1, d, k, nq = 20_000, 60_000, 64, 20, 96
    rows = (rng.standard_normal((n1, d)) * rng.uniform(0.4, 2.5, (n1, 1))).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    with HipVectorIndex(d, 0) as idx:  # (no capacity given: the appends reallocate)
        idx.append(0, rows[:n0])
        idx.set_batch_min_nq(2)
        idx.set_batch_kernel(2)
        appended, errors, calls = [n0], [], [0, 0]
        stop = threading.Event()

        def reader(t):
            try:
                while not stop.is_set():
                    upto = appended[0]  # rows that certainly exist when the call starts
                    ids, dist, cnt = idx.search(qs, k)
                    after = appended[0]
                    assert (cnt == k).all() and (ids >= 0).all() and (ids < max(after, upto) + 1024).all()
                    assert (np.diff(dist, axis=1) >= 0).all()
                    calls[t] += 1
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        def writer():
            try:
                while appended[0] + 700 <= n1 and not errors:
                    idx.append(appended[0], rows[appended[0]:appended[0] + 700])
                    appended[0] += 700
                    time.sleep(0.002)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        th = [threading.Thread(target=reader, args=(0,)), threading.Thread(target=reader, args=(1,)), threading.Thread(target=writer)]
        for t in th:
            t.start()
        th[2].join(timeout=120)
        stop.set()
        for t in th[:2]:
            t.join(timeout=60)
        assert not any(t.is_alive() for t in th), "a thread hangs"
        assert not errors, errors[:3]
        assert min(calls) > 3 and appended[0] > n0 + 20_000, (calls, appended)
        c = idx.counters()
        assert c["batch_launches"] >= sum(calls) and c["fallback_searches"] == 0
        n = appended[0]
        ids, dist, cnt = idx.search(qs, k)
        ref = oracle_mod.search_heap_many_mt(rows[:n], qs, 0, k)
        assert np.array_equal(ids, ref[0]) and np.array_equal(dist.view(np.uint64), ref[1].view(np.uint64))
