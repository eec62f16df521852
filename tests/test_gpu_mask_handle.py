"""GPU: mask handles (tsh_mask_create / tsh_search_masked / tsh_search_submit_masked, include/tostore_hip.h) -- a WHERE
row set that is uploaded once, listed ON THE DEVICE when it is selective (tostore_amd/csrc/tsh_mask.hip.h) and stays
resident across queries.  The reference has no filter on this path; the oracle is the restated exact
path over the kept, live rows, and the pointer form of the same mask must give the same bits.  The handle's own edges
are here: reuse across calls, appends after it was made (the new rows are not kept), deletes after it was made (the
kernels check the live bitmap), every route a masked search can take (row by row, tile walk, list of f32 keys, exact
sums, matrix-core epilogue, tickets, several shards, a shard with an odd row base, quarantined rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L2, IP, COS = 0, 1, 2


def _q(oracle_mod, q, metric):
    q = np.asarray(q, np.float32)
    return oracle_mod.normalize_f32(q) if metric == COS else q


def _bits(keep):
    return np.packbits(np.asarray(keep, bool), bitorder="little")


def _same(a, b):
    return np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64))


def _check(idx, oracle_mod, rows, qs, metric, k, handle, eff, thr=None, base=0):
    """search with the handle against the oracle over the rows `eff` (bool per local row) leaves."""
    ids, dist, cnt = idx.search(qs, k, thr, handle)
    om = _bits(eff)
    for i in range(len(qs)):
        e, ed = oracle_mod.search_exhaustive(rows, qs[i], metric, k, thr, om)
        assert cnt[i] == len(e), (i, cnt[i], len(e))
        assert np.array_equal(ids[i, :cnt[i]], e + base), i
        assert np.array_equal(dist[i, :cnt[i]].view(np.uint64), ed.view(np.uint64)), i
    return ids, dist, cnt


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_every_route_of_a_masked_search(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(100 + metric)
    # 500 k x 100: wide enough for the list scan's own kernel (no packed width), big enough for 16 400 kept rows to be
    # fewer than one row in 24
    n, d, k = 500_000, 100, 50
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = np.stack([_q(oracle_mod, rng.standard_normal(d), metric) for _ in range(5)])
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        for kept, want_list, want_exact in ((600, True, True), (16_384, True, True), (20_000, True, False), (250_000, False, False)):
            keep = np.zeros(n, bool)
            keep[rng.choice(n, kept, replace=False)] = True
            with idx.make_mask(_bits(keep)) as m:
                assert m.kept == kept
                c0 = idx.counters()
                got = _check(idx, oracle_mod, rows, qs, metric, k, m, keep)
                one = _check(idx, oracle_mod, rows, qs[:1], metric, k, m, keep)  # a lone query, the same handle again
                c1 = idx.counters()
                assert c1["list_scans"] - c0["list_scans"] == (6 if want_list else 0), kept
                assert c1["exact_scans"] - c0["exact_scans"] == (6 if want_exact else 0), kept
                assert c1["fallback_searches"] == c0["fallback_searches"]
                assert _same(idx.search(qs, k, None, _bits(keep)), got), "pointer form vs handle"
                assert np.array_equal(one[0][0], got[0][0])
                # tickets with the handle
                ts = [idx.submit(q, k, m) for q in qs]
                for i, t in enumerate(ts):
                    ids, dist = idx.wait(t)
                    assert np.array_equal(ids, got[0][i, :got[2][i]]) and np.array_equal(dist, got[1][i, :got[2][i]])
                # the same handle through the matrix-core path (mask applied in its epilogue, the words read in place)
                idx.set_batch_min_nq(2)
                b0 = idx.counters()["batch_launches"]
                assert _same(idx.search(qs, k, None, m), got), "masked batch with a handle vs masked scans"
                assert idx.counters()["batch_launches"] > b0
                idx.set_batch_min_nq(0)
        # a one-range mask (WHERE id BETWEEN ...), a threshold, fewer kept rows than k, no kept row at all
        keep = np.zeros(n, bool)
        keep[123_457:123_457 + 9_000] = True
        thr = {L2: float(np.sqrt(2 * d) * 0.98), IP: -1.0, COS: 0.97}[metric]
        with idx.make_mask(_bits(keep)) as m:
            _check(idx, oracle_mod, rows, qs[:3], metric, k, m, keep, thr)
        keep = np.zeros(n, bool)
        keep[[3, 77_777, n - 1]] = True
        with idx.make_mask(_bits(keep)) as m:
            _check(idx, oracle_mod, rows, qs[:2], metric, k, m, keep)
        with idx.make_mask(np.zeros(0, np.uint8)) as m:  # no byte at all: nothing is kept
            assert m.kept == 0
            assert not idx.search(qs[:2], k, None, m)[2].any()
        assert idx.counters()["fallback_searches"] == 0


def test_appends_and_deletes_after_the_handle_was_made(hip_lib, oracle_mod):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(7)
    n0, n1, d, k = 30_000, 52_000, 96, 25
    rows = rng.standard_normal((n1, d)).astype(np.float32)
    qs = rng.standard_normal((4, d)).astype(np.float32)
    with HipVectorIndex(d, L2) as idx:  # (no capacity: the appends reallocate the row store under the handle)
        idx.append(0, rows[:n0])
        idx.set_batch_min_nq(0)
        keep = np.zeros(n1, bool)
        keep[rng.choice(n0, 700, replace=False)] = True
        with idx.make_mask(_bits(keep[:n0])) as m:
            _check(idx, oracle_mod, rows[:n0], qs, L2, k, m, keep[:n0])
            # rows appended later are beyond the bitmap: not kept, whatever they hold (these are copies of the queries)
            rows[n0:n0 + 4] = qs
            idx.append(n0, rows[n0:])
            assert m.kept == 700
            _check(idx, oracle_mod, rows, qs, L2, k, m, keep)
            # ... the pointer form of the same bitmap, zero-extended, says the same
            assert _same(idx.search(qs, k, None, _bits(keep)), idx.search(qs, k, None, m))
            # rows deleted later drop out (the kernels check the live bitmap; the list still names them)
            dead = np.flatnonzero(keep)[::3]
            idx.set_deleted(dead)
            eff = keep.copy()
            eff[dead] = False
            _check(idx, oracle_mod, rows, qs, L2, k, m, eff)
            # an overwrite of a kept row is seen (the handle holds ids, not rows)
            victim = int(np.flatnonzero(eff)[5])
            rows[victim] = qs[0]
            idx.append(victim, rows[victim:victim + 1])
            eff[victim] = True
            ids, dist, cnt = _check(idx, oracle_mod, rows, qs[:1], L2, k, m, eff)
            assert ids[0, 0] == victim and dist[0, 0] == 0.0
        # a handle wider than the index: bits beyond the last row wait for their rows
        wide = np.ones(n1 + 5000, bool)
        wide[:n1] = rng.random(n1) < 0.01
        with idx.make_mask(_bits(wide)) as m:
            alive = np.ones(n1, bool)
            alive[dead] = False
            _check(idx, oracle_mod, rows, qs, L2, k, m, wide[:n1] & alive)
            more = rng.standard_normal((300, d)).astype(np.float32)
            more[0] = qs[1]
            idx.append(n1, more)
            allrows = np.concatenate([rows, more])
            ids, dist, cnt = _check(idx, oracle_mod, allrows, qs, L2, k, m, np.r_[wide[:n1] & alive, np.ones(300, bool)])
            assert ids[1, 0] == n1 and dist[1, 0] == 0.0


def test_shard_base_quarantine_and_several_shards(hip_lib, oracle_mod):
    import os

    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(11)
    n, d, k = 70_000, 160, 20
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((3, d)).astype(np.float32)
    # a shard handle whose row base is no multiple of 8 (the global bitmap is shifted bit-wise into the shard's words),
    # with a row the f32 error model cannot cover (quarantined: matched against the handle's words on the host)
    base = 12_347
    rows[4_000, 7] = np.float32(3.0e20)
    keep = rng.random(n) < 0.02
    keep[4_000] = True
    gbits = _bits(np.r_[np.zeros(base, bool), keep])
    with HipVectorIndex(d, IP, capacity_rows=n, shard_device=0, row_base=base) as idx:
        idx.append(base, rows)
        idx.set_batch_min_nq(0)
        assert idx.counters()["quarantined_rows"] == 1
        with idx.make_mask(gbits) as m:
            got = _check(idx, oracle_mod, rows, qs, IP, k, m, keep, None, base)
            assert _same(idx.search(qs, k, None, gbits), got)
            keep2 = keep.copy()
            keep2[4_000] = False
            with idx.make_mask(_bits(np.r_[np.zeros(base, bool), keep2])) as m2:
                _check(idx, oracle_mod, rows, qs, IP, k, m2, keep2, None, base)
        # a handle of another index is refused
        with HipVectorIndex(d, IP) as other:
            other.append(0, rows[:100])
            with other.make_mask(np.full(13, 255, np.uint8)) as mo:
                with pytest.raises(_ffi.TshError) as e:
                    idx.search(qs, k, None, mo)
                assert e.value.code == _ffi.TSH_E_BAD_ARG
    # several shards in one handle (an in-process multi-GPU index; on this box they share the one GPU)
    rows[4_000, 7] = np.float32(0.5)
    os.environ["TSH_SHARDS_SHARE_DEVICES"] = "1"
    _ffi.enable_test_hooks()
    try:
        with HipVectorIndex(d, L2, capacity_rows=n, n_devices=3) as idx:
            idx.append(0, rows)
            idx.set_batch_min_nq(0)
            for frac in (0.01, 0.4):
                keep = rng.random(n) < frac
                with idx.make_mask(_bits(keep)) as m:
                    assert m.kept == int(keep.sum())
                    got = _check(idx, oracle_mod, rows, qs, L2, k, m, keep)
                    assert _same(idx.search(qs, k, None, _bits(keep)), got)
                    t = idx.submit(qs[0], k, m)
                    ids, dist = idx.wait(t)
                    assert np.array_equal(ids, got[0][0]) and np.array_equal(dist, got[1][0])
    finally:
        _ffi.enable_test_hooks(False)
        del os.environ["TSH_SHARDS_SHARE_DEVICES"]


def test_many_threads_share_one_handle(hip_lib, oracle_mod):
    import threading

    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(5)
    n, d, k = 200_000, 64, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((24, d)).astype(np.float32)
    keep = rng.random(n) < 0.03
    with HipVectorIndex(d, L2, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        with idx.make_mask(_bits(keep)) as m:
            want = idx.search(qs, k, None, _bits(keep))
            out, errs = [None] * 4, []

            def run(t):
                try:
                    for _ in range(5):
                        out[t] = idx.search(qs[t * 6:(t + 1) * 6], k, None, m)
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))

            th = [threading.Thread(target=run, args=(t,)) for t in range(4)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not errs, errs
            for t in range(4):
                assert _same(out[t], tuple(x[t * 6:(t + 1) * 6] for x in want))
