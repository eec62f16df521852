"""GPU: short searches -- at most TSH_OPT_EXACT_SCAN_ROWS (16384) rows to look at: a selective mask's kept rows, a small
index or shard -- take the exact f64 sums of ALL those rows in one launch and select the k smallest exact distances
in a second (exact_scan_kernel + exact_select_kernel, tostore_amd/csrc/tsh_exact.hip.h) instead of the f32 pre-filter's
scan, select and re-rank.  Results are those of the oracle, bit for bit, like on every other path; the counters say
which path ran.  (The small-shape modules run both ways through conftest's scan_path; this module is about the exact
path's own edges.)"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L2, IP, COS = 0, 1, 2


def _q(oracle_mod, q, metric):
    q = np.asarray(q, np.float32)
    return oracle_mod.normalize_f32(q) if metric == COS else q


def _check(idx, oracle_mod, rows, qs, metric, k, mask=None, thr=None, base=0, eff=None):
    """search() against the oracle; eff: the rows the oracle may see (mask and tombstones), default = mask."""
    qs = np.atleast_2d(qs)
    ids, dist, cnt = idx.search(qs, k, thr, mask)
    om = eff if eff is not None else mask
    for i in range(len(qs)):
        e, ed = oracle_mod.search_exhaustive(rows, qs[i], metric, k, thr, om)
        assert cnt[i] == len(e), (i, cnt[i], len(e))
        assert np.array_equal(ids[i, :cnt[i]], e + base), i
        a, b = dist[i, :cnt[i]], ed
        assert np.array_equal(np.isnan(a), np.isnan(b)), i
        assert np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64)), i


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("wide", [True, False], ids=["pick", "select"])
@pytest.mark.parametrize("d", [1, 3, 5, 64, 127, 128, 129, 130, 200, 768, 1000, 2048])
def test_every_width_and_k(hip_lib, oracle_mod, metric, d, wide):
    """Pieces of 128 elements: widths below, at and across a piece, rows that end inside one, queries that ride in the
    kernel arguments (ld <= 960) and queries that do not."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(7 * d + metric)
    n = 3000 if d <= 768 else 1200
    rows = (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    qs = np.stack([_q(oracle_mod, rng.standard_normal(d), metric) for _ in range(3)])
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        idx.set_exact_select(wide)  # E2': the wide pick (default) / E2: the one-workgroup select
        c0 = idx.counters()
        for k in (1, 10, 100, n - 1, n, n + 5):
            _check(idx, oracle_mod, rows, qs, metric, k)
        c1 = idx.counters()
        assert c1["exact_scans"] - c0["exact_scans"] == c1["scan_launches"] - c0["scan_launches"] == 18
        assert c1["fallback_searches"] == 0
        # the winners are handed to the finaliser -- exactly them by the select, plus the few rows that share the cut bin
        # (1/256 of an octave of distance) by the pick: no band's worth of extra candidates either way
        extra = c1["candidates_total"] - c0["candidates_total"] - 3 * (1 + 10 + 100 + (n - 1) + n + n)
        assert extra == 0 if not wide else 0 <= extra <= 18 * 60, extra
        # (one-element rows under cosine are at distance 0 or 2, all of them: ties by the thousand, which the pick hands to
        # the select -- the only shape here that does)
        assert c1["exact_redone"] - c0["exact_redone"] == (9 if wide and d == 1 and metric == COS else 0)
        idx.set_exact_scan_rows(0)  # the same answers from the pre-filter, and the counter stands still
        _check(idx, oracle_mod, rows, qs, metric, 10)
        assert idx.counters()["exact_scans"] == c1["exact_scans"]


@pytest.mark.usefixtures("mask_form")  # the mask as a pointer, and as a device-resident handle
@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_where_the_path_ends(hip_lib, oracle_mod, metric):
    """16384 rows are the last size the exact path takes (2048 waves of eight rows there); one more row, or a lower
    TSH_OPT_EXACT_SCAN_ROWS, and the pre-filter answers."""
    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(40 + metric)
    d, k = 96, 25
    rows = rng.standard_normal((16385, d)).astype(np.float32)
    qs = np.stack([_q(oracle_mod, rng.standard_normal(d), metric) for _ in range(2)])
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows[:16384])
        idx.set_batch_min_nq(0)
        _check(idx, oracle_mod, rows[:16384], qs, metric, k)
        assert idx.counters()["exact_scans"] == 2
        idx.set_exact_scan_rows(16383)
        _check(idx, oracle_mod, rows[:16384], qs, metric, k)
        assert idx.counters()["exact_scans"] == 2
        idx.set_exact_scan_rows(16384)
        idx.append(16384, rows[16384:])
        _check(idx, oracle_mod, rows, qs, metric, k)
        assert idx.counters()["exact_scans"] == 2
        # a mask that keeps few enough of them brings the path back (as a list, or row by row when it is no list)
        keep = np.zeros(16385, bool)
        keep[rng.choice(16385, 300, replace=False)] = True
        _check(idx, oracle_mod, rows, qs, metric, k, np.packbits(keep, bitorder="little"))
        assert idx.counters()["exact_scans"] == 4
        for bad in (-1, 16385):
            with pytest.raises(_ffi.TshError) as e:
                idx.set_exact_scan_rows(bad)
            assert e.value.code == _ffi.TSH_E_BAD_ARG


@pytest.mark.usefixtures("mask_form")  # the mask as a pointer, and as a device-resident handle
@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_masks_tombstones_and_lists(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(60 + metric)
    # a small index: the mask is tested row by row (no list below 4096 rows)
    n, d, k = 3500, 100, 20
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = np.stack([_q(oracle_mod, rng.standard_normal(d), metric) for _ in range(4)])
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        alive = np.ones(n, bool)
        dead = rng.choice(n, 700, replace=False)
        idx.set_deleted(dead)
        alive[dead] = False
        _check(idx, oracle_mod, rows, qs, metric, k, None, eff=np.packbits(alive, bitorder="little"))
        for keep_frac in (0.5, 0.02, 0.001):
            keep = rng.random(n) < keep_frac
            _check(idx, oracle_mod, rows, qs, metric, k, np.packbits(keep, bitorder="little"),
                   eff=np.packbits(keep & alive, bitorder="little"))
        none = np.zeros(n, bool)
        _check(idx, oracle_mod, rows, qs[:1], metric, k, np.packbits(none, bitorder="little"))
        c = idx.counters()
        assert c["exact_scans"] == c["scan_launches"] and c["fallback_searches"] == 0
    # a big index: the kept rows as a list (below one kept row in 24); more than 16384 of them go through the f32 keys
    n, d = 400_000, 72
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = np.stack([_q(oracle_mod, rng.standard_normal(d), metric) for _ in range(3)])
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        for kept, exact in ((2000, True), (15_000, True), (16_384, True), (16_400, False)):
            keep = np.zeros(n, bool)
            keep[rng.choice(n, kept, replace=False)] = True
            m = np.packbits(keep, bitorder="little")
            c0 = idx.counters()
            _check(idx, oracle_mod, rows, qs, metric, 100, m)
            c1 = idx.counters()
            assert c1["list_scans"] - c0["list_scans"] == 3  # (16 400 of 400 000: below one row in 24, a list of f32 keys)
            assert c1["exact_scans"] - c0["exact_scans"] == (3 if exact else 0), kept
        # a block too small for k rows: the list is scanned for f32 keys as before
        idx.set_exact_scan_rows(1000)
        c0 = idx.counters()
        keep = np.zeros(n, bool)
        keep[rng.choice(n, 2000, replace=False)] = True
        _check(idx, oracle_mod, rows, qs, metric, 100, np.packbits(keep, bitorder="little"))
        c1 = idx.counters()
        assert c1["list_scans"] - c0["list_scans"] == 3 and c1["exact_scans"] == c0["exact_scans"]
        idx.set_exact_scan_rows(16384)
        # rows deleted after the mask was made stay out
        keep = np.zeros(n, bool)
        keep[100_000:103_000] = True
        m = np.packbits(keep, bitorder="little")
        dead = np.arange(100_000, 103_000, 3)
        idx.set_deleted(dead)
        keep[dead] = False
        _check(idx, oracle_mod, rows, qs, metric, 100, m, eff=np.packbits(keep, bitorder="little"))
        assert idx.counters()["fallback_searches"] == 0


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_ties_take_the_lowest_ids(hip_lib, oracle_mod, metric):
    """The k-th distance tied across more rows than k asks for: the finaliser's order is (distance, id), so the
    select takes the lowest positions of the tie -- whole index tied, a tie that straddles k, NaN distances (a NaN
    query: every distance NaN, all equal under double.compareTo)."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(80 + metric)
    d = 48
    base = rng.standard_normal((40, d)).astype(np.float32)
    rows = base[rng.integers(0, 40, 9000)]  # 40 distinct rows, ~225 copies each
    q = _q(oracle_mod, base[3] + 0.01, metric)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        for k in (1, 7, 225, 226, 500, 8999, 9000):
            _check(idx, oracle_mod, rows, q, metric, k)
        keep = rng.random(9000) < 0.3
        _check(idx, oracle_mod, rows, q, metric, 300, np.packbits(keep, bitorder="little"))
        qn = q.copy()
        qn[5] = np.nan
        ids, dist, cnt = idx.search(qn, 10)  # (cosine: a NaN denominator is "not positive": similarity 0, distance 1)
        assert cnt[0] == 10 and ids[0].tolist() == list(range(10))
        assert (dist[0] == 1.0).all() if metric == COS else np.isnan(dist[0]).all()
        _check(idx, oracle_mod, rows, qn, metric, 10)
        c = idx.counters()
        assert c["exact_scans"] == c["scan_launches"] and c["fallback_searches"] == 0
    same = np.tile(base[:1], (6000, 1))
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, same)
        _check(idx, oracle_mod, same, q, metric, 100)
        _check(idx, oracle_mod, same, np.zeros(d, np.float32), metric, 100)  # cosine: denominator 0 -> similarity 0
        assert idx.counters()["exact_scans"] == 2


def test_threshold_zero_rows_and_signed_zero(hip_lib, oracle_mod):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(5)
    d, n = 32, 2000
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[10] = 0.0
    rows[11] = -0.0
    rows[12] = rows[13] = rows[1500]
    for metric in (L2, IP, COS):
        with HipVectorIndex(d, metric) as idx:
            idx.append(0, rows)
            for q in (rows[1500], np.zeros(d, np.float32), -rows[7]):
                qq = _q(oracle_mod, q, metric)
                _, ed = oracle_mod.search_exhaustive(rows, qq, metric, 30, None, None)
                for thr in (None, float(ed[10]), float(ed[0]), float(np.nextafter(ed[0], -np.inf))):
                    _check(idx, oracle_mod, rows, qq, metric, 30, None, thr)
            assert idx.counters()["exact_scans"] == idx.counters()["scan_launches"]


def test_quarantined_rows_join_the_exact_blocks(hip_lib, oracle_mod):
    """Rows outside the f32 error model are kept out of every scan (their live bit is clear) and added with their own
    exact sums -- the exact path leaves that as it is."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(9)
    d, n = 40, 5000
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[17, 3] = np.inf
    rows[4000, 0] = np.nan
    rows[4999, 39] = 1e20
    for metric in (L2, IP, COS):
        with HipVectorIndex(d, metric) as idx:
            idx.append(0, rows)
            idx.set_batch_min_nq(0)
            qs = np.stack([_q(oracle_mod, rng.standard_normal(d), metric) for _ in range(3)])
            for k in (5, n):
                _check(idx, oracle_mod, rows, qs, metric, k)
            keep = rng.random(n) < 0.4
            keep[[17, 4999]] = True
            _check(idx, oracle_mod, rows, qs, metric, n, np.packbits(keep, bitorder="little"))
            c = idx.counters()
            assert c["quarantined_rows"] == 3 and c["safe_mode"] == 0 and c["exact_scans"] == c["scan_launches"] == 9


@pytest.mark.parametrize("wide", [True, False], ids=["pick", "select"])
def test_shard_blocks_hold_the_k_winners(hip_lib, oracle_mod, wide):
    """Shard mode: the block a rank offers holds min(k, live rows) entries, flagged exact, with global ids; merged like
    any other block."""
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    L = _ffi.lib()
    rng = np.random.default_rng(11)
    d, n, k, base = 72, 9000, 50, 123_456
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((5, d)).astype(np.float32)
    with HipVectorIndex(d, L2, shard_device=0, row_base=base) as s:
        s.append(base, rows)
        s.set_batch_min_nq(0)
        s.set_exact_select(wide)
        entries = L.tsh_default_block_entries(k)
        bb = L.tsh_candidate_block_bytes(entries)
        for mask_rows in (None, 30):
            bits = None
            mask = None
            if mask_rows:
                bits = np.zeros(n, bool)
                bits[rng.choice(n, mask_rows, replace=False)] = True
                mask = np.packbits(np.concatenate([np.zeros(base, bool), bits]), bitorder="little")
            buf = torch.zeros(5 * bb, dtype=torch.uint8, device="cuda")
            _ffi.check(L.tsh_search_shard(s._h, qs.ctypes.data_as(_ffi.p_f32), 5, k,
                                          None if mask is None else mask.ctypes.data_as(_ffi.p_u8), entries,
                                          ctypes.c_void_p(buf.data_ptr()), None))
            blk = buf.cpu().numpy()
            hdr = blk.reshape(5, bb)[:, :64].view(np.uint32)
            if mask_rows or not wide:
                assert (hdr[:, 0] == (k if not mask_rows else mask_rows)).all()
            else:  # the pick: the k winners and whatever shares the cut bin with the k-th
                assert (hdr[:, 0] >= k).all() and (hdr[:, 0] <= k + 60).all()
            assert (hdr[:, 1] == entries).all()
            assert ((hdr[:, 5] & 8) == 8).all() and ((hdr[:, 5] & 1) == 0).all()  # FLAG_EXACT, no overflow
            ids, dist, cnt = merge_candidate_blocks(L2, d, qs, k, None, blk, 1, entries)
            om = None if bits is None else np.packbits(bits, bitorder="little")
            for i in range(5):
                e, ed = oracle_mod.search_exhaustive(rows, qs[i], L2, k, None, om)
                assert cnt[i] == len(e) and np.array_equal(ids[i, :cnt[i]], e + base) and np.array_equal(dist[i, :cnt[i]], ed)
        # a block too small for k rows cannot take the exact path: the pre-filter answers and reports what it needs
        c0 = s.counters()
        small = 16
        buf = torch.zeros(L.tsh_candidate_block_bytes(small), dtype=torch.uint8, device="cuda")
        _ffi.check(L.tsh_search_shard(s._h, qs.ctypes.data_as(_ffi.p_f32), 1, k, None, small,
                                      ctypes.c_void_p(buf.data_ptr()), None))
        assert s.counters()["exact_scans"] == c0["exact_scans"]
        with pytest.raises(_ffi.TshError) as e:
            merge_candidate_blocks(L2, d, qs[:1], k, None, buf.cpu().numpy(), 1, small)
        assert e.value.code == _ffi.TSH_E_OVERFLOW


def test_ticket_api_and_threads(hip_lib, oracle_mod):
    """submit / wait and several threads on one handle: contexts (and their exact scratch) are per query in flight."""
    import threading

    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(13)
    d, n, k = 128, 10_000, 10  # config C1's shape
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((64, d)).astype(np.float32)
    want = [oracle_mod.search_exhaustive(rows, q, L2, k, None, None) for q in qs]
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        tickets = [idx.submit(q, k) for q in qs[:8]]
        for t, (e, ed) in zip(tickets, want[:8]):
            ids, dist = idx.wait(t)
            assert len(ids) == k and np.array_equal(ids, e) and np.array_equal(dist, ed)
        errs = []

        def work(lo):
            try:
                for i in range(lo, lo + 16):
                    ids, dist, cnt = idx.search(qs[i], k)
                    assert np.array_equal(ids[0, :cnt[0]], want[i][0]) and np.array_equal(dist[0, :cnt[0]], want[i][1])
            except Exception as ex:  # noqa: BLE001
                errs.append(ex)

        th = [threading.Thread(target=work, args=(lo,)) for lo in (0, 16, 32, 48)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        ids, dist, cnt = idx.search(qs, k)  # one call of 64 queries, pipelined
        for i in range(64):
            assert np.array_equal(ids[i, :cnt[i]], want[i][0]) and np.array_equal(dist[i, :cnt[i]], want[i][1])
        c = idx.counters()
        assert c["exact_scans"] == c["scan_launches"] == 8 + 64 + 64 and c["fallback_searches"] == 0


@pytest.mark.usefixtures("mask_form")
@pytest.mark.parametrize("d", [32, 64, 128])
def test_narrow_rows_get_a_list_for_the_exact_path_only(hip_lib, oracle_mod, d):
    """Rows of 32 / 64 / 128 floats are scanned by the packed kernels, which have no list form: a selective mask walks
    their tiles -- unless it keeps few enough rows for the exact path, which reads nothing but the list, also where
    one row in 24 is exceeded (10 000 of 100 000)."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(d)
    n, k = 100_000, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((3, d)).astype(np.float32)
    with HipVectorIndex(d, L2, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        for kept, listed in ((500, True), (10_000, True), (16_384, True), (17_000, False)):
            keep = np.zeros(n, bool)
            keep[rng.choice(n, kept, replace=False)] = True
            c0 = idx.counters()
            _check(idx, oracle_mod, rows, qs, L2, k, np.packbits(keep, bitorder="little"))
            c1 = idx.counters()
            assert c1["list_scans"] - c0["list_scans"] == c1["exact_scans"] - c0["exact_scans"] == (3 if listed else 0), kept
        assert idx.counters()["fallback_searches"] == 0


@pytest.mark.usefixtures("mask_form")
def test_ticket_api_builds_a_list_for_selective_masks(hip_lib, oracle_mod):
    """tsh_search_submit slices the mask itself: a selective one gets its list (and with it the exact path) like in
    tsh_search."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(17)
    n, d, k = 120_000, 96, 30
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((6, d)).astype(np.float32)
    with HipVectorIndex(d, L2, capacity_rows=n) as idx:
        idx.append(0, rows)
        for kept, listed, exact in ((1500, True, True), (4000, True, True), (30_000, False, False)):
            keep = np.zeros(n, bool)
            keep[rng.choice(n, kept, replace=False)] = True
            m = np.packbits(keep, bitorder="little")
            c0 = idx.counters()
            tickets = [idx.submit(q, k, m) for q in qs]
            for t, q in zip(tickets, qs):
                ids, dist = idx.wait(t)
                e, ed = oracle_mod.search_exhaustive(rows, q, L2, k, None, m)
                assert np.array_equal(ids, e) and np.array_equal(dist, ed)
            c1 = idx.counters()
            assert c1["list_scans"] - c0["list_scans"] == (6 if listed else 0), kept
            assert c1["exact_scans"] - c0["exact_scans"] == (6 if exact else 0), kept
        assert idx.counters()["fallback_searches"] == 0
