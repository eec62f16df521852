"""GPU parity with IRREGULAR rows in the corpus -- rows the f32 pre-filter's error model cannot cover:
non-finite elements, elements beyond 1e15, and (cosine) norms below 2^-50.

The reference has no such distinction: its f64 arithmetic simply produces inf / NaN / 0 distances for them
(ngh_graph_engine.dart:908-946) and `double.compareTo` orders those (NaN last).  The library keeps such rows out
of the scan (quarantine) and re-ranks them exactly on every search, so results stay identical to the oracle
and the rest of the shard keeps its fast path (counters: safe_mode 0, fallback_searches 0).
"""
import ctypes

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]  # small shapes: exact sums AND the f32 pre-filter (conftest)
L2, IP, COS = 0, 1, 2
METRICS = [L2, IP, COS]


def _rows(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


def _spoil(rows, metric):
    """a corpus with every kind of irregular row; returns the ids the library should quarantine"""
    d = rows.shape[1]
    rows[5, 3] = np.nan
    rows[64] = np.nan
    rows[700, 0] = np.inf
    rows[701, d - 1] = -np.inf
    rows[1500, 2] = 3e20   # finite, but its square overflows the f32 key
    rows[1501] = -2e16
    rows[4000] = 0.0
    rows[4000, 1] = 1e-30  # tiny norm: only cosine cannot bound it
    rows[4001] = 0.0       # a plain zero row is regular for every metric
    bad = [5, 64, 700, 701, 1500, 1501]
    if metric == COS:
        bad.append(4000)
    return bad


def _q(oracle, d, seed, metric):
    q = _rows(1, d, seed)[0]
    return oracle.normalize_f32(q) if metric == COS else q


def _same(ids, dist, cnt, eids, edist, tag):
    n = int(cnt)
    assert n == len(eids), f"{tag}: count {n} != {len(eids)}"
    assert np.array_equal(ids[:n], eids), f"{tag}: ids differ\n{ids[:n]}\n{eids}"
    assert np.array_equal(dist[:n].view(np.uint64), edist.view(np.uint64)) or (
        np.array_equal(np.isnan(dist[:n]), np.isnan(edist))
        and np.array_equal(dist[:n][~np.isnan(edist)], edist[~np.isnan(edist)])), f"{tag}: distances differ"


@pytest.mark.parametrize("metric", METRICS)
def test_irregular_rows_are_quarantined_not_safe_mode(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    n, d = 9000, 48
    rows = _rows(n, d, 1)
    bad = _spoil(rows, metric)
    rng = np.random.default_rng(2)
    with HipVectorIndex(d, metric) as idx:
        idx.set_batch_min_nq(2)
        idx.append(0, rows[:3000])
        idx.append(3000, rows[3000:])
        c = idx.counters()
        assert c["safe_mode"] == 0 and c["quarantined_rows"] == len(bad)
        for qi in range(3):  # the point of the quarantine: ordinary searches stay on the fast path
            q = _q(oracle_mod, d, 50 + qi, metric)
            ids, dist, cnt = idx.search(q, 10)
            eids, edist = oracle_mod.search_exhaustive(rows, q, metric, 10)
            _same(ids[0], dist[0], cnt[0], eids, edist, f"m{metric} fast")
        assert idx.counters()["fallback_searches"] == 0
        keep_bits = rng.random(n) < 0.5
        keep_bits[[5, 700, 1501]] = True
        keep_bits[[64, 701, 1500]] = False
        keep = np.packbits(keep_bits, bitorder="little")
        for qi in range(3):
            q = _q(oracle_mod, d, 10 + qi, metric)
            for k in (1, 10, 100, n - 3, n):
                for mask in (None, keep):
                    ids, dist, cnt = idx.search(q, k, None, mask)
                    eids, edist = oracle_mod.search_exhaustive(rows, q, metric, k, None, mask)
                    _same(ids[0], dist[0], cnt[0], eids, edist, f"m{metric} k{k} mask{mask is not None}")
            # thresholds: NaN distances are never `> thr`, so NaN rows pass any threshold (reference :127)
            _, e100 = oracle_mod.search_exhaustive(rows, q, metric, 100)
            for thr in (float(e100[40]), -1e300, 1e300):
                ids, dist, cnt = idx.search(q, n, thr)
                eids, edist = oracle_mod.search_exhaustive(rows, q, metric, n, thr)
                _same(ids[0], dist[0], cnt[0], eids, edist, f"m{metric} thr{thr}")
        # asynchronous form
        q = _q(oracle_mod, d, 20, metric)
        t1, t2 = idx.submit(q, 25), idx.submit(q, 25, keep)
        for t, mask in ((t1, None), (t2, keep)):
            ids, dist = idx.wait(t)
            eids, edist = oracle_mod.search_exhaustive(rows, q, metric, 25, None, mask)
            _same(ids, dist, len(ids), eids, edist, f"m{metric} async")
        # batched path (matrix cores): every query gets the quarantined rows added
        qs = np.stack([_q(oracle_mod, d, 30 + i, metric) for i in range(40)])
        before = idx.counters()["batch_launches"]
        for mask in (None, keep):
            for k in (10, 200):
                ids, dist, cnt = idx.search(qs, k, None, mask)
                for i in range(len(qs)):
                    eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, None, mask)
                    _same(ids[i], dist[i], cnt[i], eids, edist, f"m{metric} batch q{i} k{k}")
        assert idx.counters()["batch_launches"] > before


@pytest.mark.parametrize("metric", METRICS)
def test_quarantine_follows_deletes_and_overwrites(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    n, d = 5000, 32
    rows = _rows(n, d, 3)
    bad = _spoil(rows, metric)
    q = _q(oracle_mod, d, 4, metric)
    qs = np.stack([_q(oracle_mod, d, 40 + i, metric) for i in range(16)])
    with HipVectorIndex(d, metric) as idx:
        idx.set_batch_min_nq(2)
        idx.append(0, rows)
        alive = np.ones(n, bool)

        def check(tag):
            keep = np.packbits(alive, bitorder="little")
            for k in (7, n):
                ids, dist, cnt = idx.search(q, k)
                eids, edist = oracle_mod.search_exhaustive(rows, q, metric, k, None, keep)
                _same(ids[0], dist[0], cnt[0], eids, edist, f"{tag} k{k}")
            ids, dist, cnt = idx.search(qs, 20)
            for i in range(len(qs)):
                eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, 20, None, keep)
                _same(ids[i], dist[i], cnt[i], eids, edist, f"{tag} batch q{i}")

        check("initial")
        idx.set_deleted([64, 700, 10, 11])          # two quarantined rows and two regular ones
        alive[[64, 700, 10, 11]] = False
        assert idx.counters()["quarantined_rows"] == len(bad) - 2
        assert idx.counters()["deleted_rows"] == 4
        check("deleted")
        fresh = _rows(3, d, 5)
        idx.append(5, fresh[:1])                      # a quarantined row becomes regular
        rows[5] = fresh[0]
        idx.append(1500, fresh[1:3])                  # ... and two neighbours at once
        rows[1500:1502] = fresh[1:3]
        assert idx.counters()["quarantined_rows"] == len(bad) - 5
        check("overwritten")
        spoiled = _rows(2, d, 6)
        spoiled[0, 0] = np.nan
        spoiled[1, 1] = np.inf
        idx.append(2000, spoiled)                     # regular rows become irregular
        rows[2000:2002] = spoiled
        idx.append(64, spoiled[:1])                   # a deleted quarantined row is re-added, still irregular
        rows[64] = spoiled[0]
        alive[64] = True
        assert idx.counters()["quarantined_rows"] == len(bad) - 5 + 3
        assert idx.counters()["safe_mode"] == 0
        check("respoiled")
        idx.append(n, spoiled)                        # growth at the end
        rows2 = np.concatenate([rows, spoiled])
        alive2 = np.concatenate([alive, [True, True]])
        keep = np.packbits(alive2, bitorder="little")
        ids, dist, cnt = idx.search(q, 50)
        eids, edist = oracle_mod.search_exhaustive(rows2, q, metric, 50, None, keep)
        _same(ids[0], dist[0], cnt[0], eids, edist, "grown")


def test_more_irregular_rows_than_the_quarantine_holds(hip_lib, oracle_mod):
    """beyond 1024 quarantined rows per shard the remaining ones switch the shard to safe mode; still exact"""
    from tostore_amd import HipVectorIndex

    n, d = 6000, 16
    rows = _rows(n, d, 7)
    rows[100:1300, 0] = np.nan
    q = _rows(1, d, 8)[0]
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows)
        c = idx.counters()
        assert c["quarantined_rows"] == 1024 and c["safe_mode"] == 1
        for k in (10, n):
            ids, dist, cnt = idx.search(q, k)
            eids, edist = oracle_mod.search_exhaustive(rows, q, L2, k)
            _same(ids[0], dist[0], cnt[0], eids, edist, f"overflow k{k}")
        qs = _rows(12, d, 9)
        ids, dist, cnt = idx.search(qs, 10)
        for i in range(len(qs)):
            eids, edist = oracle_mod.search_exhaustive(rows, qs[i], L2, 10)
            _same(ids[i], dist[i], cnt[i], eids, edist, f"overflow batch {i}")


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n_bad", [3, 400])
def test_shard_mode_appends_quarantined_rows_to_the_device_blocks(hip_lib, oracle_mod, metric, n_bad):
    """tsh_search_shard: the quarantined rows' entries are appended to the device candidate blocks; when they do
    not fit, count > entries makes the merge ask for a retry with more entries (the blocks' overflow protocol)"""
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    n, d, k, base = 6000, 32, 10, 1003
    rng = np.random.default_rng(11)
    rows = _rows(n, d, 11)
    bad = rng.choice(n, size=n_bad, replace=False)
    for i, r in enumerate(bad):
        rows[r, int(rng.integers(0, d))] = [np.nan, np.inf, -np.inf, 1e20][i % 4]
    keep_bits = rng.random(n) < 0.5
    keep_bits[bad[::2]] = True
    keep_bits[bad[1::2]] = False
    L = _ffi.lib()
    idx = HipVectorIndex(d, metric, shard_device=0, row_base=base)
    try:
        idx.set_batch_min_nq(2)
        idx.append(base, rows)  # shard handles take GLOBAL ids
        c = idx.counters()
        assert (c["quarantined_rows"], c["safe_mode"]) == (n_bad, 0)
        for nq in (3, 24):  # 24: the batched path
            qs = np.stack([_q(oracle_mod, d, 60 + i, metric) for i in range(nq)])
            for bits in (None, keep_bits):
                mask = None if bits is None else np.packbits(np.concatenate([np.zeros(base, bool), bits]), bitorder="little")
                entries, retries = L.tsh_default_block_entries(k), 0
                while True:
                    buf = torch.empty(nq * L.tsh_candidate_block_bytes(entries), dtype=torch.uint8, device="cuda")
                    _ffi.check(L.tsh_search_shard(idx._h, qs.ctypes.data_as(_ffi.p_f32), nq, k,
                                                  None if mask is None else mask.ctypes.data_as(_ffi.p_u8), entries,
                                                  ctypes.c_void_p(buf.data_ptr()), None))
                    try:
                        ids, dist, cnt = merge_candidate_blocks(metric, d, qs, k, None, buf.cpu().numpy(), 1, entries)
                        break
                    except _ffi.TshError as e:
                        assert e.code == _ffi.TSH_E_OVERFLOW and retries == 0
                        entries, retries = int(e.needed_entries), retries + 1
                assert retries == (1 if n_bad > 200 else 0)
                omask = None if bits is None else np.packbits(bits, bitorder="little")
                for i in range(nq):
                    eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, None, omask)
                    _same(ids[i] - base, dist[i], cnt[i], eids, edist, f"shard m{metric} nq{nq} q{i} mask{bits is not None}")
        assert idx.counters()["fallback_searches"] == 0
    finally:
        idx.close()


def test_shard_mode_fallback_rewrites_the_block_and_appends_again(hip_lib, oracle_mod):
    """k > 1024 on more tiles than the select kernel can refine over (4096) takes the wide-band fallback, which
    rewrites the device block"""
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    n, d, k = 300000, 8, 1100
    rows = _rows(n, d, 13)
    rows[7, 0] = np.nan
    rows[70000, 3] = np.inf
    rows[299999, 7] = 1e20
    qs = _rows(2, d, 14)
    L = _ffi.lib()
    idx = HipVectorIndex(d, L2, shard_device=0, row_base=0)
    try:
        idx.append(0, rows)
        entries = L.tsh_default_block_entries(k)
        buf = torch.empty(2 * L.tsh_candidate_block_bytes(entries), dtype=torch.uint8, device="cuda")
        _ffi.check(L.tsh_search_shard(idx._h, qs.ctypes.data_as(_ffi.p_f32), 2, k, None, entries,
                                      ctypes.c_void_p(buf.data_ptr()), None))
        blocks = buf.cpu().numpy()
        ids, dist, cnt = merge_candidate_blocks(L2, d, qs, k, None, blocks, 1, entries)
        c = idx.counters()
        assert (c["fallback_searches"], c["quarantined_rows"], c["safe_mode"]) == (2, 3, 0)
        bb = L.tsh_candidate_block_bytes(entries)
        for i in range(2):
            blk_ids = blocks[i * bb + 64:(i + 1) * bb].view(np.int64)[::3][:int(blocks[i * bb:i * bb + 4].view(np.uint32)[0])]
            assert {7, 70000, 299999} <= set(blk_ids.tolist())  # offered by the block (they sort last for L2)
            eids, edist = oracle_mod.search_exhaustive(rows, qs[i], L2, k)
            _same(ids[i], dist[i], cnt[i], eids, edist, f"shard fallback q{i}")
    finally:
        idx.close()
