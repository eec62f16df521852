"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle.

Bar (BASELINE.json north_star): returned row ids bit-exact at a given k,
distances within 1e-5 relative.  The re-rank reproduces the reference's f64
accumulation order, so distances are in fact compared BIT-exact here.
"""
import math

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]  # small shapes: exact sums AND the f32 pre-filter (conftest)

L2, IP, COS = 0, 1, 2
METRICS = [L2, IP, COS]


def _mk(n, d, seed, normalize=False, scale=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    if scale is not None:
        x *= rng.uniform(scale[0], scale[1], size=(n, 1)).astype(np.float32)
    return x


def _prep_query(oracle, q, metric):
    q = np.ascontiguousarray(q, dtype=np.float32)
    return oracle.normalize_f32(q) if metric == COS else q


def _check(oracle, idx, rows, q, metric, k, threshold=None, keep=None, tag=""):
    ids, dist, cnt = idx.search(q, k, threshold, keep)
    eids, edist = oracle.search_exhaustive(rows, q, metric, k, threshold, keep)
    n = int(cnt[0])
    assert n == len(eids), f"{tag}: count {n} != {len(eids)}"
    assert np.array_equal(ids[0, :n], eids), f"{tag}: ids differ\n{ids[0, :n]}\n{eids}"
    a, b = dist[0, :n], edist
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) or np.array_equal(
        np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), \
        f"{tag}: distances not bit-exact, max rel {np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))}"


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("d", [4, 7, 100, 128, 384, 768, 1536])
def test_random_small(hip_lib, oracle_mod, metric, d):
    from tostore_amd import HipVectorIndex

    rows = _mk(3000, d, 100 + d, scale=(0.5, 2.0))
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        for qi in range(3):
            q = _prep_query(oracle_mod, _mk(1, d, 7 + qi)[0], metric)
            for k in (1, 10, 100):
                _check(oracle_mod, idx, rows, q, metric, k, tag=f"d{d} m{metric} k{k}")


@pytest.mark.parametrize("metric", METRICS)
def test_readme_example(hip_lib, oracle_mod, metric):
    """rows/query of /root/reference/example/lib/tostore_example.dart:387-413."""
    from tostore_amd import VectorIndexManager

    d = 128
    v1 = [i * 0.01 for i in range(d)]
    v2 = [i * 0.02 + 0.5 for i in range(d)]
    q = [i * 0.015 for i in range(d)]
    kat = {COS: [(8.881784197001252e-16, 0.9999999999999991), (0.008631337545516038, 0.991368662454484)],
           L2: [(4.155959577530976, 0.19395031806647095), (9.482193837605186, 0.09539987673310046)],
           IP: [(-268.22400006248057, 1.0), (-103.63200000291876, 1.0)]}
    m = VectorIndexManager(d, metric)
    try:
        m.insert_batch(["doc1", "doc2"], [v1, v2])
        res = m.vectorSearch(q, topK=5)
        assert len(res) == 2
        for r, (ed, es) in zip(res, kat[metric]):
            assert r.distance == ed
            assert r.score == es
        assert [r.primaryKey for r in res] == (["doc2", "doc1"] if metric == IP else ["doc1", "doc2"])
    finally:
        m.close()


def test_ef_cap_compatibility_of_the_hook(hip_lib, oracle_mod):
    """the reference returns at most ef = min(efSearch ?? meta.efSearch, max(5 topK, 32)) rows
    (/root/reference/lib/src/core/ngh_graph_engine.dart:80-82; result heap of capacity ef, :168) and none while the
    graph has no medoid (:78).  The device path returns the k best of all rows unless the maintainer asks for the old
    count (hipHonourEfCap in the hook, honourEfCap here): then the answer is the first min(topK, ef) of the same rows."""
    from tostore_amd import HipVectorIndex
    from tostore_amd.backend import HipVectorBackend

    d, n = 32, 3000
    rows = _mk(n, d, 11)
    q = _prep_query(oracle_mod, _mk(1, d, 12)[0], L2)
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows)
        eids, edist = oracle_mod.search_heap(rows, q, L2, 100)
        plain = HipVectorBackend(idx).search(query=q, topK=100)
        assert [r.nodeId for r in plain] == eids.tolist() and len(plain) == 100  # no cap: all 100
        capped = HipVectorBackend(idx, honourEfCap=True)  # meta.efSearch = 64 (model/ngh_index_meta.dart:196)
        got = capped.search(query=q, topK=100)
        assert len(got) == 64 and [r.nodeId for r in got] == eids[:64].tolist()
        assert [r.distance for r in got] == edist[:64].tolist()
        assert len(capped.search(query=q, topK=100, efSearch=200)) == 100  # the caller's ef >= topK
        assert len(capped.search(query=q, topK=100, efSearch=10)) == 10
        assert len(capped.search(query=q, topK=5)) == 5            # ef = min(64, max(25, 32)) = 32 >= 5
        assert len(capped.search(query=q, topK=5, efSearch=3)) == 3
        assert capped.search(query=q, topK=5, efSearch=0) == []
        assert HipVectorBackend(idx, honourEfCap=True, medoidNodeId=-1).search(query=q, topK=5) == []  # :78
        assert len(HipVectorBackend(idx, medoidNodeId=-1).search(query=q, topK=5)) == 5  # (not honoured: rows exist)


@pytest.mark.parametrize("metric", METRICS)
def test_ties_duplicates_and_zero_rows(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    d = 64
    base = _mk(40, d, 5)
    rows = np.concatenate([np.tile(base[:1], (500, 1)), base, np.zeros((70, d), np.float32),
                           np.tile(base[1:2], (300, 1))]).astype(np.float32)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        for q in (base[0], base[1], base[2] * 0.5, np.zeros(d, np.float32)):
            qq = _prep_query(oracle_mod, q, metric)
            for k in (3, 100, 600, len(rows), len(rows) + 5):
                _check(oracle_mod, idx, rows, qq, metric, k, tag=f"ties m{metric} k{k}")


@pytest.mark.parametrize("metric", METRICS)
def test_all_rows_identical(hip_lib, oracle_mod, metric, scan_path):
    from tostore_amd import HipVectorIndex

    d = 128
    rows = np.tile(_mk(1, d, 9), (5000, 1))
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        q = _prep_query(oracle_mod, _mk(1, d, 10)[0], metric)
        _check(oracle_mod, idx, rows, q, metric, 100, tag="identical")
        c = idx.counters()
        if scan_path == "prefilter":
            assert c["fallback_searches"] >= 1  # the wide-band path ran, still on the GPU
        else:  # 5000 rows: the exact sums of all of them, the k lowest ids of the tie -- no band to overflow
            assert c["fallback_searches"] == 0 and c["exact_scans"] == 1 and c["candidates_total"] == 100


@pytest.mark.parametrize("metric", METRICS)
def test_near_ties_at_boundary(hip_lib, oracle_mod, metric):
    """rows whose keys differ by ~1e-9 relative: f32 cannot order them, the f64 re-rank must."""
    from tostore_amd import HipVectorIndex

    d = 256
    rng = np.random.default_rng(3)
    base = rng.standard_normal(d).astype(np.float32)
    rows = np.tile(base, (4000, 1)).astype(np.float32)
    # flip the last mantissa bit of a few coordinates per row
    for i in range(len(rows)):
        cols = rng.integers(0, d, size=3)
        bits = rows[i, cols].view(np.uint32) ^ np.uint32(1)
        rows[i, cols] = bits.view(np.float32)
    q = _prep_query(oracle_mod, (base + rng.standard_normal(d).astype(np.float32) * 0.1), metric)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        for k in (10, 100):
            _check(oracle_mod, idx, rows, q, metric, k, tag=f"nearties m{metric} k{k}")


@pytest.mark.parametrize("metric", METRICS)
def test_neighbours_clustered_in_one_tile(hip_lib, oracle_mod, metric):
    """all true neighbours stored contiguously (worst case for the tile-minimum bound)."""
    from tostore_amd import HipVectorIndex

    d = 128
    rows = _mk(20000, d, 11, normalize=True)
    q0 = _mk(1, d, 12, normalize=True)[0]
    near = (q0[None, :] + 0.01 * _mk(150, d, 13)).astype(np.float32)
    rows[6400:6550] = near
    q = _prep_query(oracle_mod, q0, metric)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        _check(oracle_mod, idx, rows, q, metric, 100, tag=f"cluster m{metric}")


@pytest.mark.parametrize("metric", METRICS)
def test_nan_inf_rows(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    d = 32
    rows = _mk(500, d, 21)
    rows[3, 5] = np.nan
    rows[77, 0] = np.inf
    rows[78, 1] = -np.inf
    rows[200] = np.nan
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        c = idx.counters()  # kept out of the scan, re-ranked exactly (tests/test_gpu_irregular.py)
        assert c["safe_mode"] == 0 and c["quarantined_rows"] == 4
        q = _prep_query(oracle_mod, _mk(1, d, 22)[0], metric)
        for k in (10, 499, 500):
            _check(oracle_mod, idx, rows, q, metric, k, tag=f"naninf m{metric} k{k}")


@pytest.mark.parametrize("metric", METRICS)
def test_threshold_is_strict(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    d = 48
    rows = _mk(2000, d, 31)
    q = _prep_query(oracle_mod, _mk(1, d, 32)[0], metric)
    _, edist = oracle_mod.search_exhaustive(rows, q, metric, 50)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        thr = float(edist[20])  # exactly equal to a distance: that row is KEPT (`>` drops)
        _check(oracle_mod, idx, rows, q, metric, 50, threshold=thr, tag="thr-eq")
        ids, dist, cnt = idx.search(q, 50, thr)
        assert cnt[0] >= 21 and dist[0, cnt[0] - 1] == thr
        _check(oracle_mod, idx, rows, q, metric, 50, threshold=np.nextafter(thr, -np.inf), tag="thr-below")
        _check(oracle_mod, idx, rows, q, metric, 50, threshold=-1e30, tag="thr-none-pass")


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("d", [96, 128, 64, 32])  # 128/64/32: the packed narrow-row scan, masked variant
def test_mask_and_tombstones(hip_lib, oracle_mod, metric, d):
    from tostore_amd import HipVectorIndex

    n = 7000  # not a multiple of 64
    rows = _mk(n, d, 41)
    rng = np.random.default_rng(42)
    q = _prep_query(oracle_mod, _mk(1, d, 43)[0], metric)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows[:1000])
        idx.append(1000, rows[1000:])  # growth path
        for pct in (0.5, 0.1, 0.01, 1.0, 0.0):
            keepbits = rng.random(n) < pct
            keep = np.packbits(keepbits, bitorder="little")
            _check(oracle_mod, idx, rows, q, metric, 100, keep=keep, tag=f"mask{pct}")
        dead = rng.choice(n, size=900, replace=False)
        idx.set_deleted(dead)
        idx.set_deleted(dead[:10])  # idempotent
        alive = np.ones(n, bool)
        alive[dead] = False
        _check(oracle_mod, idx, rows, q, metric, 100, keep=np.packbits(alive, bitorder="little"),
               tag="tombstones-as-mask")
        ids, dist, cnt = idx.search(q, 100)
        eids, edist = oracle_mod.search_exhaustive(rows, q, metric, 100,
                                                   keep=np.packbits(alive, bitorder="little"))
        assert np.array_equal(ids[0, :cnt[0]], eids) and np.array_equal(dist[0, :cnt[0]], edist)
        assert idx.counters()["deleted_rows"] == 900
        # user mask AND tombstones
        keepbits = rng.random(n) < 0.3
        _check(oracle_mod, idx, rows, q, metric, 64,
               keep=np.packbits(keepbits & alive, bitorder="little"), tag="mask+tomb") if False else None
        ids, dist, cnt = idx.search(q, 64, None, np.packbits(keepbits, bitorder="little"))
        eids, edist = oracle_mod.search_exhaustive(rows, q, metric, 64,
                                                   keep=np.packbits(keepbits & alive, bitorder="little"))
        assert np.array_equal(ids[0, :cnt[0]], eids) and np.array_equal(dist[0, :cnt[0]], edist)


def test_empty_and_degenerate_calls(hip_lib, oracle_mod):
    from tostore_amd import HipVectorBackend, HipVectorIndex

    with HipVectorIndex(16, L2) as idx:
        ids, dist, cnt = idx.search(np.zeros(16, np.float32), 5)
        assert cnt[0] == 0  # empty index -> empty result (ngh_graph_engine.dart:78)
        assert HipVectorBackend(idx).search(query=np.zeros(16, np.float32), topK=5) == []
        rows = _mk(3, 16, 1)
        idx.append(0, rows)
        ids, dist, cnt = idx.search(rows[1], 0)
        assert cnt[0] == 0  # topK <= 0
        _check(oracle_mod, idx, rows, rows[1], L2, 10, tag="k>n")
        # gap: ids 3..9 absent
        idx.append(10, rows)
        full = np.zeros((13, 16), np.float32)
        full[:3] = rows
        full[10:] = rows
        keep = np.zeros(13, bool)
        keep[:3] = keep[10:] = True
        ids, dist, cnt = idx.search(rows[1], 13)
        eids, edist = oracle_mod.search_exhaustive(full, rows[1], L2, 13, keep=np.packbits(keep, bitorder="little"))
        assert np.array_equal(ids[0, :cnt[0]], eids) and np.array_equal(dist[0, :cnt[0]], edist)


@pytest.mark.parametrize("metric", METRICS)
def test_multi_query_call(hip_lib, oracle_mod, metric):
    from tostore_amd import HipVectorIndex

    d, n, nq, k = 128, 10000, 40, 10
    rows = _mk(n, d, 51, normalize=True)
    qs = np.stack([_prep_query(oracle_mod, q, metric) for q in _mk(nq, d, 52)])
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        ids, dist, cnt = idx.search(qs, k)
        for i in range(nq):
            eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k)
            assert cnt[i] == k and np.array_equal(ids[i], eids) and np.array_equal(dist[i], edist)


@pytest.mark.parametrize("metric,n,d,k", [(L2, 10000, 128, 10), (L2, 200000, 768, 100),
                                          (COS, 200000, 768, 100), (IP, 100000, 1536, 100)])
def test_config_shapes_scaled(hip_lib, oracle_mod, metric, n, d, k):
    """BASELINE.json configs C1 (full size) and C2-C4 at sizes the oracle finishes in seconds."""
    from tostore_amd import HipVectorIndex

    rows = _mk(n, d, 61, normalize=True, scale=None if metric == COS else (0.5, 2.0))
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        for qi in range(2):
            q = _prep_query(oracle_mod, _mk(1, d, 70 + qi, normalize=True)[0], metric)
            ids, dist, cnt = idx.search(q, k)
            eids, edist = oracle_mod.search_heap_mt(rows, q, metric, k)
            assert cnt[0] == k
            assert np.array_equal(ids[0], eids)
            assert np.array_equal(dist[0], edist)
            assert np.allclose(dist[0], edist, rtol=1e-5, atol=1e-6)  # the stated tolerance
        c = idx.counters()
        assert c["fallback_searches"] == 0 and c["candidates_total"] < 4 * k * 2


@pytest.mark.parametrize("metric", METRICS)
def test_async_submit_wait_matches_sync(hip_lib, oracle_mod, metric):
    """several independent single queries in flight (tsh_search_submit / tsh_search_wait)."""
    from collections import deque

    from tostore_amd import HipVectorIndex, _ffi

    d, n, k = 256, 30000, 50
    rows = _mk(n, d, 81, normalize=True)
    qs = np.stack([_prep_query(oracle_mod, q, metric) for q in _mk(24, d, 82)])
    keep = np.packbits(np.random.default_rng(83).random(n) < 0.5, bitorder="little")
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        depth = _ffi.lib().tsh_max_inflight()
        assert depth >= 4
        pend, got = deque(), []
        for i, q in enumerate(qs):
            if len(pend) == depth:
                got.append(idx.wait(pend.popleft()))
            pend.append(idx.submit(q, k, keep if i % 3 == 0 else None))
        with pytest.raises(_ffi.TshError) as e:  # one more than the handle allows
            while True:
                pend.append(idx.submit(qs[0], k))
        assert e.value.code == _ffi.TSH_E_BUSY
        while pend:
            got.append(idx.wait(pend.popleft()))
        for i, q in enumerate(qs):
            eids, edist = oracle_mod.search_exhaustive(rows, q, metric, k, None, keep if i % 3 == 0 else None)
            assert np.array_equal(got[i][0], eids) and np.array_equal(got[i][1], edist), i
        # ties flood the candidate list of an async query too (fallback inside wait)
    rows2 = np.tile(rows[:1], (9000, 1))
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows2)
        t = [idx.submit(qs[0], k), idx.submit(qs[1], k)]
        for tk, q in zip(t, qs[:2]):
            ids, dist = idx.wait(tk)
            eids, edist = oracle_mod.search_exhaustive(rows2, q, metric, k)
            assert np.array_equal(ids, eids) and np.array_equal(dist, edist)


def test_concurrent_threads_share_one_handle(hip_lib, oracle_mod):
    import threading

    from tostore_amd import HipVectorIndex

    d, n, k = 128, 40000, 20
    rows = _mk(n, d, 91)
    qs = _mk(32, d, 92)
    errs = []
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows)

        def work(lo, hi):
            try:
                for i in range(lo, hi):
                    ids, dist, cnt = idx.search(qs[i], k)
                    eids, edist = oracle_mod.search_heap(rows, qs[i], L2, k)
                    assert np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(i * 4, i * 4 + 4)) for i in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    assert not errs, errs


@pytest.mark.parametrize("d", [260, 1028, 1100, 1284, 1540, 1700, 2052, 2500, 3072, 3500, 4073, 4096])
def test_row_widths_between_kernel_variants(hip_lib, oracle_mod, d):
    """The scan kernel is compiled for a short list of chunk counts (1 KiB per lane-chunk); a row of any other
    width ends inside one chunk and leaves later ones empty -- every such chunk must contribute zeros, not the
    next row's bytes (L2 once summed (0 - next_row)^2 for d in (1024, 1280) and (1536, 1792)).  Widths up to
    4096: the reference's 16 KiB pages hold f32 vectors up to d = 4073."""
    from tostore_amd import HipVectorIndex

    n, k = 4500, 12  # >= 4096 rows: the batched path takes part too
    rows = _mk(n, d, 700 + d, normalize=False)
    qs = _mk(3, d, 701 + d, normalize=False)
    for metric in METRICS:
        with HipVectorIndex(d, metric) as idx:
            idx.set_batch_min_nq(0)
            idx.append(0, rows)
            keep = np.packbits(np.random.default_rng(d).random(n) < 0.5, bitorder="little")
            for q0 in qs:
                q = _prep_query(oracle_mod, q0, metric)
                _check(oracle_mod, idx, rows, q, metric, k, tag=f"d{d}")
                _check(oracle_mod, idx, rows, q, metric, k, keep=keep, tag=f"d{d} masked")
            idx.set_batch_min_nq(2)
            ids, dist, cnt = idx.search(np.stack([_prep_query(oracle_mod, q0, metric) for q0 in qs]), k)
            assert idx.counters()["batch_launches"] == 1
            for i, q0 in enumerate(qs):
                eids, edist = oracle_mod.search_heap(rows, _prep_query(oracle_mod, q0, metric), metric, k)
                assert np.array_equal(ids[i], eids) and np.array_equal(dist[i], edist)


@pytest.mark.parametrize("d,n", [(16, 12224), (4, 1024), (8, 65536), (100, 20480)])
def test_narrow_rows_filling_the_allocation_exactly(hip_lib, oracle_mod, d, n):
    """Lanes beyond a narrow row's end must not read `4 * lane` floats past the row start: in the last rows of an
    allocation that ends on a page boundary that address is unmapped (memory access fault, found by the
    extended fuzz run)."""
    from tostore_amd import HipVectorIndex

    rows = _mk(n, d, 900 + d, normalize=False)
    with HipVectorIndex(d, L2, capacity_rows=n) as idx:
        idx.set_batch_min_nq(0)
        idx.append(0, rows)
        idx.set_deleted([n - 1, n - 64])  # the masked variant walks the same addresses
        alive = np.ones(n, bool)
        alive[[n - 1, n - 64]] = False
        for i in (0, n - 2, n // 2):
            q = rows[i]
            ids, dist, cnt = idx.search(q, 5)
            eids, edist = oracle_mod.search_exhaustive(rows, q, L2, 5, None, np.packbits(alive, bitorder="little"))
            assert np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)


def test_many_open_indexes_share_the_device_streams(hip_lib, oracle_mod):
    """A database holds many vector indexes.  Streams (CU-masked ones above all) are finite -- about 85 open
    handles with streams of their own crashed the runtime -- so every handle on a device uses one shared
    set; two threads searching two different indexes still get exact answers."""
    import threading

    from tostore_amd import HipVectorIndex

    d, n, k = 48, 6000, 8
    rows = [_mk(n, d, 300 + i) for i in range(3)]
    qs = _mk(24, d, 310)
    handles = [HipVectorIndex(d, L2) for _ in range(150)]
    try:
        for i, h in enumerate(handles):
            h.append(0, rows[i % 3][: 200 if i >= 3 else n])
        errs = []

        def work(i):
            try:
                for _ in range(3):
                    ids, dist, cnt = handles[i].search(qs, k)
                    for j in range(len(qs)):
                        eids, edist = oracle_mod.search_heap(rows[i], qs[j], L2, k)
                        assert np.array_equal(ids[j], eids) and np.array_equal(dist[j], edist)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        ids, dist, cnt = handles[149].search(rows[149 % 3][5], 1)
        assert ids[0, 0] == 5
    finally:
        for h in handles:
            h.close()


def test_concurrent_multi_query_callers(hip_lib, oracle_mod):
    """Several threads, each handing over MANY queries per call (every call wants several of the shard's
    eight contexts; callers that each held some and waited for more used to deadlock), mixing the
    pipelined single-query path, the batched path and async tickets on one handle."""
    import threading

    from tostore_amd import HipVectorIndex

    d, n, k = 96, 30000, 15
    rows = _mk(n, d, 191)
    qs = _mk(240, d, 192)
    want = [oracle_mod.search_heap(rows, q, L2, k) for q in qs]
    errs = []
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows)

        def check(lo, ids, dist, cnt):
            for j in range(len(cnt)):
                assert cnt[j] == k and np.array_equal(ids[j], want[lo + j][0]) and np.array_equal(dist[j], want[lo + j][1])

        def pipelined(lo, hi):  # 80 queries per call, batching off for this thread's handle-wide setting
            try:
                for _ in range(3):
                    check(lo, *idx.search(qs[lo:hi], k))
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        def tickets(lo, hi):
            try:
                for i in range(lo, hi):
                    t = idx.submit(qs[i], k)
                    ids, dist = idx.wait(t)
                    assert np.array_equal(ids, want[i][0]) and np.array_equal(dist, want[i][1])
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        for min_nq in (0, 8):  # 0: every call runs pipelined single scans; 8: the 80-query calls go batched
            idx.set_batch_min_nq(min_nq)
            th = [threading.Thread(target=pipelined, args=(0, 80)), threading.Thread(target=pipelined, args=(80, 160)),
                  threading.Thread(target=pipelined, args=(160, 240)), threading.Thread(target=tickets, args=(0, 40))]
            for t in th:
                t.start()
            for t in th:
                t.join(timeout=120)
            assert not any(t.is_alive() for t in th), "deadlock"
            assert not errs, errs


@pytest.mark.parametrize("metric", METRICS)
def test_more_than_16384_tiles(hip_lib, oracle_mod, metric):
    """> 1,048,576 rows: the select kernel walks gmin[] in memory instead of registers."""
    from tostore_amd import HipVectorIndex

    d, n = 8, 1_200_000
    rows = _mk(n, d, 101, scale=(0.5, 2.0))
    rows[700_000:700_050] = rows[5]  # ties far apart
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows[:600_000])
        idx.append(600_000, rows[600_000:])
        idx.set_batch_min_nq(0)
        for qi in range(2):
            q = _prep_query(oracle_mod, _mk(1, d, 102 + qi)[0], metric)
            for k in (10, 100, 300):
                ids, dist, cnt = idx.search(q, k)
                eids, edist = oracle_mod.search_heap_mt(rows, q, metric, k)
                assert cnt[0] == k and np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)
        qs = np.stack([_prep_query(oracle_mod, x, metric) for x in _mk(12, d, 110)])
        idx.set_batch_min_nq(8)  # and the matrix-core path over the same index
        ids, dist, cnt = idx.search(qs, 20)
        for i in range(12):
            eids, edist = oracle_mod.search_heap_mt(rows, qs[i], metric, 20)
            assert np.array_equal(ids[i], eids) and np.array_equal(dist[i], edist)


@pytest.mark.parametrize("k", [513, 1000, 1024, 1025, 1500, 7000])
def test_large_k_single_query(hip_lib, oracle_mod, k):
    """k above 512 uses 2048 tile-minimum groups in the select kernel; above 1024 the fallback finds the exact
    k-th key with a radix select over all keys and re-ranks only what lies inside its band -- always exact,
    masks and tombstones included."""
    from tostore_amd import HipVectorIndex

    d, n = 16, 300_000  # 4688 tiles: more than the select kernel's short list holds
    rows = _mk(n, d, 121)
    q = _mk(1, d, 122)[0]
    with HipVectorIndex(d, L2, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        ids, dist, cnt = idx.search(q, k)
        eids, edist = oracle_mod.search_exhaustive(rows, q, L2, k)
        assert cnt[0] == k and np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)
        if k <= 1024:
            assert idx.counters()["fallback_searches"] == 0
        else:
            c = idx.counters()
            assert c["fallback_searches"] == 1 and c["candidates_total"] < k + 300  # not "every row"
            keep = np.packbits(np.random.default_rng(k).random(n) < 0.4, bitorder="little")
            idx.set_deleted(np.arange(0, n, 97))
            alive = np.unpackbits(keep, bitorder="little")[:n].astype(bool)
            alive[np.arange(0, n, 97)] = False
            ids, dist, cnt = idx.search(q, k, None, keep)
            eids, edist = oracle_mod.search_exhaustive(rows, q, L2, k, None, np.packbits(alive, bitorder="little"))
            assert cnt[0] == len(eids) and np.array_equal(ids[0, :cnt[0]], eids) and np.array_equal(dist[0, :cnt[0]], edist)


def test_appends_and_deletes_while_searching(hip_lib, oracle_mod):
    """search (shared lock) vs append / delete (exclusive lock) from different threads: every answer
    must be the exact answer for the row set at some moment between call and return."""
    import threading

    from tostore_amd import HipVectorIndex

    d, total, chunk, k = 64, 30000, 750, 15
    rows = _mk(total, d, 131)
    qs = _mk(6, d, 132)
    errs, stop = [], threading.Event()
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows[:chunk])

        def writer():
            try:
                for lo in range(chunk, total, chunk):
                    idx.append(lo, rows[lo:lo + chunk])
            except Exception as e:  # noqa: BLE001
                errs.append(e)
            finally:
                stop.set()

        def reader(qi):
            try:
                while not stop.is_set():
                    before = idx.size
                    ids, dist, cnt = idx.search(qs[qi], k)
                    after = idx.size
                    got, gd = ids[0, :cnt[0]], dist[0, :cnt[0]]
                    assert cnt[0] == k and (got < after).all()
                    exact = np.array([oracle_mod.exact_distance(qs[qi], rows[i], L2) for i in got])
                    assert np.array_equal(gd, exact)                        # distances of those very rows
                    assert all((gd[i], got[i]) <= (gd[i + 1], got[i + 1]) for i in range(k - 1))
                    _, lo_d = oracle_mod.search_heap(rows[:before], qs[qi], L2, k)
                    _, hi_d = oracle_mod.search_heap(rows[:after], qs[qi], L2, k)
                    assert hi_d[-1] <= gd[-1] <= lo_d[-1]                    # between the two snapshots
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=writer)] + [threading.Thread(target=reader, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs[:2]
        assert idx.size == total
        ids, dist, cnt = idx.search(qs[0], k)
        eids, edist = oracle_mod.search_heap(rows, qs[0], L2, k)
        assert np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("keep_frac", [0.5, 0.05, 0.004])
def test_contiguous_range_mask_needs_no_fallback(hip_lib, oracle_mod, metric, keep_frac):
    """C5 with a contiguous id range (WHERE id BETWEEN ...): few live tiles make the tile-minimum bound loose;
    the select kernel's exact-tau refinement must keep the candidate list short (no wide-band fallback)."""
    from tostore_amd import HipVectorIndex

    n, d, k = 120_000, 96, 100
    rows = _mk(n, d, 77, normalize=(metric == COS))
    rng = np.random.default_rng(int(keep_frac * 1000) + metric)
    keep = np.zeros(n, bool)
    m = int(n * keep_frac)
    start = int(rng.integers(0, n - m))
    keep[start:start + m] = True
    bits = np.packbits(keep, bitorder="little")
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        before = idx.counters()["fallback_searches"]
        for s in range(4):
            q = _prep_query(oracle_mod, rng.standard_normal(d).astype(np.float32), metric)
            _check(oracle_mod, idx, rows, q, metric, k, keep=bits, tag=f"range{keep_frac}")
        assert idx.counters()["fallback_searches"] == before


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_rows_inserted_by_topic_need_no_fallback(hip_lib, oracle_mod, metric):
    """Neighbours of a query sit in consecutive rows (documents inserted topic by topic): thousands of keys are
    below the k-th smallest tile minimum; refinement finds the exact k-th key instead of overflowing."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(5 + metric)
    n_topics, per, d, k = 150, 800, 64, 100
    centres = rng.standard_normal((n_topics, d)).astype(np.float32) * 4
    rows = (np.repeat(centres, per, axis=0) + 0.3 * rng.standard_normal((n_topics * per, d))).astype(np.float32)
    if metric == COS:
        rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        before = idx.counters()["fallback_searches"]
        for t in (0, 71, 149):
            q = _prep_query(oracle_mod, (centres[t] + 0.1 * rng.standard_normal(d)).astype(np.float32), metric)
            for kk in (10, k, 700):
                _check(oracle_mod, idx, rows, q, metric, kk, tag=f"topic{t}")
        assert idx.counters()["fallback_searches"] == before
