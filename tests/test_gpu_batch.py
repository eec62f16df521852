"""GPU: the batched matrix-core path (nq >= 8 by default) against the oracle and
against the single-query path.  Same bar: ids bit-exact, distances bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L2, IP, COS = 0, 1, 2
METRICS = [L2, IP, COS]


@pytest.fixture(autouse=True)
def _batch_from_two_queries(monkeypatch):
    """These tests are about the batched path: take it from two queries per call on, whatever the cost
    estimate of the default setting would choose for these small indexes."""
    from tostore_amd import HipVectorIndex

    orig = HipVectorIndex.__init__

    def init(self, *a, **kw):
        orig(self, *a, **kw)
        self.set_batch_min_nq(2)

    monkeypatch.setattr(HipVectorIndex, "__init__", init)


def _mk(n, d, seed, normalize=True, scale=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    if scale is not None:
        x *= rng.uniform(scale[0], scale[1], size=(n, 1)).astype(np.float32)
    return x


def _queries(oracle, nq, d, seed, metric):
    q = _mk(nq, d, seed)
    return np.stack([oracle.normalize_f32(x) for x in q]) if metric == COS else q


def _check_batch(oracle, idx, rows, qs, metric, k, thr=None, keep=None, tag=""):
    ids, dist, cnt = idx.search(qs, k, thr, keep)
    # the OpenMP oracle pays ~0.1 s of thread start-up per call on a 128-core box
    ref = oracle.search_heap_mt if rows.size > 1e8 else oracle.search_heap
    for i in range(len(qs)):
        eids, edist = ref(rows, qs[i], metric, k, thr, keep)
        n = int(cnt[i])
        assert n == len(eids), f"{tag} q{i}: count {n} != {len(eids)}"
        assert np.array_equal(ids[i, :n], eids), f"{tag} q{i}: ids differ"
        assert np.array_equal(dist[i, :n], edist), f"{tag} q{i}: distances differ"


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("d,n,nq,k", [(128, 30000, 130, 10), (100, 20000, 9, 100), (768, 30000, 130, 100),
                                      (36, 50000, 300, 37)])
@pytest.mark.parametrize("kernel", [1, 0, 2])  # 1 = bf16x3 split MFMA (default), 0 = f32 MFMA, 2 = f16
def test_batched_matches_oracle(hip_lib, oracle_mod, metric, d, n, nq, k, kernel):
    from tostore_amd import HipVectorIndex

    rows = _mk(n, d, 200 + d, scale=None if metric == COS else (0.5, 2.0))
    qs = _queries(oracle_mod, nq, d, 300 + d, metric)
    with HipVectorIndex(d, metric) as idx:
        idx.set_batch_kernel(kernel)
        idx.append(0, rows)
        _check_batch(oracle_mod, idx, rows, qs, metric, k, tag=f"m{metric} d{d}")
        c = idx.counters()
        assert c["batch_launches"] >= 1 and c["scan_launches"] == 0, c  # the MFMA path answered all of them
        # and the single-query path gives the same bytes
        idx.set_batch_min_nq(0)
        ids1, dist1, cnt1 = idx.search(qs[:12], k)
        idx.set_batch_min_nq(8)
        ids2, dist2, cnt2 = idx.search(qs[:12], k)
        assert np.array_equal(ids1, ids2) and np.array_equal(dist1, dist2) and np.array_equal(cnt1, cnt2)


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("kernel", [1, 0, 2])
def test_batched_mask_tombstones_threshold(hip_lib, oracle_mod, metric, kernel):
    from tostore_amd import HipVectorIndex

    d, n, nq, k = 64, 25000, 40, 20
    rows = _mk(n, d, 11, scale=None if metric == COS else (0.7, 1.5))
    qs = _queries(oracle_mod, nq, d, 12, metric)
    rng = np.random.default_rng(13)
    with HipVectorIndex(d, metric) as idx:
        idx.set_batch_kernel(kernel)
        idx.append(0, rows)
        keepbits = rng.random(n) < 0.3
        keep = np.packbits(keepbits, bitorder="little")
        _check_batch(oracle_mod, idx, rows, qs, metric, k, keep=keep, tag="mask")
        dead = rng.choice(n, 3000, replace=False)
        idx.set_deleted(dead)
        alive = np.ones(n, bool)
        alive[dead] = False
        ids, dist, cnt = idx.search(qs, k)
        for i in range(nq):
            eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, None, np.packbits(alive, bitorder="little"))
            assert np.array_equal(ids[i, :cnt[i]], eids) and np.array_equal(dist[i, :cnt[i]], edist)
        _, ed = oracle_mod.search_exhaustive(rows, qs[0], metric, k, None, np.packbits(alive & keepbits, bitorder="little"))
        thr = float(ed[k // 2])
        ids, dist, cnt = idx.search(qs, k, thr, keep)
        for i in range(nq):
            eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, thr,
                                                       np.packbits(alive & keepbits, bitorder="little"))
            assert np.array_equal(ids[i, :cnt[i]], eids) and np.array_equal(dist[i, :cnt[i]], edist)
        assert idx.counters()["batch_launches"] >= 3


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_batched_ties_fall_back_per_query(hip_lib, oracle_mod, metric):
    """identical rows flood every candidate list: those queries are redone by the single-query path."""
    from tostore_amd import HipVectorIndex

    d, n, nq, k = 32, 12000, 16, 50
    rows = np.tile(_mk(3, d, 21), (n // 3, 1))
    qs = _queries(oracle_mod, nq, d, 22, metric)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        _check_batch(oracle_mod, idx, rows, qs, metric, k, tag="ties")


def test_batched_mixed_bad_query_and_small_index(hip_lib, oracle_mod):
    from tostore_amd import HipVectorIndex

    d, n = 48, 9000
    rows = _mk(n, d, 31)
    qs = _queries(oracle_mod, 20, d, 32, L2)
    qs[3, 5] = np.inf  # outside the f32 error model: answered by the exact wide path
    qs[7] = 0.0
    with HipVectorIndex(d, L2) as idx:
        idx.append(0, rows)
        ids, dist, cnt = idx.search(qs, 10)
        for i in range(20):
            eids, edist = oracle_mod.search_exhaustive(rows, qs[i], L2, 10)
            assert np.array_equal(ids[i, :cnt[i]], eids)
            a, b = dist[i, :cnt[i]], edist
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    small = _mk(1000, d, 33)
    with HipVectorIndex(d, L2) as idx:  # below 4096 rows the batch path is skipped
        idx.append(0, small)
        _check_batch(oracle_mod, idx, small, qs[:10][np.isfinite(qs[:10]).all(axis=1)], L2, 5, tag="small")
        assert idx.counters()["batch_launches"] == 0


@pytest.mark.parametrize("metric", [L2, COS])
@pytest.mark.parametrize("k", [1, 2, 63, 64, 65, 255])
def test_batched_clustered_keys_and_k_edges(hip_lib, oracle_mod, metric, k):
    """rows drawn around a few dozen centres: the pre-filter keys of a query bunch into a few narrow groups (the
    workgroup-wide radix select of the sample / final select kernels sees bytes where nearly every key agrees and
    bytes where they spread), at k either side of the wave and list sizes those kernels step through"""
    from tostore_amd import HipVectorIndex

    d, n, nq = 40, 30000, 24
    rng = np.random.default_rng(77 + k)
    centres = rng.standard_normal((48, d)).astype(np.float32)
    rows = (centres[rng.integers(0, 48, n)] + 2e-3 * rng.standard_normal((n, d))).astype(np.float32)
    rows[::997] = centres[0]  # a sprinkle of exact duplicates
    qs = _queries(oracle_mod, nq, d, 78, metric)
    qs[:8] = (centres[:8] + 1e-3 * rng.standard_normal((8, d))).astype(np.float32)  # queries inside a cluster
    if metric == COS:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        _check_batch(oracle_mod, idx, rows, qs, metric, k, tag=f"clustered k{k}")


@pytest.mark.parametrize("k", [300, 1000])
def test_batched_large_k(hip_lib, oracle_mod, k):
    """k above the thread count of the per-query select kernels (group minima, 4 per thread)."""
    from tostore_amd import HipVectorIndex

    d, n, nq = 32, 60000, 12
    rows = _mk(n, d, 41)
    qs = _queries(oracle_mod, nq, d, 42, COS)
    with HipVectorIndex(d, COS) as idx:
        idx.append(0, rows)
        _check_batch(oracle_mod, idx, rows, qs, COS, k, tag=f"k{k}")
        c = idx.counters()
        assert c["batch_launches"] >= 1 and c["scan_launches"] == 0, c


@pytest.mark.parametrize("metric", METRICS)
def test_bf16_planes_follow_appends_overwrites_and_growth(hip_lib, oracle_mod, metric):
    """The bf16 (hi, lo) copy of the rows is built by the first batched search and must track later appends,
    overwrites of existing ids and reallocation of the row store."""
    from tostore_amd import HipVectorIndex

    d, nq, k = 96, 32, 30
    rows = _mk(30_000, d, 21, normalize=(metric == COS))
    qs = _queries(oracle_mod, nq, d, 22, metric)
    with HipVectorIndex(d, metric) as idx:  # capacity grows on demand
        idx.append(0, rows[:6000])
        _check_batch(oracle_mod, idx, rows[:6000], qs, metric, k, tag="first")
        idx.append(6000, rows[6000:9000])                     # append inside / past the capacity
        _check_batch(oracle_mod, idx, rows[:9000], qs, metric, k, tag="appended")
        rows[100:164] = rows[20_000:20_064]                   # overwrite existing ids
        idx.append(100, rows[100:164])
        _check_batch(oracle_mod, idx, rows[:9000], qs, metric, k, tag="overwritten")
        idx.append(9000, rows[9000:])                         # forces a reallocation of the row store
        _check_batch(oracle_mod, idx, rows, qs, metric, k, tag="grown")
        idx.set_batch_kernel(0)
        _check_batch(oracle_mod, idx, rows, qs, metric, k, tag="f32 kernel")
        idx.set_batch_kernel(2)                               # planes rebuilt in the other format
        _check_batch(oracle_mod, idx, rows, qs, metric, k, tag="f16 kernel")
        c_before = idx.counters()
        assert c_before["scan_launches"] == 0
        idx.append(30_000, rows[:500] * np.float32(40.0))     # larger magnitudes: the f16 scale must follow
        _check_batch(oracle_mod, idx, np.concatenate([rows, rows[:500] * np.float32(40.0)]), qs, metric, k, tag="f16 rescale")
        c = idx.counters()
        assert c["batch_launches"] > c_before["batch_launches"]
        # IP / cosine: still answered by the batched path alone.  L2 keys carry |v|^2: with fp16 FORCED onto rows
        # whose norms now span a factor 40, the band (it scales with the largest row) holds more
        # rows than a candidate list does, and the queries are handed to the scans -- exact, and what the automatic
        # choice avoids by taking bf16x3 for such a corpus (test_auto_kernel_choice)
        assert metric == L2 or c["scan_launches"] == 0


def test_bf16x3_wide_dynamic_range(hip_lib, oracle_mod):
    """Rows whose components span many binades (hi + lo must carry each element's own exponent)."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(8)
    n, d, nq = 12_000, 200, 24
    rows = (rng.standard_normal((n, d)) * np.exp2(rng.integers(-40, 20, (n, 1)))).astype(np.float32)
    rows[:, ::7] *= np.float32(2.0 ** -30)
    qs = (rng.standard_normal((nq, d)) * np.exp2(rng.integers(-10, 10, (nq, 1)))).astype(np.float32)
    for metric in METRICS:
        qm = np.stack([oracle_mod.normalize_f32(x) for x in qs]) if metric == COS else qs
        for kernel in (1, 2):  # f16: rows far below the largest magnitude and small queries must stay exact
            with HipVectorIndex(d, metric) as idx:
                idx.set_batch_kernel(kernel)
                idx.append(0, rows)
                _check_batch(oracle_mod, idx, rows, qm, metric, 25, tag=f"range m{metric} kernel{kernel}")


def test_auto_kernel_choice(hip_lib, oracle_mod):
    """TSH_OPT_BATCH_KERNEL auto: fp16 keys for cosine, for inner product (a row's band is its own) and for L2 corpora
    whose band's shared term stays small against the shortest rows' key spacing (norms a factor 32 apart: yes; norms
    2^-6 .. 2^6, or a zero row: bf16x3) -- results exact either way."""
    from tostore_amd import HipVectorIndex

    n, d, nq, k = 12_000, 64, 40, 20
    unit = _mk(n, d, 71)
    spread = unit * np.exp2(np.random.default_rng(72).integers(-6, 7, (n, 1))).astype(np.float32)
    wide = unit * np.random.default_rng(75).uniform(0.1, 3.2, (n, 1)).astype(np.float32)  # a factor 32 between the norms
    for metric, rows, want in ((COS, spread, 2), (IP, unit, 2), (L2, unit * np.float32(3.0), 2), (IP, spread, 2),
                               (L2, spread, 1), (L2, wide, 2), (IP, wide, 2)):
        qs = _queries(oracle_mod, nq, d, 73, metric)
        with HipVectorIndex(d, metric) as idx:
            assert idx.counters()["batch_kernel_last"] == -1
            idx.append(0, rows)
            _check_batch(oracle_mod, idx, rows, qs, metric, k, tag=f"auto m{metric}")
            assert idx.counters()["batch_kernel_last"] == want
    for metric, want in ((IP, 2), (L2, 1)):  # a zero row: the smallest norm is 0 -- L2's shortest rows have no key spacing to speak of
        with HipVectorIndex(d, metric) as idx:
            rows = unit.copy()
            rows[17] = 0.0
            idx.append(0, rows)
            _check_batch(oracle_mod, idx, rows, _queries(oracle_mod, nq, d, 74, metric), metric, k, tag="zero row")
            assert idx.counters()["batch_kernel_last"] == want


def test_default_switches_to_batched_by_estimated_cost(hip_lib, oracle_mod):
    """TSH_OPT_BATCH_MIN_NQ = 1 (default): 300 k x 768 rows -- two queries are cheaper as two pipelined scans,
    three are cheaper as one batched pass; the answers are the same bytes either way."""
    from tostore_amd import HipVectorIndex

    n, d, k = 300_000, 768, 10
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((3, d)).astype(np.float32)
    with HipVectorIndex(d, L2, capacity_rows=n) as idx:
        idx.set_batch_min_nq(1)
        idx.append(0, rows)
        a = idx.search(qs[:2], k)
        c = idx.counters()
        assert c["batch_launches"] == 0 and c["scan_launches"] == 2
        b = idx.search(qs, k)
        c = idx.counters()
        assert c["batch_launches"] == 1 and c["scan_launches"] == 2
        assert np.array_equal(a[0], b[0][:2]) and np.array_equal(a[1], b[1][:2])
        eids, edist = oracle_mod.search_heap_mt(rows, qs[2], L2, k)
        assert np.array_equal(b[0][2], eids) and np.array_equal(b[1][2], edist)


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_device_finaliser_edges(hip_lib, oracle_mod, metric):
    """The batched path's results are finalised on the device (distance from the exact sums, strict threshold, compareTo
    order with the id as tie-break, cut to k): duplicate rows (equal distances, ids decide), a zero row (cosine: the
    denominator rule), a threshold that EQUALS a returned distance (kept: the drop is strict), a threshold below
    everything (empty lists), slots past the count, and k of 1 / of a whole list."""
    from tostore_amd import HipVectorIndex

    d, n, nq = 96, 20000, 33
    rows = _mk(n, d, 71, scale=None if metric == COS else (0.8, 1.2))
    rows[100:140] = rows[7]          # forty copies of one row: ties inside every list that holds them
    rows[5000] = 0.0                 # cosine: mag_b = 0 -> similarity 0 -> distance 1
    qs = _queries(oracle_mod, nq, d, 72, metric)
    qs[3] = rows[7] if metric != COS else oracle_mod.normalize_f32(rows[7])  # the copies are this query's top hits
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        for k in (1, 10, 50):
            _check_batch(oracle_mod, idx, rows, qs, metric, k, tag=f"k{k}")
        k = 50
        _, ed = oracle_mod.search_exhaustive(rows, qs[3], metric, k)
        for thr in (float(ed[20]), float(ed[0]), float(np.nextafter(ed[0], -np.inf))):
            _check_batch(oracle_mod, idx, rows, qs, metric, k, thr=thr, tag=f"thr{thr}")
        ids, dist, cnt = idx.search(qs, k, float(np.nextafter(ed[0], -np.inf)))
        assert cnt[3] == 0 and np.all(ids[3] == -1) and np.all(np.isnan(dist[3]))
        c = idx.counters()
        assert c["batch_launches"] >= 7 and c["fallback_searches"] == 0


def test_two_callers_overlap_bit_exact(hip_lib, oracle_mod):
    """Two host threads with batched calls in flight on one handle (a shard keeps two scratch sets, the calls'
    GPU work is one in-order sequence): every call's answers equal the oracle's."""
    import threading
    from tostore_amd import HipVectorIndex

    d, n, nq, k, metric = 128, 60000, 96, 20, COS
    rows = _mk(n, d, 81)
    sets = [_queries(oracle_mod, nq, d, 82 + t, metric) for t in range(2)]
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows)
        idx.search(sets[0][:8], k)  # planes built
        got, err = [[], []], []

        def caller(t):
            try:
                for _ in range(6):
                    got[t].append(idx.search(sets[t], k))
            except Exception as e:  # noqa: BLE001
                err.append(repr(e))

        th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        for t in range(2):
            ref = [oracle_mod.search_heap(rows, sets[t][i], metric, k) for i in range(nq)]
            for ids, dist, cnt in got[t]:
                for i in range(nq):
                    assert cnt[i] == len(ref[i][0])
                    assert np.array_equal(ids[i, :cnt[i]], ref[i][0]) and np.array_equal(dist[i, :cnt[i]], ref[i][1])
        assert idx.counters()["fallback_searches"] == 0


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_batched_range_mask_takes_its_sample_inside_the_range(hip_lib, oracle_mod, metric, kernel):
    """A WHERE clause that keeps ONE id range leaves the first rows -- the default sample -- without a kept row: the
    batched path then moves its sample window to where the kept rows are (round 4; before, every query of such a call
    overflowed and was redone by a scan of its own).  Ranges at the start, in the middle and at the end of the rows;
    a mask kept in two clusters; the filtered pass runs on both sides of the window."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(17 + metric)
    n, d, k, nq = 150_000, 128, 20, 48
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(2)
        idx.set_batch_kernel(kernel)
        for spans in ([(0, 15_000)], [(70_001, 90_003)], [(n - 12_345, n)], [(30_000, 33_000), (120_000, 140_000)]):
            keep = np.zeros(n, bool)
            for a, b in spans:
                keep[a:b] = True
            mask = np.packbits(keep, bitorder="little")
            c0 = idx.counters()
            ids, dist, cnt = idx.search(qs, k, None, mask)
            c1 = idx.counters()
            ref = oracle_mod.search_heap_many_mt(rows, qs, metric, k, None, mask)
            assert np.array_equal(cnt, ref[2]) and np.array_equal(ids, ref[0])
            assert np.array_equal(dist.view(np.uint64), ref[1].view(np.uint64))
            assert c1["batch_launches"] > c0["batch_launches"]
            assert c1["scan_launches"] == c0["scan_launches"], (spans, c1["scan_launches"] - c0["scan_launches"])


@pytest.mark.parametrize("metric", [L2, IP])
def test_hub_rows_bound(hip_lib, oracle_mod, metric):
    """TSH_OPT_BATCH_HUB = 1 (round 6; off by default): the batched path also scores the index's 4096 hub rows -- the
    shortest (L2) / the longest (inner product) -- densely, as a gathered fp16 copy, and takes the smaller of the sample's
    threshold and their k-th smallest key, a bound by construction.  Same answers as the oracle; hub rows that are
    tombstoned, overwritten (the copy is dropped and rebuilt) or outgrown by appends change nothing about that."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(900 + metric)
    n, d, nq, k = 90_000, 96, 70, 60
    rows = _mk(n + 40_000, d, 901 + metric, scale=(0.5, 2.0))
    qs = _queries(oracle_mod, nq, d, 902, metric)
    with HipVectorIndex(d, metric) as idx:
        idx.append(0, rows[:n])
        idx.set_batch_min_nq(2)
        idx.set_batch_kernel(2)
        idx.set_batch_group(False)  # (the norm-grouped plane is the default cure for crowded rows: the hub bound is the other)
        off = idx.search(qs, k)
        idx.set_batch_hub(True)
        r0 = idx.counters()["bytes_resident"]
        _check_batch(oracle_mod, idx, rows[:n], qs, metric, k, tag="hub")
        on = idx.search(qs, k)
        assert idx.counters()["bytes_resident"] - r0 >= 4096 * 4 * 64  # (the hub rows' fp16 copy exists: the bound ran)
        assert all(np.array_equal(a, b) for a, b in zip(on, off))
        # the most extreme rows are tombstoned: they are hub rows, and their keys must leave the bound
        nrm = np.linalg.norm(rows[:n].astype(np.float64), axis=1)
        order = np.argsort(nrm if metric == L2 else -nrm)
        dead = order[:300]
        idx.set_deleted(dead)
        keep = np.ones(n, bool)
        keep[dead] = False
        ids, dist, cnt = idx.search(qs, k)
        for i in range(0, nq, 7):
            e, ed = oracle_mod.search_heap(rows[:n], qs[i], metric, k, None, np.packbits(keep, bitorder="little"))
            assert np.array_equal(ids[i, :cnt[i]], e) and np.array_equal(dist[i, :cnt[i]], ed)
        # a hub row is overwritten by a far-away one: the copy of the old contents must not bound anything
        victim = int(order[400])
        rows[victim] = rows[victim] * np.float32(40.0 if metric == L2 else 0.01)
        idx.append(victim, rows[victim:victim + 1])
        # ... and the shard grows by more than a quarter (the hub is rebuilt), with new extreme rows among the new ones
        rows[n:n + 50] *= np.float32(0.2 if metric == L2 else 3.0)
        idx.append(n, rows[n:])
        keep = np.ones(len(rows), bool)
        keep[dead] = False
        ids, dist, cnt = idx.search(qs, k)
        for i in range(0, nq, 7):
            e, ed = oracle_mod.search_heap(rows, qs[i], metric, k, None, np.packbits(keep, bitorder="little"))
            assert np.array_equal(ids[i, :cnt[i]], e) and np.array_equal(dist[i, :cnt[i]], ed)
        assert idx.counters()["fallback_searches"] == 0


@pytest.mark.parametrize("metric", [L2, IP])
def test_norm_grouped_plane(hip_lib, oracle_mod, metric):
    """TSH_OPT_BATCH_GROUP (round 6; on by default): the fp16 plane of an L2 / inner-product index holds its rows by norm
    inside blocks of 8192 (plane_group_kernel, tsh_batch_f16.hip.h) -- the key kernel works in plane positions and lists
    its survivors under the rows' ids.  Same answers as the oracle and as the plane in row order, through everything
    that indexes by row: tombstones, caller masks (Bernoulli, one id range: the sample window moves in whole blocks),
    mask handles, appends that complete a block, overwrites inside an old block, a threshold, both query-tile shapes."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(950 + metric)
    n0, n1, d, k = 41_000, 70_000, 72, 40
    rows = _mk(n1, d, 951 + metric, scale=(0.3, 3.0))
    for nq in (40, 200):
        qs = _queries(oracle_mod, nq, d, 952 + nq, metric)
        with HipVectorIndex(d, metric) as idx:
            idx.append(0, rows[:n0])
            idx.set_batch_min_nq(2)
            idx.set_batch_kernel(2)
            r0 = idx.counters()["bytes_resident"]
            _check_batch(oracle_mod, idx, rows[:n0], qs, metric, k, tag=f"grouped nq={nq}")
            grouped = idx.search(qs, k)
            r1 = idx.counters()["bytes_resident"]
            idx.set_batch_group(False)
            plain = idx.search(qs, k)
            assert all(np.array_equal(a, b) for a, b in zip(grouped, plain))
            idx.set_batch_group(True)
            # (the two maps, 8 B per position, are part of the resident bytes)
            assert r1 - r0 >= n0 * 8
            # tombstones: the extreme rows of every block (they lead the grouped blocks)
            nrm = np.linalg.norm(rows[:n0].astype(np.float64), axis=1)
            order = np.argsort(nrm if metric == L2 else -nrm)
            dead = order[:500]
            idx.set_deleted(dead)
            alive = np.ones(n1, bool)
            alive[dead] = False

            def check(nrows, keep=None, thr=None, handle=False):
                eff = alive[:nrows] if keep is None else alive[:nrows] & keep[:nrows]
                em = np.packbits(eff, bitorder="little")
                bits = None if keep is None else np.packbits(keep[:nrows], bitorder="little")
                c0 = idx.counters()
                if handle:
                    with idx.make_mask(bits) as m:
                        ids, dist, cnt = idx.search(qs, k, thr, m)
                else:
                    ids, dist, cnt = idx.search(qs, k, thr, bits)
                c1 = idx.counters()
                assert c1["batch_launches"] > c0["batch_launches"] and c1["fallback_searches"] == c0["fallback_searches"]
                ref = oracle_mod.search_heap_many_mt(rows[:nrows], qs, metric, k, thr, em)
                assert np.array_equal(cnt, ref[2])
                for i in range(nq):
                    assert np.array_equal(ids[i, :cnt[i]], ref[0][i, :cnt[i]]), i
                    assert np.array_equal(dist[i, :cnt[i]].view(np.uint64), ref[1][i, :cnt[i]].view(np.uint64)), i

            check(n0)
            keep = rng.random(n1) < 0.3
            check(n0, keep)
            check(n0, keep, handle=True)
            rng_keep = np.zeros(n1, bool)
            rng_keep[19_000:33_000] = True  # one id range that starts and ends inside blocks
            check(n0, rng_keep)
            check(n0, rng_keep, handle=True)
            # appends: the tail block is completed (8 192 x 5 = 40 960 < 41 000 < 49 152) and more blocks follow
            idx.append(n0, rows[n0:n1])
            check(n1)
            check(n1, keep)
            # overwrites inside old blocks: a short row becomes a long one and the other way round
            for victim, f in ((int(order[700]), 30.0), (int(order[-5]), 0.02)):
                rows[victim] = rows[victim] * np.float32(f if metric == L2 else 1.0 / f)
                idx.append(victim, rows[victim:victim + 1])
                alive[victim] = True
            check(n1)
            ed = oracle_mod.search_heap(rows[:n1], qs[0], metric, k, None, np.packbits(alive, bitorder="little"))[1]
            check(n1, None, float(ed[len(ed) // 2]))


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_listed_batch_behind_selective_masks(hip_lib, oracle_mod, metric):
    """Listed mode (round 6): a batched call behind a mask that keeps at most 16 384 rows, a small part of the shard, scores a
    GATHERED fp16 copy of the kept rows -- no pass over the shard, no copy of the whole shard built -- and lists its
    candidates under the rows' ids.  Masks by pointer and by handle, Bernoulli and one id range, fewer kept rows than k, rows
    deleted after the handle was made, both query-tile shapes, a threshold; answers equal to the oracle's.  And the cost
    model behind TSH_OPT_BATCH_MIN_NQ = 1: four queries behind such a mask go one by one (their kept rows' exact sums), sixty
    go to the matrix cores."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(1200 + metric)
    n, d, k = 300_000, 64, 30
    rows = _mk(n, d, 1201 + metric, scale=None if metric == COS else (0.5, 2.0))
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(2)
        r0 = idx.counters()["bytes_resident"]
        alive = np.ones(n, bool)

        def check(nq, keep, thr=None, handle=False, seed=0):
            qs = _queries(oracle_mod, nq, d, 1300 + seed, metric)
            bits = np.packbits(keep, bitorder="little")
            em = np.packbits(keep & alive, bitorder="little")
            c0 = idx.counters()
            if handle:
                with idx.make_mask(bits) as m:
                    got = idx.search(qs, k, thr, m)
            else:
                got = idx.search(qs, k, thr, bits)
            c1 = idx.counters()
            assert c1["batch_launches"] > c0["batch_launches"] and c1["fallback_searches"] == c0["fallback_searches"]
            ref = oracle_mod.search_heap_many_mt(rows, qs, metric, k, thr, em)
            assert np.array_equal(got[2], ref[2])
            for i in range(nq):
                c = got[2][i]
                assert np.array_equal(got[0][i, :c], ref[0][i, :c]), i
                assert np.array_equal(got[1][i, :c].view(np.uint64), ref[1][i, :c].view(np.uint64)), i
            return qs, got

        keep_a = rng.random(n) < 0.005  # ~1500 rows
        check(40, keep_a)
        check(300, keep_a, handle=True, seed=1)
        keep_b = np.zeros(n, bool)
        keep_b[rng.choice(n, 12_000, replace=False)] = True
        qs, got = check(70, keep_b, seed=2)
        ed = oracle_mod.search_heap(rows, qs[0], metric, k, None, np.packbits(keep_b, bitorder="little"))[1]
        check(70, keep_b, thr=float(ed[len(ed) // 2]), handle=True, seed=2)
        keep_c = np.zeros(n, bool)
        keep_c[123_457:123_457 + 3_000] = True  # WHERE id BETWEEN ...
        check(33, keep_c, seed=3)
        keep_d = np.zeros(n, bool)
        keep_d[[5, 77_777, n - 1] + list(range(200_000, 200_011))] = True  # fewer kept rows than k
        check(20, keep_d, handle=True, seed=4)
        # no copy of the whole shard was built for any of these calls (300 k rows of 64 fp16 would be 38 MB)
        assert idx.counters()["bytes_resident"] - r0 < 30_000_000
        # rows deleted after a handle was made: the list still names them, the kernels look at their live bits
        with idx.make_mask(np.packbits(keep_b, bitorder="little")) as m:
            dead = np.flatnonzero(keep_b)[::4]
            idx.set_deleted(dead)
            alive[dead] = False
            qs = _queries(oracle_mod, 50, d, 1400, metric)
            got = idx.search(qs, k, None, m)
            ref = oracle_mod.search_heap_many_mt(rows, qs, metric, k, None, np.packbits(keep_b & alive, bitorder="little"))
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint64), ref[1].view(np.uint64))
        # the same selective mask with the library free to choose: a few queries one by one, many batched
        idx.set_batch_min_nq(1)
        bits = np.packbits(keep_a, bitorder="little")
        c0 = idx.counters()
        four = idx.search(_queries(oracle_mod, 4, d, 1500, metric), k, None, bits)
        c1 = idx.counters()
        assert c1["batch_launches"] == c0["batch_launches"] and c1["exact_scans"] - c0["exact_scans"] == 4
        sixty = idx.search(_queries(oracle_mod, 60, d, 1501, metric), k, None, bits)
        c2 = idx.counters()
        assert c2["batch_launches"] == c1["batch_launches"] + 1 and c2["scan_launches"] == c1["scan_launches"]
        assert (four[2] == k).all() and (sixty[2] == k).all()
