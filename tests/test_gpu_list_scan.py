"""GPU: selective row masks are scanned as a compacted LIST of row ids (scan_list_kernel + the select kernel on
list-ordered keys): every width class, metric and edge the tile-walking masked scan is
tested with, at selectivities where the list path runs (below one kept row in 24)."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]  # small shapes: exact sums AND the f32 pre-filter (conftest)


def _check(idx, oracle_mod, rows, qs, metric, k, mask, thr=None, base=0):
    ids, dist, cnt = idx.search(qs, k, thr, mask)
    for i in range(len(qs)):
        e, ed = oracle_mod.search_exhaustive(rows, qs[i], metric, k, thr, None if mask is None else _local(mask, base, len(rows)))
        assert cnt[i] == len(e), (i, cnt[i], len(e))
        assert np.array_equal(ids[i, :cnt[i]], e + base), i
        assert np.array_equal(dist[i, :cnt[i]].view(np.uint64), ed.view(np.uint64)), i


def _local(mask, base, n):
    bits = np.unpackbits(np.asarray(mask, np.uint8), bitorder="little")[base:base + n]
    return np.packbits(bits, bitorder="little")


@pytest.mark.usefixtures("mask_form")  # the mask as a pointer, and as a device-resident handle
@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [100, 384, 768, 1000, 1536, 2048])
def test_selective_masks_every_width(hip_lib, oracle_mod, metric, d, scan_path):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(d + metric)
    n, k = 40_000, 20
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((9, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)  # every query scans on its own
        for keep in (0.004, 0.03):
            mask = np.packbits(rng.random(n) < keep, bitorder="little")
            c0 = idx.counters()
            _check(idx, oracle_mod, rows, qs, metric, k, mask)      # a call of several queries (one list for all)
            _check(idx, oracle_mod, rows, qs[:1], metric, k, mask)  # a lone query
            c1 = idx.counters()
            assert c1["fallback_searches"] == c0["fallback_searches"]
            # the path under test is the one that ran: every one of these scans took the compacted list
            assert c1["list_scans"] - c0["list_scans"] == c1["scan_launches"] - c0["scan_launches"] == len(qs) + 1
            # ... as f32 keys of the listed rows, or (the product's choice for so few rows) as their exact sums
            assert c1["exact_scans"] - c0["exact_scans"] == (len(qs) + 1 if scan_path != "prefilter" else 0)
        # a mild mask (one kept row in 10 > 1 in 24) walks the tiles: no list scan -- unless its 4000 kept rows are few
        # enough for the exact path, which reads nothing but its list at any selectivity
        mask = np.packbits(rng.random(n) < 0.1, bitorder="little")
        c0 = idx.counters()
        _check(idx, oracle_mod, rows, qs[:2], metric, k, mask)
        c1 = idx.counters()
        assert c1["scan_launches"] - c0["scan_launches"] == 2
        assert c1["list_scans"] - c0["list_scans"] == c1["exact_scans"] - c0["exact_scans"] == (2 if scan_path != "prefilter" else 0)
        # fewer kept rows than k, and a threshold
        mask = np.zeros(n, bool)
        mask[[5, 77, 20_000, n - 1]] = True
        mask = np.packbits(mask, bitorder="little")
        _check(idx, oracle_mod, rows, qs[:3], metric, k, mask)
        thr = {0: float(np.sqrt(d) * 1.3), 1: -1.0, 2: 0.98}[metric]
        mask = np.packbits(rng.random(n) < 0.02, bitorder="little")
        _check(idx, oracle_mod, rows, qs[:3], metric, k, mask, thr)


@pytest.mark.usefixtures("mask_form")
def test_tombstones_ties_ranges_and_shards(hip_lib, oracle_mod, scan_path):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(3)
    n, d, k = 60_000, 256, 15
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((5, d)).astype(np.float32)
    base = 7_000  # a shard handle: global ids, global mask
    gmask = np.zeros(base + n, bool)
    gmask[base + 30_000:base + 31_500] = True  # WHERE id BETWEEN ...: one contiguous run (2.5 % of the rows)
    with HipVectorIndex(d, 0, capacity_rows=n, shard_device=0, row_base=base) as idx:
        idx.append(base, rows)
        idx.set_batch_min_nq(0)
        m = np.packbits(gmask, bitorder="little")
        _check(idx, oracle_mod, rows, qs, 0, k, m, None, base)
        # rows deleted after the mask was made stay out
        dead = np.arange(base + 30_000, base + 31_500, 3)
        idx.set_deleted(dead)
        g2 = gmask.copy()
        g2[dead] = False
        ids, dist, cnt = idx.search(qs, k, None, m)
        for i in range(len(qs)):
            e, ed = oracle_mod.search_exhaustive(rows, qs[i], 0, k, None, _local(np.packbits(g2, bitorder="little"), base, n))
            assert np.array_equal(ids[i, :cnt[i]], e + base) and np.array_equal(dist[i, :cnt[i]], ed)
    # ties wider than the candidate list inside a selective mask: the wide-band fallback works on list-ordered keys
    same = np.tile(rows[:1], (n, 1))
    same[::7] = rows[::7]
    with HipVectorIndex(d, 0, capacity_rows=n) as idx:
        idx.append(0, same)
        idx.set_batch_min_nq(0)
        keep = np.zeros(n, bool)
        keep[rng.choice(n, 2_000, replace=False)] = True
        keep[::7] = False  # only duplicates of row 0 are kept: 2000-way tie
        m = np.packbits(keep, bitorder="little")
        c0 = idx.counters()
        ids, dist, cnt = idx.search(rows[0], 10, None, m)
        want = np.flatnonzero(keep)[:10]
        assert cnt[0] == 10 and ids[0].tolist() == want.tolist() and np.all(dist[0] == 0.0)
        if scan_path == "prefilter":
            assert idx.counters()["fallback_searches"] > c0["fallback_searches"]
        else:  # 2000 kept rows: their exact sums, the ten lowest ids of the tie
            assert idx.counters()["fallback_searches"] == c0["fallback_searches"]
            assert idx.counters()["exact_scans"] == c0["exact_scans"] + 1
            # (the wide pick meets a cut bin of 2000 tied rows: finished by the one-workgroup select, which ranks by id)
            assert idx.counters()["exact_redone"] - c0["exact_redone"] == (1 if scan_path == "exact" else 0)


def test_bench_hook_measures_the_list_kernel(hip_lib):
    """tsh_bench_scan with a selective mask times the kernel a search with that mask runs."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(0)
    n, d = 200_000, 768
    rows = rng.standard_normal((n, d)).astype(np.float32)
    with HipVectorIndex(d, 0, capacity_rows=n) as idx:
        idx.append(0, rows)
        sel = np.packbits(rng.random(n) < 0.01, bitorder="little")
        mild = np.packbits(rng.random(n) < 0.5, bitorder="little")
        t_sel, t_mild = idx.bench_scan(rows[0], 20, sel), idx.bench_scan(rows[0], 20, mild)
        assert 0 < t_sel < t_mild
