"""CPU: the oracle (C restatement) against the committed golden fixtures, the
second (NumPy) restatement, and hand-derivable / README known answers."""
import hashlib
import json
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
L2, IP, COS = 0, 1, 2


def _bits(a):
    return np.asarray(a, np.float64).view(np.uint64)


def test_readme_and_hand_kats(oracle_mod):
    with open(os.path.join(GOLD, "kat.json")) as f:
        kat = json.load(f)
    d = 128
    rows = np.stack([oracle_mod.to_float32([i * 0.01 for i in range(d)], d),
                     oracle_mod.to_float32([i * 0.02 + 0.5 for i in range(d)], d)])
    q = oracle_mod.to_float32([i * 0.015 for i in range(d)], d)
    for c in kat["readme"]:
        qq = oracle_mod.normalize_f32(q) if c["metric"] == COS else q
        ids, dist = oracle_mod.search_exhaustive(rows, qq, c["metric"], c["k"])
        assert ids.tolist() == c["ids"]
        assert [float(x).hex() for x in dist] == c["dist_hex"]
        assert [oracle_mod.distance_to_score(x, c["metric"]) for x in dist] == c["score"]
    # SURVEY.md section 8c seed values
    cos = [c for c in kat["readme"] if c["metric"] == COS][0]
    assert cos["dist"] == [8.881784197001252e-16, 0.008631337545516038]
    assert cos["score"] == [0.9999999999999991, 0.991368662454484]
    l2 = [c for c in kat["readme"] if c["metric"] == L2][0]
    assert l2["dist"] == [4.155959577530976, 9.482193837605186]
    for h in kat["hand"]:
        ids, dist = oracle_mod.search_exhaustive(np.asarray(h["rows"], np.float32),
                                                 np.asarray(h["query"], np.float32), h["metric"], 1)
        assert [float(x) for x in dist] == h["dist"], h["name"]
        assert math.copysign(1, dist[0]) == math.copysign(1, h["dist"][0])


def test_random_small_fixture(oracle_mod):
    z = np.load(os.path.join(GOLD, "random_small.npz"))
    with open(os.path.join(GOLD, "random_small.json")) as f:
        meta = json.load(f)
    rows, queries, keep = z["rows"], z["queries"], z["keep"]
    assert len(meta) == 3 * 3 * 3 * 2
    for m in meta:
        q = queries[m["query"]]
        if m["metric"] == COS:
            q = oracle_mod.normalize_f32(q)
        kp = keep if m["mask"] else None
        for fn in (oracle_mod.search_exhaustive, oracle_mod.search_heap):
            ids, dist = fn(rows, q, m["metric"], m["k"], None, kp)
            assert np.array_equal(ids, z[m["key"] + "_ids"]), m["key"]
            assert np.array_equal(_bits(dist), _bits(z[m["key"] + "_dist"])), m["key"]
            thr = float.fromhex(m["threshold_hex"])
            ids, dist = fn(rows, q, m["metric"], m["k"], thr, kp)
            assert np.array_equal(ids, z[m["key"] + "_thr_ids"]), m["key"]
            assert np.array_equal(_bits(dist), _bits(z[m["key"] + "_thr_dist"])), m["key"]
            assert (dist <= thr).all()  # strict '>' drops; equality is kept
    for metric in (L2, IP, COS):
        q = oracle_mod.normalize_f32(queries[0]) if metric == COS else queries[0]
        ids, dist = oracle_mod.search_exhaustive(z["bad_rows"], q, metric, 80)
        assert np.array_equal(ids, z[f"bad_m{metric}_ids"])
        a, b = dist, z[f"bad_m{metric}_dist"]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
        # double.compareTo: NaN sorts last
        nan_at = np.flatnonzero(np.isnan(a))
        assert len(nan_at) == 0 or nan_at[0] == len(a) - len(nan_at)


def test_config_c1_fixture(oracle_mod):
    """BASELINE.json configs[0]: 10k x 128 f32, k=10, the reference's own CPU-runnable case."""
    with open(os.path.join(GOLD, "config_c1.json")) as f:
        g = json.load(f)
    rows = np.random.Generator(np.random.Philox(20260612)).standard_normal((10000, 128)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    qs = np.random.Generator(np.random.Philox(20260613)).standard_normal((4, 128)).astype(np.float32)
    if hashlib.sha256(rows.tobytes()).hexdigest() != g["rows_sha256"]:
        pytest.skip("this NumPy build draws a different Philox stream than the fixture's")
    for c in g["cases"]:
        q = oracle_mod.normalize_f32(qs[c["query"]]) if c["metric"] == COS else qs[c["query"]]
        ids, dist = oracle_mod.search_heap(rows, q, c["metric"], c["k"])
        assert ids.tolist() == c["ids"]
        assert [float(x).hex() for x in dist] == c["dist_hex"]


def test_two_restatements_agree(oracle_mod):
    from oracle import np_oracle as npo

    rng = np.random.default_rng(5)
    for d in (1, 3, 16, 130):
        rows = (rng.standard_normal((400, d)) * rng.uniform(0.1, 10, (400, 1))).astype(np.float32)
        q = rng.standard_normal(d).astype(np.float32)
        assert np.array_equal(oracle_mod.normalize_f32(q), npo.normalize_f32(q))
        vals = rng.standard_normal(d + 3) * 1e3
        for dim in (d, d + 5, max(1, d - 1)):
            assert np.array_equal(oracle_mod.to_float32(vals, dim), npo.to_float32(vals, dim))
        for metric in (L2, IP, COS):
            assert np.array_equal(_bits(oracle_mod.all_distances(q, rows, metric)),
                                  _bits(npo.all_distances(q, rows, metric)))
            a = oracle_mod.search_exhaustive(rows, q, metric, 33)
            b = npo.search_exhaustive(rows, q, metric, 33)
            c = oracle_mod.search_heap_mt(rows, q, metric, 33, threads=3)
            assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1]))
            assert np.array_equal(a[0], c[0]) and np.array_equal(_bits(a[1]), _bits(c[1]))
        # the many-query OpenMP variant (recall checks over >= 1000 queries) is the same arithmetic
        qs = rng.standard_normal((9, d)).astype(np.float32)
        keep = np.packbits(rng.random(400) < 0.5, bitorder="little")
        for metric in (L2, IP, COS):
            ids, dist, cnt = oracle_mod.search_heap_many_mt(rows, qs, metric, 21, None, keep, threads=3)
            for i in range(9):
                e_ids, e_dist = oracle_mod.search_exhaustive(rows, qs[i], metric, 21, None, keep)
                assert cnt[i] == len(e_ids) and np.array_equal(ids[i, :cnt[i]], e_ids)
                assert np.array_equal(_bits(dist[i, :cnt[i]]), _bits(e_dist))


def test_two_restatements_agree_on_irregular_rows(oracle_mod):
    """rows with NaN / inf / huge / tiny values: the reference's f64 arithmetic gives them inf / NaN / 0 distances and
    `double.compareTo` orders those (NaN last, never `> threshold`) -- what the library's quarantined rows must
    reproduce (tests/test_gpu_irregular.py); here the C and the NumPy restatement agree on it bit for bit"""
    from oracle import np_oracle as npo

    rng = np.random.default_rng(11)
    d = 12
    rows = rng.standard_normal((300, d)).astype(np.float32)
    rows[3, 1] = np.nan
    rows[10] = np.nan
    rows[50, 0] = np.inf
    rows[51, 5] = -np.inf
    rows[52, 2] = np.inf
    rows[52, 3] = -np.inf      # inf - inf
    rows[100, 4] = 3e20        # finite; its square overflows f32, not f64
    rows[101] = -2e16
    rows[150] = 0.0
    rows[150, 7] = 1e-30       # tiny norm
    rows[151] = 0.0            # zero row: cosine similarity defined as 0
    keep = np.packbits(rng.random(300) < 0.7, bitorder="little")
    for qi in range(4):
        q = rng.standard_normal(d).astype(np.float32)
        for metric in (L2, IP, COS):
            qq = oracle_mod.normalize_f32(q) if metric == COS else q
            a = oracle_mod.all_distances(qq, rows, metric)
            b = npo.all_distances(qq, rows, metric)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(_bits(a[~np.isnan(a)]), _bits(b[~np.isnan(b)]))
            if metric == COS:  # `denom > 0 ? dot / denom : 0` (ngh_graph_engine.dart:945): a NaN denominator means similarity 0
                assert a[3] == 1.0 and a[10] == 1.0
            else:              # (row 52, inf and -inf: inf for L2, inf or NaN by the query's signs for IP)
                assert np.isnan(a[3]) and np.isnan(a[10])
            if metric == L2:
                assert a[50] == np.inf and a[51] == np.inf and np.isfinite(a[100]) and a[100] > 1e20
            if metric == COS:
                assert a[151] == 1.0 and np.isfinite(a[150])
            for k in (5, 300):
                for thr in (None, float(np.nanmedian(a)), -1e300):
                    for mask in (None, keep):
                        x = oracle_mod.search_exhaustive(rows, qq, metric, k, thr, mask)
                        y = npo.search_exhaustive(rows, qq, metric, k, thr, mask)
                        assert np.array_equal(x[0], y[0]), (metric, k, thr)
                        assert np.array_equal(np.isnan(x[1]), np.isnan(y[1]))
                        assert np.array_equal(_bits(x[1][~np.isnan(x[1])]), _bits(y[1][~np.isnan(y[1])]))
                        nan_at = np.flatnonzero(np.isnan(x[1]))
                        assert len(nan_at) == 0 or nan_at[0] == len(x[1]) - len(nan_at)  # NaN sorts last ...
                        if thr is not None and mask is None and k == 300 and metric != COS:
                            assert {3, 10} <= set(x[0].tolist())                          # ... and passes any threshold


def test_properties_on_all_distances(oracle_mod):
    """size-independent properties: symmetry of L2, IP linearity in sign, cosine in [0,2]."""
    rng = np.random.default_rng(9)
    rows = rng.standard_normal((200, 64)).astype(np.float32)
    q = rng.standard_normal(64).astype(np.float32)
    d_l2 = oracle_mod.all_distances(q, rows, L2)
    assert all(oracle_mod.exact_distance(rows[i], q, L2) == d_l2[i] for i in range(0, 200, 17))
    d_ip = oracle_mod.all_distances(q, rows, IP)
    assert np.array_equal(oracle_mod.all_distances(-q, rows, IP), -d_ip)
    d_c = oracle_mod.all_distances(oracle_mod.normalize_f32(q), rows, COS)
    assert (d_c >= -1e-15).all() and (d_c <= 2 + 1e-15).all()
    # sums relate to distances exactly as the host finaliser assumes
    s0, _ = oracle_mod.exact_sums(q, rows[3], L2)
    assert math.sqrt(s0) == d_l2[3]
    s0, _ = oracle_mod.exact_sums(q, rows[3], IP)
    assert -s0 == d_ip[3]


def test_compare_and_score_edges(oracle_mod):
    from oracle import np_oracle as npo

    nan, inf = math.nan, math.inf
    vals = [-inf, -1.5, -0.0, 0.0, 1e-300, 2.0, inf, nan]
    for i, a in enumerate(vals):
        for j, b in enumerate(vals):
            want = (i > j) - (i < j)
            assert oracle_mod.compare_double(a, b) == want == npo.compare_double(a, b)
    assert oracle_mod.compare_double(nan, nan) == 0
    for metric in (L2, IP, COS):
        for x in (0.0, -0.0, 0.5, 1.0, 1.5, 2.0, -3.0, 1e300, nan, inf, 8.881784197001252e-16):
            a, b = oracle_mod.distance_to_score(x, metric), npo.distance_to_score(x, metric)
            assert (math.isnan(a) and math.isnan(b)) or a == b, (metric, x, a, b)
    assert oracle_mod.distance_to_score(nan, COS) == 1.0  # num.clamp: NaN compares greatest
    assert oracle_mod.distance_to_score(1.0, COS) == 0.0 and math.copysign(1, oracle_mod.distance_to_score(1.0, COS)) == 1


def test_rawvec_page_format(oracle_mod):
    from oracle import np_oracle as npo

    with open(os.path.join(GOLD, "rawvec_pages.json")) as f:
        g = json.load(f)
    # sizer values quoted in SURVEY.md section 8a row A7
    assert oracle_mod.vectors_per_raw_page(16384, 128, 4) == g["sizer"]["d128"] == 31
    assert oracle_mod.vectors_per_raw_page(16384, 768, 4) == g["sizer"]["d768"] == 5
    assert oracle_mod.vectors_per_raw_page(16384, 1536, 4) == g["sizer"]["d1536"] == 2
    v = np.asarray(g["tiny_vectors"], np.float32)
    page = oracle_mod.rawvec_page_build(v, 1, 256)
    assert page.hex() == g["tiny_page_hex"] == npo.rawvec_page_build(v, 1, 256).hex()
    assert page[:4] == b"TPG2" and page[4:6] == b"\x14\x00" and page[6] == 8
    assert int.from_bytes(page[12:16], "little") == oracle_mod.crc32(page[20:20 + int.from_bytes(page[8:12], "little")])
    got, prec = oracle_mod.rawvec_page_parse(page, 4, 2)
    assert prec == 1 and np.array_equal(got.view(np.uint32), v.view(np.uint32))
    bad = bytearray(page)
    bad[30] ^= 1
    assert oracle_mod.rawvec_page_parse(bytes(bad), 4, 2) is None  # CRC mismatch
    rng = np.random.default_rng(7)
    for ent in g["pages"]:
        vv = (rng.standard_normal((ent["vectors_per_page"], ent["dims"])) * 0.5).astype(np.float32)
        pc = oracle_mod.rawvec_page_build(vv, ent["precision"], 16384)
        assert hashlib.sha256(pc).hexdigest() == ent["page_sha256"]
        assert int.from_bytes(pc[12:16], "little") == ent["payload_crc32"]
        parsed, _ = oracle_mod.rawvec_page_parse(pc, ent["dims"], ent["vectors_per_page"])
        assert hashlib.sha256(parsed.tobytes()).hexdigest() == ent["decoded_sha256"]
        if ent["precision"] == 1:
            assert np.array_equal(parsed, vv)
        elif ent["precision"] == 2:
            assert np.abs(parsed - np.clip(vv, -1, 1)).max() <= 0.5 / 127 + 1e-7
    assert oracle_mod.crc32(b"123456789") == 0xCBF43926  # the CRC-32/IEEE check value
    # node id -> (partition, page, slot): model/ngh_index_meta.dart:480-490
    assert oracle_mod.rawvec_locate(0, 5, 1024) == (0, 1, 0)
    assert oracle_mod.rawvec_locate(5 * 1024 - 1, 5, 1024) == (0, 1024, 4)
    assert oracle_mod.rawvec_locate(5 * 1024, 5, 1024) == (1, 1, 0)


@pytest.mark.parametrize("n,sd,k", [(300, 8, 64), (250, 6, 32), (200, 4, 200), (120, 3, 16)])
def test_pq_train_restatements_agree(oracle_mod, n, sd, k):
    """N4 training: C and NumPy restatements of trainPqSubspace, written separately, agree bit for bit."""
    from oracle import np_oracle as npo

    rng = np.random.default_rng(n)
    data = (rng.standard_normal((n, sd)) + rng.integers(0, 5, (n, 1)) * 2.5).astype(np.float32)
    init = rng.integers(0, n, size=k).astype(np.int32)
    for iters in (1, 10):
        a = oracle_mod.pq_train(data, 1, k, iters, init)[0]
        b = npo.pq_train_subspace(data, k, iters, init)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,sd,k", [(60, 8, 60), (99, 4, 99), (37, 6, 37), (50, 3, 20), (5, 4, 1)])
def test_pq_train_kmeanspp_restatements_agree(oracle_mod, n, sd, k):
    """N3: the trainer for first batches of fewer than 100 vectors (VectorQuantizer.train, ref
    core/vector_quantizer.dart:81-350: k-means++ seeding + Lloyd on squared distances): the C and the NumPy
    restatement, written separately, agree bit for bit -- Float32x4 path (sd % 4 == 0) and scalar path."""
    from oracle import np_oracle as npo

    rng = np.random.default_rng(1000 + n)
    data = (rng.standard_normal((n, sd)) + rng.integers(0, 4, (n, 1)) * 3.0).astype(np.float32)
    data[n // 2] = data[0]  # a duplicated sample: a zero distance inside the seeding
    first = int(rng.integers(0, n))
    draws = rng.random(max(k - 1, 0))
    for iters in (1, 10):
        a = oracle_mod.pq_train_pp(data, 1, k, iters, [first], draws)[0]
        b = npo.pq_train_subspace_pp(data, k, iters, first, draws)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # hand-checkable: k = n well separated points, draws that walk the cumulative distances -> every sample is chosen once
    pts = (np.arange(6, dtype=np.float32)[:, None] * 10.0 + np.zeros((6, 4), np.float32))
    c = oracle_mod.pq_train_pp(pts, 1, 6, 10, [0], np.full(5, 0.999999))[0]
    assert sorted(c[:, 0].tolist()) == [0.0, 10.0, 20.0, 30.0, 40.0, 50.0]
    # all samples identical: total distance 0 -> the last sample is selected every time (:160-169)
    same = np.ones((4, 4), np.float32)
    assert np.array_equal(oracle_mod.pq_train_pp(same, 1, 4, 3, [2], np.full(3, 0.5))[0], same)


def test_ngh_directory_writer_reader_round_trip(oracle_mod, tmp_path):
    """N1 fixtures: the directory writer restatement (oracle/ngh_dir.py) and the reader restatement agree,
    across partition files and dir_N buckets, for every stored precision."""
    from oracle import ngh_dir

    rng = np.random.default_rng(9)
    v = (rng.standard_normal((1200, 96)) * 0.5).astype(np.float32)
    for precision in (1, 0, 2):
        root = tmp_path / f"ngh{precision}"
        meta = ngh_dir.write_ngh_dir(str(root), v, metric=1, precision=precision, max_partition_file_size=16384 * 4,
                                     deleted=[0, 7, 1199], max_entries_per_dir=2)
        assert precision == 2 or (root / "rawvec" / "dir_1" / "p2.ngh").exists()
        m2, vec, dead = ngh_dir.read_ngh_dir(str(root), 2)
        assert m2 == meta and np.flatnonzero(dead).tolist() == [0, 7, 1199]
        if precision != 2:
            assert np.array_equal(vec, v)
        else:  # int8 pages: clamp to [-1,1], *127 rounded half away from zero, read back as /127
            c = np.clip(v.astype(np.float64), -1, 1) * 127
            q = np.where(c < 0, -np.floor(-c + 0.5), np.floor(c + 0.5))
            assert np.array_equal(vec, (q / 127.0).astype(np.float32))
