#!/usr/bin/env python
"""Generates tests/golden/*.json|*.npz.

The reference ships no golden vectors for vectorSearch and cannot run here
(pure Dart, no SDK), so these fixtures are produced by the two independent
restatements in oracle/ (C and NumPy).  A fixture is only written when BOTH
agree bit for bit on every distance and id.  "PARITY UNPINNED" by the
reference itself -- see oracle/vs_oracle.h.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from oracle import np_oracle as npo  # noqa: E402

L2, IP, COS = 0, 1, 2


def both(rows, q, metric, k, thr=None, keep=None):
    a = oracle.search_exhaustive(rows, q, metric, k, thr, keep)
    b = npo.search_exhaustive(rows, q, metric, k, thr, keep)
    assert np.array_equal(a[0], b[0]), "restatements disagree on ids"
    assert np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)), "restatements disagree on distances"
    return a


def hexf(x):
    return float(x).hex()


def kat_file():
    """README / example vectors (/root/reference/example/lib/tostore_example.dart:387-413,
    README.md:615-625) and hand-derivable cases."""
    d = 128
    v1 = [i * 0.01 for i in range(d)]
    v2 = [i * 0.02 + 0.5 for i in range(d)]
    q = [i * 0.015 for i in range(d)]
    rows = np.stack([oracle.to_float32(v1, d), oracle.to_float32(v2, d)])
    assert np.array_equal(rows, np.stack([npo.to_float32(v1, d), npo.to_float32(v2, d)]))
    qf = oracle.to_float32(q, d)
    cases = []
    for metric in (L2, IP, COS):
        qq = oracle.normalize_f32(qf) if metric == COS else qf
        if metric == COS:
            assert np.array_equal(qq, npo.normalize_f32(qf))
        ids, dist = both(rows, qq, metric, 5)
        cases.append({"name": "readme_example", "metric": metric, "dim": d,
                      "rows_formula": ["i*0.01", "i*0.02+0.5"], "query_formula": "i*0.015", "k": 5,
                      "ids": ids.tolist(), "dist_hex": [hexf(x) for x in dist],
                      "dist": [float(x) for x in dist],
                      "score": [oracle.distance_to_score(x, metric) for x in dist]})
    hand = [
        {"name": "l2_3_4_5", "metric": L2, "rows": [[3, 4, 0, 0]], "query": [0, 0, 0, 0], "dist": [5.0], "score": [1.0 / 6.0]},
        {"name": "cos_orthogonal", "metric": COS, "rows": [[0, 1, 0, 0]], "query": [1, 0, 0, 0], "dist": [1.0], "score": [0.0]},
        {"name": "cos_zero_row", "metric": COS, "rows": [[0, 0, 0, 0]], "query": [1, 0, 0, 0], "dist": [1.0], "score": [0.0]},
        {"name": "cos_same_dir", "metric": COS, "rows": [[2, 0, 0, 0]], "query": [1, 0, 0, 0], "dist": [0.0], "score": [1.0]},
        {"name": "cos_opposite", "metric": COS, "rows": [[-3, 0, 0, 0]], "query": [1, 0, 0, 0], "dist": [2.0], "score": [0.0]},
        {"name": "ip_negated", "metric": IP, "rows": [[1, 2, 3, 4]], "query": [1, 1, 1, 1], "dist": [-10.0],
         "score": [1.0 / (1.0 + np.exp(-10.0))]},
        {"name": "ip_zero", "metric": IP, "rows": [[0, 0, 0, 0]], "query": [1, 1, 1, 1], "dist": [-0.0], "score": [0.5]},
    ]
    for h in hand:
        r = np.asarray(h["rows"], np.float32)
        qq = np.asarray(h["query"], np.float32)
        ids, dist = both(r, qq, h["metric"], 1)
        assert [float(x) for x in dist] == h["dist"], (h["name"], dist)
        sc = [oracle.distance_to_score(x, h["metric"]) for x in dist]
        assert np.allclose(sc, h["score"], rtol=0, atol=1e-15), (h["name"], sc)
        assert sc == [npo.distance_to_score(x, h["metric"]) for x in dist]
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump({"readme": cases, "hand": hand}, f, indent=1)


def random_file():
    """Seeded small corpora with edge content; inputs stored explicitly."""
    rng = np.random.default_rng(20260612)
    n, d = 600, 24
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[10] = rows[11] = rows[12] = rows[300]          # duplicates -> ties broken by id
    rows[50] = 0.0                                       # zero vector
    rows[51] = -0.0
    rows[60] *= 1e-3
    rows[61] *= 1e3
    queries = rng.standard_normal((3, d)).astype(np.float32)
    queries[2] = rows[300]                               # exact hit (distance 0 / ties)
    keep = np.packbits(rng.random(n) < 0.4, bitorder="little")
    arrays = {"rows": rows, "queries": queries, "keep": keep}
    meta = []
    for metric in (L2, IP, COS):
        for qi in range(3):
            q = oracle.normalize_f32(queries[qi]) if metric == COS else queries[qi]
            for k in (1, 10, 700):
                for use_keep in (False, True):
                    kp = keep if use_keep else None
                    ids, dist = both(rows, q, metric, k, None, kp)
                    thr = float(dist[min(4, len(dist) - 1)])
                    ids_t, dist_t = both(rows, q, metric, k, thr, kp)
                    key = f"m{metric}_q{qi}_k{k}_{'mask' if use_keep else 'all'}"
                    arrays[key + "_ids"] = ids
                    arrays[key + "_dist"] = dist
                    arrays[key + "_thr_ids"] = ids_t
                    arrays[key + "_thr_dist"] = dist_t
                    meta.append({"key": key, "metric": metric, "query": qi, "k": k, "mask": use_keep,
                                 "threshold_hex": hexf(thr)})
    # non-finite content: NaN sorts last (double.compareTo), inf before it
    bad = rows[:80].copy()
    bad[3, 1] = np.nan
    bad[7] = np.inf
    bad[9, 0] = -np.inf
    arrays["bad_rows"] = bad
    for metric in (L2, IP, COS):
        q = oracle.normalize_f32(queries[0]) if metric == COS else queries[0]
        ids, dist = both(bad, q, metric, 80)
        arrays[f"bad_m{metric}_ids"] = ids
        arrays[f"bad_m{metric}_dist"] = dist
    np.savez_compressed(os.path.join(HERE, "random_small.npz"), **arrays)
    with open(os.path.join(HERE, "random_small.json"), "w") as f:
        json.dump(meta, f)


def config_c1_file():
    """BASELINE.json config C1: 10k x 128 f32, L2, k=10, single query.  Inputs are
    regenerated from the seed (checksummed); expected outputs stored."""
    import hashlib

    rng = np.random.Generator(np.random.Philox(20260612))
    rows = rng.standard_normal((10000, 128)).astype(np.float32)
    rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    qs = np.random.Generator(np.random.Philox(20260613)).standard_normal((4, 128)).astype(np.float32)
    out = {"rows_sha256": hashlib.sha256(rows.tobytes()).hexdigest(),
           "queries_sha256": hashlib.sha256(qs.tobytes()).hexdigest(), "cases": []}
    for metric in (L2, IP, COS):
        for qi in range(4):
            q = oracle.normalize_f32(qs[qi]) if metric == COS else qs[qi]
            ids, dist = both(rows, q, metric, 10)
            out["cases"].append({"metric": metric, "query": qi, "k": 10, "ids": ids.tolist(),
                                 "dist_hex": [hexf(x) for x in dist]})
    with open(os.path.join(HERE, "config_c1.json"), "w") as f:
        json.dump(out, f)


def pages_file():
    """Raw-vector pages written by the byte-exact page-writer restatements (C and
    NumPy must produce identical bytes).  ref: ngh_page.dart:310-450, btree_page.dart:132-234."""
    import hashlib

    rng = np.random.default_rng(7)
    out = []
    for prec in (1, 0, 2):
        for dims in (4, 128, 768):
            bpe = {0: 8, 1: 4, 2: 1}[prec]
            vpp = oracle.vectors_per_raw_page(16384, dims, bpe)
            assert vpp == npo.vectors_per_raw_page(16384, dims, bpe)
            v = (rng.standard_normal((vpp, dims)) * 0.5).astype(np.float32)
            pc = oracle.rawvec_page_build(v, prec, 16384)
            pn = npo.rawvec_page_build(v, prec, 16384)
            assert pc == pn, "page writers disagree"
            parsed, p2 = oracle.rawvec_page_parse(pc, dims, vpp)
            assert p2 == prec
            out.append({"precision": prec, "dims": dims, "vectors_per_page": vpp,
                        "page_sha256": hashlib.sha256(pc).hexdigest(),
                        "payload_crc32": int.from_bytes(pc[12:16], "little"),
                        "decoded_sha256": hashlib.sha256(parsed.tobytes()).hexdigest(), "seed_note": "rng(7) sequential"})
    # one tiny page stored verbatim (hex) so the format is pinned without regenerating
    v = np.array([[1.0, -2.0, 0.5, 0.25], [3.0, 4.0, -0.0, 1e-3]], np.float32)
    page = oracle.rawvec_page_build(v, 1, 256)
    assert page == npo.rawvec_page_build(v, 1, 256)
    with open(os.path.join(HERE, "rawvec_pages.json"), "w") as f:
        json.dump({"sizer": {"d128": 31, "d768": 5, "d1536": 2}, "pages": out,
                   "tiny_page_hex": page.hex(), "tiny_vectors": v.tolist()}, f)


if __name__ == "__main__":
    oracle.build()
    kat_file()
    random_file()
    config_c1_file()
    pages_file()
    print("golden fixtures written to", HERE)
