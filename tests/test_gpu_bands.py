"""GPU: every pre-filter key kernel against the error bound the host claims for it, on ADVERSARIAL rows.

The exhaustive path returns bit-exact ids because each candidate generator (the f32 scan, and the f32-MFMA /
bf16x3 / fp16 batched key kernels) is paired with a proven bound D on |key - exact key| and everything within
tau + 2 D of the k-th key is re-ranked in f64.  Gaussian test data never comes near those
bounds.  Here, per kernel, metric and row width:

  1. candidates: rows nearly parallel to the query (so sum |q_i v_i| ~ |q| |v|, the case the bounds are tight for),
     all products of one sign, the operands sitting just below / just above the rounding midpoints of the format
     the kernel rounds them to (fp16: 10 mantissa bits; bf16 hi + lo: 16) -- every operand rounding errs the same
     way within a row -- and with exact keys spread over a few D around a boundary K;
  2. the kernel's ACTUAL keys of all candidates are read back (tsh_probe_*_keys) and compared with the exact f64
     keys: max |key - exact| / D must be <= 1, and is recorded -- a bound that only holds by luck would show as a
     ratio near (or above) 1;
  3. the worst cases are selected: G = the k rows truly better than K whose approximate keys look WORST, B = rows
     truly worse than K whose approximate keys look BEST.  In a corpus of G + B + far-away filler rows the true top k is exactly G while
     the approximate ranking prefers B -- the search must still return G, bit for bit, without any fallback.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L2, IP, COS = 0, 1, 2
SCAN = -1
K = 50
N_CAND = 8192
RESULTS = []


def _exact_keys(rows, q, metric):
    r64, q64 = rows.astype(np.float64), q.astype(np.float64)
    if metric == L2:
        d = r64 - q64
        return np.einsum("ij,ij->i", d, d)
    dot = r64 @ q64
    if metric == IP:
        return -dot
    return -dot / np.sqrt(np.einsum("ij,ij->i", r64, r64))


def _patterned(rng, shape, path, up):
    """Positive f32 values 2^e (1 + m), e in {0, -1}, whose low mantissa bits sit just below (up = False) or just
    above (up = True) the rounding midpoint of the format `path` rounds operands to."""
    e = rng.integers(-1, 1, size=shape)
    if path == 2:  # fp16: 10 mantissa bits kept; bit 11 set = midpoint, -/+ 2^-21
        j = rng.integers(200, 248, size=shape)  # (room for _nudge_to in both directions)
        m = j * 2.0 ** -10 + 2.0 ** -11 + (2.0 ** -21 if up else -2.0 ** -21)
    elif path == 1:  # bf16 hi + lo: 16 mantissa bits kept; low 7 bits 0b1000001 / 0b0111111
        hi = rng.integers(0, 1 << 14, size=shape)  # a quarter of the range: values stay within [1, 1.25)
        m = (hi * 128 + (0x41 if up else 0x3F)) * 2.0 ** -23
    else:  # f32 operands are exact: only the accumulation rounds
        m = rng.integers(0, 1 << 21, size=shape) * 2.0 ** -23
    x = np.ldexp(1.0 + m, e)
    assert np.array_equal(x.astype(np.float32).astype(np.float64), x)
    return x.astype(np.float32)


def _nudge_to(rows, q, metric, path, target, rng, passes=8):
    """Move each row's exact key to target[i]: per pass one random element of the row takes as much of the remaining
    difference as it can, in whole steps of the lowest mantissa bit ABOVE the bits that carry the rounding pattern
    (so the pattern survives) and without leaving the pattern's value range."""
    step_bit = {2: -10, 1: -16}.get(path, -23)
    fmax = {2: 900.0, 1: float(1 << 14)}.get(path, float(1 << 23)) - 1.0  # mantissa field, in steps
    rows = rows.copy()
    n, d = rows.shape
    for it in range(passes):
        diff = target - _exact_keys(rows, q, metric)
        if not np.any(diff):
            break
        if metric == COS:
            r64 = rows.astype(np.float64)
            dots, nrms = r64 @ q.astype(np.float64), np.sqrt(np.einsum("ij,ij->i", r64, r64))
        for i in np.nonzero(diff)[0]:
            c = int(rng.integers(0, d))
            x = float(rows[i, c])
            e = np.floor(np.log2(x))
            unit = 2.0 ** (e + step_bit)
            if metric == L2:  # (q - x - dx)^2 - (q - x)^2 = diff: the root of smaller size, or as close as x = q gets
                a = float(q[c]) - x
                disc = a * a + diff[i]
                dx = a - np.copysign(np.sqrt(disc), a) if disc > 0 else a
            elif metric == IP:  # d key / d x = -q
                dx = diff[i] / -float(q[c])
            else:  # key = -q.v / |v|: d key / d x = -(q_c - (q.v) x / |v|^2) / |v|  (small for v near q: many passes)
                g = -(float(q[c]) - dots[i] * x / nrms[i] ** 2) / nrms[i]
                if abs(g) < 1e-12:
                    continue
                dx = diff[i] / g
            f = np.floor((x / 2.0 ** e - 1.0) / 2.0 ** step_bit)  # the field's current value
            steps = np.clip(np.round(dx / unit), -f, fmax - f)
            nx = x + steps * unit
            rows[i, c] = np.float32(nx)
            assert float(rows[i, c]) == nx
    return rows


def _candidates(rng, d, metric, path, delta_abs_guess, normalize):
    """A query with four distinct values (a fixed random assignment to the positions), two base rows -- operands
    just below / just above the rounding midpoints -- brought to the SAME exact key, and N_CAND rows that permute a
    base row's elements among positions of equal query value: the exact key of a permuted row is the base row's (so
    is its norm), what changes is how the kernel's roundings fall.  A last nudge spreads the exact keys over +-3 D."""
    classes = rng.integers(0, 4, size=d)
    q = _patterned(rng, 4, path, up=False)[classes]
    if normalize is not None:
        q = normalize(q)
    v_dn, v_up = _patterned(rng, d, path, up=False), _patterned(rng, d, path, up=True)
    kb = float(_exact_keys(v_dn[None, :], q, metric)[0])
    v_up = _nudge_to(v_up[None, :], q, metric, path, np.array([kb]), rng, passes=3000)[0]
    assert abs(float(_exact_keys(v_up[None, :], q, metric)[0]) - kb) < 0.05 * delta_abs_guess
    cand = np.empty((N_CAND, d), np.float32)
    members = [np.nonzero(classes == c)[0] for c in range(4)]
    for i in range(N_CAND):
        src = v_up if i & 1 else v_dn
        for m in members:
            cand[i, m] = src[rng.permutation(m)]
    return q, _nudge_to(cand, q, metric, path, np.full(N_CAND, kb), rng), kb


def _open(d, metric, rows, path):
    from tostore_amd import HipVectorIndex

    idx = HipVectorIndex(d, metric, capacity_rows=len(rows))
    idx.append(0, rows)
    if path != SCAN:
        idx.set_batch_kernel(path)
        idx.set_batch_min_nq(2)
    else:
        idx.set_batch_min_nq(0)
    return idx


def _keys_and_bound(idx, q, others, metric, path, exact):
    """(approximate keys of the first query, per-key bound D as an absolute number per row)"""
    if path == SCAN:
        keys, eps_rel, delta_abs = idx.probe_scan_keys(q)
        if metric == L2:  # relative bound: eps_rel = 3 eps, |key - s| <= eps s
            return keys.astype(np.float64), (eps_rel / 3.0) * np.abs(exact) + delta_abs
        return keys.astype(np.float64), np.full(len(exact), delta_abs / 2.0 / 1.0001)
    keys, d2 = idx.probe_batch_keys(np.stack([q] + list(others)), K)
    return keys[0].astype(np.float64), np.full(len(exact), float(d2[0]) / 2.0 / 1.0001)


@pytest.mark.parametrize("d", [768, 1536])
@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("path", [SCAN, 0, 1, 2])
def test_band_holds_on_adversarial_rows(hip_lib, oracle_mod, path, metric, d):
    rng = np.random.default_rng(1000 * (path + 2) + 10 * metric + d)
    # a first look at the bound's size for this shape (any patterned rows will do)
    q0 = _patterned(rng, d, path, up=False)
    some = np.stack([_patterned(rng, d, path, up=False) for _ in range(4096)])
    if metric == COS:
        q0 = oracle_mod.normalize_f32(q0)
    with _open(d, metric, some, path) as idx:
        _, bound = _keys_and_bound(idx, q0, [some[1], some[2], some[3]], metric, path, _exact_keys(some, q0, metric))
    guess = float(np.median(bound))

    # ---- 1. candidates around a boundary, 2. the kernel's actual keys of all of them --------------------------
    q, cand, kb = _candidates(rng, d, metric, path, guess, oracle_mod.normalize_f32 if metric == COS else None)
    others = [_patterned(rng, d, path, up=True) for _ in range(3)]
    if metric == COS:
        others = [oracle_mod.normalize_f32(o) for o in others]
    # a first reading with every candidate AT the boundary tells how large this kernel's errors really are; the exact
    # keys are then spread over just that much, so that the approximate ranking around K is decided by the errors
    with _open(d, metric, cand, path) as idx:
        keys0, bound0 = _keys_and_bound(idx, q, others, metric, path, _exact_keys(cand, q, metric))
    emax = float(np.max(np.abs(keys0 - _exact_keys(cand, q, metric))))
    spread = min(1.9 * float(np.median(bound0)), max(1.5 * emax, 1e-3 * float(np.median(bound0))))
    cand = _nudge_to(cand, q, metric, path, kb + rng.uniform(-1.0, 1.0, size=N_CAND) * spread, rng)
    exact = _exact_keys(cand, q, metric)
    with _open(d, metric, cand, path) as idx:
        keys, bound = _keys_and_bound(idx, q, others, metric, path, exact)
    err = keys - exact
    ratio = np.abs(err) / bound
    assert np.all(np.isfinite(keys))
    assert ratio.max() <= 1.0, "a key is further from the exact value than the claimed bound: ratio %.3f" % ratio.max()

    # ---- 3. the worst cases around the boundary: G truly better but looking worst, B truly worse but looking best
    better = np.nonzero((exact < kb) & (exact > kb - spread))[0]
    worse = np.nonzero((exact > kb) & (exact < kb + spread))[0]
    assert len(better) >= 4 * K and len(worse) >= 4 * K, (len(better), len(worse))
    g = better[np.argsort(-keys[better])[:K]]  # truly better than K, with the WORST-looking approximate keys
    b = worse[np.argsort(keys[worse])[:2 * K]]  # truly worse than K, with the BEST-looking approximate keys
    n_total, n_sample = 40960, 8192
    filler = np.stack([_patterned(rng, d, path, up=bool(i & 1)) for i in range(1024)])
    filler = np.tile(filler, (n_total // 1024, 1))[:n_total] * rng.choice([-1.0, 1.0], size=(n_total, d)).astype(np.float32)
    rows = filler
    # half of G and B inside the sample the batched path draws its threshold from, half in the filtered pass
    pos_g = np.concatenate([rng.choice(n_sample, K // 2, replace=False), n_sample + rng.choice(n_total - n_sample, K - K // 2, replace=False)])
    left = np.setdiff1d(np.arange(n_total), pos_g)
    pos_b = rng.choice(left, len(b), replace=False)
    rows[pos_g] = cand[g]
    rows[pos_b] = cand[b]
    ex_all = _exact_keys(rows, q, metric)
    true_top = set(np.argsort(ex_all, kind="stable")[:K].tolist())
    assert true_top == set(pos_g.tolist())  # the construction: G is the true top k
    queries = np.stack([q] + others)
    with _open(d, metric, rows, path) as idx:
        c0 = idx.counters()
        keys_all, bound_all = _keys_and_bound(idx, q, others, metric, path, ex_all)
        tau = np.sort(keys_all)[K - 1]  # k-th smallest APPROXIMATE key over all rows
        usage = float(np.max((keys_all[pos_g] - tau) / (2.0 * bound_all[pos_g])))
        looks_better = int(np.sum(keys_all[pos_b] < np.max(keys_all[pos_g])))
        if path == SCAN:
            got = [idx.search(x, K) for x in queries]
            ids = np.concatenate([x[0] for x in got])
            dist = np.concatenate([x[1] for x in got])
            cnt = np.concatenate([x[2] for x in got])
        else:
            ids, dist, cnt = idx.search(queries, K)
        c1 = idx.counters()
    e_ids, e_dist, e_cnt = oracle_mod.search_heap_many_mt(rows, queries, metric, K)
    assert np.array_equal(cnt, e_cnt) and np.array_equal(ids, e_ids), "ids differ from the oracle's"
    assert np.array_equal(dist.view(np.uint64), e_dist.view(np.uint64)), "distances differ from the oracle's"
    assert set(ids[0].tolist()) == set(pos_g.tolist())
    assert c1["fallback_searches"] == c0["fallback_searches"], "a query took the wide-band fallback"
    if path != SCAN:
        assert c1["batch_launches"] > c0["batch_launches"] and c1["batch_kernel_last"] == path
    # the adversity was real: rows outside the top k out-ranked rows inside it by their approximate keys,
    # and the true neighbours sat inside the band, not at its edge by luck
    assert looks_better > 0 or ratio.max() < 0.02, "the construction did not invert any approximate ranking"
    assert usage <= 1.0, "a true neighbour lies outside tau + 2 D: band usage %.3f" % usage
    RESULTS.append({"path": {SCAN: "f32 scan", 0: "f32 MFMA", 1: "bf16x3", 2: "fp16"}[path],
                    "metric": ["l2", "ip", "cosine"][metric], "dim": d, "max_abs_err_over_bound": float(ratio.max()),
                    "band_usage_of_true_neighbours": usage, "outsiders_ranked_above_a_true_neighbour": looks_better,
                    "candidates_per_query": (c1["candidates_total"] - c0["candidates_total"]) / max(c1["searches"] - c0["searches"], 1)})


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d", [768, 200])
def test_per_row_band_of_the_fp16_keys(hip_lib, oracle_mod, metric, d):
    """Round 5: the fp16 keys of an L2 / inner-product index carry a band PER ROW, alpha_q |v| + beta_q (the operand
    roundings act on the products: 2^-10 |q| |v| a row), and the batched path widens every row's key by its own.
    (1) The claim, row by row, on rows whose operand roundings all err one way and whose norms span a factor 16:
    2 |key - exact| <= alpha2 |v| + beta2.  (2) A corpus whose k-th neighbour sits in a crowd of SHORT rows a few
    fp16 errors apart, with long rows elsewhere: ids and distances are the oracle's, no fallback, and the short rows'
    band is a fraction of the longest row's -- which is what every row carried before."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(77 + 10 * metric + d)
    q = _patterned(rng, d, 2, up=False)
    base = np.stack([_patterned(rng, d, 2, up=bool(i & 1)) for i in range(4096)])
    # 1, 1/2 ... 1/16 (exact, the rounding pattern survives), and a tail down to 2^-26 of the longest row: those rows'
    # elements sit in fp16's subnormal steps, which are absolute -- the shared term of the band has to carry them
    expo = np.where(rng.random(len(base)) < 0.8, rng.integers(0, 5, size=len(base)), rng.integers(5, 27, size=len(base)))
    scale = np.ldexp(1.0, -expo).astype(np.float32)
    rows = base * scale[:, None]
    others = [_patterned(rng, d, 2, up=True) for _ in range(3)]
    exact = _exact_keys(rows, q, metric)
    nrm = np.sqrt(np.einsum("ij,ij->i", rows.astype(np.float64), rows.astype(np.float64)))
    with _open(d, metric, rows, 2) as idx:
        keys, d2 = idx.probe_batch_keys(np.stack([q] + others), K)
        a2, b2 = idx.probe_batch_row_band(4)
    err2 = 2.0 * np.abs(keys[0].astype(np.float64) - exact)
    claim = float(a2[0]) * nrm + float(b2[0])
    ratio = err2 / claim
    assert ratio.max() <= 1.0, "a key is further from its exact value than its row's band: %.3f" % ratio.max()
    assert float(d2[0]) >= claim.max() * 0.999  # the one-number bound of the old probe covers the longest row
    short = nrm < nrm.max() / 8
    assert claim[short].max() < 0.45 * float(d2[0]), "short rows do not carry a materially narrower band"
    # the per-row term is what the errors follow: the worst short row uses its own band about as much as the worst long one
    use_short, use_long = float(ratio[short].max()), float(ratio[~short].max())

    # ---- (2) a crowd of short rows around the k-th key, long rows far away ------------------------------------
    n, k = 40960, K
    g = rng.standard_normal((n, d)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g *= rng.uniform(1.0, 2.0, size=(n, 1)).astype(np.float32)  # long rows, random directions
    qq = rng.standard_normal(d).astype(np.float32)
    qq /= np.linalg.norm(qq)
    t = (qq * np.float32(0.45) if metric == L2 else qq * np.float32(2.5))  # L2: a short row near q; IP: the longest, along q
    crowd = 400
    pert = rng.standard_normal((crowd, d)).astype(np.float32) * np.float32(2e-4 if metric == L2 else 1e-3)
    pos = rng.choice(n, crowd, replace=False)
    g[pos] = t[None, :] * (1.0 + rng.uniform(-3e-4, 3e-4, size=(crowd, 1)).astype(np.float32)) + pert
    queries = np.stack([qq] + [x / np.linalg.norm(x) for x in rng.standard_normal((3, d)).astype(np.float32)])
    with _open(d, metric, g, 2) as idx:
        c0 = idx.counters()
        ids, dist, cnt = idx.search(queries, k)
        c1 = idx.counters()
    e_ids, e_dist, e_cnt = oracle_mod.search_heap_many_mt(g, queries, metric, k)
    assert np.array_equal(cnt, e_cnt) and np.array_equal(ids, e_ids), "ids differ from the oracle's"
    assert np.array_equal(dist.view(np.uint64), e_dist.view(np.uint64)), "distances differ from the oracle's"
    assert set(ids[0].tolist()) <= set(pos.tolist())  # the neighbours of the first query are in the crowd
    assert c1["fallback_searches"] == c0["fallback_searches"] and c1["batch_kernel_last"] == 2
    RESULTS.append({"path": "fp16 per-row band", "metric": ["l2", "ip"][metric], "dim": d,
                    "max_abs_err_over_bound": float(ratio.max()), "short_rows_use_of_their_band": use_short,
                    "long_rows_use_of_their_band": use_long,
                    "short_row_band_over_longest_row_band": float(claim[short].max() / float(d2[0])),
                    "candidates_per_query": (c1["candidates_total"] - c0["candidates_total"]) / max(c1["searches"] - c0["searches"], 1)})


@pytest.mark.parametrize("metric", [L2, IP])
def test_widely_spread_norms_stay_on_fp16_keys(hip_lib, oracle_mod, metric):
    """Round 6: a corpus whose row norms are U(0.1, 3.2) -- a factor 32 between the shortest
    and the longest row -- used to be sent to bf16x3 keys by the automatic choice (2.2 x slower) because every row
    carried the longest row's band.  With a band per row the choice looks at the band's SHARED term only: such a corpus
    stays on the fp16 kernel, ids and distances are the oracle's, nothing falls back, and the candidate lists are no
    longer than bf16x3's (whose length is the corpus's own near-ties: every query's neighbours are the shortest rows)."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(500 + metric)
    n, d, nq, k = 120_000, 768, 96, 100
    g = rng.standard_normal((n, d)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g *= rng.uniform(0.1, 3.2, size=(n, 1)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    e_ids, e_dist, e_cnt = oracle_mod.search_heap_many_mt(g, qs, metric, k)
    per_query = {}
    for kernel in (3, 1):  # the automatic choice, then bf16x3 forced
        with HipVectorIndex(d, metric, capacity_rows=n) as idx:
            idx.append(0, g)
            idx.set_batch_min_nq(2)
            idx.set_batch_kernel(kernel)
            idx.search(qs, k)  # (builds the converted copy of the rows)
            c0 = idx.counters()
            ids, dist, cnt = idx.search(qs, k)
            c1 = idx.counters()
            assert c1["batch_kernel_last"] == (2 if kernel == 3 else 1), "the automatic choice must be fp16 here"
            assert np.array_equal(cnt, e_cnt) and np.array_equal(ids, e_ids), "ids differ from the oracle's"
            assert np.array_equal(dist.view(np.uint64), e_dist.view(np.uint64)), "distances differ from the oracle's"
            assert c1["fallback_searches"] == c0["fallback_searches"] and c1["batch_launches"] > c0["batch_launches"]
            per_query[kernel] = (c1["candidates_total"] - c0["candidates_total"]) / nq
    assert per_query[3] <= 1.25 * per_query[1] + 40, per_query
    RESULTS.append({"path": "fp16 keys, norms U(0.1, 3.2) (automatic choice)", "metric": ["l2", "ip"][metric], "dim": d,
                    "candidates_per_query": per_query[3], "candidates_per_query_bf16x3": per_query[1]})


def test_write_band_report():
    """(runs last in this file) the measured ratios, for profiles/: gpurun_out/band_ratios.json"""
    if not RESULTS:
        pytest.skip("no band case ran in this process")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "band_ratios.json"), "w") as f:
        json.dump(RESULTS, f, indent=1)
    worst = max(r.get("max_abs_err_over_bound", 0.0) for r in RESULTS)
    assert worst <= 1.0
