"""bench_check.py -- every use bench.py makes of the CPU oracle lives here: the recall / bit-exactness checks of the
GPU answers and the `cpu_baseline` legs.  The oracle (oracle/) is test infrastructure, never the thing measured:
bench.py's timed code cannot reach it -- while a timed region is open (bench.timed_region sets TSH_BENCH_TIMED=1)
importing this module raises, and so does every function in it.  Nothing under tostore_amd/ imports either.

Layout of an answer everywhere below: (ids[nq, k] int64, dist[nq, k] float64, cnt[nq] int32), the C ABI's."""
import os
import time

import numpy as np

TIMED_ENV = "TSH_BENCH_TIMED"


def _untimed():
    if os.environ.get(TIMED_ENV) == "1":
        raise RuntimeError("the CPU oracle was reached from inside a timed region of bench.py")


_untimed()  # importing it there is already wrong
import oracle  # noqa: E402  (the one place bench code touches oracle/)


def mt_threads():
    _untimed()
    return oracle.mt_max_threads()


def compare(got, ref):
    """-> (recall@k over the reference's entries, ids and distances bit-identical and in the same order)."""
    g_ids, g_dist, g_cnt = got
    r_ids, r_dist, r_cnt = ref
    nq = len(r_cnt)
    hits = sum(len(set(g_ids[i, :g_cnt[i]].tolist()) & set(r_ids[i, :r_cnt[i]].tolist())) for i in range(nq))
    same = all(g_cnt[i] == r_cnt[i] and np.array_equal(g_ids[i, :g_cnt[i]], r_ids[i, :r_cnt[i]])
               and np.array_equal(g_dist[i, :g_cnt[i]], r_dist[i, :r_cnt[i]]) for i in range(nq))
    return hits / max(int(np.sum(r_cnt[:nq])), 1), bool(same)


def check_answers(host_rows, queries, metric, k, got, row_mask=None):
    """The GPU's answers `got` for `queries` against the exhaustive oracle (OpenMP over query groups; the same
    per-(query, row) arithmetic as the single-thread form).  -> dict for the JSON line."""
    _untimed()
    m = len(queries)
    t0 = time.perf_counter()
    ref = oracle.search_heap_many_mt(host_rows, queries, metric, k, None, row_mask)
    rec, same = compare(tuple(x[:m] for x in got), ref)
    return {"recall_at_k": rec, "ids_and_distances_bit_exact": same, "checked_queries": m,
            "oracle_seconds": time.perf_counter() - t0}


def order_keys(d):
    """double.compareTo as integers (NaN greatest, -0 < +0), for merging oracle answers of row chunks."""
    b = np.ascontiguousarray(d, np.float64).view(np.int64)
    key = np.where(b < 0, ~b, b | np.int64(-2 ** 63)).view(np.uint64)
    return np.where(np.isnan(d), np.uint64(2 ** 64 - 1), key)


def oracle_topk_stream(chunks, queries, metric, k, row_mask=None):
    """The exhaustive CPU oracle over a corpus that arrives as (first row id, rows) chunks (a sharded corpus is
    never whole in one place): the oracle's own top k of every chunk, merged by (compareTo order, row id).
    -> (ids[nq,k], dist[nq,k], cnt[nq])."""
    _untimed()
    nq = len(queries)
    acc = [([], []) for _ in range(nq)]
    bits = None if row_mask is None else np.unpackbits(np.asarray(row_mask, np.uint8), bitorder="little")
    for r0, rows in chunks:
        keep = None if bits is None else np.packbits(bits[r0:r0 + len(rows)], bitorder="little")
        ids, dist, cnt = oracle.search_heap_many_mt(rows, queries, metric, k, None, keep)
        for q in range(nq):
            acc[q][0].append(ids[q, :cnt[q]] + r0)
            acc[q][1].append(dist[q, :cnt[q]])
    out_ids = np.full((nq, k), -1, np.int64)
    out_dist = np.full((nq, k), np.nan, np.float64)
    out_cnt = np.zeros(nq, np.int32)
    for q in range(nq):
        i, d = np.concatenate(acc[q][0]), np.concatenate(acc[q][1])
        order = np.lexsort((i, order_keys(d)))[:k]
        out_cnt[q] = len(order)
        out_ids[q, :len(order)], out_dist[q, :len(order)] = i[order], d[order]
    return out_ids, out_dist, out_cnt


def cpu_sample_size(per_query_s, budget_s, pool):
    """Queries of the single-thread cpu_baseline leg: what fits the budget, 2..32, never more than the pool."""
    return int(max(1, min(pool, max(2, min(32, budget_s / max(per_query_s, 1e-3))))))


def cpu_baseline_single(host_rows, queries, metric, k, row_mask, budget_s):
    """The reference's shape of the work -- one isolate, one query at a time, every row -- restated in C
    (oracle/vs_oracle.c), timed on a bounded sample of the bench's own queries.
    -> (refs [(ids, dist)] of the sampled queries, seconds)."""
    _untimed()
    nqp = len(queries)
    t1 = time.perf_counter()
    oracle.search_heap(host_rows, queries[0], metric, k, None, row_mask)
    n_cpu = cpu_sample_size(time.perf_counter() - t1, budget_s, nqp)
    t1 = time.perf_counter()
    refs = [oracle.search_heap(host_rows, queries[i % nqp], metric, k, None, row_mask) for i in range(n_cpu)]
    return refs, time.perf_counter() - t1


def cpu_baseline_sample_of_rows(chunks, n_total, queries, metric, k, row_mask, budget_s):
    """A corpus too big for a host copy: the single-thread oracle over the first chunks of it, scaled by the row
    ratio (the scan is linear in the rows).  -> (queries/s over the WHOLE corpus, queries timed, rows sampled)."""
    _untimed()
    sample = np.concatenate([rows for _, rows in (c for _, c in zip(range(8), chunks))])
    smask = None if row_mask is None else row_mask[:(len(sample) + 7) // 8]
    nqp = len(queries)
    t1 = time.perf_counter()
    m = 0
    while m < 2 or (time.perf_counter() - t1 < budget_s and m < 32):
        oracle.search_heap(sample, queries[m % nqp], metric, k, None, smask)
        m += 1
    return m / (time.perf_counter() - t1) * len(sample) / n_total, m, len(sample)


def cpu_baseline_mt(host_rows, queries, metric, k, row_mask, m):
    _untimed()
    t1 = time.perf_counter()
    for i in range(m):
        oracle.search_heap_mt(host_rows, queries[i % len(queries)], metric, k, None, row_mask)
    return {"value": m / (time.perf_counter() - t1), "unit": "queries/s", "cores": oracle.mt_max_threads(),
            "kind": "port", "sample": "%d queries, OpenMP" % m}


def recall_leg(host_rows, queries, metric, k, row_mask, search, want, seconds):
    """recall@k over `want` queries (>= 1000): 100-query chunks -- oracle, then the GPU's
    answers of the same queries through `search(queries) -> answer` -- until the oracle's time budget is used.
    -> dict(recall_at_k, recall_queries, bit_exact, cpu_baseline_mt_batched)."""
    _untimed()
    nr, t_or, hits, tot, same = 0, 0.0, 0.0, 0, True
    while nr < want and t_or < seconds:
        m = min(100, want - nr)
        t1 = time.perf_counter()
        rr = oracle.search_heap_many_mt(host_rows, queries[nr:nr + m], metric, k, None, row_mask)
        t_or += time.perf_counter() - t1
        rec, ok = compare(search(queries[nr:nr + m]), rr)
        hits += rec * int(np.sum(rr[2]))
        tot += int(np.sum(rr[2]))
        same &= ok
        nr += m
    return {"recall_at_k": hits / max(tot, 1), "recall_queries": nr, "bit_exact": bool(same),
            "cpu_baseline_mt_batched": {"value": nr / max(t_or, 1e-9), "unit": "queries/s", "cores": oracle.mt_max_threads(),
                                        "kind": "port", "sample": "%d queries, OpenMP over query groups" % nr}}


def compare_refs(got, refs):
    """got: list of single-query answers; refs: [(ids, dist)] from cpu_baseline_single / oracle_topk_stream rows.
    -> (hits, total, bit-exact)."""
    hits, tot, exact = 0, 0, True
    for (ids, dd, cnt), (r_ids, r_dist) in zip(got, refs):
        g = ids[0, :cnt[0]]
        hits += len(set(g.tolist()) & set(r_ids.tolist()))
        tot += len(r_ids)
        exact &= bool(np.array_equal(g, r_ids) and np.array_equal(dd[0, :cnt[0]], r_dist))
    return hits, tot, exact


def c1_cpu_context(host_rows, qs, metric, k, ref_ids):
    """C1's CPU side: the single-thread baseline on 20 queries, and -- labelled context, SURVEY section 8d / N3 --
    what the reference's OWN search (the approximate NGH graph walk this build replaces) returns on the same rows and
    queries.  A restatement (oracle/ngh_ann.c), not the reference: its PQ training draws come from NumPy instead of
    Dart's Random(42), nothing here was produced by a Dart VM, so the numbers are properties of the restatement and
    must not be quoted as ToStore's."""
    _untimed()
    out = {}
    t1 = time.perf_counter()
    for i in range(20):
        oracle.search_heap(host_rows, qs[i], metric, k)
    out["cpu_baseline"] = {"value": 20 / (time.perf_counter() - t1), "unit": "queries/s", "cores": 1,
                           "kind": "port", "sample": "20 queries, oracle/vs_oracle.c single thread"}
    try:
        n, d = host_rows.shape
        t1 = time.perf_counter()
        ann = oracle.NghAnnIndex(d, metric, host_rows[:2500])
        for b0 in range(2500, n, 2500):
            ann.insert_batch(host_rows[b0:b0 + 2500])
        t_build = time.perf_counter() - t1
        nqa = 200
        ann.counters()
        t1 = time.perf_counter()
        found = [ann.search(qs[i], k)[0] for i in range(nqa)]
        t_search = time.perf_counter() - t1
        ctr = ann.counters()
        hit = sum(len(np.intersect1d(found[i], ref_ids[i][:k])) for i in range(nqa))
        out["reference_ann_restated"] = {
            "label": "restatement of the reference's NGH graph search (PQ/ADC beam search + exact re-rank), CPU, "
                     "1 thread, in memory; unverifiable here: Dart's PRNG differs, no Dart VM in the image",
            "recall_at_k": hit / float(nqa * k), "ms_per_query": 1e3 * t_search / nqa, "queries": nqa,
            "build_seconds": t_build, "adc_evaluations_per_query": ctr["adc_evaluations"] / nqa,
            "hops_per_query": ctr["hops"] / nqa,
            "defaults": "M=%d K=%d R=64 efSearch=64 efConstruction=128 alpha=1.2" % (ann.subspaces, ann.centroids)}
        ann.close()
    except Exception as e:  # context only: never fails the line
        out["reference_ann_restated"] = {"error": repr(e)}
    return out


def oracle_answers(host_rows, queries, metric, k, row_mask=None):
    _untimed()
    return oracle.search_heap_many_mt(host_rows, queries, metric, k, None, row_mask)
