#!/usr/bin/env python
"""Context measurement: what the reference's OWN search path returns.

Builds the restated NGH index (oracle/ngh_ann.c: PQ training, incremental Vamana-style insert,
beam search over PQ codes, exact re-rank of max(2k, 20) candidates) on the same synthetic data as
bench.py, and reports its recall@k against the exhaustive-exact oracle and its CPU latency; with a
GPU present, the exhaustive HIP path's latency on the same rows and queries is printed beside it.
Statistical, not bit-level: Dart's PRNG seeds the reference's PQ training (see ngh_ann.c header).

  python tests/probes/reference_ann_probe.py [--rows 10000 --dim 128 --metric l2 --k 10 --queries 500]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--metric", choices=["l2", "ip", "cosine"], default="l2")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=500)
    ap.add_argument("--batch", type=int, default=5000, help="rows per writeChanges batch")
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    metric = {"l2": 0, "ip": 1, "cosine": 2}[a.metric]
    rng = np.random.Generator(np.random.Philox(20260612))
    x = rng.standard_normal((a.rows, a.dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)  # the demo's distribution (example/lib/tostore_example.dart:728-747)
    qs = np.random.Generator(np.random.Philox(20260613)).standard_normal((a.queries, a.dim)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    t = time.perf_counter()
    ann = oracle.NghAnnIndex(a.dim, metric, x[:a.batch])
    for s in range(a.batch, a.rows, a.batch):
        ann.insert_batch(x[s:s + a.batch])
    t_build = time.perf_counter() - t
    print(f"config: {a.rows} x {a.dim} f32, {a.metric}, k={a.k}, {a.queries} queries; index: M={ann.subspaces} "
          f"K={ann.centroids} R=64 efSearch=64 efConstruction=128 alpha=1.2 (reference defaults)")
    print(f"restated reference build: {t_build:.1f} s single thread ({a.rows / t_build:.0f} inserts/s), "
          f"mean out-degree {ann.mean_degree:.1f}")
    ann.counters()
    t = time.perf_counter()
    got = [ann.search(q, a.k) for q in qs]
    t_ann = (time.perf_counter() - t) / a.queries
    c = ann.counters()
    t = time.perf_counter()
    ref = [oracle.search_heap(x, q, metric, a.k) for q in qs]
    t_exact = (time.perf_counter() - t) / a.queries
    hits = sum(len(set(g[0].tolist()) & set(r[0].tolist())) for g, r in zip(got, ref))
    print(f"restated reference ANN search: recall@{a.k} = {hits / (a.queries * a.k):.3f} vs exhaustive-exact; "
          f"{t_ann * 1e3:.3f} ms/query (C restatement, 1 thread, in memory - no page I/O); "
          f"{c['adc_evaluations'] / a.queries:.0f} ADC evaluations, {c['hops'] / a.queries:.0f} hops per query")
    print(f"exhaustive-exact oracle on the CPU (1 thread): {t_exact * 1e3:.3f} ms/query, recall 1.0 by definition")
    if not a.no_gpu:
        try:
            import torch
            if torch.cuda.is_available():
                from tostore_amd import HipVectorIndex
                with HipVectorIndex(a.dim, metric) as idx:
                    idx.append(0, x)
                    idx.set_batch_min_nq(0)
                    idx.search(qs[0], a.k)
                    lat = []
                    same = True
                    for q, r in zip(qs, ref):
                        t = time.perf_counter()
                        ids, dist, cnt = idx.search(q, a.k)
                        lat.append(time.perf_counter() - t)
                        same &= bool(np.array_equal(ids[0, :cnt[0]], r[0]) and np.array_equal(dist[0, :cnt[0]], r[1]))
                    lat = np.sort(np.array(lat))
                    t = time.perf_counter()
                    idx.search(qs, a.k)
                    tp = (time.perf_counter() - t) / a.queries
                    print(f"exhaustive HIP path on the GPU: recall@{a.k} = 1.0 (ids+distances bit-exact vs oracle: {same}); "
                          f"one at a time p50 {lat[len(lat) // 2] * 1e6:.0f} us; {1 / tp:.0f} queries/s pipelined")
        except ImportError:
            pass


if __name__ == "__main__":
    main()
