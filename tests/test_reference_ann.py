"""CPU: the restated reference ANN (oracle/ngh_ann.c).  It has no golden vectors to
meet (Dart's PRNG seeds the reference's PQ training), so the tests pin what the algorithm guarantees:
exact re-rank distances, ordering, bounds on the result count, determinism, and that it is an ANN -- some,
not all, of the true neighbours."""
import numpy as np
import pytest

L2, IP, COS = 0, 1, 2


def _data(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_restated_reference_ann_invariants(oracle_mod, metric):
    n, d, k = 3000, 32, 10
    x, qs = _data(n, d, 1), _data(40, d, 2)
    ann = oracle_mod.NghAnnIndex(d, metric, x[:1500])
    ann.insert_batch(x[1500:])  # a second writeChanges batch
    again = oracle_mod.NghAnnIndex(d, metric, x[:1500])
    again.insert_batch(x[1500:])
    assert ann.size == n and 1.0 < ann.mean_degree <= 64.0
    hits = 0
    for q in qs:
        ids, dist = ann.search(q, k)
        ids2, dist2 = again.search(q, k)
        assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)  # deterministic given the seeds
        assert 0 < len(ids) <= k and len(set(ids.tolist())) == len(ids)
        # phase 3 of NghGraphEngine.search: exact distances, ascending (ngh_graph_engine.dart:122-134)
        assert [oracle_mod.exact_distance(q, x[i], metric) for i in ids] == dist.tolist()
        assert all(oracle_mod.compare_double(dist[i], dist[i + 1]) <= 0 for i in range(len(dist) - 1))
        eids, _ = oracle_mod.search_heap(x, q, metric, k)
        hits += len(set(ids.tolist()) & set(eids.tolist()))
    recall = hits / (len(qs) * k)
    assert 0.05 < recall < 1.0, recall  # approximate: finds neighbours, misses some
    ann.close()
    again.close()


def test_restated_reference_ann_result_count_rules(oracle_mod):
    """ef = min(efSearch, max(5k, 32)) bounds the beam, max(2k, 20) the re-rank pool (ngh_graph_engine.dart:83,115);
    a threshold drops results strictly above it."""
    n, d = 2500, 16
    x = _data(n, d, 5)
    ann = oracle_mod.NghAnnIndex(d, L2, x)
    q = _data(1, d, 6)[0]
    ids, dist = ann.search(q, 100)            # default efSearch 64 < k: at most 64 come back
    assert len(ids) <= 64
    ids, dist = ann.search(q, 100, ef_search=400)
    assert 64 < len(ids) <= 100
    ids3, dist3 = ann.search(q, 3)            # pool of 20 re-ranked, best 3 returned
    assert len(ids3) == 3
    thr = float(dist[len(dist) // 2])
    idt, dt = ann.search(q, 100, ef_search=400, threshold=thr)
    assert len(idt) > 0 and dt.max() <= thr and np.array_equal(idt, ids[:len(idt)])
    ann.close()


def test_restated_reference_ann_small_first_batch(oracle_mod):
    """A first batch of fewer than 100 vectors trains with VectorQuantizer.train (k-means++ seeding, k = n centroids:
    ref core/vector_index_manager.dart:842-849, core/vector_quantizer.dart:81-350); later batches are encoded with that
    codebook.  With k = n every training vector is its own centroid in every sub-space it is distinct in, so the ADC
    distance of a first-batch vector to itself is 0 and it comes back first."""
    d = 16
    x = _data(900, d, 11)
    ann = oracle_mod.NghAnnIndex(d, L2, x[:60])
    assert ann.centroids == 60 and ann.codebook.shape == (ann.subspaces, 60, d // ann.subspaces)
    ann.insert_batch(x[60:])
    assert ann.size == 900
    hits = 0
    for i in range(0, 60, 7):
        ids, dist = ann.search(x[i], 5, ef_search=200)
        hits += int(len(ids) > 0 and ids[0] == i and dist[0] == 0.0)
    assert hits >= 8  # (9 probes; the graph walk is approximate)
    with pytest.raises(ValueError):
        oracle_mod.NghAnnIndex(d, L2, x[:0])
    ann.close()
