"""GPU: BASELINE.json's configurations at their FULL sizes (the other GPU tests use corpora the oracle finishes in
seconds for hundreds of queries).  1 M x 768 rows are generated on the device; the oracle (all host cores) answers
a sample of the queries, the rest is held to size-independent properties: every path of the library gives the same
answer for the same query, a stored row finds itself first, masked and unmasked answers nest.
  C2  1M x 768 f32, L2, k = 100, single queries        (HBM-bound scan)
  C3  1M x 768 f32, cosine, k = 100, 1024-query batch   (matrix-core path)
  C5  1M x 768 f32 + device-side row bitmask"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, K = 1_000_000, 768, 100


@pytest.fixture(scope="module")
def corpora(hip_lib):
    import torch

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(20260612)
    x = torch.empty((N, D), dtype=torch.float32, device=dev)
    for s in range(0, N, 131072):
        e = min(N, s + 131072)
        t = torch.randn((e - s, D), generator=g, device=dev)
        t /= t.norm(dim=1, keepdim=True)
        x[s:e] = t
    scale = torch.rand((N, 1), generator=g, device=dev) * 1.5 + 0.5
    torch.cuda.synchronize()
    host_unit = x.cpu().numpy()
    yield {"dev_unit": x, "scale": scale, "host_unit": host_unit}


def _index(metric, dev_rows):
    import torch

    from tostore_amd import HipVectorIndex

    idx = HipVectorIndex(D, metric, capacity_rows=N, shard_device=0, row_base=0)
    torch.cuda.synchronize()
    idx.append_device(0, N, dev_rows.data_ptr())
    torch.cuda.synchronize()
    return idx


def _queries(oracle, nq, metric, seed):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, D)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    if metric == 2:
        q = np.stack([oracle.normalize_f32(v) for v in q])
    return np.ascontiguousarray(q)


def _same(a, b):
    return (np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0])
            and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)))


def test_c2_single_queries_full_size(corpora, oracle_mod):
    import torch

    rows_dev = corpora["dev_unit"] * corpora["scale"]  # L2 / IP corpora: norms in [0.5, 2)
    torch.cuda.synchronize()
    host = rows_dev.cpu().numpy()
    qs = _queries(oracle_mod, 80, 0, 1)
    with _index(0, rows_dev) as idx:
        del rows_dev
        idx.set_batch_min_nq(0)  # every query scans on its own
        got = idx.search(qs, K)
        ref = oracle_mod.search_heap_many_mt(host, qs[:64], 0, K)  # (the pass / fail record carries the parity evidence)
        assert _same(tuple(x[:64] for x in got), ref), "pipelined single-query scans vs oracle (64 queries)"
        one = idx.search(qs[65], K)  # a lone query (no pipelining) gives the same answer as inside a group
        assert np.array_equal(one[0][0], got[0][65]) and np.array_equal(one[1][0], got[1][65])
        pend = [idx.submit(qs[i], K) for i in range(66, 70)]  # asynchronous form
        for i, t in zip(range(66, 70), pend):
            ids, dist = idx.wait(t)
            assert np.array_equal(ids, got[0][i]) and np.array_equal(dist, got[1][i])
        # a stored row finds itself first, at distance 0
        probe = host[[0, 499_999, N - 1]]
        ids, dist, cnt = idx.search(probe, 5)
        assert ids[:, 0].tolist() == [0, 499_999, N - 1] and not dist[:, 0].any()
        # the batched path answers the same queries identically
        idx.set_batch_min_nq(2)
        assert _same(idx.search(qs, K), got), "matrix-core batch vs single-query scans (L2)"
        # ... and as part of a 1024-query call (256-query tiles), on the fp16 plane grouped by norm (the default for L2 /
        # inner product) and on the plane in row order: the same bits for all 1024
        qs_big = np.concatenate([qs, _queries(oracle_mod, 944, 0, 9)])
        big = idx.search(qs_big, K)
        assert idx.counters()["batch_kernel_last"] == 2
        assert _same(tuple(x[:80] for x in big), got), "the first 80 of a 1024-query L2 batch vs single-query scans"
        idx.set_batch_group(False)
        assert _same(idx.search(qs_big, K), big), "fp16 plane in row order vs grouped by norm (1024 queries, L2)"
        idx.set_batch_group(True)
        c = idx.counters()
        assert c["fallback_searches"] == 0 and c["batch_launches"] > 0 and c["safe_mode"] == 0


def test_c3_batch_of_1024_full_size(corpora, oracle_mod):
    host = corpora["host_unit"]
    qs = _queries(oracle_mod, 1024, 2, 2)
    with _index(2, corpora["dev_unit"]) as idx:
        got = idx.search(qs, K)  # one call: matrix-core path (auto = fp16 keys for cosine)
        assert idx.counters()["batch_kernel_last"] == 2 and idx.counters()["batch_launches"] == 1
        sample = np.r_[0:96, 464:560, 960:1024]
        ref = oracle_mod.search_heap_many_mt(host, qs[sample], 2, K)
        assert _same(tuple(x[sample] for x in got), ref), "1024-query batch vs oracle (256 sampled queries)"
        assert (got[2] == K).all()
        # every key kernel, and the single-query pipeline, give the batch's answer bit for bit
        for kern in (1, 0):
            idx.set_batch_kernel(kern)
            part = idx.search(qs[256:512], K)
            assert _same(part, tuple(x[256:512] for x in got)), "key kernel %d" % kern
        idx.set_batch_kernel(3)
        idx.set_batch_min_nq(0)
        assert _same(idx.search(qs[600:632], K), tuple(x[600:632] for x in got)), "single-query scans vs batch (cosine)"
        assert idx.counters()["fallback_searches"] == 0


@pytest.mark.parametrize("keep,kind", [(0.01, "bernoulli"), (0.03, "bernoulli"), (0.5, "bernoulli"), (1.0, "bernoulli"),
                                       (0.1, "range")])
def test_c5_row_mask_full_size(corpora, oracle_mod, keep, kind):
    import torch

    rows_dev = corpora["dev_unit"] * corpora["scale"]
    torch.cuda.synchronize()
    host = rows_dev.cpu().numpy()
    rng = np.random.default_rng(int(keep * 100))
    if kind == "range":  # WHERE id BETWEEN ...: one contiguous run of node ids
        bits = np.zeros(N, bool)
        start = int(rng.integers(0, N - int(N * keep)))
        bits[start:start + int(N * keep)] = True
    else:
        bits = rng.random(N) < keep
    mask = np.packbits(bits, bitorder="little")
    qs = _queries(oracle_mod, 40, 0, 3)
    with _index(0, rows_dev) as idx:
        del rows_dev
        idx.set_batch_min_nq(0)
        c0 = idx.counters()
        got = idx.search(qs, K, None, mask)  # masked scans: bytes read scale with the mask
        c1 = idx.counters()
        # selective masks (below one kept row in 24) are scanned as a compacted list of row ids, the others by tiles
        assert (c1["list_scans"] - c0["list_scans"] == 40) == (keep < 1 / 24), "which scan kernel ran"
        # ... a list of at most 16 384 rows (keep 1 %: 10 k) is answered from its exact sums, a longer one (keep 3 %:
        # 30 k rows, still fewer than one in 24) by scan_list_kernel's f32 keys + select + re-rank
        assert c1["exact_scans"] - c0["exact_scans"] == (40 if keep * N <= 16_384 else 0), "exact path / f32 keys of the list"
        if keep == 0.03:
            assert c1["list_scans"] - c0["list_scans"] == 40 and c1["exact_scans"] == c0["exact_scans"]
        ref = oracle_mod.search_heap_many_mt(host, qs[:32], 0, K, None, mask)
        assert _same(tuple(x[:32] for x in got), ref), "masked single-query scans vs oracle (32 queries)"
        assert bits[got[0]].all(), "a masked-out row came back"
        # the same mask as a HANDLE (uploaded once, listed on the device, resident): same scans, same kernels, same bits
        with idx.make_mask(mask) as mh:
            assert mh.kept == int(bits.sum())
            c2 = idx.counters()
            assert _same(idx.search(qs, K, None, mh), got), "mask handle vs mask pointer (single-query scans)"
            c3 = idx.counters()
            for key in ("list_scans", "exact_scans", "scan_launches"):
                assert c3[key] - c2[key] == c1[key] - c0[key], key
            t = idx.submit(qs[5], K, mh)
            ids, dist = idx.wait(t)
            assert np.array_equal(ids, got[0][5]) and np.array_equal(dist, got[1][5])
            idx.set_batch_min_nq(2)
            assert _same(idx.search(qs, K, None, mh), got), "masked batch with a handle vs masked scans"
        idx.set_batch_min_nq(2)  # the same calls through the matrix-core path (mask applied in its epilogue)
        assert _same(idx.search(qs, K, None, mask), got), "masked batch vs masked scans"
        # nesting: the unmasked top-k restricted to kept rows is a prefix-subsequence of the masked answer
        full = idx.search(qs[:4], K)
        for i in range(4):
            kept_of_full = [r for r in full[0][i].tolist() if bits[r]]
            assert kept_of_full == got[0][i][:len(kept_of_full)].tolist()
        assert idx.counters()["fallback_searches"] == 0


C4_D, C4_PER, C4_K, C4_NQ = 1536, 1_250_000, 100, 16


def c4_shard_rows(r, per, d, dev):
    """Shard r of the tests' C4 corpus, on the device: unit directions scaled by U(0.5, 2), seeded by r alone (the
    same function lives in tests/_c4_worker.py, whose processes regenerate their own shard)."""
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(20260614 + 1000 * r)
    x = torch.empty((per, d), dtype=torch.float32, device=dev)
    for s in range(0, per, 65536):
        e = min(per, s + 65536)
        t = torch.randn((e - s, d), generator=g, device=dev)
        t /= t.norm(dim=1, keepdim=True)
        t *= torch.rand((e - s, 1), generator=g, device=dev) * 1.5 + 0.5
        x[s:e] = t
    torch.cuda.synchronize()
    return x


def _order_keys(d):
    """double.compareTo as integers (NaN greatest, -0 < +0): merging the oracle's answers over row chunks."""
    b = np.ascontiguousarray(d, np.float64).view(np.int64)
    key = np.where(b < 0, ~b, b | np.int64(-2 ** 63)).view(np.uint64)
    return np.where(np.isnan(d), np.uint64(2 ** 64 - 1), key)


@pytest.fixture(scope="module")
def c4(hip_lib, oracle_mod):
    """TWO of C4's eight row-range shards at their real size (1.25 M x 1536 f32 = 7.7 GB each, inner product): host
    copies for the oracle, 16 queries, and the oracle's answers -- per shard (what a rank alone must find among its own
    rows) and over the 2.5 M rows (what the merged answer must be)."""
    import torch

    dev = torch.device("cuda", 0)
    hosts = []
    for r in range(2):
        x = c4_shard_rows(r, C4_PER, C4_D, dev)
        hosts.append(x.cpu().numpy())
        del x
        torch.cuda.empty_cache()
    rng = np.random.default_rng(3)
    qs = rng.standard_normal((C4_NQ, C4_D)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True).astype(np.float32)
    qs[0] = hosts[1][C4_PER - 1]  # the last row of the second shard: finds itself first (largest inner product
    qs[0] *= np.float32(4.0)      # with itself among unit-direction rows of norm < 2 needs a long query: scaled)
    per_shard = []
    for r in range(2):
        ids, dist, cnt = oracle_mod.search_heap_many_mt(hosts[r], qs, 1, C4_K)
        assert (cnt == C4_K).all()
        per_shard.append((ids + r * C4_PER, dist, cnt))
    ids = np.concatenate([p[0] for p in per_shard], axis=1)
    dist = np.concatenate([p[1] for p in per_shard], axis=1)
    m_ids, m_dist = np.empty((C4_NQ, C4_K), np.int64), np.empty((C4_NQ, C4_K), np.float64)
    for q in range(C4_NQ):
        order = np.lexsort((ids[q], _order_keys(dist[q])))[:C4_K]
        m_ids[q], m_dist[q] = ids[q][order], dist[q][order]
    yield {"hosts": hosts, "queries": qs, "per_shard": per_shard,
           "merged": (m_ids, m_dist, np.full(C4_NQ, C4_K, np.int32))}


def _c4_shard(r, host_rows):
    from tostore_amd import HipVectorIndex

    idx = HipVectorIndex(C4_D, 1, capacity_rows=C4_PER, shard_device=0, row_base=r * C4_PER)
    for s in range(0, C4_PER, 250_000):  # (host to device in pieces: no 7.7 GB staging copy)
        idx.append(r * C4_PER + s, host_rows[s:s + 250_000])
    return idx


def test_c4_one_rank_at_size_over_real_rccl(c4, hip_lib):
    """One rank of C4's eight, as it runs in the deployment: tsh_index_create_shard (global ids from row_base =
    1.25 M), tsh_comm_create + tsh_search_sharded over REAL RCCL in a world of one -- every query its own 7.7 GB scan,
    the library's own groups, then the same call on the shard's matrix cores -- all 16 answers against the oracle over
    that shard's rows."""
    from tostore_amd.sharded import CommSearcher

    qs, ref = c4["queries"], c4["per_shard"][1]
    with _c4_shard(1, c4["hosts"][1]) as idx, CommSearcher(idx, 1, 0, CommSearcher.unique_id(), 0) as cs:
        idx.set_batch_min_nq(0)
        scans = cs.search(qs, C4_K)
        assert _same(scans, ref), "16 single-query scans through tsh_search_sharded (W = 1, real RCCL) vs oracle"
        assert scans[0][0][0] == 2 * C4_PER - 1  # the stored row finds itself, under its GLOBAL id
        t = cs.timeline()
        assert t["transport"] == "rccl" and t["calls"] == 1 and t["groups"] >= 2 and t["retries"] == 0
        assert _same(cs.search(qs[3], C4_K), tuple(x[3:4] for x in ref)), "a lone query"
        idx.set_batch_min_nq(2)
        g0 = cs.timeline()["groups"]
        assert _same(cs.search(qs, C4_K), ref), "the same call on the matrix cores vs oracle"
        assert cs.timeline()["groups"] - g0 == 1  # (ranks that batch: one group, one batched call on the shard)
        # ... and as the head of a 1040-query call: two groups of 512 and one of 16, a batched call on the shard each
        big = np.concatenate([qs, np.random.default_rng(5).standard_normal((1024, C4_D)).astype(np.float32)])
        g0 = cs.timeline()["groups"]
        got = cs.search(big, C4_K)
        assert cs.timeline()["groups"] - g0 == 3
        assert _same(tuple(x[:len(qs)] for x in got), ref), "the first 16 of a 1040-query sharded call vs oracle"
        assert (got[2] == C4_K).all()
        c = idx.counters()
        assert c["batch_launches"] >= 4 and c["fallback_searches"] == 0 and c["safe_mode"] == 0


def test_c4_two_ranks_at_size_over_the_rccl_branch(c4, hip_lib, tmp_path):
    """Both real-size shards, one process each, through tsh_comm_create + tsh_search_sharded over the library's RCCL
    branch with W = 2 (tests/fake_rccl: the two ranks share this box's one GPU): the all-gather of the candidate blocks,
    every rank's merge of its slice, the result all-gather -- all 16 answers, single-query and batched, on BOTH ranks
    against the oracle over the 2.5 M rows."""
    import os

    from test_gpu_comm_rccl_multirank import ROOT, run_ranks

    ids, dist, cnt = c4["merged"]
    refs = str(tmp_path / "c4_refs.npz")
    np.savez(refs, queries=c4["queries"], ids=ids, dist=dist, per=C4_PER, dim=C4_D, k=C4_K)
    rcs, outs = run_ranks(2, [os.path.join(ROOT, "tests", "_c4_worker.py"), refs, "@TMP@/uid"], timeout=900)
    out = "\n".join(outs)
    assert all(rc == 0 for rc in rcs), out[-6000:]
    assert "MISMATCH" not in out and out.count(" ok\n") == 2 * 5, out[-6000:]


def test_c4_two_of_eight_shards_at_size(c4, hip_lib, oracle_mod):
    """C4 (10 M x 1536 f32, inner product, k = 100, rows split over 8 GPUs): TWO of the eight row-range shards at
    their real size (1.25 M x 1536 each, 7.7 GB), both on the one GPU of the test box and in ONE process, through
    the block-level entry points -- tsh_index_create_shard with global row ids, tsh_search_shard into device candidate
    blocks, tsh_merge_candidates over the concatenated blocks (what the all-gather delivers).  All 16 queries against
    the oracle over the 2.5 M rows; pipelined scans == matrix-core batch, bit for bit."""
    import ctypes

    import torch

    from tostore_amd import _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    L = _ffi.lib()
    d, per, k, ip = C4_D, C4_PER, C4_K, 1
    qs, nq = c4["queries"], C4_NQ
    shards = []
    try:
        for r in range(2):
            shards.append(_c4_shard(r, c4["hosts"][r]))
        entries = L.tsh_default_block_entries(k)
        bb = L.tsh_candidate_block_bytes(entries)

        def answer(batch_min_nq):
            bufs = []
            for idx in shards:
                idx.set_batch_min_nq(batch_min_nq)
                buf = torch.empty(nq * bb, dtype=torch.uint8, device="cuda")
                _ffi.check(L.tsh_search_shard(idx._h, qs.ctypes.data_as(_ffi.p_f32), nq, k, None, entries,
                                              ctypes.c_void_p(buf.data_ptr()), None))
                bufs.append(buf)
            return merge_candidate_blocks(ip, d, qs, k, None, torch.cat(bufs).cpu().numpy(), 2, entries)

        scans = answer(0)    # every query its own HBM scan of each shard
        batch = answer(2)    # all sixteen in one matrix-core pass per shard
        assert _same(scans, batch), "pipelined scans vs batched path over two shards"
        assert shards[0].counters()["batch_launches"] >= 1
        assert _same(scans, c4["merged"]), "two merged shards vs oracle over 2.5 M x 1536 (16 queries)"
        assert scans[0][0][0] == 2 * per - 1  # global id of the stored row, from the shard with row_base = 1.25 M
    finally:
        for idx in shards:
            idx.close()
