"""GPU: cold start from an on-disk NGH index directory -- meta.json, raw-vector
partitions and graph tombstones laid out as /root/reference/lib/src/core/path_manager.dart:275-324 and
core/ngh_partition_manager.dart:409-531 write them (writer restated in oracle/ngh_dir.py)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _keep_bits(dead):
    return np.packbits(~dead, bitorder="little")


def _check_search(oracle_mod, idx, vec, dead, metric, k=40, seed=0):
    rng = np.random.default_rng(seed)
    for _ in range(3):
        q = rng.standard_normal(vec.shape[1]).astype(np.float32)
        if metric == 2:
            q = oracle_mod.normalize_f32(q)
        ids, dist, cnt = idx.search(q, k)
        eids, edist = oracle_mod.search_exhaustive(vec, q, metric, k, keep=_keep_bits(dead))
        assert cnt[0] == len(eids)
        assert np.array_equal(ids[0][:cnt[0]], eids) and np.array_equal(dist[0][:cnt[0]], edist)


@pytest.mark.parametrize("metric,precision,dims,n", [(0, 1, 96, 3000), (2, 1, 768, 400), (1, 0, 64, 1500),
                                                     (2, 2, 128, 2500), (0, 1, 100, 777)])
def test_open_ngh_directory(hip_lib, oracle_mod, tmp_path, metric, precision, dims, n):
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(n)
    v = (rng.standard_normal((n, dims)) * 0.4).astype(np.float32)
    deleted = sorted(set(rng.integers(0, n, n // 20).tolist()) | {0, n - 1})
    root = tmp_path / "ngh"
    meta = ngh_dir.write_ngh_dir(str(root), v, metric=metric, precision=precision, max_partition_file_size=16384 * 8,
                                 deleted=deleted, max_entries_per_dir=3)
    _, vec, dead = ngh_dir.read_ngh_dir(str(root), 3)  # what the reference's reader yields per node id
    assert dead.sum() == len(deleted)
    idx, info = HipVectorIndex.open_ngh(str(root), max_entries_per_dir=3)
    with idx:
        assert (info["dimensions"], info["metric"], info["precision"]) == (dims, metric, precision)
        assert info["next_node_id"] == n and info["rows_loaded"] == n and idx.size == n
        assert info["tombstones"] == len(deleted) == info["deleted_count"]
        assert info["total_vectors"] == meta["totalVectors"] and info["files_read"] >= 2
        assert idx.counters()["deleted_rows"] == len(deleted)
        _check_search(oracle_mod, idx, vec, dead, metric)


def test_open_ngh_missing_and_short_files(hip_lib, oracle_mod, tmp_path):
    """A missing graph file / page reads as "no flags", as in the reference.  A missing raw-vector file, and
    raw-vector pages past the end of a file, are where the reference's reader makes up zero vectors
    (ngh_partition_manager.dart:270-281): here those ids are ABSENT rows -- never returned -- and counted."""
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(4)
    n, dims = 2000, 96
    v = rng.standard_normal((n, dims)).astype(np.float32)
    root = tmp_path / "ngh"
    ngh_dir.write_ngh_dir(str(root), v, metric=0, max_partition_file_size=16384 * 8, deleted=[3, 400, 600, 1999],
                          skip_rawvec_partitions=(1,), skip_graph_partitions=(0,))
    p2 = root / "rawvec" / "dir_0" / "p2.ngh"
    with open(p2, "r+b") as f:
        f.truncate(16384 * 4)  # pages 4.. of partition 2 vanish
    _, vec, dead = ngh_dir.read_ngh_dir(str(root))
    # graph partition 0 = ids 0..503 (63 slots x 8 pages), rawvec partition 1 = ids 336..671 (42 x 8)
    assert not dead[3] and not dead[400] and dead[600] and dead[1999]
    assert not vec[336:672].any() and vec[335].any() and vec[672].any()
    absent = ~vec.any(axis=1)  # the reader restatement yields zero vectors exactly for the ids not on disk
    assert absent[336:672].all() and absent.sum() > 336
    idx, info = HipVectorIndex.open_ngh(str(root))
    with idx:
        assert info["rows_loaded"] == n - int(absent.sum()) and info["tombstones"] == int(dead.sum())
        assert info["files_absent"] == 1 and info["pages_absent"] == int(absent.sum()) // 42
        _check_search(oracle_mod, idx, vec, dead | absent, 0, k=60)
        # the zero query ranks zero vectors first: none of the made-up rows may come back
        ids, dist, cnt = idx.search(np.zeros(dims, np.float32), 100)
        assert not absent[ids[0][:cnt[0]]].any()


def test_open_ngh_defaults_and_errors(hip_lib, oracle_mod, tmp_path):
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(2)
    v = rng.standard_normal((500, 64)).astype(np.float32)
    root = tmp_path / "ngh"
    meta = ngh_dir.write_ngh_dir(str(root), v, metric=2)
    # fromJson defaults: cosine, float32, 16 KiB pages, 16 MiB partitions, maxDegree 64
    slim = {k: meta[k] for k in ("name", "tableName", "fieldName", "dimensions", "timestamps", "nextNodeId")}
    slim["nextNodeId"] = 500.0  # (json[...] as num).toInt()
    (root / "meta.json").write_text(json.dumps(slim, indent=2))
    idx, info = HipVectorIndex.open_ngh(str(root))
    with idx:
        assert (info["metric"], info["precision"], info["page_size"], info["max_degree"]) == (2, 1, 16384, 64)
        assert info["max_partition_file_size"] == 16 << 20 and info["rows_loaded"] == 500
        _check_search(oracle_mod, idx, v, np.zeros(500, bool), 2)
    # corrupt graph page -> TSH_E_FORMAT (BTreePageIO.parsePageBytes throws, btree_page.dart:226-230)
    g = root / "graph" / "dir_0" / "p0.ngh"
    raw = bytearray(g.read_bytes())
    raw[16384 + 200] ^= 1
    g.write_bytes(bytes(raw))
    with pytest.raises(_ffi.TshError) as e:
        HipVectorIndex.open_ngh(str(root))
    assert e.value.code == _ffi.TSH_E_FORMAT
    (root / "meta.json").write_text("[1,2,3]")
    with pytest.raises(_ffi.TshError) as e:
        HipVectorIndex.open_ngh(str(root))
    assert e.value.code == _ffi.TSH_E_FORMAT
    os.remove(root / "meta.json")
    with pytest.raises(_ffi.TshError) as e:
        HipVectorIndex.open_ngh(str(root))
    assert e.value.code == _ffi.TSH_E_IO


def test_open_ngh_empty_index(hip_lib, tmp_path):
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex

    root = tmp_path / "ngh"
    ngh_dir.write_ngh_dir(str(root), np.zeros((0, 32), np.float32), metric=0)
    idx, info = HipVectorIndex.open_ngh(str(root))
    with idx:
        assert info["rows_loaded"] == 0 and idx.size == 0
        ids, dist, cnt = idx.search(np.zeros(32, np.float32), 5)
        assert cnt[0] == 0


def _merged_shard_search(shards, q, k, metric, dims):
    """The exchange of a sharded call without its transport: every shard's candidate blocks (tsh_search_shard, device
    memory), concatenated as an all-gather would deliver them, merged by tsh_merge_candidates."""
    import ctypes

    import torch

    from tostore_amd import _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    L = _ffi.lib()
    q = np.ascontiguousarray(np.atleast_2d(q), np.float32)
    entries = L.tsh_default_block_entries(k)
    bb = L.tsh_candidate_block_bytes(entries)
    bufs = []
    for idx in shards:
        buf = torch.empty(len(q) * bb, dtype=torch.uint8, device="cuda")
        _ffi.check(L.tsh_search_shard(idx._h, q.ctypes.data_as(_ffi.p_f32), len(q), k, None, entries,
                                      ctypes.c_void_p(buf.data_ptr()), None))
        bufs.append(buf)
    return merge_candidate_blocks(metric, dims, q, k, None, torch.cat(bufs).cpu().numpy(), len(shards), entries)


@pytest.mark.parametrize("metric,precision,dims,n,world", [(0, 1, 96, 3000, 3), (2, 0, 64, 1500, 4), (1, 2, 128, 2500, 3),
                                                           (0, 1, 100, 777, 8)])
def test_open_ngh_shards_one_rank_at_a_time(hip_lib, oracle_mod, tmp_path, metric, precision, dims, n, world):
    """tsh_index_open_ngh_shard: every rank of a one-process-per-GPU deployment opens ITS node-id range of <index>/ngh
    -- [rank * ceil(N / W), ...), which the addressing of ngh_index_meta.dart:480-490 makes a run of partition files
    and pages -- and the W shards, merged, answer what the whole-index handle answers.  The directory has 8 data pages
    per partition file, so the ranges begin inside files and inside pages (96 dims: 42 vectors per page, 336 per file;
    rank 1 of 3 starts at id 1000 = file 2, page 7, slot 34)."""
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(n + world)
    v = (rng.standard_normal((n, dims)) * 0.4).astype(np.float32)
    deleted = sorted(set(rng.integers(0, n, n // 15).tolist()) | {0, n - 1, (n + world - 1) // world})
    root = tmp_path / "ngh"
    ngh_dir.write_ngh_dir(str(root), v, metric=metric, precision=precision, max_partition_file_size=16384 * 8,
                          deleted=deleted, max_entries_per_dir=3)
    _, vec, dead = ngh_dir.read_ngh_dir(str(root), 3)
    per = (n + world - 1) // world
    whole, winfo = HipVectorIndex.open_ngh(str(root), max_entries_per_dir=3)
    shards = []
    try:
        assert (winfo["row_base"], winfo["row_end"]) == (0, n)
        loaded = tombs = 0
        for r in range(world):
            idx, info = HipVectorIndex.open_ngh_shard(str(root), world, r, device=0, max_entries_per_dir=3)
            shards.append(idx)
            lo, hi = min(n, r * per), min(n, (r + 1) * per)
            assert (info["row_base"], info["row_end"]) == (lo, hi) and idx.row_base == lo
            assert info["rows_loaded"] == hi - lo and info["pages_absent"] == 0 and info["files_absent"] == 0
            assert info["tombstones"] == int(dead[lo:hi].sum())
            assert info["next_node_id"] == n and info["dimensions"] == dims and info["metric"] == metric
            assert idx.size == (hi if hi > lo else 0)
            # a rank reads its own files only: fewer of them than the whole index has (once there are several per rank)
            if world <= 4 and n >= 1500:
                assert info["files_read"] < winfo["files_read"]
            loaded += info["rows_loaded"]
            tombs += info["tombstones"]
        assert loaded == n and tombs == len(deleted) == winfo["tombstones"]
        k = 40
        for _ in range(4):
            q = rng.standard_normal(dims).astype(np.float32)
            if metric == 2:
                q = oracle_mod.normalize_f32(q)
            want = whole.search(q, k)
            got = _merged_shard_search([s for s in shards if s.size], q, k, metric, dims)
            assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])
            assert np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64))
            eids, edist = oracle_mod.search_exhaustive(vec, q, metric, k, keep=_keep_bits(dead))
            assert np.array_equal(got[0][0][:got[2][0]], eids) and np.array_equal(got[1][0][:got[2][0]], edist)
    finally:
        whole.close()
        for s in shards:
            s.close()


def test_open_ngh_shard_census_and_errors(hip_lib, oracle_mod, tmp_path):
    """A shard's census covers ITS range only: the missing partition file and the truncated one of
    test_open_ngh_missing_and_short_files are counted by the ranks whose ids they hold, and nobody else."""
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(4)
    n, dims = 2000, 96
    v = rng.standard_normal((n, dims)).astype(np.float32)
    root = tmp_path / "ngh"
    ngh_dir.write_ngh_dir(str(root), v, metric=0, max_partition_file_size=16384 * 8, deleted=[3, 400, 600, 1999],
                          skip_rawvec_partitions=(1,))
    with open(root / "rawvec" / "dir_0" / "p2.ngh", "r+b") as f:
        f.truncate(16384 * 4)  # pages 4.. of partition 2 (ids 798 .. 1007) vanish
    _, vec, dead = ngh_dir.read_ngh_dir(str(root))
    absent = ~vec.any(axis=1)
    whole, winfo = HipVectorIndex.open_ngh(str(root))
    whole.close()
    tot_pages = tot_files = tot_rows = 0
    for r in range(4):  # ranges of 500 ids: [0, 500) holds the missing file's 336 .. 499, [500, 1000) the rest + the short file
        idx, info = HipVectorIndex.open_ngh_shard(str(root), 4, r, device=0)
        with idx:
            lo, hi = r * 500, (r + 1) * 500
            assert info["rows_loaded"] == 500 - int(absent[lo:hi].sum())
            assert info["tombstones"] == int(dead[lo:hi].sum())
            tot_rows += info["rows_loaded"]
            # rank 2's first ids, 1000 .. 1007, sit in the truncated file's last page; rank 3 has no hole at all
            assert info["pages_absent"] == {2: 1, 3: 0}.get(r, info["pages_absent"]) and (r < 2 or info["files_absent"] == 0)
            q = rng.standard_normal(dims).astype(np.float32)
            ids, dist, cnt = idx.search(q, 30)
            keep = np.zeros(n, bool)
            keep[lo:hi] = True
            eids, edist = oracle_mod.search_exhaustive(vec, q, 0, 30, keep=_keep_bits(dead | absent | ~keep))
            assert np.array_equal(ids[0][:cnt[0]], eids) and np.array_equal(dist[0][:cnt[0]], edist)
        tot_files += info["files_absent"]
    assert tot_rows == winfo["rows_loaded"] and tot_files == 2  # (the missing file straddles ranks 0 and 1)
    for bad in ((0, 0), (3, 3), (2, -1)):
        with pytest.raises(_ffi.TshError) as e:
            HipVectorIndex.open_ngh_shard(str(root), bad[0], bad[1], device=0)
        assert e.value.code == _ffi.TSH_E_BAD_ARG
    # more ranks than rows: the ranks past the end hold nothing and answer nothing
    small = tmp_path / "small"
    ngh_dir.write_ngh_dir(str(small), v[:5], metric=0)
    idx, info = HipVectorIndex.open_ngh_shard(str(small), 8, 7, device=0)
    with idx:
        assert (info["row_base"], info["row_end"], info["rows_loaded"]) == (5, 5, 0) and idx.size == 0
        assert idx.search(v[0], 3)[2][0] == 0


def test_open_ngh_into_a_handle_over_several_devices(hip_lib, oracle_mod, tmp_path):
    """tsh_index_open_ngh(n_devices = 3): the loader's pipeline (pages read and decoded on the pool while a helper thread appends the
    batch before) feeds a handle whose appends are routed to three shards (here sharing the one GPU); many partition files, a
    missing one among them; the same answers as the one-device handle and as the oracle."""
    from oracle import ngh_dir
    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(21)
    n, dims = 9000, 80
    v = (rng.standard_normal((n, dims)) * 0.5).astype(np.float32)
    deleted = sorted(set(rng.integers(0, n, 300).tolist()))
    root = tmp_path / "ngh"
    ngh_dir.write_ngh_dir(str(root), v, metric=0, max_partition_file_size=16384 * 6, deleted=deleted, skip_rawvec_partitions=[4])
    os.environ["TSH_SHARDS_SHARE_DEVICES"] = "1"
    _ffi.enable_test_hooks()
    try:
        one, info1 = HipVectorIndex.open_ngh(str(root))
        many, info3 = HipVectorIndex.open_ngh(str(root), n_devices=3)
        with one, many:
            assert info3["rows_loaded"] == info1["rows_loaded"] < n and info3["files_absent"] == info1["files_absent"] == 1
            assert info3["tombstones"] == info1["tombstones"]
            qs = rng.standard_normal((5, dims)).astype(np.float32)
            a, b = one.search(qs, 30), many.search(qs, 30)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
            ids = b[0]
            assert not np.isin(ids[ids >= 0], deleted).any()  # (the one-device handle is held to the oracle by the tests above)
    finally:
        _ffi.enable_test_hooks(False)
        del os.environ["TSH_SHARDS_SHARE_DEVICES"]
