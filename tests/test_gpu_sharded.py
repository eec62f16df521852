"""GPU: row-range shards (tsh_index_create_shard + tsh_search_shard) produce
device candidate blocks whose host merge equals one exhaustive search over
the whole corpus.  Two shards live on the one GPU of the test box; the
all-gather itself is covered by tests/test_distributed_gloo.py."""
import ctypes

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]  # small shapes: exact sums AND the f32 pre-filter (conftest)
L2, IP, COS = 0, 1, 2


def _shard_blocks(torch, idx, qs, k, entries, mask=None):
    from tostore_amd import _ffi

    L = _ffi.lib()
    bb = L.tsh_candidate_block_bytes(entries)
    buf = torch.empty(len(qs) * bb, dtype=torch.uint8, device="cuda")
    mp = None
    if mask is not None:
        mp = mask.ctypes.data_as(_ffi.p_u8)
    q = np.ascontiguousarray(qs, dtype=np.float32)
    _ffi.check(L.tsh_search_shard(idx._h, q.ctypes.data_as(_ffi.p_f32), len(qs), k, mp, entries,
                                  ctypes.c_void_p(buf.data_ptr()), None))
    return buf


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("nq", [3, 24])  # 24 takes the batched matrix-core path inside each shard
def test_two_shards_merge_to_global_answer(hip_lib, oracle_mod, metric, nq):
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    rng = np.random.default_rng(5)
    n, d, k = 20003, 96, 30
    split = 9997  # row_base of shard 1 is not a multiple of 8 or 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[split - 1] = rows[split] = rows[17]  # ties across the shard boundary
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == COS:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    keep = np.packbits(rng.random(n) < 0.6, bitorder="little")
    s0 = HipVectorIndex(d, metric, shard_device=0, row_base=0)
    s1 = HipVectorIndex(d, metric, shard_device=0, row_base=split)
    try:
        s0.append(0, rows[:split])
        s1.append(split, rows[split:])  # shard handles take GLOBAL ids
        entries = _ffi.lib().tsh_default_block_entries(k)
        for mask in (None, keep):
            b0 = _shard_blocks(torch, s0, qs, k, entries, mask)
            b1 = _shard_blocks(torch, s1, qs, k, entries, mask)
            allb = torch.cat([b0, b1]).cpu().numpy()
            ids, dist, cnt = merge_candidate_blocks(metric, d, qs, k, None, allb, 2, entries)
            for i in range(nq):
                eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, None, mask)
                assert cnt[i] == len(eids) and np.array_equal(ids[i, :cnt[i]], eids)
                assert np.array_equal(dist[i, :cnt[i]], edist)
    finally:
        s0.close()
        s1.close()


def test_truncated_block_asks_every_rank_to_retry(hip_lib, oracle_mod, scan_path):
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    d, n, k = 32, 6000, 20
    rows = np.tile(np.random.default_rng(1).standard_normal((1, d)).astype(np.float32), (n, 1))  # all tied
    q = np.random.default_rng(2).standard_normal(d).astype(np.float32)
    with HipVectorIndex(d, L2, shard_device=0, row_base=1000) as s:
        s.append(1000, rows)
        entries = _ffi.lib().tsh_default_block_entries(k)
        blk = _shard_blocks(torch, s, q[None], k, entries).cpu().numpy()
        if scan_path != "prefilter":  # 6000 rows: the block holds the k lowest ids of the tie, nothing is truncated
            ids, dist, cnt = merge_candidate_blocks(L2, d, q, k, None, blk, 1, entries)
        else:
            with pytest.raises(_ffi.TshError) as e:
                merge_candidate_blocks(L2, d, q, k, None, blk, 1, entries)
            assert e.value.code == _ffi.TSH_E_OVERFLOW and e.value.needed_entries >= n
            blk = _shard_blocks(torch, s, q[None], k, e.value.needed_entries).cpu().numpy()
            ids, dist, cnt = merge_candidate_blocks(L2, d, q, k, None, blk, 1, e.value.needed_entries)
        assert ids[0].tolist() == list(range(1000, 1000 + k))  # ties -> lowest global ids


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_in_process_multi_shard_handle(hip_lib, oracle_mod, metric, monkeypatch):
    """tsh_index_create(n_devices=3): the one-process deployment shape (a Dart server with several
    GPUs).  TSH_SHARDS_SHARE_DEVICES=1 lets the three shards share this box's single GPU."""
    from tostore_amd import HipVectorIndex, _ffi

    monkeypatch.setenv("TSH_SHARDS_SHARE_DEVICES", "1")
    _ffi.enable_test_hooks(False)  # (whatever an earlier test of this process left)
    with pytest.raises(_ffi.TshError):  # the variable alone changes nothing: three shards need three GPUs ...
        HipVectorIndex(40, metric, capacity_rows=100, n_devices=3)
    _ffi.enable_test_hooks()  # ... until the process itself asks for the test hooks
    rng = np.random.default_rng(9)
    n, d, k = 10_000, 40, 25
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((10, d)).astype(np.float32)
    if metric == COS:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    with HipVectorIndex(d, metric, capacity_rows=n, n_devices=3) as idx:
        idx.append(0, rows[:3000])       # inside shard 0
        idx.append(3000, rows[3000:7777])  # straddles shard 0 -> 1 -> 2 boundaries (3392 rows per shard)
        idx.append(7777, rows[7777:])
        assert idx.size == n
        dead = rng.choice(n, 700, replace=False)
        idx.set_deleted(dead)
        alive = np.ones(n, bool)
        alive[dead] = False
        keep = np.packbits(rng.random(n) < 0.7, bitorder="little")
        both = np.packbits(alive & np.unpackbits(keep, bitorder="little")[:n].astype(bool), bitorder="little")
        for nq in (1, 10):
            ids, dist, cnt = idx.search(qs[:nq], k, None, keep)
            for i in range(nq):
                eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, None, both)
                assert cnt[i] == len(eids) and np.array_equal(ids[i, :cnt[i]], eids)
                assert np.array_equal(dist[i, :cnt[i]], edist)
        t = idx.submit(qs[0], k)  # asynchronous form over all shards
        ids, dist = idx.wait(t)
        eids, edist = oracle_mod.search_exhaustive(rows, qs[0], metric, k, None, np.packbits(alive, bitorder="little"))
        assert np.array_equal(ids, eids) and np.array_equal(dist, edist)
        c = idx.counters()
        assert c["rows"] == n and c["deleted_rows"] == 700
    _ffi.enable_test_hooks(False)
