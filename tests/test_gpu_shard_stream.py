"""GPU: the progressive shard search (tsh_search_shard_begin / _progress / _end, ABI 4) -- the scans of all queries
of a call as ONE pipeline on a library thread, blocks final in query order -- leaves exactly the blocks
tsh_search_shard leaves, on both routes (single-query scans, matrix cores), with masks, ties that overflow a block,
an empty shard; progress is monotonic and answers for a prefix as soon as that prefix is final."""
import ctypes

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]  # small shapes: exact sums AND the f32 pre-filter (conftest)
L2, IP, COS = 0, 1, 2


def _blocks(torch, idx, qs, k, entries, mask=None):
    from tostore_amd import _ffi

    L = _ffi.lib()
    bb = L.tsh_candidate_block_bytes(entries)
    buf = torch.zeros(len(qs) * bb, dtype=torch.uint8, device="cuda")
    mp = None if mask is None else mask.ctypes.data_as(_ffi.p_u8)
    q = np.ascontiguousarray(qs, dtype=np.float32)
    _ffi.check(L.tsh_search_shard(idx._h, q.ctypes.data_as(_ffi.p_f32), len(qs), k, mp, entries,
                                  ctypes.c_void_p(buf.data_ptr()), None))
    return buf.cpu().numpy()


def _stream_blocks(torch, idx, qs, k, entries, mask=None, step=0, wants=None):
    """-> (blocks, the done counts _progress reported for `wants`)."""
    from tostore_amd import _ffi

    L = _ffi.lib()
    bb = L.tsh_candidate_block_bytes(entries)
    nq = len(qs)
    buf = torch.zeros(nq * bb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    q = np.ascontiguousarray(qs, dtype=np.float32).copy()
    m = None if mask is None else mask.copy()
    st = ctypes.c_void_p()
    _ffi.check(L.tsh_search_shard_begin(idx._h, q.ctypes.data_as(_ffi.p_f32), nq, k,
                                        None if m is None else m.ctypes.data_as(_ffi.p_u8), entries,
                                        ctypes.c_void_p(buf.data_ptr()), step, ctypes.byref(st)))
    q[:] = np.nan  # the inputs were consumed by _begin: scribbling over them must not matter
    if m is not None:
        m[:] = 0
    seen, prefix = [], {}
    try:
        for w in (wants if wants is not None else [nq]):
            done = ctypes.c_int32(-1)
            _ffi.check(L.tsh_search_shard_progress(st, w, ctypes.byref(done)))
            assert done.value >= min(w, nq)
            seen.append(done.value)
            if w < nq:  # the prefix that was reported final must not change afterwards
                prefix[w] = buf[: w * bb].cpu().numpy().copy()
    finally:
        _ffi.check(L.tsh_search_shard_end(st))
    out = buf.cpu().numpy()
    for w, p in prefix.items():
        assert np.array_equal(out[: w * bb], p), "a block reported final changed later"
    return out, seen


def _check(oracle_mod, rows, qs, k, metric, blocks, entries, base=0, mask=None):
    from tostore_amd.sharded import merge_candidate_blocks

    ids, dist, cnt = merge_candidate_blocks(metric, rows.shape[1], qs, k, None, blocks, 1, entries)
    local = None
    if mask is not None:
        bits = np.unpackbits(mask, bitorder="little")[base:base + len(rows)]
        local = np.packbits(bits, bitorder="little")
    for i in range(len(qs)):
        eids, edist = oracle_mod.search_exhaustive(rows, qs[i], metric, k, None, local)
        assert cnt[i] == len(eids) and np.array_equal(ids[i, :cnt[i]], eids + base), i
        assert np.array_equal(dist[i, :cnt[i]], edist), i


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("min_nq", [0, 1])  # 0: every query its own scan; 1: steps of 24 go to the matrix cores
def test_stream_equals_plain_shard_search(hip_lib, oracle_mod, metric, min_nq):
    import torch

    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(11 + metric)
    n, d, k, nq, base = 30011, 96, 25, 53, 4099
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == COS:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    keep = np.packbits(rng.random(base + n) < 0.5, bitorder="little")
    sparse = np.packbits(rng.random(base + n) < 0.01, bitorder="little")  # the list scan
    with HipVectorIndex(d, metric, shard_device=0, row_base=base) as s:
        s.append(base, rows)
        s.set_batch_min_nq(min_nq)
        entries = _ffi.lib().tsh_default_block_entries(k)
        for mask in (None, keep, sparse):
            plain = _blocks(torch, s, qs, k, entries, mask)
            wants = [1, 7, 24, 25, 48, nq, nq + 5]
            got, seen = _stream_blocks(torch, s, qs, k, entries, mask, step=24, wants=wants)
            assert seen == sorted(seen) and seen[-1] == nq
            _check(oracle_mod, rows, qs, k, metric, got, entries, base, mask)
            _check(oracle_mod, rows, qs, k, metric, plain, entries, base, mask)
        # step 0 = the whole call; ending a stream nobody asked about waits for it
        got, _ = _stream_blocks(torch, s, qs, k, entries, None, step=0, wants=[])
        _check(oracle_mod, rows, qs, k, metric, got, entries, base)
        c = s.counters()
        assert c["fallback_searches"] == 0


def test_stream_ties_overflow_like_the_plain_call(hip_lib, oracle_mod, scan_path):
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    d, n, k = 32, 6000, 20
    rows = np.tile(np.random.default_rng(1).standard_normal((1, d)).astype(np.float32), (n, 1))  # all tied
    qs = np.random.default_rng(2).standard_normal((9, d)).astype(np.float32)
    with HipVectorIndex(d, L2, shard_device=0, row_base=1000) as s:
        s.append(1000, rows)
        s.set_batch_min_nq(0)
        entries = _ffi.lib().tsh_default_block_entries(k)
        blk, _ = _stream_blocks(torch, s, qs, k, entries, None, step=3, wants=[3, 6, 9])
        if scan_path != "prefilter":  # 6000 rows: the block holds the k lowest ids of the tie, nothing overflows
            ids, dist, cnt = merge_candidate_blocks(L2, d, qs, k, None, blk, 1, entries)
        else:
            with pytest.raises(_ffi.TshError) as e:
                merge_candidate_blocks(L2, d, qs, k, None, blk, 1, entries)
            assert e.value.code == _ffi.TSH_E_OVERFLOW and e.value.needed_entries >= n
            blk, _ = _stream_blocks(torch, s, qs, k, e.value.needed_entries, None, step=3)
            ids, dist, cnt = merge_candidate_blocks(L2, d, qs, k, None, blk, 1, e.value.needed_entries)
        assert all(ids[i].tolist() == list(range(1000, 1000 + k)) for i in range(9))


def test_stream_empty_shard_and_bad_arguments(hip_lib, scan_path):
    import torch

    from tostore_amd import HipVectorIndex, _ffi

    L = _ffi.lib()
    d, k = 16, 5
    qs = np.ones((4, d), np.float32)
    entries = L.tsh_default_block_entries(k)
    bb = L.tsh_candidate_block_bytes(entries)
    with HipVectorIndex(d, L2, shard_device=0, row_base=77) as s:
        blk, seen = _stream_blocks(torch, s, qs, k, entries, None, step=2, wants=[2, 4])
        hdr = blk.reshape(4, bb)[:, :16].view(np.uint32)
        assert seen[-1] == 4 and (hdr[:, 0] == 0).all() and (hdr[:, 1] == entries).all()
        buf = torch.zeros(4 * bb, dtype=torch.uint8, device="cuda")
        st = ctypes.c_void_p()
        qp = qs.ctypes.data_as(_ffi.p_f32)
        dp = ctypes.c_void_p(buf.data_ptr())
        assert L.tsh_search_shard_begin(None, qp, 4, k, None, entries, dp, 0, ctypes.byref(st)) == _ffi.TSH_E_BAD_ARG
        assert L.tsh_search_shard_begin(s._h, qp, 0, k, None, entries, dp, 0, ctypes.byref(st)) == _ffi.TSH_E_BAD_ARG
        assert L.tsh_search_shard_begin(s._h, qp, 4, k, None, entries, None, 0, ctypes.byref(st)) == _ffi.TSH_E_BAD_ARG
        assert L.tsh_search_shard_begin(s._h, qp, 4, k, None, entries, dp, -1, ctypes.byref(st)) == _ffi.TSH_E_BAD_ARG
        assert L.tsh_search_shard_begin(s._h, qp, 4, k, None, entries, dp, 0, None) == _ffi.TSH_E_BAD_ARG
        assert not st.value
        assert L.tsh_search_shard_progress(None, 1, None) == _ffi.TSH_E_BAD_ARG
        assert L.tsh_search_shard_end(None) == _ffi.TSH_OK
    with HipVectorIndex(d, L2, capacity_rows=10, n_devices=1) as whole:  # (a plain handle has one shard: accepted)
        whole.append(0, np.ones((10, d), np.float32))
        blk, seen = _stream_blocks(torch, whole, qs, k, entries, None, step=0)
        # (ten rows: the pre-filter offers every one of them, the exact path the k winners -- two waves of rows are too few
        # for the wide pick's bound, the one-workgroup select ranks them)
        assert (blk.reshape(4, bb)[:, :4].view(np.uint32)[:, 0] == (k if scan_path != "prefilter" else 10)).all()


def test_appends_wait_for_a_running_stream(hip_lib, oracle_mod):
    """The stream holds the shard share-locked while it runs: an append started meanwhile lands after it, and the
    stream's blocks are those of the rows it started with."""
    import threading

    import torch

    from tostore_amd import HipVectorIndex, _ffi

    rng = np.random.default_rng(3)
    n, d, k, nq = 50000, 128, 10, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    more = rng.standard_normal((500, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    with HipVectorIndex(d, L2, shard_device=0, row_base=0) as s:
        s.append(0, rows)
        s.set_batch_min_nq(0)
        entries = _ffi.lib().tsh_default_block_entries(k)
        L = _ffi.lib()
        bb = L.tsh_candidate_block_bytes(entries)
        buf = torch.zeros(nq * bb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        st = ctypes.c_void_p()
        _ffi.check(L.tsh_search_shard_begin(s._h, qs.ctypes.data_as(_ffi.p_f32), nq, k, None, entries,
                                            ctypes.c_void_p(buf.data_ptr()), 8, ctypes.byref(st)))
        done = ctypes.c_int32(0)
        _ffi.check(L.tsh_search_shard_progress(st, 1, ctypes.byref(done)))  # it is running (and holds the shard)
        t = threading.Thread(target=lambda: s.append(n, more))
        t.start()
        _ffi.check(L.tsh_search_shard_end(st))
        t.join()
        assert s.size == n + 500
        _check(oracle_mod, rows, qs, k, L2, buf.cpu().numpy(), entries)
