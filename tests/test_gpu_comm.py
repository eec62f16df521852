"""GPU: the RCCL exchange behind the C ABI (tsh_comm_*, tsh_search_sharded).  A gpurun box has ONE GPU and RCCL
does not allow two ranks on one device, so what can be executed here is a communicator of world size 1: the full
code path -- dlopen of librccl, ncclCommInitRank, shard scan into device blocks, ncclAllGather, D2H, host merge,
overflow retry -- with nothing but the cross-rank traffic missing.  N > 1 ranks are the driver's 8-GPU run
(bench.py --exchange capi) and, on the CPU, the gloo tests of the same block format and merge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=["exchange_when_final", "exchange_ahead"])
def exchange_ahead(request, hip_lib):
    """Both ways tsh_search_sharded launches a group's all-gather over real RCCL (TSH_OPT_EXCHANGE_AHEAD)."""
    from tostore_amd import _ffi

    _ffi.check(hip_lib.tsh_index_set_option(None, _ffi.TSH_OPT_EXCHANGE_AHEAD, request.param))
    yield request.param
    _ffi.check(hip_lib.tsh_index_set_option(None, _ffi.TSH_OPT_EXCHANGE_AHEAD, 0))


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_world_of_one_equals_plain_search(hip_lib, oracle_mod, metric, exchange_ahead):
    from tostore_amd import HipVectorIndex
    from tostore_amd.sharded import CommSearcher

    rng = np.random.default_rng(metric)
    n, d, k = 30_000, 96, 50
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[100] = rows[200] = rows[7]  # ties
    qs = rng.standard_normal((70, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle_mod.normalize_f32(q) for q in qs])
    keep = np.packbits(rng.random(n + 5000) < 0.4, bitorder="little")
    uid = CommSearcher.unique_id()
    assert len(uid) == 128
    with HipVectorIndex(d, metric, capacity_rows=n, shard_device=0, row_base=5000) as idx:
        idx.append(5000, rows)
        with CommSearcher(idx, 1, 0, uid) as cs:
            for queries, mask in ((qs[0], None), (qs[:3], keep), (qs, None)):  # single, few (pipelined), batched
                ids, dist, cnt = cs.search(queries, k, None, mask)
                e_ids, e_dist, e_cnt = idx.search(queries, k, None, mask)
                assert np.array_equal(cnt, e_cnt) and np.array_equal(ids, e_ids)
                assert np.array_equal(dist.view(np.uint64), e_dist.view(np.uint64))
            # against the oracle too (global ids = 5000 + local)
            ids, dist, cnt = cs.search(qs[1], k)
            eids, edist = oracle_mod.search_heap(rows, qs[1], metric, k)
            assert np.array_equal(ids[0], eids + 5000) and np.array_equal(dist[0], edist)
            # ties wider than a block: every row identical -> the merge asks for more entries, the call retries
            same = np.tile(rows[:1], (3000, 1))
            with HipVectorIndex(d, metric, capacity_rows=3000, shard_device=0, row_base=0) as flat:
                flat.append(0, same)
                with CommSearcher(flat, 1, 0, CommSearcher.unique_id()) as cs2:
                    ids, dist, cnt = cs2.search(qs[2], 10)
                    assert cnt[0] == 10 and ids[0].tolist() == list(range(10))


def test_argument_errors(hip_lib):
    import ctypes

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import CommSearcher

    L = _ffi.lib()
    c = ctypes.c_void_p()
    assert L.tsh_comm_create(None, 1, 0, 0, ctypes.byref(c)) == _ffi.TSH_E_BAD_ARG
    uid = CommSearcher.unique_id()
    assert L.tsh_comm_create(ctypes.create_string_buffer(uid, 128), 2, 2, 0, ctypes.byref(c)) == _ffi.TSH_E_BAD_ARG
    assert L.tsh_comm_destroy(None) == 0
    with HipVectorIndex(8, 0, capacity_rows=128, n_devices=1) as idx, HipVectorIndex(8, 0, shard_device=0) as sh:
        with CommSearcher(sh, 1, 0, uid) as cs:
            cnt = np.zeros(1, np.int32)
            assert L.tsh_search_sharded(None, cs._c, None, 1, 1, 0.0, None, None, None, None) == _ffi.TSH_E_BAD_ARG
            ids, dist, cnt = cs.search(np.zeros(8, np.float32), 3)  # empty shard: empty answer
            assert cnt[0] == 0
