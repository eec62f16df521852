"""CPU, world_size 2 over gloo: the N>1 data path of tostore_amd.sharded --
per-rank candidate blocks, all-gather, host merge -- equals one exhaustive
search over the whole corpus.  (The blocks are built from oracle sums here;
on the GPU box they come from tsh_search_shard.)"""
import os
import socket
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle
    from tostore_amd import _ffi
    from tostore_amd.sharded import merge_candidate_blocks

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)  # same corpus on every rank
        n, d, k = 3001, 48, 25
        rows = rng.standard_normal((n, d)).astype(np.float32)
        rows[1500] = rows[1501] = rows[2]  # cross-shard ties -> global id order
        qs = rng.standard_normal((3, d)).astype(np.float32)
        if metric == 2:
            qs = np.stack([oracle.normalize_f32(q) for q in qs])
        per = (n + world - 1) // world
        lo, hi = rank * per, min(n, (rank + 1) * per)
        entries = _ffi.lib().tsh_default_block_entries(k)
        bb = _ffi.lib().tsh_candidate_block_bytes(entries)
        mine = bytearray(len(qs) * bb)
        for qi, q in enumerate(qs):
            ids, _ = oracle.search_exhaustive(rows[lo:hi], q, metric, k)
            struct.pack_into("<8IqqI", mine, qi * bb, len(ids), entries, 0, 0, 0, 0, k, metric, lo, hi - lo, 0)
            for i, lid in enumerate(ids):
                s0, s1 = oracle.exact_sums(q, rows[lo + int(lid)], metric)
                struct.pack_into("<qdd", mine, qi * bb + 64 + 24 * i, lo + int(lid), s0, s1)
        t_mine = torch.frombuffer(mine, dtype=torch.uint8)
        t_all = torch.empty(world * len(mine), dtype=torch.uint8)
        dist.all_gather_into_tensor(t_all, t_mine)
        ids, dd, cnt = merge_candidate_blocks(metric, d, qs, k, None, t_all.numpy(), world, entries)
        ok = True
        for qi, q in enumerate(qs):
            eids, edist = oracle.search_exhaustive(rows, q, metric, k)
            ok &= cnt[qi] == k and np.array_equal(ids[qi], eids) and np.array_equal(dd[qi], edist)
        out_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_allgather_merge_world2(metric):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, metric, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
