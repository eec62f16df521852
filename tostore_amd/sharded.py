"""Row-range sharded search, one process per GPU.

Rank g holds global rows [base_g, base_g + n_g) in its own `HipVectorIndex`
shard.  A query is scanned by every rank; each rank's fixed-size candidate
block (exact f64 sums of every row that can be in ITS top k) stays in device
memory, the blocks are all-gathered with `torch.distributed` (backend "nccl"
is RCCL over xGMI on ROCm; k'*24 B per rank per query, latency-bound), and the
host merge (`tsh_merge_candidates`: sqrt / negate / 1-cos, threshold, ordering,
top-k cut) runs on whichever ranks want the answer.  The reference has no
counterpart (it has no distributed compute at all); the
result is identical to one un-sharded index over the same rows.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import numpy as np

from . import _ffi


def merge_candidate_blocks(metric: int, dim: int, queries: np.ndarray, k: int,
                           distance_threshold: Optional[float], blocks: np.ndarray, n_blocks: int,
                           entries: int):
    """Host merge of `n_blocks` x nq candidate blocks (uint8 array, layout [block][query]).

    Returns (ids[nq,k], dist[nq,k], count[nq]); raises _ffi.TshError(TSH_E_OVERFLOW) with
    `.needed_entries` set when a block was truncated."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    if q.ndim == 1:
        q = q[None, :]
    nq = q.shape[0]
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    bb = _ffi.lib().tsh_candidate_block_bytes(entries)
    if blocks.size != n_blocks * nq * bb:
        raise ValueError("blocks buffer has the wrong size")
    kk = max(int(k), 0)
    ids = np.full((nq, max(kk, 1)), -1, dtype=np.int64)
    dist = np.full((nq, max(kk, 1)), np.nan, dtype=np.float64)
    cnt = np.zeros(nq, dtype=np.int32)
    need = ctypes.c_int32(entries)
    thr = math.nan if distance_threshold is None else float(distance_threshold)
    rc = _ffi.lib().tsh_merge_candidates(metric, dim, q.ctypes.data_as(_ffi.p_f32), nq, int(k), thr,
                                         blocks.ctypes.data_as(ctypes.c_void_p), n_blocks, entries,
                                         ids.ctypes.data_as(_ffi.p_i64), dist.ctypes.data_as(_ffi.p_f64),
                                         cnt.ctypes.data_as(_ffi.p_i32), ctypes.byref(need))
    if rc != _ffi.TSH_OK:
        err = _ffi.TshError(rc, _ffi.last_error())
        err.needed_entries = need.value
        raise err
    return ids[:, :kk], dist[:, :kk], cnt


class ShardedSearcher:
    """All-gather + merge around one local shard.  `group=None` = default process group."""

    def __init__(self, shard_index, group=None):
        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        self.index = shard_index
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bufs = {}

    def _buffers(self, nq: int, entries: int, slot: int = 0):
        """(mine, all, host) byte buffers for nq blocks per rank: views of per-(entries, slot) buffers that only
        grow, so a short last group does not allocate (pinned allocations cost about a millisecond)."""
        t = self._torch
        bb = _ffi.lib().tsh_candidate_block_bytes(entries)
        key = (entries, slot)
        cur = self._bufs.get(key)
        if cur is None or cur[0] < nq:
            cap = max(nq, 64, 0 if cur is None else cur[0])
            if len(self._bufs) > 8:
                self._bufs.clear()
            cur = (cap, t.empty(cap * bb, dtype=t.uint8, device="cuda"),
                   t.empty(self.world * cap * bb, dtype=t.uint8, device="cuda"),
                   t.empty(self.world * cap * bb, dtype=t.uint8, pin_memory=True))
            self._bufs[key] = cur
        _, mine, allb, host = cur
        return mine[: nq * bb], allb[: self.world * nq * bb], host[: self.world * nq * bb]

    def _scan(self, q: np.ndarray, k: int, mp, entries: int, slot: int):
        """This rank's shard: candidate blocks for the queries of one group, left in device memory."""
        mine, _, _ = self._buffers(q.shape[0], entries, slot)
        _ffi.check(_ffi.lib().tsh_search_shard(self.index._h, q.ctypes.data_as(_ffi.p_f32), q.shape[0], int(k), mp,
                                               entries, ctypes.c_void_p(mine.data_ptr()), None))

    def _exchange_merge(self, q: np.ndarray, k: int, thr, entries: int, slot: int, mine=None):
        t = self._torch
        own, allb, host = self._buffers(q.shape[0], entries, slot)
        mine = own if mine is None else mine
        if self._dist.is_initialized() and self._dist.get_backend(self.group) == "gloo":
            # test setups (several ranks on one GPU): exchange through host memory
            mine_h = mine.cpu()
            self._dist.all_gather_into_tensor(host, mine_h, group=self.group)
            return merge_candidate_blocks(self.index.metric, self.index.dim, q, k, thr, host.numpy(),
                                          self.world, entries)
        if self._dist.is_initialized():
            self._dist.all_gather_into_tensor(allb, mine, group=self.group)
        else:
            allb = mine
        host[: allb.numel()].copy_(allb, non_blocking=True)
        t.cuda.current_stream().synchronize()
        return merge_candidate_blocks(self.index.metric, self.index.dim, q, k, thr, host[: allb.numel()].numpy(),
                                      self.world, entries)

    def search_many(self, queries, k: int, distance_threshold: Optional[float] = None, row_mask=None,
                    group: int = 8):
        """A stream of independent queries, exchanged in groups: this rank's scans of ALL the queries run as one
        pipeline on a library thread (tsh_search_shard_begin: the GPU sees no group boundaries), and the calling
        thread all-gathers and merges every group as soon as its blocks are final (collectives stay on one thread,
        in one order on every rank).  Same results as search()."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, kk = q.shape[0], max(int(k), 0)
        ids = np.full((nq, kk), -1, dtype=np.int64)
        dist = np.full((nq, kk), np.nan, dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        L = _ffi.lib()
        entries = L.tsh_default_block_entries(int(k))
        row_mask, mp = self.index.mask_arg(row_mask)  # GLOBAL mask: one bit per row id below this shard's end
        group = max(1, int(group))
        spans = [(s, min(nq, s + group)) for s in range(0, nq, group)]
        if not spans:
            return ids, dist, cnt
        bb = L.tsh_candidate_block_bytes(entries)
        key = ("many", entries)
        cur = self._bufs.get(key)
        if cur is None or cur[0] < nq:
            cur = (max(nq, 64), self._torch.empty(max(nq, 64) * bb, dtype=self._torch.uint8, device="cuda"))
            self._bufs[key] = cur
            self._torch.cuda.current_stream().synchronize()
        mine_all = cur[1]
        st = ctypes.c_void_p()
        _ffi.check(L.tsh_search_shard_begin(self.index._h, q.ctypes.data_as(_ffi.p_f32), nq, int(k), mp, entries,
                                            ctypes.c_void_p(mine_all.data_ptr()), group, ctypes.byref(st)))
        try:
            for g, (lo, hi) in enumerate(spans):
                _ffi.check(L.tsh_search_shard_progress(st, hi, None))  # this group's blocks are final
                try:
                    i, d, c = self._exchange_merge(q[lo:hi], k, distance_threshold, entries, g % 2,
                                                   mine_all[lo * bb: hi * bb])
                except _ffi.TshError as e:
                    if e.code != _ffi.TSH_E_OVERFLOW:
                        raise
                    _ffi.check(L.tsh_search_shard_progress(st, nq, None))  # nothing overlaps a retry
                    i, d, c = self.search(q[lo:hi], k, distance_threshold, row_mask)  # ties: bigger blocks
                ids[lo:hi], dist[lo:hi], cnt[lo:hi] = i, d, c
        finally:
            rc = L.tsh_search_shard_end(st)
        _ffi.check(rc)
        return ids, dist, cnt

    def search(self, queries, k: int, distance_threshold: Optional[float] = None, row_mask=None):
        t = self._torch
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        L = _ffi.lib()
        entries = L.tsh_default_block_entries(int(k))
        row_mask, mp = self.index.mask_arg(row_mask)  # GLOBAL mask: one bit per row id below this shard's end
        for _attempt in range(3):
            self._scan(q, k, mp, entries, 0)
            try:
                return self._exchange_merge(q, k, distance_threshold, entries, 0)
            except _ffi.TshError as e:
                if e.code != _ffi.TSH_E_OVERFLOW:
                    raise
                entries = int(e.needed_entries)  # identical on every rank: all saw the same headers
        raise RuntimeError("candidate blocks kept overflowing")


class CommSearcher:
    """The same exchange through the library's OWN entry points (include/tostore_hip.h: tsh_comm_*,
    tsh_search_sharded) -- what a host without torch (one Dart process per GPU) calls.  The library pipelines the
    call itself (groups of queries: a helper thread scans group g + 1 while group g is all-gathered and merged, each
    rank merging its slice of the group's queries).

    RCCL transport: rank 0 makes the 128-byte id with `CommSearcher.unique_id()`, the host ships it to the other
    ranks over whatever channel it has, every rank constructs its searcher (collective) and then calls `search`
    with the same queries (collective).
    Host transport (`allgather=`): a callable `(send: np.ndarray[uint8], recv: np.ndarray[uint8]) -> None` that
    places every rank's bytes at recv[rank * len(send):]; `CommSearcher.over_torch(...)` builds one from a
    torch.distributed group (gloo: several ranks may share one GPU)."""

    python_groups = False  # search_many: the library forms the groups itself

    def __init__(self, shard_index, world: int, rank: int, unique_id: Optional[bytes] = None, device: int = -1,
                 allgather=None):
        self.index = shard_index
        self.world = world
        self.rank = rank
        self._c = ctypes.c_void_p()
        self._cb = None
        self.callback_error = None
        if allgather is not None:
            def _cb(_user, send, recv, nbytes):
                try:
                    src = np.ctypeslib.as_array(ctypes.cast(send, _ffi.p_u8), shape=(nbytes,))
                    dst = np.ctypeslib.as_array(ctypes.cast(recv, _ffi.p_u8), shape=(nbytes * world,))
                    allgather(src, dst)
                    return 0
                except Exception as e:  # noqa: BLE001 -- must not unwind through the C frames
                    self.callback_error = e
                    return -1
            self._cb = _ffi.ALLGATHER_FN(_cb)  # kept alive as long as the communicator
            _ffi.check(_ffi.lib().tsh_comm_create_host(world, rank, device, ctypes.cast(self._cb, ctypes.c_void_p),
                                                       None, ctypes.byref(self._c)))
            return
        if unique_id is None or len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of tsh_comm_unique_id")
        buf = ctypes.create_string_buffer(unique_id, 128)
        _ffi.check(_ffi.lib().tsh_comm_create(buf, world, rank, device, ctypes.byref(self._c)))

    @classmethod
    def over_torch(cls, shard_index, group=None, device: int = -1):
        """Host transport over a torch.distributed process group (CPU tensors: gloo)."""
        import torch
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def allgather(send, recv):
            dist.all_gather_into_tensor(torch.from_numpy(recv), torch.from_numpy(send), group=group)

        return cls(shard_index, world, rank, None, device, allgather=allgather)

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _ffi.check(_ffi.lib().tsh_comm_unique_id(buf))
        return buf.raw

    def set_group(self, queries_per_exchange: int) -> None:
        """Queries per exchange (0 = by the size of the call).  Same value on every rank."""
        _ffi.check(_ffi.lib().tsh_comm_set_group(self._c, int(queries_per_exchange)))

    def timeline(self, reset: bool = False) -> dict:
        """Where this rank's tsh_search_sharded time went (tsh_comm_timeline, include/tostore_hip.h): sums of
        microseconds per phase since the communicator was made or last reset."""
        t = _ffi.TshCommTimeline()
        _ffi.check(_ffi.lib().tsh_comm_get_timeline(self._c, ctypes.byref(t), 1 if reset else 0))
        d = {k: getattr(t, k) for k, _ in t._fields_ if k != "reserved"}
        d["transport"] = {0: "rccl", 1: "host callback", 2: "TSH_RCCL_LIB"}.get(d["transport"], d["transport"])
        return d

    def close(self) -> None:
        if self._c:
            _ffi.lib().tsh_comm_destroy(self._c)
            self._c = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def search(self, queries, k: int, distance_threshold: Optional[float] = None, row_mask=None, shard=...):
        """Collective.  `shard=None` (tests) passes a NULL handle: this rank fails locally and stays in the
        collective."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, kk = q.shape[0], max(int(k), 0)
        ids = np.empty((nq, max(kk, 1)), dtype=np.int64)
        dist = np.empty((nq, max(kk, 1)), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        thr = math.nan if distance_threshold is None else float(distance_threshold)
        row_mask, mp = self.index.mask_arg(row_mask)
        h = self.index._h if shard is ... else shard
        _ffi.check(_ffi.lib().tsh_search_sharded(h, self._c, q.ctypes.data_as(_ffi.p_f32), nq, int(k), thr, mp,
                                                 ids.ctypes.data_as(_ffi.p_i64), dist.ctypes.data_as(_ffi.p_f64),
                                                 cnt.ctypes.data_as(_ffi.p_i32)))
        return ids[:, :kk], dist[:, :kk], cnt

    def search_many(self, queries, k: int, distance_threshold: Optional[float] = None, row_mask=None, group: int = 0):
        """One collective call; the library forms the groups itself (same signature as
        ShardedSearcher.search_many; `group` > 0 overrides the library's choice)."""
        if group:
            self.set_group(group)
        try:
            return self.search(queries, k, distance_threshold, row_mask)
        finally:
            if group:
                self.set_group(0)
