"""Host-side mirror of the seam the HIP path plugs into.

`HipVectorIndex` wraps one `tsh_index*` handle (the device-resident copy of an
index's raw-vector column).  `HipVectorBackend.search` has the signature and
result shape of `NghGraphEngine.search`
(/root/reference/lib/src/core/ngh_graph_engine.dart:67-135; result class
:26-40), so `VectorIndexManager` can call it where the reference calls the
graph engine (/root/reference/lib/src/core/vector_index_manager.dart:538-548).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import METRIC_COSINE, METRIC_IP, METRIC_L2  # noqa: F401


@dataclass
class NghSearchResult:
    """ref: core/ngh_graph_engine.dart:26-40 (nodeId, distance, primaryKey)."""
    nodeId: int
    distance: float
    primaryKey: Optional[str] = None


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class HipMask:
    """A WHERE row set that lives on the device (`tsh_mask*`, include/tostore_hip.h): the keep bitmap is uploaded once
    and -- when it is selective -- compacted into the list of kept row ids on the device; searches that pass the
    handle (`HipVectorIndex.search(..., row_mask=handle)`) do no host work on the mask.  The natural row sets are the
    ones that serve many queries in the reference: a WHERE's primary keys mapped through the pk -> nodeId tree
    (/root/reference/lib/src/core/vector_index_manager.dart:1223-1378) and the complement of the tombstones
    (/root/reference/lib/src/core/ngh_page.dart:105-108).  Rows appended after the mask was made are not kept; rows
    deleted later are dropped as always.  Destroy before the index (close / context manager)."""

    def __init__(self, index: "HipVectorIndex", bits):
        bits = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
        self._h = ctypes.c_void_p()
        self.index = index
        _ffi.check(_ffi.lib().tsh_mask_create(index._h, bits.ctypes.data_as(_ffi.p_u8), bits.shape[0],
                                              ctypes.byref(self._h)))

    @property
    def kept(self) -> int:
        n = _ffi.lib().tsh_mask_kept(self._h)
        if n < 0:
            _ffi.check(int(n))
        return int(n)

    def close(self) -> None:
        if self._h:
            _ffi.lib().tsh_mask_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipVectorIndex:
    """Owner of one `tsh_index*`.  Destroy exactly once (close / context manager)."""

    def __init__(self, dim: int, metric: int, capacity_rows: int = 0, n_devices: int = 1,
                 *, shard_device: Optional[int] = None, row_base: int = 0):
        self._h = ctypes.c_void_p()
        L = _ffi.lib()
        if shard_device is None:
            _ffi.check(L.tsh_index_create(dim, metric, capacity_rows, n_devices, ctypes.byref(self._h)))
        else:
            _ffi.check(L.tsh_index_create_shard(dim, metric, capacity_rows, shard_device, row_base,
                                                ctypes.byref(self._h)))
        self.dim, self.metric, self.row_base = dim, metric, row_base

    @classmethod
    def open_ngh(cls, ngh_dir: str, max_entries_per_dir: int = 500, n_devices: int = 1):
        """Cold start from `<index>/ngh` (meta.json + rawvec + graph tombstones) as the reference lays it
        out (/root/reference/lib/src/core/path_manager.dart:275-324).  Returns (index, info dict)."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        info = _ffi.TshNghInfo()
        _ffi.check(_ffi.lib().tsh_index_open_ngh(str(ngh_dir).encode(), max_entries_per_dir, n_devices,
                                                 ctypes.byref(self._h), ctypes.byref(info)))
        self.dim, self.metric, self.row_base = info.dimensions, info.metric, 0
        return self, {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}

    @classmethod
    def open_ngh_shard(cls, ngh_dir: str, world: int, rank: int, device: int = -1, max_entries_per_dir: int = 500):
        """Cold start of ONE rank's row range [rank * ceil(N / world), ...) of `<index>/ngh` as a shard handle
        (global ids): only the partition files / pages that hold the range are read
        (/root/reference/lib/src/model/ngh_index_meta.dart:480-490).  Returns (index, info dict)."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        info = _ffi.TshNghInfo()
        _ffi.check(_ffi.lib().tsh_index_open_ngh_shard(str(ngh_dir).encode(), max_entries_per_dir, device, world, rank,
                                                       ctypes.byref(self._h), ctypes.byref(info)))
        self.dim, self.metric, self.row_base = info.dimensions, info.metric, info.row_base
        return self, {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}

    def make_mask(self, bits) -> "HipMask":
        """A device-resident row set for many searches (bit i of `bits`, LSB first, keeps GLOBAL row id i)."""
        return HipMask(self, bits)

    # -- lifetime -----------------------------------------------------------
    def close(self) -> None:
        if self._h:
            _ffi.lib().tsh_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data feed (mirrors _writeRawVector / deleteBatch) --------------------
    def append(self, first_row_id: int, rows) -> None:
        rows = _f32c(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be n x {self.dim}")
        _ffi.check(_ffi.lib().tsh_index_append(self._h, first_row_id, rows.shape[0],
                                               rows.ctypes.data_as(_ffi.p_f32)))

    def append_device(self, first_row_id: int, n_rows: int, device_ptr: int) -> None:
        _ffi.check(_ffi.lib().tsh_index_append_device(self._h, first_row_id, n_rows,
                                                      ctypes.c_void_p(device_ptr)))

    def set_deleted(self, ids: Sequence[int]) -> None:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        _ffi.check(_ffi.lib().tsh_index_set_deleted(self._h, ids.ctypes.data_as(_ffi.p_i64), ids.shape[0]))

    def load_rawvec_file(self, path: str, page_size: int, precision: int, first_row_id: int,
                         max_rows: int) -> int:
        out = ctypes.c_int64(0)
        _ffi.check(_ffi.lib().tsh_index_load_rawvec_file(self._h, path.encode(), page_size, precision,
                                                         first_row_id, max_rows, ctypes.byref(out)))
        return out.value

    def pq_encode(self, first_row_id: int, n_rows: int, codebook, subspaces: int, centroids: int = 256) -> np.ndarray:
        """PQ codes (n_rows x subspaces, uint8) of resident rows; mirrors batchPqEncode
        (/root/reference/lib/src/core/compute_tasks.dart:2292-2326)."""
        cb = _f32c(codebook).reshape(-1)
        sub_dim = self.dim // subspaces
        if cb.shape[0] != subspaces * centroids * sub_dim:
            raise ValueError("codebook must hold subspaces*centroids*(dim//subspaces) floats")
        codes = np.empty((n_rows, subspaces), dtype=np.uint8)
        _ffi.check(_ffi.lib().tsh_index_pq_encode(self._h, first_row_id, n_rows, cb.ctypes.data_as(_ffi.p_f32),
                                                  subspaces, centroids, codes.ctypes.data_as(_ffi.p_u8)))
        return codes

    @staticmethod
    def pq_train(samples, subspaces: int, init_index, centroids: int = 256, iterations: int = 10,
                 device: int = 0) -> np.ndarray:
        """Codebook (subspaces x centroids x subDim, float32) from host samples; mirrors the per-sub-space
        trainPqSubspace tasks (/root/reference/lib/src/core/compute_tasks.dart:2135-2266) given the
        initial sample indices the reference draws from Random(42 + m)."""
        s = _f32c(samples)
        if s.ndim != 2:
            raise ValueError("samples must be n x dim")
        n, dim = s.shape
        init = np.ascontiguousarray(init_index, dtype=np.int32)
        if init.size != subspaces * centroids:
            raise ValueError("init_index must hold subspaces*centroids indices")
        out = np.empty((subspaces, centroids, dim // subspaces), dtype=np.float32)
        _ffi.check(_ffi.lib().tsh_pq_train(device, s.ctypes.data_as(_ffi.p_f32), n, dim, subspaces, centroids,
                                           iterations, init.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                           out.ctypes.data_as(_ffi.p_f32)))
        return out

    @property
    def size(self) -> int:
        return _ffi.lib().tsh_index_size(self._h)

    def counters(self) -> dict:
        c = _ffi.TshCounters()
        _ffi.check(_ffi.lib().tsh_get_counters(self._h, ctypes.byref(c)))
        return {k: getattr(c, k) for k, _ in c._fields_}

    # -- search ---------------------------------------------------------------
    def mask_arg(self, row_mask):
        """(array kept alive by the caller, pointer) of a keep mask for the C ABI.  The ABI carries no mask
        length: the library reads ceil(size / 8) bytes (bit i = GLOBAL row id i), so a short buffer is
        rejected here, for every entry point that takes a mask."""
        if row_mask is None:
            return None, None
        row_mask = np.ascontiguousarray(row_mask, dtype=np.uint8).reshape(-1)
        need = (self.size + 7) // 8
        if row_mask.shape[0] < need:
            raise ValueError(f"row_mask needs {need} bytes (one bit per row id below {self.size}), got {row_mask.shape[0]}")
        return row_mask, row_mask.ctypes.data_as(_ffi.p_u8)

    def search(self, queries, k: int, distance_threshold: Optional[float] = None, row_mask=None):
        """Raw C-ABI search: (ids[nq,k], dist[nq,k], count[nq])."""
        q = _f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError(f"queries must be nq x {self.dim}")
        nq, kk = q.shape[0], max(int(k), 0)
        # (the library fills the slots past a query's count with -1 / NaN itself: pre-filling 1.6 MB here cost a
        # 1024-query call 0.3 ms)
        ids = np.empty((nq, max(kk, 1)), dtype=np.int64)
        dist = np.empty((nq, max(kk, 1)), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        thr = math.nan if distance_threshold is None else float(distance_threshold)
        if isinstance(row_mask, HipMask):  # a mask handle: resident on the device, nothing to prepare
            _ffi.check(_ffi.lib().tsh_search_masked(self._h, q.ctypes.data_as(_ffi.p_f32), nq, int(k), thr, row_mask._h,
                                                    ids.ctypes.data_as(_ffi.p_i64), dist.ctypes.data_as(_ffi.p_f64),
                                                    cnt.ctypes.data_as(_ffi.p_i32)))
            return ids[:, :kk], dist[:, :kk], cnt
        row_mask, mp = self.mask_arg(row_mask)
        _ffi.check(_ffi.lib().tsh_search(self._h, q.ctypes.data_as(_ffi.p_f32), nq, int(k), thr, mp,
                                         ids.ctypes.data_as(_ffi.p_i64), dist.ctypes.data_as(_ffi.p_f64),
                                         cnt.ctypes.data_as(_ffi.p_i32)))
        return ids[:, :kk], dist[:, :kk], cnt

    # -- asynchronous single-query form (several queries in flight) -----------------
    def submit(self, query, k: int, row_mask=None) -> tuple:
        q = _f32c(query).reshape(-1)
        if q.shape[0] != self.dim:
            raise ValueError(f"query must have {self.dim} elements")
        t = ctypes.c_int32(-1)
        if isinstance(row_mask, HipMask):
            _ffi.check(_ffi.lib().tsh_search_submit_masked(self._h, q.ctypes.data_as(_ffi.p_f32), int(k), row_mask._h,
                                                           ctypes.byref(t)))
            return (t.value, int(k))
        row_mask, mp = self.mask_arg(row_mask)
        _ffi.check(_ffi.lib().tsh_search_submit(self._h, q.ctypes.data_as(_ffi.p_f32), int(k), mp,
                                                ctypes.byref(t)))
        return (t.value, int(k))

    def ready(self, ticket: tuple) -> bool:
        """True once the ticket's kernels have finished (wait() will not block on the GPU).  Never blocks."""
        rc = _ffi.lib().tsh_search_ready(self._h, ticket[0])
        if rc < 0:
            _ffi.check(rc)
        return rc == 1

    def wait(self, ticket: tuple, distance_threshold: Optional[float] = None):
        t, k = ticket
        ids = np.empty(k, dtype=np.int64)
        dist = np.empty(k, dtype=np.float64)
        cnt = ctypes.c_int32(0)
        thr = math.nan if distance_threshold is None else float(distance_threshold)
        _ffi.check(_ffi.lib().tsh_search_wait(self._h, t, thr, ids.ctypes.data_as(_ffi.p_i64),
                                              dist.ctypes.data_as(_ffi.p_f64), ctypes.byref(cnt)))
        return ids[:cnt.value], dist[:cnt.value]

    def set_batch_min_nq(self, nq: int) -> None:
        """When search() uses the batched matrix-core path: 0 never, 1 by estimated cost (default), n >= 2 from n
        queries per call on.  Results are identical."""
        _ffi.check(_ffi.lib().tsh_index_set_option(self._h, _ffi.TSH_OPT_BATCH_MIN_NQ, int(nq)))

    def set_exact_scan_rows(self, rows: int) -> None:
        """Single-query searches with at most `rows` rows to look at (a selective mask's kept rows, a small index) take
        the exact sums of all of them and select among the exact distances (two dispatches, no f32 pre-filter):
        0 never, default and maximum 16384.  Results are identical."""
        _ffi.check(_ffi.lib().tsh_index_set_option(self._h, _ffi.TSH_OPT_EXACT_SCAN_ROWS, int(rows)))

    def set_exact_select(self, wide: bool) -> None:
        """What follows the exact scan of a short search: True (default) the wide pick (one workgroup per 256 rows, the
        cut from the scan's key histogram, k rows plus the cut bin's few others), False the one-workgroup select that
        ranks exactly k rows.  Results are identical."""
        _ffi.check(_ffi.lib().tsh_index_set_option(self._h, _ffi.TSH_OPT_EXACT_SELECT, 1 if wide else 0))

    def set_batch_hub(self, on: bool) -> None:
        """Batched searches (fp16 keys, L2 / inner product) also bound the k-th key by the index's hub rows -- the 4096
        shortest (L2) / longest (inner product) -- beside the sample's estimate (default off: it pays for itself on no corpus measured).  Results are identical."""
        _ffi.check(_ffi.lib().tsh_index_set_option(self._h, _ffi.TSH_OPT_BATCH_HUB, 1 if on else 0))

    def set_batch_group(self, on: bool) -> None:
        """The fp16 copy of an L2 / inner-product index holds its rows by norm inside blocks of 8192 (default on) or in
        row order.  Results are identical; the copy is rebuilt by the next batched search."""
        _ffi.check(_ffi.lib().tsh_index_set_option(self._h, _ffi.TSH_OPT_BATCH_GROUP, 1 if on else 0))

    def set_batch_kernel(self, kind: int) -> None:
        """Batched pre-filter keys: 0 f32 MFMA, 1 bf16x3, 2 f16, 3 auto (default: f16 for cosine, bf16x3
        otherwise).  Results are identical."""
        _ffi.check(_ffi.lib().tsh_index_set_option(self._h, _ffi.TSH_OPT_BATCH_KERNEL, int(kind)))

    def bench_batch(self, queries, k: int, iters: int = 3):
        """(avg microseconds of the matrix-core passes, algorithmic flops) for one batch."""
        q = _f32c(queries)
        us, fl = ctypes.c_double(0), ctypes.c_double(0)
        _ffi.check(_ffi.lib().tsh_bench_batch(self._h, q.ctypes.data_as(_ffi.p_f32), q.shape[0], int(k), iters,
                                              ctypes.byref(us), ctypes.byref(fl)))
        return us.value, fl.value

    def probe_scan_keys(self, query):
        """(f32 key of every row from the scan kernel, eps_rel, delta_abs): tests of the error model."""
        q = _f32c(query).reshape(-1)
        keys = np.empty(self.size - self.row_base, dtype=np.float32)
        er, da = ctypes.c_float(0), ctypes.c_float(0)
        _ffi.check(_ffi.lib().tsh_probe_scan_keys(self._h, q.ctypes.data_as(_ffi.p_f32), keys.ctypes.data_as(_ffi.p_f32),
                                                  ctypes.byref(er), ctypes.byref(da)))
        return keys, er.value, da.value

    def probe_batch_keys(self, queries, k: int):
        """(keys[nq, rows] of the batched key kernel in use, delta2[nq]): tests of the error model."""
        q = _f32c(queries)
        keys = np.empty((q.shape[0], self.size - self.row_base), dtype=np.float32)
        d2 = np.empty(q.shape[0], dtype=np.float32)
        _ffi.check(_ffi.lib().tsh_probe_batch_keys(self._h, q.ctypes.data_as(_ffi.p_f32), q.shape[0], int(k),
                                                   keys.ctypes.data_as(_ffi.p_f32), d2.ctypes.data_as(_ffi.p_f32)))
        return keys, d2

    def probe_batch_row_band(self, nq: int):
        """(alpha2[nq], beta2[nq]) of the last probe_batch_keys call: 2 |key - exact| <= alpha2 |v| + beta2 per row."""
        a2 = np.empty(nq, dtype=np.float32)
        b2 = np.empty(nq, dtype=np.float32)
        _ffi.check(_ffi.lib().tsh_probe_batch_row_band(self._h, int(nq), a2.ctypes.data_as(_ffi.p_f32),
                                                       b2.ctypes.data_as(_ffi.p_f32)))
        return a2, b2

    def bench_scan(self, query, iters: int = 20, row_mask=None) -> float:
        q = _f32c(query)
        out = ctypes.c_double(0)
        row_mask, mp = self.mask_arg(row_mask)
        _ffi.check(_ffi.lib().tsh_bench_scan(self._h, q.ctypes.data_as(_ffi.p_f32), iters, mp,
                                             ctypes.byref(out)))
        return out.value


class HipVectorBackend:
    """Drop-in for `NghGraphEngine.search` backed by a `HipVectorIndex`.

    ref: core/ngh_graph_engine.dart:67-135.  The scan is exhaustive (ef -> infinity), so by default `efSearch`
    changes nothing and the k=100 / default-ef trap of :80-82,168 (at most 64 rows returned) does not exist here.
    `honourEfCap=True` (the hook's `hipHonourEfCap`, tostore_amd/dart/hip_vector_hook.dart) brings the reference's
    row COUNT back for callers that depend on it: the answer is cut to `min(topK, ef)` rows with
    `ef = min(efSearch ?? metaEfSearch, max(5 topK, 32))` (:80-82; the result heap holds ef entries, :168), and an
    index whose graph has no entry point (`medoidNodeId < 0`, :78) answers nothing.  The rows are still the exact
    nearest ones -- which the reference's own ef rows need not be.
    """

    def __init__(self, index: HipVectorIndex, *, honourEfCap: bool = False, metaEfSearch: int = 64,
                 medoidNodeId: int = 0):
        self.index = index
        self.honourEfCap = honourEfCap
        self.metaEfSearch = metaEfSearch  # NghIndexMeta.efSearch (model/ngh_index_meta.dart:196: default 64)
        self.medoidNodeId = medoidNodeId  # NghIndexMeta.medoidNodeId (-1 until the first insert picked one)

    def search(self, *, query, topK: int, efSearch: Optional[int] = None,
               distanceThreshold: Optional[float] = None, rowMask=None) -> list:
        # ref: :78  `if (meta.totalVectors == 0 || meta.medoidNodeId < 0) return const []`
        if self.index.size == 0 or topK <= 0:
            return []
        if self.honourEfCap:
            if self.medoidNodeId < 0:
                return []
            ef = min(self.metaEfSearch if efSearch is None else efSearch, max(topK * 5, 32))  # :80-82
            topK = min(topK, max(ef, 0))
            if topK <= 0:
                return []
        ids, dist, cnt = self.index.search(query, topK, distanceThreshold, rowMask)
        n = int(cnt[0])
        return [NghSearchResult(nodeId=int(ids[0, i]), distance=float(dist[0, i])) for i in range(n)]
