"""ctypes loader for the C restatement (oracle/vs_oracle.c).

TEST INFRASTRUCTURE ONLY: import this package from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package (tostore_amd/).  PARITY UNPINNED, see vs_oracle.h.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
L2, IP, COSINE = 0, 1, 2

_c_i64 = ctypes.c_int64
_c_f32p = ctypes.POINTER(ctypes.c_float)
_c_f64p = ctypes.POINTER(ctypes.c_double)
_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> None:
    """Compile the restatement with gcc (see oracle/Makefile)."""
    so = os.path.join(_HERE, "libvs_oracle.so")
    src = os.path.join(_HERE, "vs_oracle.c")
    if (not force and os.path.exists(so)
            and os.path.getmtime(so) >= os.path.getmtime(src)
            and os.path.exists(os.path.join(_HERE, "libvs_oracle_mt.so"))
            and os.path.getmtime(os.path.join(_HERE, "libvs_oracle_mt.so"))
            >= os.path.getmtime(os.path.join(_HERE, "vs_oracle_mt.c"))
            and os.path.exists(os.path.join(_HERE, "libngh_ann.so"))
            and os.path.getmtime(os.path.join(_HERE, "libngh_ann.so"))
            >= os.path.getmtime(os.path.join(_HERE, "ngh_ann.c"))):
        return
    subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None
_lib_mt = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "libvs_oracle.so"))
        L.vso_to_float32.argtypes = [_c_f64p, _c_i64, ctypes.c_int, _c_f32p]
        L.vso_to_float32.restype = None
        L.vso_normalize_f32.argtypes = [_c_f32p, ctypes.c_int, _c_f32p]
        L.vso_normalize_f32.restype = None
        for name in ("vso_l2_distance", "vso_inner_product", "vso_cosine_similarity"):
            f = getattr(L, name)
            f.argtypes = [_c_f32p, _c_f32p, ctypes.c_int]
            f.restype = ctypes.c_double
        L.vso_exact_distance.argtypes = [_c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int]
        L.vso_exact_distance.restype = ctypes.c_double
        L.vso_exact_sums.argtypes = [_c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, _c_f64p, _c_f64p]
        L.vso_exact_sums.restype = None
        L.vso_distance_to_score.argtypes = [ctypes.c_double, ctypes.c_int]
        L.vso_distance_to_score.restype = ctypes.c_double
        L.vso_compare_double.argtypes = [ctypes.c_double, ctypes.c_double]
        L.vso_compare_double.restype = ctypes.c_int
        for name in ("vso_search_exhaustive", "vso_search_heap"):
            f = getattr(L, name)
            f.argtypes = [_c_f32p, _c_i64, ctypes.c_int, ctypes.c_int, _c_f32p, _c_i64,
                          ctypes.c_double, _c_u8p, _c_i64p, _c_f64p]
            f.restype = _c_i64
        L.vso_all_distances.argtypes = [_c_f32p, _c_i64, ctypes.c_int, ctypes.c_int,
                                        _c_f32p, _c_f64p]
        L.vso_all_distances.restype = None
        L.vso_pq_encode.argtypes = [_c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f32p, _c_i64, ctypes.c_int,
                                    _c_u8p]
        L.vso_pq_encode.restype = None
        L.vso_pq_train_subspace.argtypes = [_c_f32p, _c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_int32), _c_f32p]
        L.vso_pq_train_subspace.restype = None
        L.vso_pq_train_subspace_pp.argtypes = [_c_f32p, _c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int32,
                                               _c_f64p, _c_f32p]
        L.vso_pq_train_subspace_pp.restype = None
        L.vso_crc32.argtypes = [_c_u8p, ctypes.c_size_t]
        L.vso_crc32.restype = ctypes.c_uint32
        L.vso_vectors_per_raw_page.argtypes = [ctypes.c_int] * 3
        L.vso_vectors_per_raw_page.restype = ctypes.c_int
        L.vso_rawvec_page_build.argtypes = [_c_f32p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, _c_u8p]
        L.vso_rawvec_page_build.restype = ctypes.c_int
        L.vso_rawvec_page_parse.argtypes = [_c_u8p, ctypes.c_int, ctypes.c_int, _c_f32p,
                                            ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.vso_rawvec_page_parse.restype = ctypes.c_int
        L.vso_ngh_meta_page_build.argtypes = [ctypes.c_int, ctypes.c_int, _c_i64, _c_i64,
                                              ctypes.c_int, _c_u8p]
        L.vso_ngh_meta_page_build.restype = ctypes.c_int
        L.vso_rawvec_locate.argtypes = [_c_i64, ctypes.c_int, _c_i64, _c_i64p, _c_i64p,
                                        ctypes.POINTER(ctypes.c_int)]
        L.vso_rawvec_locate.restype = None
        _lib = L
    return _lib


def lib_mt() -> ctypes.CDLL:
    global _lib_mt
    if _lib_mt is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "libvs_oracle_mt.so"))
        L.vso_mt_max_threads.restype = ctypes.c_int
        L.vso_search_heap_mt.argtypes = [_c_f32p, _c_i64, ctypes.c_int, ctypes.c_int,
                                         _c_f32p, _c_i64, ctypes.c_double, _c_u8p,
                                         ctypes.c_int, _c_i64p, _c_f64p]
        L.vso_search_heap_mt.restype = _c_i64
        L.vso_search_heap_many_mt.argtypes = [_c_f32p, _c_i64, ctypes.c_int, ctypes.c_int, _c_f32p, _c_i64, _c_i64,
                                              ctypes.c_double, _c_u8p, ctypes.c_int, _c_i64p, _c_f64p, _c_i64p]
        L.vso_search_heap_many_mt.restype = _c_i64
        _lib_mt = L
    return _lib_mt


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


def to_float32(values, dim: int) -> np.ndarray:
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty(dim, dtype=np.float32)
    lib().vso_to_float32(_p(v, _c_f64p), v.shape[0], dim, _p(out, _c_f32p))
    return out


def normalize_f32(v) -> np.ndarray:
    v = _f32(v)
    out = np.empty_like(v)
    lib().vso_normalize_f32(_p(v, _c_f32p), v.shape[0], _p(out, _c_f32p))
    return out


def exact_distance(a, b, metric: int) -> float:
    a, b = _f32(a), _f32(b)
    return lib().vso_exact_distance(_p(a, _c_f32p), _p(b, _c_f32p), a.shape[0], metric)


def exact_sums(a, b, metric: int):
    a, b = _f32(a), _f32(b)
    s0, s1 = ctypes.c_double(), ctypes.c_double()
    lib().vso_exact_sums(_p(a, _c_f32p), _p(b, _c_f32p), a.shape[0], metric, ctypes.byref(s0), ctypes.byref(s1))
    return s0.value, s1.value


def distance_to_score(distance: float, metric: int) -> float:
    return lib().vso_distance_to_score(float(distance), metric)


def compare_double(a: float, b: float) -> int:
    return lib().vso_compare_double(a, b)


def all_distances(query, rows, metric: int) -> np.ndarray:
    rows, query = _f32(rows), _f32(query)
    n, d = rows.shape
    out = np.empty(n, dtype=np.float64)
    lib().vso_all_distances(_p(rows, _c_f32p), n, d, metric, _p(query, _c_f32p),
                            _p(out, _c_f64p))
    return out


def _search(fn, rows, query, metric, k, threshold, keep, *extra):
    rows, query = _f32(rows), _f32(query)
    n, d = rows.shape if rows.ndim == 2 else (0, query.shape[0])
    k = int(k)
    ids = np.empty(max(k, 1), dtype=np.int64)
    dist = np.empty(max(k, 1), dtype=np.float64)
    thr = math.nan if threshold is None else float(threshold)
    kp = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        assert keep.shape[0] >= (n + 7) // 8
        kp = _p(keep, _c_u8p)
    m = fn(_p(rows, _c_f32p), n, d, metric, _p(query, _c_f32p), k, thr, kp, *extra,
           _p(ids, _c_i64p), _p(dist, _c_f64p))
    if m < 0:
        raise MemoryError("oracle allocation failed")
    return ids[:m].copy(), dist[:m].copy()


def search_exhaustive(rows, query, metric: int, k: int, threshold=None, keep=None):
    return _search(lib().vso_search_exhaustive, rows, query, metric, k, threshold, keep)


def search_heap(rows, query, metric: int, k: int, threshold=None, keep=None):
    return _search(lib().vso_search_heap, rows, query, metric, k, threshold, keep)


def search_heap_mt(rows, query, metric: int, k: int, threshold=None, keep=None, threads=0):
    return _search(lib_mt().vso_search_heap_mt, rows, query, metric, k, threshold, keep,
                   int(threads) if threads > 0 else mt_max_threads())


def search_heap_many_mt(rows, queries, metric: int, k: int, threshold=None, keep=None, threads=0):
    """Exact top-k of many queries (OpenMP over query groups); returns (ids[nq,k], dist[nq,k], count[nq])."""
    rows, queries = _f32(rows), _f32(queries)
    n, d = rows.shape
    nq = queries.shape[0]
    ids = np.full((nq, max(k, 1)), -1, dtype=np.int64)
    dist = np.full((nq, max(k, 1)), np.nan, dtype=np.float64)
    cnt = np.zeros(nq, dtype=np.int64)
    thr = math.nan if threshold is None else float(threshold)
    kp = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        kp = _p(keep, _c_u8p)
    rc = lib_mt().vso_search_heap_many_mt(_p(rows, _c_f32p), n, d, metric, _p(queries, _c_f32p), nq, int(k), thr, kp,
                                          int(threads) if threads > 0 else mt_max_threads(), _p(ids, _c_i64p), _p(dist, _c_f64p), _p(cnt, _c_i64p))
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    return ids[:, :k], dist[:, :k], cnt


def cpu_quota() -> int:
    """CPUs the container may actually use (cgroup v2 cpu.max), 0 when unlimited or unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max" and int(quota) > 0 and int(period) > 0:
            return max(1, int(quota) // int(period))
        return 0
    except (OSError, ValueError):
        pass
    try:  # cgroup v1 (-1 = unlimited)
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    return 0


def mt_max_threads() -> int:
    """Threads of the multi-threaded checker: OpenMP's default, capped by the container's CPU quota (256
    runnable threads on a 16-CPU quota spend most of their time throttled and would be reported as 256 cores)."""
    t = lib_mt().vso_mt_max_threads()
    q = cpu_quota()
    return min(t, q) if q > 0 else t


def pq_encode(codebook, subspaces: int, centroids: int, sub_dim: int, vectors) -> np.ndarray:
    cb, v = _f32(codebook).reshape(-1), _f32(vectors)
    assert cb.shape[0] == subspaces * centroids * sub_dim and v.shape[1] >= subspaces * sub_dim
    codes = np.empty((v.shape[0], subspaces), dtype=np.uint8)
    lib().vso_pq_encode(_p(cb, _c_f32p), subspaces, centroids, sub_dim, _p(v, _c_f32p), v.shape[0], v.shape[1],
                        _p(codes, _c_u8p))
    return codes


def pq_train(samples, subspaces: int, k: int, iterations: int, init_index) -> np.ndarray:
    """Codebook (subspaces x k x subDim) trained like the reference's isolate tasks, one
    trainPqSubspace per sub-space, from caller-supplied initial sample indices (subspaces x k)."""
    s = _f32(samples)
    n, dim = s.shape
    sd = dim // subspaces
    init = np.ascontiguousarray(init_index, dtype=np.int32).reshape(subspaces, k)
    out = np.empty((subspaces, k, sd), dtype=np.float32)
    for m in range(subspaces):
        sub = np.ascontiguousarray(s[:, m * sd:(m + 1) * sd])
        lib().vso_pq_train_subspace(_p(sub, _c_f32p), n, sd, k, iterations,
                                    init[m].ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _p(out[m], _c_f32p))
    return out


def pq_train_pp(samples, subspaces: int, k: int, iterations: int, first_index, draws) -> np.ndarray:
    """Codebook (subspaces x k x subDim) trained like VectorQuantizer.train -- the k-means++ trainer the reference
    uses when the first batch holds fewer than 100 vectors (ref: core/vector_quantizer.dart:81-350).  Dart's
    Random(42) is not reproducible here: `first_index[m]` stands for its nextInt(n) and `draws[m][c - 1]` for the
    nextDouble() that picks centroid c of sub-space m."""
    s = _f32(samples)
    n, dim = s.shape
    sd = dim // subspaces
    first = np.ascontiguousarray(first_index, dtype=np.int32).reshape(subspaces)
    u = np.ascontiguousarray(draws, dtype=np.float64).reshape(subspaces, max(k - 1, 0))
    out = np.empty((subspaces, k, sd), dtype=np.float32)
    for m in range(subspaces):
        sub = np.ascontiguousarray(s[:, m * sd:(m + 1) * sd])
        um = np.ascontiguousarray(u[m]) if k > 1 else np.zeros(1)
        lib().vso_pq_train_subspace_pp(_p(sub, _c_f32p), n, sd, k, iterations, int(first[m]), _p(um, _c_f64p), _p(out[m], _c_f32p))
    return out


def crc32(data: bytes) -> int:
    buf = np.frombuffer(data, dtype=np.uint8)
    return lib().vso_crc32(_p(buf, _c_u8p), buf.shape[0])


def vectors_per_raw_page(page_size: int, dims: int, bpe: int) -> int:
    return lib().vso_vectors_per_raw_page(page_size, dims, bpe)


def rawvec_page_build(vectors, precision: int, page_size: int) -> bytes:
    v = _f32(vectors)
    count, dims = v.shape
    out = np.zeros(page_size, dtype=np.uint8)
    rc = lib().vso_rawvec_page_build(_p(v, _c_f32p), count, dims, precision, page_size,
                                     _p(out, _c_u8p))
    if rc != 0:
        raise ValueError("page overflow")
    return out.tobytes()


def rawvec_page_parse(page: bytes, dims: int, max_vectors: int):
    buf = np.frombuffer(page, dtype=np.uint8)
    out = np.zeros((max_vectors, dims), dtype=np.float32)
    prec = ctypes.c_int(-1)
    n = lib().vso_rawvec_page_parse(_p(buf, _c_u8p), buf.shape[0], dims, _p(out, _c_f32p),
                                    max_vectors, ctypes.byref(prec))
    if n < 0:
        return None
    return out[:n].copy(), prec.value


def ngh_meta_page_build(partition_no: int, category: int, total_entries: int,
                        file_size: int, page_size: int) -> bytes:
    out = np.zeros(page_size, dtype=np.uint8)
    rc = lib().vso_ngh_meta_page_build(partition_no, category, total_entries, file_size,
                                       page_size, _p(out, _c_u8p))
    assert rc == 0
    return out.tobytes()


def rawvec_locate(node_id: int, vectors_per_page: int, pages_per_partition: int):
    part, page = ctypes.c_int64(), ctypes.c_int64()
    slot = ctypes.c_int()
    lib().vso_rawvec_locate(node_id, vectors_per_page, pages_per_partition,
                            ctypes.byref(part), ctypes.byref(page), ctypes.byref(slot))
    return part.value, page.value, slot.value


# ---- N3: the reference's own ANN path, restated (oracle/ngh_ann.c) ------------------------------
_lib_ann = None


def _ann():
    global _lib_ann
    if _lib_ann is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "libngh_ann.so"))
        L.vso_ann_create.argtypes = [ctypes.c_int] * 7 + [ctypes.c_double, _c_f32p]
        L.vso_ann_create.restype = ctypes.c_void_p
        L.vso_ann_destroy.argtypes = [ctypes.c_void_p]
        L.vso_ann_destroy.restype = None
        L.vso_ann_insert_batch.argtypes = [ctypes.c_void_p, _c_f32p, _c_i64]
        L.vso_ann_insert_batch.restype = None
        L.vso_ann_search.argtypes = [ctypes.c_void_p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     _c_i64p, _c_f64p]
        L.vso_ann_search.restype = _c_i64
        L.vso_ann_size.argtypes = [ctypes.c_void_p]
        L.vso_ann_size.restype = _c_i64
        L.vso_ann_counters.argtypes = [ctypes.c_void_p, _c_i64p, _c_i64p, ctypes.c_int]
        L.vso_ann_counters.restype = None
        L.vso_ann_mean_degree.argtypes = [ctypes.c_void_p]
        L.vso_ann_mean_degree.restype = ctypes.c_double
        _lib_ann = L
    return _lib_ann


class NghAnnIndex:
    """The reference's index as it builds and searches it (NghGraphEngine + VectorQuantizer), restated on
    the CPU for context numbers only.  `first_batch` plays the first writeChanges call: its first <= 2500
    rows train the codebook (100 or more: trainPqSubspace per sub-space; fewer: VectorQuantizer.train with
    k-means++ seeding; random draws from `seed` because Dart's PRNG is not reproducible here).  ref: core/vector_index_manager.dart:300-420,725-850, core/ngh_graph_engine.dart."""

    def __init__(self, dim: int, metric: int, first_batch, *, subspaces: Optional[int] = None, max_degree: int = 64,
                 ef_search: int = 64, ef_construction: int = 128, prune_alpha: float = 1.2, seed: int = 42):
        fb = _f32(first_batch)
        self.dim, self.metric = dim, metric
        self.subspaces = subspaces or min(max(dim // 8, 8), 128)  # NghIndexMeta.autoPqSubspaces
        samples = fb[:2500]
        n = samples.shape[0]
        if n < 1:
            raise ValueError("an empty first batch trains nothing (vector_index_manager.dart:739)")
        self.centroids = min(256, n)
        rng = np.random.default_rng(seed)
        if n >= 100:  # isolate tasks, one per sub-space (vector_index_manager.dart:744-841)
            init = rng.integers(0, n, size=(self.subspaces, self.centroids)).astype(np.int32)
            self.codebook = pq_train(samples, self.subspaces, self.centroids, 10, init)
        else:  # VectorQuantizer.train: k-means++ seeding, k = n centroids (vector_index_manager.dart:842-849)
            first = rng.integers(0, n, size=self.subspaces).astype(np.int32)
            draws = rng.random((self.subspaces, max(self.centroids - 1, 0)))
            self.codebook = pq_train_pp(samples, self.subspaces, self.centroids, 10, first, draws)
        cb = np.ascontiguousarray(self.codebook.reshape(-1), dtype=np.float32)
        self._h = _ann().vso_ann_create(dim, metric, self.subspaces, self.centroids, max_degree, ef_search,
                                        ef_construction, prune_alpha, _p(cb, _c_f32p))
        self.insert_batch(fb)

    def insert_batch(self, vectors) -> None:
        v = _f32(vectors)
        _ann().vso_ann_insert_batch(self._h, _p(v, _c_f32p), v.shape[0])

    def search(self, query, k: int, ef_search: int = -1, threshold=None):
        q = np.ascontiguousarray(query, dtype=np.float32)
        ids = np.empty(max(k, 1), np.int64)
        dist = np.empty(max(k, 1), np.float64)
        n = _ann().vso_ann_search(self._h, _p(q, _c_f32p), k, ef_search, (math.nan if threshold is None else float(threshold)), _p(ids, _c_i64p),
                                  _p(dist, _c_f64p))
        return ids[:n].copy(), dist[:n].copy()

    def counters(self, reset: bool = True):
        a, h = ctypes.c_int64(), ctypes.c_int64()
        _ann().vso_ann_counters(self._h, ctypes.byref(a), ctypes.byref(h), 1 if reset else 0)
        return {"adc_evaluations": a.value, "hops": h.value}

    @property
    def size(self) -> int:
        return _ann().vso_ann_size(self._h)

    @property
    def mean_degree(self) -> float:
        return _ann().vso_ann_mean_degree(self._h)

    def close(self) -> None:
        if self._h:
            _ann().vso_ann_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
