"""Host finaliser alone (no device): tsh_merge_candidates over synthetic candidate blocks shaped like config C3's
(1024 queries, ~127 candidates each, cosine, 768 dims).  Prints microseconds per query with the library's pool
(tools/README.md).  TSH_HOST_THREADS=1 gives the single-thread cost."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tostore_amd import _ffi

L = _ffi.lib()
nq, k, dim, cands, metric = 1024, 100, int(os.environ.get("DIM", "768")), int(os.environ.get("CANDS", "127")), int(os.environ.get("METRIC", "2"))
entries = L.tsh_default_block_entries(k)
bb = L.tsh_candidate_block_bytes(entries)
rng = np.random.default_rng(1)
q = rng.standard_normal((nq, dim)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
blocks = np.zeros(nq * bb, dtype=np.uint8)
hdr = 64
for i in range(nq):
    b = blocks[i * bb:(i + 1) * bb]
    h = b[:hdr].view(np.uint32)
    h[0] = cands      # count
    h[1] = entries    # entries
    h[6] = k
    h[7] = metric
    e = b[hdr:hdr + cands * 24].view(np.int64).reshape(cands, 3)
    e[:, 0] = rng.permutation(1_000_000)[:cands]
    e[:, 1] = rng.uniform(0.05, 0.2, cands).view(np.int64)   # s0 = dot
    e[:, 2] = rng.uniform(0.99, 1.01, cands).view(np.int64)  # s1 = |v|^2
ids = np.empty((nq, k), np.int64)
dist = np.empty((nq, k), np.float64)
cnt = np.empty(nq, np.int32)
need = ctypes.c_int32(0)
def run():
    rc = L.tsh_merge_candidates(metric, dim, q.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), nq, k, float("nan"),
                                blocks.ctypes.data_as(ctypes.c_void_p), 1, entries,
                                ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                dist.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(need))
    assert rc == 0, (rc, _ffi.last_error())
run()
assert cnt.min() == k and (np.diff(dist, axis=1) >= 0).all()
best = 1e9
for _ in range(20):
    t = time.perf_counter(); run(); best = min(best, time.perf_counter() - t)
print("%d queries: %.1f us per call, %.2f us per query (wall, pool of TSH_HOST_THREADS=%s)" % (
    nq, best * 1e6, best * 1e6 / nq, os.environ.get("TSH_HOST_THREADS", "auto")))
