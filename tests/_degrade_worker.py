"""Worker of tests/test_gpu_degrade.py: a batched search whose device allocations fail (TSH_TEST_FAIL_ALLOC_OVER,
read once per process -- hence a process of its own) must still answer, bit-exact, and say in tsh_counters which
path ran.  argv: expect = "planes" | "scans" """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from tostore_amd import HipVectorIndex, _ffi  # noqa: E402

_ffi.enable_test_hooks()  # TSH_TEST_FAIL_ALLOC_OVER is obeyed only in a process that asked for the test hooks

expect = sys.argv[1]
rng = np.random.default_rng(5)
n, d, k, nq = 50_000, 128, 20, 64
ok = True
for metric in (0, 1, 2):
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    if metric == 2:
        qs = np.stack([oracle.normalize_f32(q) for q in qs])
    mask = np.packbits(rng.random(n) < 0.5, bitorder="little")
    with HipVectorIndex(d, metric, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(2)  # batched path from two queries on
        for m, thr in ((None, None), (mask, None)):
            ids, dist, cnt = idx.search(qs, k, thr, m)
            ref = oracle.search_heap_many_mt(rows, qs, metric, k, thr, m)
            same = bool(np.array_equal(cnt, ref[2]) and np.array_equal(ids, ref[0])
                        and np.array_equal(dist.view(np.uint64), ref[1].view(np.uint64)))
            ok &= same
        c = idx.counters()
        if expect == "planes":
            good = c["batch_plane_fallbacks"] == 2 and c["batch_scan_fallbacks"] == 0 and c["batch_kernel_last"] == 0 \
                and c["batch_launches"] == 2
        else:
            good = c["batch_scan_fallbacks"] == 2 and c["batch_launches"] == 0 and c["scan_launches"] >= 2 * nq
        ok &= good
        print("metric %d: results %s, counters %s: %s" % (metric, "ok" if same else "MISMATCH", "ok" if good else "WRONG",
                                                          {x: c[x] for x in ("batch_plane_fallbacks", "batch_scan_fallbacks",
                                                                             "batch_kernel_last", "batch_launches",
                                                                             "scan_launches")}))
sys.exit(0 if ok else 1)
