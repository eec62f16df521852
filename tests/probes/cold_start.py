#!/usr/bin/env python
"""Cold start from <index>/ngh, timed: tsh_index_open_ngh (the whole index) and tsh_index_open_ngh_shard
(one rank of eight) on an index directory written by the oracle's restated writer (oracle/ngh_dir.py: test
infrastructure -- this is a probe, not the product).  The files are in the page cache when they are opened (just written);
a cold disk adds its own read time.  One process per library: TSH_LIB_PATH selects a variant for an A/B.
  python tests/probes/cold_start.py [rows=1000000] [dim=768] [reps=3]"""
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
root = os.environ.get("TSH_COLD_DIR", "/tmp/tsh_cold_%d_%d" % (rows, dim))

import torch  # noqa: E402  (device bring-up as the tests have it)

torch.cuda.init()
from tostore_amd import HipVectorIndex  # noqa: E402

if not os.path.exists(os.path.join(root, "meta.json")):
    from oracle import ngh_dir

    shutil.rmtree(root, ignore_errors=True)
    rng = np.random.default_rng(1)
    v = rng.standard_normal((rows, dim), dtype=np.float32)
    t = time.time()
    ngh_dir.write_ngh_dir(root, v, metric=0, deleted=list(range(0, rows, 1000)))
    print("wrote %s in %.1f s" % (root, time.time() - t), flush=True)
    del v
raw_bytes = 0
for d, _, fs in os.walk(os.path.join(root, "rawvec")):
    raw_bytes += sum(os.path.getsize(os.path.join(d, f)) for f in fs)
all_bytes = raw_bytes
for d, _, fs in os.walk(os.path.join(root, "graph")):
    all_bytes += sum(os.path.getsize(os.path.join(d, f)) for f in fs)
print("library", os.environ.get("TSH_LIB_PATH", "shipped"), "| index", rows, "x", dim, "| rawvec files %.2f GB, rawvec + graph %.2f GB" % (raw_bytes / 1e9, all_bytes / 1e9))
q = np.random.default_rng(2).standard_normal(dim).astype(np.float32)
for what in ("whole", "rank 3 of 8"):
    ts = []
    for r in range(reps + 1):
        t = time.perf_counter()
        if what == "whole":
            idx, info = HipVectorIndex.open_ngh(root)
        else:
            idx, info = HipVectorIndex.open_ngh_shard(root, world=8, rank=3)
        dt = time.perf_counter() - t
        ids, dist, cnt = idx.search(q[None, :], 10)
        n = info["rows_loaded"] if isinstance(info, dict) else info.rows_loaded
        idx.close()
        if r:  # (the first pass also pays the library's one-time set-up)
            ts.append(dt)
    share = 1.0 if what == "whole" else n / rows
    best, med = min(ts), sorted(ts)[len(ts) // 2]
    print("%-12s rows %8d  open %.3f s best / %.3f s median  = %.2f GB/s of rawvec page bytes (best), %.2f incl. graph pages; first hit %d"
          % (what, n, best, med, raw_bytes * share / best / 1e9, all_bytes * share / best / 1e9, int(ids[0, 0])))
