"""Round 5: one shape on the exact path, for profilers: n x d rows (no mask), single-query searches one after the
other.  python tools/r5_exact_one.py --rows 16384 --dim 768 --queries 200"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=200)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--metric", type=int, default=0)
a = ap.parse_args()
import torch  # noqa: F401,E402

from tostore_amd import HipVectorIndex  # noqa: E402

rng = np.random.default_rng(2)
rows = rng.standard_normal((a.rows, a.dim)).astype(np.float32)
qs = rng.standard_normal((a.queries, a.dim)).astype(np.float32)
with HipVectorIndex(a.dim, a.metric, capacity_rows=a.rows) as idx:
    idx.append(0, rows)
    idx.set_batch_min_nq(0)
    for q in qs:
        idx.search(q, a.k)
    print(idx.counters()["exact_scans"], "exact scans")
