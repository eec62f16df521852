#!/usr/bin/env python
"""Cold-start rate of tsh_index_open_ngh: writes an index directory with the writer restatement
(oracle/ngh_dir.py), then times opening it (meta.json + raw-vector pages + graph flags -> HBM).
  python tests/probes/cold_start_probe.py [rows=200000] [dim=768]"""
import os, sys, time, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ngh_dir
from tostore_amd import HipVectorIndex
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rng = np.random.default_rng(0)
v = rng.standard_normal((n, d)).astype(np.float32)
root = tempfile.mkdtemp(prefix="ngh_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    t = time.perf_counter()
    ngh_dir.write_ngh_dir(root, v, metric=0, max_partition_file_size=(256 + 32 * 16) << 20, deleted=list(range(0, n, 1000)))
    tw = time.perf_counter() - t
    size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(root) for f in fs)
    torch.cuda.init()
    for rep in range(2):
        t = time.perf_counter()
        idx, info = HipVectorIndex.open_ngh(root)
        dt = time.perf_counter() - t
        q = v[12345]
        ids, dist, cnt = idx.search(q, 5)
        assert ids[0, 0] == 12345 and info["rows_loaded"] == n and info["tombstones"] == len(range(0, n, 1000))
        idx.close()
        print(f"open_ngh {n} x {d} f32: {size / 1e9:.2f} GB on disk (page cache) in {dt:.2f} s = {size / dt / 1e9:.2f} GB/s, "
              f"{n / dt / 1e3:.0f} k rows/s (writer took {tw:.1f} s)", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
