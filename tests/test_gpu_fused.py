"""GPU: the single-dispatch experiment (tsh_fused.hip.h: scan + select + re-rank in ONE kernel, the last
workgroup to finish its tiles does the tail; enabled with TSH_FUSED=1, off by default because it measured slower
than the three-kernel pipeline -- DESIGN.md).  It must still be exact: a process with TSH_FUSED=1 runs the same
cases as the default pipeline, both bit-identical to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [  # (rows, dim, k)
    (10_000, 128, 10),    # config C1
    (777, 32, 5), (5000, 64, 100), (3000, 100, 7), (4097, 300, 33), (20_000, 768, 100), (2500, 1000, 50),
    (1500, 1536, 20), (600, 4096, 9), (130, 770, 200), (64, 16, 64), (40_000, 96, 1000),
]


def _run_cases(oracle, expect_fused):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(42)
    for n, d, k in CASES:
        rows = rng.standard_normal((n, d)).astype(np.float32)
        rows[n // 3] = rows[n // 2]  # a tie
        dead = rng.choice(n, size=max(1, n // 50), replace=False)
        keepbits = rng.random(n) < 0.3
        mask = np.packbits(keepbits, bitorder="little")
        for metric in (0, 1, 2):
            q = rng.standard_normal(d).astype(np.float32)
            if metric == 2:
                q = oracle.normalize_f32(q)
            with HipVectorIndex(d, metric) as idx:
                idx.set_batch_min_nq(0)
                idx.append(0, rows)
                # dense scan, one query and a pipelined group
                ids, dist, cnt = idx.search(q, k)
                eids, edist = oracle.search_heap(rows, q, metric, k)
                assert cnt[0] == len(eids) and np.array_equal(ids[0, :cnt[0]], eids), (n, d, k, metric)
                assert np.array_equal(dist[0, :cnt[0]], edist), (n, d, k, metric)
                qs = rng.standard_normal((9, d)).astype(np.float32)
                if metric == 2:
                    qs = np.stack([oracle.normalize_f32(x) for x in qs])
                ids, dist, cnt = idx.search(qs, k)
                for i in range(len(qs)):
                    eids, edist = oracle.search_heap(rows, qs[i], metric, k)
                    assert np.array_equal(ids[i, :cnt[i]], eids) and np.array_equal(dist[i, :cnt[i]], edist)
                # masked scan: tombstones and a caller mask, with a threshold
                idx.set_deleted(dead.tolist())
                live = keepbits.copy()
                live[dead] = False
                thr = float(np.median(oracle.all_distances(q, rows[:200], metric)))
                ids, dist, cnt = idx.search(q, k, thr, mask)
                eids, edist = oracle.search_heap(rows, q, metric, k, thr, np.packbits(live, bitorder="little"))
                assert cnt[0] == len(eids) and np.array_equal(ids[0, :cnt[0]], eids), (n, d, k, metric, "masked")
                assert np.array_equal(dist[0, :cnt[0]], edist)
                c = idx.counters()
                assert c["fallback_searches"] == 0 or k >= 200
                if expect_fused:
                    assert c["fused_launches"] == c["scan_launches"] > 0, (n, d, k, c)
                else:
                    assert c["fused_launches"] == 0


def test_three_kernel_pipeline_is_bit_exact(hip_lib, oracle_mod):
    if os.environ.get("TSH_FUSED") == "1":
        pytest.skip("TSH_FUSED is set")
    _run_cases(oracle_mod, False)


def test_fused_single_dispatch_on_the_same_cases(hip_lib):
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch; torch.cuda.is_available() and torch.cuda.init()\n"
            "import oracle; oracle.build()\n"
            "import test_gpu_fused as t; t._run_cases(oracle, True); t._big(oracle); print('ok')\n") % (
                ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, TSH_FUSED="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def _big(oracle_mod):
    """TSH_FUSED=1: big scans keep the three-kernel pipeline, a selective mask on the same shard is fused."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(1)
    n, d, k = 60_000, 768, 10  # 184 MB: above the fused limit
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    with HipVectorIndex(d, 0) as idx:
        idx.set_batch_min_nq(0)
        idx.append(0, rows)
        ids, dist, cnt = idx.search(q, k)
        eids, edist = oracle_mod.search_heap(rows, q, 0, k)
        assert np.array_equal(ids[0], eids) and np.array_equal(dist[0], edist)
        assert idx.counters()["fused_launches"] == 0
        keep = np.zeros(n, bool)
        keep[1000:1600] = True  # a selective mask makes the same shard's scan short: fused
        ids, dist, cnt = idx.search(np.stack([q, q]), k, None, np.packbits(keep, bitorder="little"))
        eids, edist = oracle_mod.search_heap(rows, q, 0, k, None, np.packbits(keep, bitorder="little"))
        assert np.array_equal(ids[0], eids) and np.array_equal(dist[1], edist)
        assert idx.counters()["fused_launches"] == 2
