"""GPU: PQ batch encode of resident rows -- codes bit-exact with the
oracle's restatement of batchPqEncode (/root/reference/lib/src/core/compute_tasks.dart:2292-2326)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,subspaces", [(64, 8), (128, 16), (768, 96), (100, 12), (36, 9), (40, 2)])
def test_pq_codes_bit_exact(hip_lib, oracle_mod, dim, subspaces):
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(dim)
    K, sd = 256, dim // subspaces
    n = 3000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    cb = rng.standard_normal((subspaces, K, sd)).astype(np.float32)
    cb[1, 200] = cb[1, 3]                     # duplicate centroid: the lower index must win
    rows[5, sd:2 * sd] = cb[1, 3]             # exact hit on the duplicated centroid
    rows[9, 0] = np.nan                       # NaN distance never beats +inf: code 0
    cb[0, 17, 0] = np.nan
    with HipVectorIndex(dim, 0) as idx:
        idx.append(0, rows[:1000])
        idx.append(1000, rows[1000:])
        got = idx.pq_encode(0, n, cb, subspaces, K)
        want = oracle_mod.pq_encode(cb, subspaces, K, sd, rows)
        assert got.dtype == np.uint8 and np.array_equal(got, want)
        assert got[5, 1] == 3 and got[9, 0] == 0
        part = idx.pq_encode(1234, 500, cb, subspaces, K)  # a sub-range
        assert np.array_equal(part, want[1234:1734])


def test_pq_fewer_centroids_and_shards(hip_lib, oracle_mod, monkeypatch):
    from tostore_amd import HipVectorIndex

    monkeypatch.setenv("TSH_SHARDS_SHARE_DEVICES", "1")
    rng = np.random.default_rng(3)
    dim, M, K = 48, 6, 100
    rows = rng.standard_normal((5000, dim)).astype(np.float32)
    cb = rng.standard_normal((M, K, 8)).astype(np.float32)
    with HipVectorIndex(dim, 2, capacity_rows=5000, n_devices=2) as idx:
        idx.append(0, rows)
        got = idx.pq_encode(100, 4800, cb, M, K)  # spans both shards
        assert np.array_equal(got, oracle_mod.pq_encode(cb, M, K, 8, rows[100:4900]))


def test_pq_encode_rate(hip_lib, oracle_mod):
    """not an assertion on speed, just a recorded figure: 200k x 768 rows, M = 96."""
    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(1)
    n, dim, M = 200_000, 768, 96
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    cb = rng.standard_normal((M, 256, 8)).astype(np.float32)
    with HipVectorIndex(dim, 2, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.pq_encode(0, 1000, cb, M)
        t = time.perf_counter()
        got = idx.pq_encode(0, n, cb, M)
        dt = time.perf_counter() - t
        print(f"\npq_encode {n}x{dim} M={M}: {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M vectors/s")
        sample = rng.choice(n, 300, replace=False)
        assert np.array_equal(got[sample], oracle_mod.pq_encode(cb, M, 256, 8, rows[sample]))
