"""GPU: a batched tsh_search on a device too full for its allocations degrades instead of returning TSH_E_OOM
(VERDICT round 3, weak item 4): first to the f32 MFMA kernel on the rows as stored (no converted copy), then to
pipelined single-query scans; results stay bit-exact and tsh_counters says which path ran."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("expect,over", [("planes", str(8 << 20)), ("scans", "1")])
def test_injected_allocation_failures(hip_lib, expect, over):
    # 50 k x 128: the fp16 copy is 12.8 MB, the largest scratch buffer 4 MB -- 8 MB fails the copy only, 1 byte everything
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_degrade_worker.py"), expect], cwd=ROOT,
                       env=dict(os.environ, TSH_TEST_FAIL_ALLOC_OVER=over), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    assert p.stdout.count("results ok, counters ok") == 3, p.stdout


def test_device_filled_with_ballast(hip_lib, oracle_mod):
    """The real thing: the device is filled with a ballast allocation until the copy of the rows (then the batch
    scratch) does not fit."""
    import torch

    from tostore_amd import HipVectorIndex

    rng = np.random.default_rng(11)
    n, d, k = 400_000, 256, 50
    rows = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((512, d)).astype(np.float32)
    ref = oracle_mod.search_heap_many_mt(rows, qs, 0, k)

    def same(got, m):
        return bool(np.array_equal(got[2], ref[2][:m]) and np.array_equal(got[0], ref[0][:m])
                    and np.array_equal(got[1].view(np.uint64), ref[1][:m].view(np.uint64)))

    with HipVectorIndex(d, 0, capacity_rows=n) as idx:
        idx.append(0, rows)
        idx.set_batch_min_nq(0)
        assert same(idx.search(qs[:64], k), 64)  # every single-query context exists before the device fills up
        idx.set_batch_min_nq(2)
        ballast = []
        try:
            torch.cuda.synchronize()
            free, _ = torch.cuda.mem_get_info()
            # the fp16 copy is 205 MB, a 64-query call's scratch about 10 MB: leave 100 MB
            ballast.append(torch.empty(max(0, free - (100 << 20)), dtype=torch.uint8, device="cuda"))
            got = idx.search(qs[:64], k)
            c = idx.counters()
            assert same(got, 64)
            assert c["batch_plane_fallbacks"] == 1 and c["batch_scan_fallbacks"] == 0 and c["batch_kernel_last"] == 0, c
            # now leave (almost) nothing: a 512-query call's scratch (26 MB of sample keys alone) cannot grow
            free, _ = torch.cuda.mem_get_info()
            if free > (6 << 20):
                ballast.append(torch.empty(free - (4 << 20), dtype=torch.uint8, device="cuda"))
            got = idx.search(qs, k)
            c2 = idx.counters()
            assert same(got, 512)
            assert c2["batch_scan_fallbacks"] == 1 and c2["scan_launches"] - c["scan_launches"] >= 512, c2
        finally:
            del ballast
            torch.cuda.empty_cache()
        # with room again the library returns to the converted copy on its own (after the 64 calls it waits)
        for _ in range(66):
            got = idx.search(qs[:64], k)
        c3 = idx.counters()
        assert same(got, 64) and c3["batch_kernel_last"] in (1, 2), c3
