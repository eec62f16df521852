"""Round 5: one rank's share of the headline at N = 8 (125 k x 768, L2, k = 100) through tsh_search_sharded over real
RCCL in a world of one, 20-query calls -- the shape of side.shard_of_8 without the rest of bench.py.  Run it under
`rocprofv3 --kernel-trace` to see the scans of a call back to back (tools/attic/r5_s8_trace.py reads the trace)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=125000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--nq", type=int, default=20)
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--group", type=int, default=0)
    ap.add_argument("--ahead", type=int, default=0)
    a = ap.parse_args()
    import torch

    torch.cuda.init()
    from tostore_amd import HipVectorIndex
    from tostore_amd.sharded import CommSearcher

    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randn((a.rows, a.dim), generator=g, device="cuda")
    x /= x.norm(dim=1, keepdim=True)
    x *= torch.rand((a.rows, 1), generator=g, device="cuda") * 1.5 + 0.5
    idx = HipVectorIndex(a.dim, 0, capacity_rows=a.rows, shard_device=0, row_base=0)
    torch.cuda.synchronize()
    idx.append_device(0, a.rows, x.data_ptr())
    idx.set_batch_min_nq(0)
    from tostore_amd import _ffi
    _ffi.check(_ffi.lib().tsh_index_set_option(None, _ffi.TSH_OPT_EXCHANGE_AHEAD, a.ahead))
    cs = CommSearcher(idx, 1, 0, CommSearcher.unique_id(), 0)
    rng = np.random.default_rng(2)
    qs = rng.standard_normal((a.nq * 8, a.dim)).astype(np.float32)
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    for i in range(5):
        cs.search_many(qs[: a.nq], a.k, group=a.group)
    cs.timeline(reset=True)
    ts = []
    for i in range(a.calls):
        q = qs[(i % 8) * a.nq:(i % 8 + 1) * a.nq]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cs.search_many(q, a.k, group=a.group)
        ts.append(time.perf_counter() - t0)
    tl = cs.timeline()
    ts = np.asarray(ts) * 1e6
    print("per call us: median %.1f min %.1f  -> %.2f us per query" % (np.median(ts), ts.min(), np.median(ts) / a.nq))
    print({k: round(v / a.calls, 1) for k, v in tl.items() if k.endswith("_us")})
    cs.close()
    idx.close()


if __name__ == "__main__":
    main()
