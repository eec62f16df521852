"""Debug helper: why do some queries of a large-k batch fall back?  Prints the
per-query header statistics of one batched shard search."""
import ctypes
import struct
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
from tostore_amd import HipVectorIndex, _ffi  # noqa: E402

torch.cuda.init()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, d, nq = 1_000_000, 768, 1024
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn((n, d), generator=g, device="cuda"); x /= x.norm(dim=1, keepdim=True)
idx = HipVectorIndex(d, 2, capacity_rows=n, shard_device=0, row_base=0)
idx.append_device(0, n, x.data_ptr())
q = torch.randn((nq, d), generator=g, device="cuda"); q /= q.norm(dim=1, keepdim=True)
qs = np.ascontiguousarray(q.cpu().numpy())
L = _ffi.lib()
entries = L.tsh_default_block_entries(k)
bb = L.tsh_candidate_block_bytes(entries)
buf = torch.empty(nq * bb, dtype=torch.uint8, device="cuda")
_ffi.check(L.tsh_search_shard(idx._h, qs.ctypes.data_as(_ffi.p_f32), nq, k, None, entries, ctypes.c_void_p(buf.data_ptr()), None))
h = buf.cpu().numpy()
cnts, totals, flags = [], [], []
for i in range(nq):
    c, e, tau, band, tiles, fl, kk, m = struct.unpack_from("<8I", h, i * bb)
    cnts.append(c); totals.append(tiles); flags.append(fl)
cnts, totals, flags = np.array(cnts), np.array(totals), np.array(flags)
print("entries", entries, "count: min/mean/max", cnts.min(), cnts.mean(), cnts.max())
print("list totals (B1 survivors + sample): min/mean/max", totals.min(), totals.mean(), totals.max())
print("flags nonzero:", int((flags != 0).sum()), "counters", idx.counters())
