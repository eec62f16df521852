import sys, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.init()
from tostore_amd import HipVectorIndex
rng = np.random.default_rng(0)
rows = rng.standard_normal((5000, 64)).astype(np.float32)
idxs = []
for i in range(int(sys.argv[1])):
    ix = HipVectorIndex(64, 0); ix.append(0, rows); ix.search(rows[:3], 5); idxs.append(ix)
    if i % 10 == 9: print("open", i + 1, flush=True)
for ix in idxs:
    ids, d, c = ix.search(rows[7], 3); assert ids[0, 0] == 7
print("ok", len(idxs))
