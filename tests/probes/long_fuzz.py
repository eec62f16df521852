#!/usr/bin/env python
"""Extended run of the seeded fuzz cases of tests/test_gpu_fuzz.py (many more seeds than the test suite).
  python tests/probes/long_fuzz.py [minutes=8]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import oracle
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as f
budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 480
t0 = time.time()
seed = 100
n_single = n_batch = n_seq = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(50_000 + seed)
    for case in range(20):
        f._one_case(oracle, rng, f"long {seed}/{case}")
        n_single += 1
    rng = np.random.default_rng(90_000 + seed)
    for case in range(4):
        f._batch_case(oracle, rng, f"long {seed}/{case}")
        n_batch += 1
    rng = np.random.default_rng(130_000 + seed)
    for case in range(3):
        f._sequence_case(oracle, rng, f"long {seed}/{case}", steps=30)
        n_seq += 1
    seed += 1
print(f"long fuzz ok: {n_single} single-path cases, {n_batch} batched cases, {n_seq} call sequences of 30 steps, "
      f"{seed - 100} seeds, {time.time() - t0:.0f} s")
