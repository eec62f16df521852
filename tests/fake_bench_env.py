"""CPU stand-in for bench.py's device side (bench.Env): answers searches with the oracle.  Test code only --
the product path never does that.  Used by tests/test_bench_logic.py in process, and by the ranks that
`python bench.py --gpus N` starts when TSH_BENCH_ENV=tests.fake_bench_env:FakeEnv is set (the launcher test:
a machine without a GPU can then run the whole multi-rank flow over gloo)."""
import os

import numpy as np


class FakeMask:
    """Stands for a HipMask: the bitmap, usable wherever a row mask is."""

    def __init__(self, bits):
        self.bits, self.closed = np.asarray(bits, np.uint8), False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.closed = True


class FakeIndex:
    def __init__(self, oracle, d, metric, rows, lo):
        self.o, self.dim, self.metric, self.rows, self.lo = oracle, d, metric, np.ascontiguousarray(rows), lo
        self.c = {"searches": 0, "scan_launches": 0, "batch_launches": 0, "fallback_searches": 0,
                  "candidates_total": 0, "scan_us_sum": 0.0, "scan_us_samples": 0, "batch_kernel_last": -1,
                  "list_scans": 0, "exact_scans": 0}
        self.min_nq, self.kernel, self.closed = 1, 3, False
        self.pending, self.next_ticket = {}, 0

    @property
    def size(self):
        return self.lo + len(self.rows)

    def _mask(self, row_mask):
        if row_mask is None:
            return None
        if isinstance(row_mask, FakeMask):
            assert not row_mask.closed
            row_mask = row_mask.bits
        assert len(row_mask) >= (self.size + 7) // 8
        bits = np.unpackbits(np.asarray(row_mask, np.uint8), bitorder="little")[self.lo:self.lo + len(self.rows)]
        return np.packbits(bits, bitorder="little")

    def search(self, queries, k, thr=None, row_mask=None):
        assert not self.closed
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        assert q.shape[1] == self.dim and q.shape[0] >= 1
        ids, dist, cnt = self.o.search_heap_many_mt(self.rows, q, self.metric, k, thr, self._mask(row_mask), threads=2)
        ids = np.where(ids >= 0, ids + self.lo, ids)
        self.c["searches"] += len(q)
        self.c["candidates_total"] += int(cnt.sum())
        if len(q) > 1 and self.min_nq >= 1:
            self.c["batch_launches"] += 2
            self.c["batch_kernel_last"] = 2 if self.kernel == 3 else self.kernel
        else:
            self.c["scan_launches"] += len(q)
            self.c["scan_us_sum"] += 10.0 * ((len(q) + 3) // 4)
            self.c["scan_us_samples"] += (len(q) + 3) // 4
        return ids, dist, cnt

    def submit(self, q, k, row_mask=None):
        t, self.next_ticket = self.next_ticket, self.next_ticket + 1
        self.pending[t] = self.search(q, k, None, row_mask)
        return (t, k)

    def wait(self, ticket, thr=None):
        ids, dist, cnt = self.pending.pop(ticket[0])
        return ids[0, :cnt[0]], dist[0, :cnt[0]]

    def make_mask(self, bits):
        return FakeMask(bits)

    def counters(self):
        return dict(self.c)

    def bench_scan(self, q, iters=20, row_mask=None):
        return 10.0

    def bench_batch(self, qs, k, iters=3):
        return 100.0, 2.0 * len(qs) * len(self.rows) * self.dim

    def set_batch_min_nq(self, v):
        self.min_nq = v

    def set_batch_kernel(self, v):
        self.kernel = v

    def close(self):
        self.closed = True


class FakeSearcher:
    """Stands for ShardedSearcher: answers over the WHOLE corpus, as the merge of all ranks would."""

    def __init__(self, whole):
        self.whole = whole
        self.groups = []
        self.tl = dict.fromkeys(("calls", "queries", "groups", "retries", "call_us", "reserve_us", "wait_scan_us", "scan_us",
                                 "exchange_wait_us", "gather_us", "slice_d2h_us", "merge_us", "result_gather_us",
                                 "copy_out_us", "retry_scan_us", "pre_enqueue_us"), 0)

    def _book(self, nq, dt):
        t = self.tl
        t["calls"] += 1
        t["queries"] += nq
        t["groups"] += 1
        t["call_us"] += dt * 1e6
        for key, share in (("wait_scan_us", 0.6), ("exchange_wait_us", 0.2), ("merge_us", 0.1), ("result_gather_us", 0.05),
                           ("copy_out_us", 0.04)):
            t[key] += dt * 1e6 * share
        t["gather_us"] += dt * 1e6 * 0.15
        t["slice_d2h_us"] += dt * 1e6 * 0.05
        t["scan_us"] += dt * 1e6 * 0.9

    def timeline(self, reset=False):
        out = dict(self.tl)
        if reset:
            self.tl = dict.fromkeys(self.tl, 0)
        return out

    def search(self, q, k, thr=None, row_mask=None):
        import time

        t0 = time.perf_counter()
        r = self.whole.search(q, k, thr, row_mask)
        self._book(len(np.atleast_2d(q)), time.perf_counter() - t0)
        return r

    def close(self):
        pass

    def search_many(self, qs, k, thr=None, row_mask=None, group=8):
        import time

        assert group >= 0 and len(qs) >= 1
        self.groups.append((len(qs), group))
        t0 = time.perf_counter()
        r = self.whole.search(qs, k, thr, row_mask)
        self._book(len(qs), time.perf_counter() - t0)
        return r


class FakeEnv:
    """`oracle` given: in-process stand-in with stubbed collectives (rank 0 of `world`).  Constructed by bench.py
    itself (TSH_BENCH_ENV): rank / world from the environment, collectives over a real gloo group."""

    def __init__(self, oracle, world=1):
        self.dist = None
        if not hasattr(oracle, "search_heap"):  # bench.py passes its parsed arguments
            import oracle as oracle_mod

            oracle_mod.build()
            self.world, self.rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
            oracle = oracle_mod
            if os.environ.get("TSH_BENCH_HANG_RANK") == str(self.rank):
                import time

                time.sleep(3600)
            if self.world > 1:
                import torch.distributed as dist

                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
                self.dist = dist
            if os.environ.get("TSH_BENCH_FAIL_RANK") == str(self.rank):
                raise RuntimeError("rank %d told to fail (launcher test)" % self.rank)
        else:
            self.world, self.rank = world, 0
        self.o = oracle
        self.fences = 0
        self.made = []
        self.last_searcher = None
        self.searchers = []  # the headline's first, then the side legs'
        self.exchange, self.exchange_note = "stand-in", None

    def corpus(self, n, d, metric):
        rng = np.random.default_rng(7 + metric)
        x = rng.standard_normal((n, d)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
        if metric != 2:
            x *= (rng.random((n, 1)) * 1.5 + 0.5).astype(np.float32)
        return x

    def build_index(self, d, metric, n, lo, hi, keep_host=False, n_devices=1):
        self.n_devices_seen = n_devices  # (one handle over several devices: bench.py --in-process)
        corpus = self.corpus(n, d, metric)
        idx = FakeIndex(self.o, d, metric, corpus[lo:hi], lo)
        self.made.append(idx)
        self._whole = FakeIndex(self.o, d, metric, corpus, 0)
        return idx, (corpus[lo:hi] if keep_host else None)

    def oracle_chunks(self, n, d, metric, lo=0, hi=None):
        corpus = self.corpus(n, d, metric)
        hi = n if hi is None else hi
        for r0 in range(lo, hi, 1000):  # several chunks: the merge across chunks is exercised
            yield r0, corpus[r0:min(hi, r0 + 1000)]

    def searcher(self, idx):
        if self.world == 1:
            return None
        self.last_searcher = FakeSearcher(self._whole)
        self.searchers.append(self.last_searcher)
        return self.last_searcher

    def shard_comm(self, idx):
        self.searchers.append(FakeSearcher(idx))
        return self.searchers[-1]

    def max_inflight(self):
        return 8

    def fence(self):
        self.fences += 1
        if self.dist is not None:
            self.dist.barrier()

    def reduce_max(self, x):
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_int(self, x):
        if self.dist is None:
            return int(x)
        import torch

        t = torch.tensor([int(x)], dtype=torch.int64)
        self.dist.broadcast(t, src=0)
        return int(t.item())

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        box = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, box, dst=0)
        return box

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


