#!/usr/bin/env python
"""bench.py -- BASELINE.json headline metric: kNN queries/sec (+ recall@k) on
1M x 768 f32 brute force, L2, k=100, single query per step (config C2).

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks started for it)
  python bench.py --gpus N --config c4                     (10M x 1536 inner product, k = 100: BASELINE.json C4)

A step = one single-query search through the C-ABI: K1 scan (reads every stored
row once, the query rides in the kernel arguments), K2 select, K4 f64 re-rank
(results stored straight into pinned host memory), host merge.  Independent
queries are handed to the library in groups (--group, default 64) and the
library keeps several of them in flight: scans run back to back, a query's
select/re-rank overlap the next scan on reserved CUs.  The corpus is resident
in HBM before the timed region.  N > 1: the SAME corpus is row-range sharded
over the ranks (strong scaling; every rank generates only its own rows); every
rank scans its shard, candidate blocks are all-gathered over RCCL, each rank
merges its slice of the queries (tsh_search_sharded, include/tostore_hip.h).

Timing: W warm-up steps, then a timed region of EXACTLY K steps between a
barrier + device synchronise on both sides, MAX over ranks.  That region is
repeated (--repeats, default: enough regions for about 2000 timed steps, at
most 25) and the MEDIAN region is the one reported: a 20-step region lasts
9 ms, one region alone is noise.  Every region's time is in the line.

Prints ONE JSON line on rank 0.  Besides the headline it carries
`roofline`, `cpu_baseline` and a `side` object with the other BASELINE.json
configurations measured in the same process (C1 latency, C3 1024-query
batches, C5 masked scans); a side leg that fails reports its error text and
never takes the headline down.

N > 1 lines carry `exchange_timeline` (where every rank's tsh_search_sharded time
went, phase by phase, against ms_per_step), `host_cpu` per rank, and a `side.C4_per_rank`
leg at BASELINE.json's C4 per-rank shape (1.25 M x 1536 per rank, inner product): the strong
(C2) and the weak (C4) point of the scaling curve from one command.

The CPU oracle is reached through bench_check.py only, and only outside timed regions
(`timed_region` below; bench_check refuses to load or run inside one).

The device side sits behind `Env` so that tests/test_bench_logic.py can drive
every line of the arithmetic below on a machine without a GPU.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import contextlib


@contextlib.contextmanager
def timed_region():
    """Everything measured runs inside one of these: bench_check.py (the oracle) refuses to load or run meanwhile."""
    prev = os.environ.get("TSH_BENCH_TIMED")
    os.environ["TSH_BENCH_TIMED"] = "1"
    try:
        yield
    finally:
        if prev is None:
            del os.environ["TSH_BENCH_TIMED"]
        else:
            os.environ["TSH_BENCH_TIMED"] = prev


HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
F32_MFMA_PEAK_TF = 157.3  # same guide: dense f32 MFMA
F16_MFMA_PEAK_TF = 2500.0  # same guide: dense bf16 / f16 MFMA
METRICS = {"l2": 0, "ip": 1, "cosine": 2}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of --steps steps each; the median one is reported (0 = auto)")
    ap.add_argument("--config", default="c2", choices=["c2", "c4"],
                    help="BASELINE.json configuration: c2 = 1M x 768 L2 k=100 (the headline), c4 = 10M x 1536 inner "
                         "product k=100 (row-sharded over --gpus ranks); --rows / --dim / --k / --metric override")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--metric", default=None, choices=list(METRICS))
    ap.add_argument("--inflight", type=int, default=8,
                    help="independent single-query searches kept in flight (1 = strictly one at a time)")
    ap.add_argument("--group", type=int, default=64,
                    help="N=1: hand the library this many independent queries per call (its own C++ pipeline "
                         "keeps tsh_max_inflight() in flight); 0 = drive submit/wait from Python")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per step through the batched matrix-core path (config C3: --batch 1024 "
                         "--metric cosine); 0 = the headline single-query workload")
    ap.add_argument("--mask-keep", type=float, default=0.0,
                    help="config C5: row mask keeping this fraction of the rows (0 = no mask)")
    ap.add_argument("--batch-kernel", type=int, choices=[0, 1, 2, 3], default=3,
                    help="C3 pre-filter keys: 0 f32 MFMA, 1 bf16x3, 2 fp16, 3 auto = the library default (fp16 for "
                         "cosine, bf16x3 otherwise); results are identical")
    ap.add_argument("--mask-kind", choices=["bernoulli", "range"], default="bernoulli",
                    help="C5 mask shape: i.i.d. Bernoulli(keep) per row, or one contiguous id range of keep*rows rows")
    ap.add_argument("--recall-queries", type=int, default=1000,
                    help="N=1: queries whose GPU answer is compared with the exhaustive CPU oracle (all host cores)")
    ap.add_argument("--recall-seconds", type=float, default=60.0,
                    help="the recall leg stops taking further 100-query chunks after this many seconds of oracle time")
    ap.add_argument("--lat-queries", type=int, default=1000,
                    help="queries of the one-at-a-time latency leg (p50 / p99 / max in the line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unit-rows", action="store_true",
                    help="corpus rows stay L2-normalised for every metric (the demo's own recipe; default: L2 / IP rows are "
                         "also scaled by U(0.5, 2) so that the three metrics rank differently, SURVEY.md section 8d)")
    ap.add_argument("--norm-range", default="0.5,2",
                    help="L2 / IP corpora: row norms are U(lo, hi) (default 0.5,2: SURVEY.md section 8d); 0.1,3.2 is the 'widely "
                         "spread norms' corpus of VERDICT round 5, item 7 (a factor 32 between the shortest and the longest row)")
    ap.add_argument("--hub", action="store_true",
                    help="--batch: with the hub rows' bound (TSH_OPT_BATCH_HUB = 1) beside the sample's threshold, for A/B runs")
    ap.add_argument("--plane-in-row-order", action="store_true",
                    help="--batch: the fp16 copy of an L2 / inner-product corpus in row order (TSH_OPT_BATCH_GROUP = 0) instead of "
                         "grouped by norm inside blocks of 8192 rows, for A/B runs")
    ap.add_argument("--no-side", action="store_true", help="skip the side legs (C1 / C3 / C5)")
    ap.add_argument("--side", default="c5,s8,c4s8,c1,c3",
                    help="side legs to run, comma separated (s8 = side.shard_of_8: one rank's share of the headline at N = 8 "
                         "through tsh_search_sharded; c4s8 = side.C4_shard_of_8: one rank's share of BASELINE.json's C4, "
                         "--c4-rows-per-rank x 1536 inner product, the same way)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--exchange", choices=["auto", "torch", "capi"], default="auto",
                    help="N > 1: who all-gathers the candidate blocks -- capi: the library's own entry points "
                         "(tsh_comm_* / tsh_search_sharded: RCCL inside the library, what a host without torch uses; the "
                         "communicator id travels by torch broadcast here); torch: torch.distributed around "
                         "tsh_search_shard + tsh_merge_candidates; auto (default): capi, checked against torch on a "
                         "few queries first, torch if that fails")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="N > 1 started without WORLD_SIZE: seconds before the ranks this process started are given up")
    ap.add_argument("--ranks-share-gpu", action="store_true", help="testing: every rank uses cuda:0")
    ap.add_argument("--fake-rccl", action="store_true",
                    help="testing (implies --ranks-share-gpu): the library's RCCL branch over tests/fake_rccl, the "
                         "stand-in that lets several ranks share one GPU (TSH_RCCL_LIB); torch's own collectives run over gloo")
    ap.add_argument("--c4-rows-per-rank", type=int, default=1_250_000,
                    help="N > 1: rows per rank of the side.C4_per_rank leg (BASELINE.json C4: 10 M x 1536 over 8 GPUs); 0 = skip")
    ap.add_argument("--c3-check", type=int, default=1024, help="side.C3: queries of one batch checked against the oracle")
    ap.add_argument("--c5-check", type=int, default=100, help="side.C5: queries checked per selectivity")
    ap.add_argument("--in-process", action="store_true",
                    help="--gpus N in ONE process: tsh_index_create(n_devices = N) -- row-range shards on N devices, one host "
                         "thread per shard, candidate blocks copied back, host merge, NO collective: the deployment an embedded "
                         "single-process database uses first (the reference searches from one isolate of one process: "
                         "lib/src/core/vector_index_manager.dart:538).  Same line; config.sharding says which exchange ran")
    ap.add_argument("--shards-share-gpu", action="store_true",
                    help="testing (with --in-process): the N shards share cuda:0 (TSH_SHARDS_SHARE_DEVICES behind the library's "
                         "test-hook opt-in): a rehearsal of the code path on a one-GPU box, not a scaling figure")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the N>1 code path (process group, all-gather, merge) even with one rank")
    a = ap.parse_args(argv)
    preset = {"c2": (1_000_000, 768, 100, "l2"), "c4": (10_000_000, 1536, 100, "ip")}[a.config]
    a.rows = preset[0] if a.rows is None else a.rows
    a.dim = preset[1] if a.dim is None else a.dim
    a.k = preset[2] if a.k is None else a.k
    a.metric = preset[3] if a.metric is None else a.metric
    a.norm_lo, a.norm_hi = (float(x) for x in a.norm_range.split(","))
    if a.fake_rccl:
        a.ranks_share_gpu = True
    if a.ranks_share_gpu and a.backend == "nccl":
        a.backend = "gloo"  # RCCL refuses two ranks on one device
    return a


# ------------------------------------------------------------------ sizing (pure)
def auto_repeats(steps, repeats=0):
    """Timed regions per run: about 2000 timed steps in total, 3..25 regions."""
    if repeats > 0:
        return repeats
    return int(max(3, min(25, math.ceil(2000 / max(steps, 1)))))


def query_pool_size(steps, warmup, recall_queries, batch=0):
    """Distinct queries generated up front.  Every later index into the pool is taken modulo its
    length, and the pool is never smaller than what the baseline / recall legs read."""
    return int(max(warmup + steps, 64, recall_queries, 2 * batch, 1))


def library_schedule(nq, rows_per_shard, dim, batched=False):
    """tsh_search_sharded's own schedule of queries per exchange (sharded_schedule, tsh_host_sync.h), for the
    report: uniform groups beyond 128 queries; up to 128 shrinking groups when the shards scan query by query (batched =
    False: the headline's TSH_OPT_BATCH_MIN_NQ = 0), one group when the ranks batch and the call pays for it."""
    if nq > 128:
        g = 512 if nq >= 1024 else 256
        return [min(g, nq - q) for q in range(0, nq, g)]
    scan_us = float(rows_per_shard) * ((dim + 3) // 4 * 4) * 4.0 / 6.5e6
    if batched and nq >= 2 and nq * (scan_us + 25.0) > 300.0 + 0.825 * scan_us * ((nq + 127) // 128):
        return [int(nq)]
    g_min = int(min(128.0, max(4.0, math.ceil(150.0 / max(scan_us, 1.0)))))
    out, rem = [], int(nq)
    while rem > 0:
        g = rem if rem < 2 * g_min else max(g_min, (rem + 1) // 2)
        out.append(g)
        rem -= g
    return out


def library_group(nq, rows_per_shard=125_000, dim=768):
    """The largest group of that schedule."""
    return max(library_schedule(max(int(nq), 1), rows_per_shard, dim))


def sharded_group(group, count):
    """N > 1: up to --group queries per all-gather / merge; short runs use smaller groups so that
    at least four of them pipeline (scan of group g+1 behind the exchange of group g)."""
    return int(max(1, max(16, min(group or 64, count // 4))))


def clean_json(o):
    """NaN / inf -> null (json.dumps would print a bare NaN, which strict parsers reject)."""
    if isinstance(o, dict):
        return {str(k): clean_json(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [clean_json(v) for v in o]
    if isinstance(o, (np.floating, float)):
        o = float(o)
        return o if math.isfinite(o) else None
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


def dumps(o):
    return json.dumps(clean_json(o), allow_nan=False)


def make_queries(nq, d, metric, seed=20260613):
    rng = np.random.Generator(np.random.Philox(seed))
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    if metric == 2:  # caller-side _normalizeFloat32 (vector_index_manager.dart:516-520)
        from tostore_amd import normalize_float32
        q = np.stack([normalize_float32(x) for x in q])
    return np.ascontiguousarray(q)


def make_mask(n, keep, kind, seed=20260614):
    """C5: WHERE pre-filter as a row bitmask, LSB first."""
    if kind == "range":  # e.g. WHERE id BETWEEN ...: one contiguous run of node ids
        keepbits = np.zeros(n, bool)
        m = int(n * keep)
        start = int(np.random.Generator(np.random.Philox(seed)).integers(0, max(1, n - m)))
        keepbits[start:start + m] = True
    else:
        keepbits = np.random.Generator(np.random.Philox(seed)).random(n) < keep
    return np.packbits(keepbits, bitorder="little"), int(keepbits.sum())


# ------------------------------------------------------------------ device side
CORPUS_SEED = 20260612
CORPUS_CHUNK = 131072  # rows per generator chunk: chunk c depends on (seed, c) only, so a rank generates just its own


class Env:
    """Everything the benchmark needs from the GPU box.  tests/test_bench_logic.py substitutes a
    CPU stand-in with the same methods."""

    def __init__(self, a):
        import torch

        self.torch = torch
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if a.ranks_share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        if a.in_process:
            if self.world != 1:
                raise RuntimeError("--in-process is one process: start it without a launcher")
            if a.shards_share_gpu:
                os.environ["TSH_SHARDS_SHARE_DEVICES"] = "1"
            elif torch.cuda.device_count() < a.gpus:
                raise RuntimeError("--in-process --gpus %d: this box has %d GPU(s) (one GPU: --shards-share-gpu)"
                                   % (a.gpus, torch.cuda.device_count()))
        elif self.world != a.gpus and self.world == 1 and a.gpus > 1:
            raise RuntimeError("--gpus %d needs %d ranks (main() starts them when WORLD_SIZE is not set)" % (a.gpus, a.gpus))
        if torch.cuda.device_count() <= self.local_rank:
            raise RuntimeError("rank %d wants cuda:%d, this box has %d GPU(s) (one GPU: --ranks-share-gpu)"
                               % (self.rank, self.local_rank, torch.cuda.device_count()))
        if self.world > 1:  # one rank of several: a share of the CPUs (see launch_ranks)
            torch.set_num_threads(max(1, int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cpus() // self.world))
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        self.exchange = None
        self.exchange_note = None
        if self.world > 1 or a.force_sharded:
            import torch.distributed as dist

            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist
        self.ctl = self.dev if a.backend == "nccl" else torch.device("cpu")  # where small control tensors live
        from tostore_amd import _ffi

        assert _ffi.lib().tsh_device_count() >= 1, "libtostore_hip.so sees no device"
        self._ffi = _ffi
        if a.in_process and a.shards_share_gpu:  # (the library obeys the variable only in a process that asked for its test hooks)
            _ffi.enable_test_hooks()

    def corpus_chunks(self, n, d, metric, lo=0, hi=None):
        """Recipe of the reference's demo (/root/reference/example/lib/tostore_example.dart:728-747):
        i.i.d. N(0,1) components, rows L2-normalised, stored f32; for L2/IP each row is also scaled
        by U(0.5,2) so the three metrics rank differently.  Counter-based per
        chunk of CORPUS_CHUNK rows: yields (first row id, device rows) covering [lo, hi) and touches
        no other chunk, so every rank generates only its own shard and any rank can regenerate any row."""
        torch = self.torch
        hi = n if hi is None else hi
        g = torch.Generator(device=self.dev)
        for c in range(lo // CORPUS_CHUNK, (max(hi, lo + 1) - 1) // CORPUS_CHUNK + 1):
            s, e = c * CORPUS_CHUNK, min(n, (c + 1) * CORPUS_CHUNK)
            if e <= lo or s >= hi:
                continue
            g.manual_seed(CORPUS_SEED + c)
            x = torch.randn((e - s, d), generator=g, device=self.dev, dtype=torch.float32)
            x /= x.norm(dim=1, keepdim=True)
            if metric != 2 and not self.a.unit_rows:
                x *= torch.rand((e - s, 1), generator=g, device=self.dev) * (self.a.norm_hi - self.a.norm_lo) + self.a.norm_lo
            a0, a1 = max(s, lo), min(e, hi)
            yield a0, x[a0 - s:a1 - s].contiguous()

    def build_index(self, d, metric, n, lo, hi, keep_host=False, n_devices=1):
        """Shard handle holding global rows [lo, hi) of the n-row corpus, filled chunk by chunk (device to device, on
        the library's stream); -> (index, host copy of those rows or None).  n_devices > 1: ONE handle over that many
        devices (tsh_index_create: the library routes every chunk's rows to the shard that owns them, host to device)."""
        from tostore_amd import HipVectorIndex

        if n_devices > 1:
            idx = HipVectorIndex(d, metric, capacity_rows=hi - lo, n_devices=n_devices)
        else:
            idx = HipVectorIndex(d, metric, capacity_rows=hi - lo, shard_device=self.local_rank, row_base=lo)
        host = np.empty((hi - lo, d), np.float32) if keep_host else None
        for r0, x in self.corpus_chunks(n, d, metric, lo, hi):
            self.torch.cuda.synchronize()  # the library copies on its own stream: the producer must be done
            if n_devices > 1:
                xh = x.cpu().numpy()
                idx.append(r0, xh)
                if host is not None:
                    host[r0 - lo:r0 - lo + x.shape[0]] = xh
                del x
                continue
            idx.append_device(r0, x.shape[0], x.data_ptr())
            if host is not None:
                host[r0 - lo:r0 - lo + x.shape[0]] = x.cpu().numpy()
            del x
        self.torch.cuda.synchronize()
        self.torch.cuda.empty_cache()
        return idx, host

    def oracle_chunks(self, n, d, metric, lo=0, hi=None):
        """(first row id, host rows) over the WHOLE corpus (or its rows [lo, hi)), for the recall check of a sharded run."""
        for r0, x in self.corpus_chunks(n, d, metric, lo, n if hi is None else hi):
            yield r0, x.cpu().numpy()

    def searcher(self, idx):
        """N > 1: the exchange around this rank's shard (collective)."""
        if self.dist is None:
            return None
        from tostore_amd.sharded import CommSearcher, ShardedSearcher

        want = self.a.exchange
        fake = bool(self.a.fake_rccl and os.environ.get("TSH_RCCL_LIB"))  # tests/fake_rccl: the RCCL branch with ranks sharing a GPU
        if fake:  # (the library obeys the variable only in a process that asked for its test hooks)
            self._ffi.enable_test_hooks()
        if want == "torch":
            self.exchange = "torch.distributed all_gather_into_tensor (%s) + tsh_merge_candidates" % self.a.backend
            return ShardedSearcher(idx)
        if self.a.backend == "gloo" and not fake:  # several ranks on one GPU: the library's protocol over a host transport
            self.exchange = "tsh_search_sharded over a host transport (gloo)"
            return CommSearcher.over_torch(idx, device=self.local_rank)
        # Every step that can fail on one rank alone is followed by an agreement, so that the ranks take the same
        # branch: a rank that skipped a collective the others entered would hang the job until --launch-timeout
        cs, err = None, None
        try:
            box = [CommSearcher.unique_id() if self.rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            box, err = [None], repr(e)
        self.dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            err = err or "rank 0 could not make a communicator id"
        else:
            try:
                cs = CommSearcher(idx, self.world, self.rank, box[0], self.local_rank)  # collective inside the library
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        bad = self.reduce_max(1.0 if err else 0.0)
        if bad == 0.0 and want == "auto":
            # the library's exchange against torch's on a few queries, before anything is timed
            qs = make_queries(6, idx.dim, idx.metric, seed=20260617)
            got = None
            try:
                got = cs.search(qs, 10)
            except Exception as e:  # noqa: BLE001 -- (a rank failing locally stays in the library's collective)
                err = repr(e)
            bad = self.reduce_max(1.0 if err else 0.0)
            if bad == 0.0:
                exp = ShardedSearcher(idx).search(qs, 10)  # collective: entered by all ranks or (above) by none
                if not (np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])):
                    err = "tsh_search_sharded and the torch exchange disagree"
                bad = self.reduce_max(1.0 if err else 0.0)
        if bad == 0.0:
            self.exchange = "tsh_search_sharded (%s inside the library)" % ("tests/fake_rccl stand-in" if fake else "RCCL")
            return cs
        if want == "capi":
            raise RuntimeError("tsh_search_sharded is not usable: %s" % (err or "another rank failed"))
        if cs is not None:
            cs.close()
        self.exchange = "torch.distributed all_gather_into_tensor (%s) + tsh_merge_candidates" % self.a.backend
        self.exchange_note = "tsh_search_sharded failed its check on some rank (%s): fell back" % (err or "another rank")
        return ShardedSearcher(idx)

    def shard_comm(self, idx):
        """A communicator of ONE rank over real RCCL around `idx` (side.shard_of_8): tsh_comm_create + tsh_search_sharded
        exactly as a rank of an N-GPU job runs them, the collective included (an all-gather in a world of one)."""
        from tostore_amd.sharded import CommSearcher

        return CommSearcher(idx, 1, 0, CommSearcher.unique_id(), self.local_rank)

    def gather_objects(self, obj):
        """Every rank's object, on rank 0 (None elsewhere)."""
        if self.dist is None:
            return [obj]
        box = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, box, dst=0)
        return box

    def max_inflight(self):
        return int(self._ffi.lib().tsh_max_inflight())

    def fence(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce_max(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.ctl)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_int(self, x):
        if self.dist is None:
            return int(x)
        t = self.torch.tensor([int(x)], dtype=self.torch.int64, device=self.ctl)
        self.dist.broadcast(t, src=0)
        return int(t.item())

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------ helpers shared by the legs
def profiled_kernel_us(key):
    """A kernel's average duration under rocprofv3 --kernel-trace --stats, from the committed summaries (profiles/
    kernel_us.json: written from the profile files it names) -> dict or None.  The line's own figures come from HIP
    events in THIS run; the profiler's stand beside them where a leg's frac has been disputed."""
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_us.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def pmc_traffic(rows, d):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes (not measured in this run:
    counters need their own rocprofv3 passes) -> (bytes or None, source or None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get("%dx%d" % (rows, d))
        if ent:
            return ent["traffic_bytes"], ent.get("source")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def batch_roofline(kind, gemm_us, flops):
    tf = flops / (gemm_us * 1e-6) / 1e12 if gemm_us and gemm_us > 0 else float("nan")
    if kind == 0:
        return {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / F32_MFMA_PEAK_TF, "traffic": None,
                "kernel": "tsh::batch_score_kernel (sample + filtered passes)",
                "kernel_us": gemm_us, "algorithmic_flops_per_launch": flops}
    if kind == 2:
        return {"bound": "mfma", "achieved": tf, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf / F16_MFMA_PEAK_TF, "traffic": None,
                "kernel": "tsh::batch_score_f16pp_kernel: sample + filtered passes",
                "kernel_us": gemm_us, "algorithmic_flops_per_launch": flops, "vs_f32_mfma_peak": tf / F32_MFMA_PEAK_TF}
    # three bf16 MFMAs per algorithmic multiply-add: the ceiling for ALGORITHMIC flops is 2500 / 3
    return {"bound": "mfma", "achieved": tf, "peak": F16_MFMA_PEAK_TF / 3, "unit": "TFLOP/s",
            "frac": tf / (F16_MFMA_PEAK_TF / 3), "traffic": None,
            "kernel": "tsh::batch_score_bf16x3_kernel (sample + filtered passes)",
            "kernel_us": gemm_us, "algorithmic_flops_per_launch": flops,
            "executed_bf16_tflops": 3 * tf, "bf16_dense_peak": F16_MFMA_PEAK_TF, "vs_f32_mfma_peak": tf / F32_MFMA_PEAK_TF}


BATCH_DTYPE = {0: "f32", 1: "f32 as bf16 hi+lo (3 bf16 MFMAs per product), f64 rerank",
               2: "fp16 pre-filter keys (1 f16 MFMA per product), f64 rerank"}
BATCH_KERNEL_NAME = {0: "f32 MFMA", 1: "bf16x3", 2: "f16"}


def two_callers_batch(env, idx, qs, nq, k, steps):
    """The same calls from TWO host threads (a shard keeps two scratch sets: one call's host preparation and
    finalisation overlap the other's GPU work; the GPU side stays one in-order sequence).  Wall clock over all
    calls, no median: what a host that keeps two batches in flight gets."""
    import threading

    per = max(2, (steps + 1) // 2)
    err = []

    def caller(t):
        try:
            for i in range(per):
                idx.search(qs[((i + t) % 2) * nq:((i + t) % 2 + 1) * nq], k)
        except Exception as e:  # noqa: BLE001 -- reported below
            err.append(repr(e))

    for rep in range(2):  # the first round warms the second scratch set
        th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
        env.fence()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        env.fence()
        dt = time.perf_counter() - t0
    if err:
        return {"error": err[0]}
    return {"value": 2 * per * nq / dt, "unit": "queries/s", "calls": 2 * per, "ms_per_call": dt / (2 * per) * 1e3,
            "timing": "wall clock over all calls of both threads"}


def measure_batch(env, idx, host_rows, metric, n, d, k, nq, steps, warm, kernel, check_queries):
    """One step = one call with nq queries through the matrix-core path."""
    steps, warm = max(1, steps), max(1, warm)
    qs = make_queries(nq * 2, d, metric, seed=20260615)
    idx.set_batch_min_nq(1)
    idx.set_batch_kernel(kernel)
    for i in range(warm):
        idx.search(qs[(i % 2) * nq:(i % 2 + 1) * nq], k)
    env.fence()
    per_step = []
    with timed_region():
        t0 = time.perf_counter()
        for i in range(steps):
            t1 = time.perf_counter()
            ids, dist, cnt = idx.search(qs[(i % 2) * nq:(i % 2 + 1) * nq], k)  # synchronous: results are on the host
            per_step.append(time.perf_counter() - t1)
        env.fence()
        mean_step = (time.perf_counter() - t0) / steps
        gemm_us, flops = idx.bench_batch(qs[:nq], k, iters=3)
        two = two_callers_batch(env, idx, qs, nq, k, steps)
    # the container's CPU quota freezes the process for milliseconds now and then (tools/attic/throttle_probe.py): the
    # MEDIAN step is the measurement; mean, p99 and the slowest step are reported beside it
    srt = np.sort(np.asarray(per_step))
    elapsed = float(np.median(per_step)) * steps
    ran = idx.counters()["batch_kernel_last"]  # what auto resolved to
    out = {"value": nq * steps / elapsed, "unit": "queries/s", "steps": steps, "warmup": warm,
           "ms_per_step": elapsed / steps * 1e3, "ms_per_step_mean": mean_step * 1e3,
           "ms_per_step_p99": float(srt[min(len(srt) - 1, int(math.ceil(len(srt) * 0.99)) - 1)]) * 1e3,
           "ms_per_step_max": float(srt[-1]) * 1e3, "timing": "median step",
           "queries_per_step": nq, "callers": 1,
           "two_callers": two,
           "dtype": BATCH_DTYPE.get(ran, str(ran)), "batch_kernel": BATCH_KERNEL_NAME.get(ran, str(ran)),
           "roofline": batch_roofline(ran, gemm_us, flops),
           "key_passes_share_of_step": gemm_us * 1e-3 / (elapsed / steps * 1e3)}
    if host_rows is not None and check_queries > 0:
        import bench_check

        m = min(check_queries, nq)
        base = ((steps - 1) % 2) * nq
        out.update(bench_check.check_answers(host_rows, qs[base:base + m], metric, k, (ids, dist, cnt)))
    c = idx.counters()
    out["counters"] = {k2: c[k2] for k2 in ("batch_launches", "scan_launches", "fallback_searches")}
    out["counters"]["candidates_per_query"] = c["candidates_total"] / max(c["searches"], 1)
    return out


def bench_batch(a, env, idx, host_rows, metric):
    """Config C3 as the main line (--batch N): a side measurement, single GPU only."""
    assert env.world == 1, "the batched benchmark is single-GPU"
    n, d, k, nq = a.rows, a.dim, a.k, a.batch
    if a.hub and hasattr(idx, "set_batch_hub"):
        idx.set_batch_hub(True)
    if a.plane_in_row_order and hasattr(idx, "set_batch_group"):
        idx.set_batch_group(False)
    r = measure_batch(env, idx, host_rows, metric, n, d, k, nq, min(a.steps, 20), min(a.warmup, 3), a.batch_kernel, 128)
    out = {"metric": "kNN queries/sec, %dx%d f32 %s k=%d, %d-query batch (matrix-core path)" % (n, d, a.metric, k, nq),
           "value": r["value"], "unit": "queries/s", "n_gpus": 1, "steps": r["steps"], "warmup": r["warmup"],
           "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": r["dtype"], "data": "synthetic",
           "config": {"workload": "C3: %dx%d f32, %s, k=%d, %d-query batch" % (n, d, a.metric, k, nq),
                      "batch_kernel": r["batch_kernel"], "hub_bound": bool(a.hub),
                      "plane": "row order" if a.plane_in_row_order else "grouped by norm (L2 / inner product)"}}
    for key in ("roofline", "recall_at_k", "ids_and_distances_bit_exact", "checked_queries", "counters",
                "key_passes_share_of_step", "callers", "two_callers", "ms_per_step_mean", "ms_per_step_p99",
                "ms_per_step_max"):
        if key in r:
            out[key] = r[key]
    idx.close()
    return dumps(out)


# ------------------------------------------------------------------ side legs (N = 1, after the headline)
def side_c5(env, idx, host_rows, queries, metric, n, d, k, check=100):
    """C5: the headline corpus behind a WHERE pre-filter bitmask -- Bernoulli keep
    1 / 10 / 50 / 100 % and one contiguous id range of 10 %."""
    out = {"workload": "C5: %dx%d f32, L2, k=%d, device-side row bitmask" % (n, d, k),
           "note": "value = every query scans its kept rows on its own (masked HBM scan, pipelined, 64 queries per "
                   "call); library_default_path = the same calls with the library free to choose (64 queries: the matrix "
                   "cores -- over a gathered copy of the kept rows when the mask keeps at most 16 384 of them, over the "
                   "whole shard with the mask in the epilogue otherwise)"}
    idx.set_batch_min_nq(0)
    cnt = 1024
    sel = [i % len(queries) for i in range(cnt)]
    m = max(0, min(check, cnt))
    for keep, kind in ((0.01, "bernoulli"), (0.10, "bernoulli"), (0.50, "bernoulli"), (1.00, "bernoulli"), (0.10, "range")):
        mask, kept = make_mask(n, keep, kind)
        idx.search(queries[sel[:32]], k, None, mask)
        env.fence()
        c0 = idx.counters()
        got = [None] * (cnt // 64)
        per_call = []
        with timed_region():
            t0 = time.perf_counter()
            for g in range(cnt // 64):
                tc = time.perf_counter()
                got[g] = idx.search(queries[sel[g * 64:g * 64 + 64]], k, None, mask)  # (synchronous: results are back)
                per_call.append(time.perf_counter() - tc)
            env.fence()
            el_mean = (time.perf_counter() - t0) / (cnt // 64)
            # the MEDIAN call (side.C3 reports its median step): sixteen calls of about a millisecond, and one of them frozen
            # for the rest of a cgroup CPU period (host_cpu.throttled_*) moved the mean by a third in runs of this round
            el = sorted(per_call)[len(per_call) // 2] * (cnt // 64)
            c1 = idx.counters()
            scan_us = idx.bench_scan(queries[0], iters=20, row_mask=mask)
        # which kernel scanned: tsh_counters says (selective masks are scanned as a compacted list of row ids)
        listed = c1.get("list_scans", 0) - c0.get("list_scans", 0)
        scans = c1["scan_launches"] - c0["scan_launches"]
        exact = c1.get("exact_scans", 0) - c0.get("exact_scans", 0)
        kernel = "tsh::scan_list_kernel" if listed > 0 and listed >= scans else \
                 ("tsh::scan_kernel<MASKED>" if listed == 0 else "tsh::scan_list_kernel / tsh::scan_kernel<MASKED>")
        if exact > 0 and exact >= scans:  # few enough kept rows: their exact f64 sums in one launch, no f32 keys
            kernel = "tsh::exact_scan_kernel (+ exact_select_kernel: two dispatches per query)"
        useful = float(kept) * d * 4 + (4.0 * kept if listed > 0 else n / 8)  # (the list's ids instead of the mask's words)
        ent = {"value": cnt / el, "unit": "queries/s", "ms_per_step": el / cnt * 1e3, "timing": "median call",
               "ms_per_step_mean": el_mean / 64 * 1e3, "kept_rows": kept, "mask": kind,
               "roofline": {"bound": "hbm", "achieved": useful / (scan_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": useful / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "kernel": kernel, "kernel_us": scan_us, "list_scans": int(listed), "exact_scans": int(exact),
                            "scan_launches": int(scans),
                            "kernel_us_source": "HIP events around back-to-back launches of the one kernel (tsh_bench_scan)",
                            "algorithmic_bytes_per_launch": useful},
               "fallback_searches": c1["fallback_searches"] - c0["fallback_searches"]}
        prof = profiled_kernel_us("C5.keep_%g%%%s" % (keep * 100, "_range" if kind == "range" else ""))
        if prof and prof.get("avg_us"):  # the same kernel inside real searches, under rocprofv3 (committed summary)
            ent["roofline"]["kernel_us_rocprofv3"] = prof["avg_us"]
            ent["roofline"]["frac_rocprofv3"] = useful / (prof["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            ent["roofline"]["rocprofv3_source"] = prof.get("source")
        tail = tuple(np.concatenate([g[j] for g in got])[cnt - m:] for j in range(3)) if m else None
        # the same 64-query calls with the library's own choice of path (cost model: for a call of this size it
        # scores all queries in one matrix-core pass -- over the kept rows' gathered copy behind a selective mask, over the
        # shard with the mask in the epilogue otherwise)
        tail2 = None
        try:
            idx.set_batch_min_nq(1)
            idx.search(queries[sel[:64]], k, None, mask)
            env.fence()
            c0 = idx.counters()
            per_call = []
            with timed_region():
                t0 = time.perf_counter()
                for g in range(cnt // 64):
                    tc = time.perf_counter()
                    got[g] = idx.search(queries[sel[g * 64:g * 64 + 64]], k, None, mask)  # (synchronous: results are back)
                    per_call.append(time.perf_counter() - tc)
                env.fence()
                el2 = time.perf_counter() - t0
            c1 = idx.counters()
            # the MEDIAN call, like side.C3's median step: sixteen calls of half a millisecond, and one of them frozen for
            # the rest of a cgroup CPU period (host_cpu.throttled_*) moved the mean by a third in one run of this round
            med = sorted(per_call)[len(per_call) // 2]
            ent["library_default_path"] = {"value": 64 / med, "unit": "queries/s", "ms_per_step": med / 64 * 1e3,
                                           "timing": "median call", "ms_per_step_mean": el2 / cnt * 1e3,
                                           "ms_per_call_max": max(per_call) * 1e3,
                                           "batch_launches": c1["batch_launches"] - c0["batch_launches"],
                                           "scan_launches": c1["scan_launches"] - c0["scan_launches"]}
            tail2 = tuple(np.concatenate([g[j] for g in got])[cnt - m:] for j in range(3)) if m else None
        except Exception as e:  # noqa: BLE001
            ent["library_default_path"] = {"error": repr(e)}
        finally:
            idx.set_batch_min_nq(0)
        # the same mask as a HANDLE (tsh_mask_create: uploaded once, its rows listed on the device, resident): the same
        # pipelined 64-query calls, and one query at a time -- where the pointer form's host passes over the bitmap
        # (slice, count, list: every call) are more than half of what the caller waits for
        tail3 = None
        try:
            with idx.make_mask(mask) as mh:
                idx.search(queries[sel[:64]], k, None, mh)
                env.fence()
                per_call = []
                with timed_region():
                    t0 = time.perf_counter()
                    for g in range(cnt // 64):
                        tc = time.perf_counter()
                        got[g] = idx.search(queries[sel[g * 64:g * 64 + 64]], k, None, mh)
                        per_call.append(time.perf_counter() - tc)
                    env.fence()
                    el3_mean = (time.perf_counter() - t0) / (cnt // 64)
                    el3 = sorted(per_call)[len(per_call) // 2] * (cnt // 64)  # (the median call, as the leg above)
                    lone = {}
                    for form, arg in (("pointer", mask), ("handle", mh)):
                        lat = []
                        for i in range(200):
                            t1 = time.perf_counter()
                            idx.search(queries[sel[i]], k, None, arg)
                            lat.append(time.perf_counter() - t1)
                        lat = np.sort(np.asarray(lat)) * 1e6
                        lone[form] = {"p50": float(lat[len(lat) // 2]), "p99": float(lat[int(len(lat) * 0.99)])}
                ent["mask_handle"] = {"value": cnt / el3, "unit": "queries/s", "ms_per_step": el3 / cnt * 1e3,
                                      "timing": "median call", "ms_per_step_mean": el3_mean / 64 * 1e3, "one_at_a_time_us": lone}
                tail3 = tuple(np.concatenate([g[j] for g in got])[cnt - m:] for j in range(3)) if m else None
        except Exception as e:  # noqa: BLE001
            ent["mask_handle"] = {"error": repr(e)}
        if host_rows is not None and m:
            import bench_check

            ref = bench_check.oracle_answers(host_rows, queries[sel[cnt - m:]], metric, k, mask)
            ent["recall_at_k"], ent["ids_and_distances_bit_exact"] = bench_check.compare(tail, ref)
            ent["checked_queries"] = m
            if tail2 is not None:
                ent["library_default_path"]["ids_and_distances_bit_exact"] = bench_check.compare(tail2, ref)[1]
            if tail3 is not None:
                ent["mask_handle"]["ids_and_distances_bit_exact"] = bench_check.compare(tail3, ref)[1]
        out["keep_%d%%%s" % (round(keep * 100), "" if kind == "bernoulli" else "_" + kind)] = ent
    return out


def side_c1(env, with_oracle):
    """C1: the reference's own CPU-runnable size, 10k x 128, L2, k = 10, one query at a time."""
    n, d, k, metric = 10_000, 128, 10, 0
    idx, host_rows = env.build_index(d, metric, n, 0, n, keep_host=with_oracle)
    try:
        qs = make_queries(1000, d, metric, seed=20260616)
        idx.set_batch_min_nq(0)
        for i in range(50):
            idx.search(qs[i], k)
        lat = []
        with timed_region():
            for i in range(1000):
                t1 = time.perf_counter()
                idx.search(qs[i], k)
                lat.append(time.perf_counter() - t1)
            lat = np.sort(np.asarray(lat)) * 1e6
            env.fence()
            t0 = time.perf_counter()
            for g0 in range(0, 1000, 64):
                idx.search(qs[g0:g0 + 64], k)
            env.fence()
            el = time.perf_counter() - t0
            scan_us = idx.bench_scan(qs[0], iters=50)
        out = {"workload": "C1: %dx%d f32, L2, k=%d, single query" % (n, d, k),
               "latency_us": {"p50": float(lat[len(lat) // 2]), "p99": float(lat[int(len(lat) * 0.99)]),
                              "mean": float(lat.mean()), "queries": len(lat)},
               "value": 1000 / el, "unit": "queries/s (pipelined, 64 per call)", "scan_kernel_us": scan_us}
        if host_rows is not None:
            import bench_check

            got = idx.search(qs, k)
            ref = bench_check.oracle_answers(host_rows, qs, metric, k)
            out["recall_at_k"], out["ids_and_distances_bit_exact"] = bench_check.compare(got, ref)
            out["checked_queries"] = len(qs)
            out.update(bench_check.c1_cpu_context(host_rows, qs, metric, k, ref[0]))
        return out
    finally:
        idx.close()


def side_c3(env, a, with_oracle):
    """C3: 1M x 768 cosine, k = 100, 1024-query batches on the matrix cores: the default key kernel and
    the f32-MFMA variant."""
    n, d, k, metric, nq = a.rows, a.dim, a.k, 2, 1024
    idx, host_rows = env.build_index(d, metric, n, 0, n, keep_host=with_oracle)
    try:
        out = {"workload": "C3: %dx%d f32, cosine, k=%d, %d-query batch (matrix-core path)" % (n, d, k, nq)}
        r = measure_batch(env, idx, host_rows, metric, n, d, k, nq, 10, 2, 3, a.c3_check)
        out.update(r)
        try:
            out["f32_mfma_variant"] = measure_batch(env, idx, host_rows, metric, n, d, k, nq, 3, 1, 0, 16)
        except Exception as e:  # noqa: BLE001
            out["f32_mfma_variant"] = {"error": repr(e)}
        small = {}
        for m in (16, 128):
            try:
                rr = measure_batch(env, idx, None, metric, n, d, k, m, 20, 3, 3, 0)
                small["%d_queries" % m] = {"value": rr["value"], "ms_per_step": rr["ms_per_step"]}
            except Exception as e:  # noqa: BLE001
                small["%d_queries" % m] = {"error": repr(e)}
        out["smaller_calls"] = small
        return out
    finally:
        idx.close()


# ------------------------------------------------------------------ main flow
def make_env(a):
    """The device side.  TSH_BENCH_ENV=module:Class (tests only) substitutes a stand-in, so that the rank processes
    `bench.py --gpus N` starts can run on a machine without a GPU (tests/test_bench_launcher.py)."""
    hook = os.environ.get("TSH_BENCH_ENV")
    if not hook:
        return Env(a)
    import importlib

    mod, cls = hook.split(":")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    return getattr(importlib.import_module(mod), cls)(a)


TIMELINE_MAIN = ("reserve_us", "pre_enqueue_us", "wait_scan_us", "exchange_wait_us", "merge_us", "result_gather_us",
                 "copy_out_us", "retry_scan_us")  # the calling thread's phases of tsh_search_sharded: they add up to call_us


def exchange_timeline(per_rank, steps_total, ms_per_step):
    """The N > 1 line's account of where the time went: every rank's tsh_comm_timeline over the timed regions
    (sums of microseconds -> ms per step), the calling thread's phases summed and set against ms_per_step, and the
    slowest rank of every phase (a slow first 8-GPU run must say where it was slow)."""
    ranks = []
    for r in per_rank:
        t = r.get("timeline")
        if not t:
            ranks.append({"rank": r["rank"], "timeline": None, "host_cpu": r.get("host_cpu")})
            continue
        ent = {"rank": r["rank"], "calls": t["calls"], "groups": t["groups"], "retries": t["retries"],
               "host_cpu": r.get("host_cpu")}
        for key in ("call_us",) + TIMELINE_MAIN + ("scan_us", "gather_us", "slice_d2h_us"):
            ent[key.replace("_us", "_ms_per_step")] = t[key] * 1e-3 / max(steps_total, 1)
        ent["phases_sum_ms_per_step"] = sum(t[key] for key in TIMELINE_MAIN) * 1e-3 / max(steps_total, 1)
        if "closing_fence_ms_per_step" in r:
            ent["closing_fence_ms_per_step"] = r["closing_fence_ms_per_step"]
            ent["harness_ms_per_step"] = r["run_ms_per_step"] - ent["call_ms_per_step"]  # Python around the library call
            # what this rank can account for of a step: its phases inside the library + the region's closing fence
            ent["accounted_ms_per_step"] = ent["phases_sum_ms_per_step"] + r["closing_fence_ms_per_step"]
        ranks.append(ent)
    have = [e for e in ranks if e.get("calls") is not None]
    out = {"unit": "ms per step (sums over the timed regions / timed steps), per rank",
           "phases": "calling thread: reserve | pre_enqueue (the exchange's launch, ahead of its blocks) | wait_scan | exchange_wait (= block all-gather + this rank's slice to the "
                     "host) | merge | result_gather | copy_out (+ retry_scan); scan runs beside them on a helper "
                     "thread; gather + slice_d2h are the device-side split of exchange_wait",
           "ms_per_step": ms_per_step, "ranks": ranks}
    if have:
        worst = max(have, key=lambda e: e["call_ms_per_step"])
        out["slowest_rank"] = worst["rank"]
        out["phases_sum_ms_per_step"] = worst["phases_sum_ms_per_step"]
        out["call_ms_per_step"] = worst["call_ms_per_step"]
        out["phases_sum_over_ms_per_step"] = worst["phases_sum_ms_per_step"] / ms_per_step if ms_per_step > 0 else None
        if "accounted_ms_per_step" in worst and ms_per_step > 0:
            # (ms_per_step is the MEDIAN region, the accounts are means over all regions)
            acc = [e["accounted_ms_per_step"] / ms_per_step for e in have if "accounted_ms_per_step" in e]
            out["accounted_over_ms_per_step"] = {"rank0": acc[0], "min": min(acc), "max": max(acc)}
        out["max_over_ranks"] = {key: max(e[key] for e in have) for key in have[0] if key.endswith("_ms_per_step")}
    return out


def measure_single(env, a, idx, searcher, queries, k, row_mask, steps, warmup, repeats):
    """The headline measurement: W warm-up steps, then `repeats` timed regions of exactly `steps` single-query
    searches each, every region between a barrier + device synchronise on both sides, MAX over ranks.
    -> dict(regions, scan_us, scan samples, per-rank records on rank 0, cgroup stats)."""
    nqp = len(queries)
    py_groups = getattr(searcher, "python_groups", True)  # False: the library forms the groups (tsh_search_sharded)
    batches = {}

    def one(i):
        q = queries[i % nqp]
        if searcher is not None:
            return searcher.search(q, k, None, row_mask)
        return idx.search(q, k, None, row_mask)

    def run(first, count):
        """`count` single-query searches, several of them in flight.  Each query still streams the
        whole (shard of the) corpus on its own; only the select / re-rank / copy tail of one query
        overlaps the scan of the next."""
        if count <= 0:
            return
        if a.inflight <= 1:
            for i in range(count):
                one(first + i)
        elif searcher is not None:
            # N > 1: groups of queries share one all-gather + one merge call, and the next group's
            # shard scans run while this group is exchanged and merged
            qb = batches.get((first, count))
            if qb is None:
                qb = queries[[(first + j) % nqp for j in range(count)]]
            searcher.search_many(qb, k, None, row_mask, group=sharded_group(a.group, count) if py_groups else 0)
        elif a.group > 0:
            for g0 in range(0, count, a.group):
                sel = [(first + g0 + j) % nqp for j in range(min(a.group, count - g0))]
                idx.search(queries[sel], k, None, row_mask)
        else:
            from collections import deque
            pend = deque()
            for i in range(count):
                if len(pend) == a.inflight:
                    idx.wait(pend.popleft())
                pend.append(idx.submit(queries[(first + i) % nqp], k, row_mask))
            while pend:
                idx.wait(pend.popleft())

    run(0, warmup)
    regions = []
    if searcher is not None and a.inflight > 1:  # the regions' query batches exist before their clocks start
        for r in range(repeats):
            first = warmup + r * steps
            batches[(first, steps)] = np.ascontiguousarray(queries[[(first + j) % nqp for j in range(steps)]])
    env.fence()
    has_tl = hasattr(searcher, "timeline")
    if has_tl:
        searcher.timeline(reset=True)
    cg0, t_cg0 = cgroup_cpu_stat(), time.perf_counter()
    cpu0 = time.process_time()
    c0 = idx.counters()
    run_s = fence_s = 0.0
    with timed_region():
        for r in range(repeats):
            env.fence()
            t0 = time.perf_counter()
            run(warmup + r * steps, steps)
            t_run = time.perf_counter()
            env.fence()  # (inside the region, as the contract asks: barrier + device synchronise)
            t_end = time.perf_counter()
            run_s += t_run - t0
            fence_s += t_end - t_run
            regions.append(env.reduce_max(t_end - t0))
    c1 = idx.counters()
    cpu1 = time.process_time()
    cg1, t_cg1 = cgroup_cpu_stat(), time.perf_counter()
    ns = c1["scan_us_samples"] - c0["scan_us_samples"]
    scan_us = (c1["scan_us_sum"] - c0["scan_us_sum"]) / ns if ns > 0 else float("nan")
    mine = {"rank": env.rank, "timeline": searcher.timeline() if has_tl else None,
            "host_cpu": {"cpus_busy": (cpu1 - cpu0) / max(t_cg1 - t_cg0, 1e-9), "cpu_s": cpu1 - cpu0,
                         "wall_s": t_cg1 - t_cg0, "note": "this rank's process, all threads, over its timed regions"},
            "scan_us": scan_us,
            # this rank's timed regions, split at the end of its last search: the closing fence (waiting for the other
            # ranks + the device) is part of every region
            "run_ms_per_step": run_s * 1e3 / max(steps * repeats, 1),
            "closing_fence_ms_per_step": fence_s * 1e3 / max(steps * repeats, 1)}
    per_rank = env.gather_objects(mine) if env.world > 1 else [mine]
    host_cpu = None
    if cg0 and cg1:  # host side of the timed regions, all ranks together: CPUs busy, and whether the quota throttled
        host_cpu = {"cpus_busy": (cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) / 1e6 / max(t_cg1 - t_cg0, 1e-9),
                    "throttled_periods": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                    "throttled_ms": (cg1.get("throttled_usec", cg1.get("throttled_time", 0)) -
                                     cg0.get("throttled_usec", cg0.get("throttled_time", 0))) / 1e3,
                    "wall_s": t_cg1 - t_cg0}
    return {"regions": regions, "scan_us": scan_us, "scan_samples": int(ns), "per_rank": per_rank, "host_cpu": host_cpu,
            "one": one, "py_groups": py_groups}


def side_c4_per_rank(env, a):
    """N > 1: BASELINE.json's C4 at its per-rank shape -- every rank holds --c4-rows-per-rank (1.25 M) x 1536 rows,
    inner product, k = 100, so 8 ranks are the whole 10 M x 1536 configuration and fewer ranks the same per-GPU load
    (the WEAK-scaling point beside the headline's strong one).  Single queries through the exchange like the headline,
    then 1024-query calls (every shard on its matrix cores).  Collective: every rank runs it."""
    d, k, metric = 1536, 100, METRICS["ip"]
    per = int(a.c4_rows_per_rank)
    n = per * env.world
    lo, hi = env.rank * per, (env.rank + 1) * per
    # this rank's shard is the one step of the leg that can fail on one rank alone (7.7 GB of rows): agree on it before
    # any rank enters the leg's collectives -- a rank waiting in one for a rank that gave up would hang the whole job
    # (and the headline with it) until the launcher's timeout
    idx, err = None, 0.0
    try:
        idx, _ = env.build_index(d, metric, n, lo, hi, keep_host=False)
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("[bench] rank %d: C4 per-rank shard not built: %r\n" % (env.rank, e))
        err = 1.0
    if env.reduce_max(err) > 0.0:
        if idx is not None:
            idx.close()
        return {"error": "a rank could not build its %d x %d shard: leg skipped on every rank" % (per, d)} if env.rank == 0 else None
    searcher = None
    try:
        steps, warmup = max(a.steps, 20), min(max(a.warmup, 2), 10)
        queries = make_queries(max(1024, steps + warmup), d, metric, seed=20260618)
        idx.set_batch_min_nq(0)  # (before the first sharded call: the ranks tell each other whether they batch, library_schedule)
        searcher = env.searcher(idx)
        repeats = max(3, min(10, auto_repeats(steps, a.repeats)))
        m = measure_single(env, a, idx, searcher, queries, k, None, steps, warmup, repeats)
        elapsed = float(np.median(m["regions"]))
        scan_us = env.reduce_max(m["scan_us"] if math.isfinite(m["scan_us"]) else idx.bench_scan(queries[0], iters=10))
        shard_bytes = float(hi - lo) * d * 4
        out = None
        # 1024 queries per call: the library hands each shard's share to its matrix cores
        idx.set_batch_min_nq(1)
        bt = []
        searcher.search(queries[:1024], k)
        if hasattr(searcher, "timeline"):
            searcher.timeline(reset=True)
        run_s = fence_s = 0.0
        with timed_region():
            for i in range(3):
                env.fence()
                t0 = time.perf_counter()
                got_b = searcher.search(queries[:1024], k)
                t_run = time.perf_counter()
                env.fence()
                run_s += t_run - t0
                fence_s += time.perf_counter() - t_run
                bt.append(env.reduce_max(time.perf_counter() - t0))
        mine_b = {"rank": env.rank, "timeline": searcher.timeline() if hasattr(searcher, "timeline") else None,
                  "run_ms_per_step": run_s * 1e3 / 3, "closing_fence_ms_per_step": fence_s * 1e3 / 3}
        per_rank_b = env.gather_objects(mine_b)
        idx.set_batch_min_nq(0)
        got_s = searcher.search(queries[:2], k)
        if env.rank == 0:
            ms = elapsed / steps * 1e3
            out = {"workload": "C4 per-rank shape: %d ranks x %d x %d f32, ip, k=%d (= %dx%d in all), single query per step"
                               % (env.world, per, d, k, n, d),
                   "value": steps / elapsed, "unit": "queries/s", "scaling": "weak", "n_gpus": env.world, "steps": steps,
                   "warmup": warmup, "ms_per_step": ms,
                   "timed_regions": {"count": repeats, "seconds": [float(x) for x in m["regions"]]},
                   "roofline": {"bound": "hbm", "achieved": shard_bytes / (scan_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": shard_bytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                "kernel": "tsh::scan_kernel", "kernel_us": scan_us, "algorithmic_bytes_per_launch": shard_bytes,
                                "note": "per rank: its shard's bytes over its scan's duration"},
                   "exchange_timeline": exchange_timeline(m["per_rank"], steps * repeats, ms),
                   "batch_1024": {"value": 1024 / float(np.median(bt)), "unit": "queries/s",
                                  "ms_per_call": float(np.median(bt)) * 1e3, "calls": len(bt),
                                  "exchange_timeline": exchange_timeline(per_rank_b, len(bt), float(np.median(bt)) * 1e3)},
                   "single_and_batched_agree": bool(np.array_equal(got_s[0], got_b[0][:2]) and
                                                    np.array_equal(got_s[1], got_b[1][:2]))}
        # parity on sampled queries: rank 0 regenerates the corpus chunk by chunk for the oracle, all ranks search
        n_chk = 0
        if env.rank == 0 and not a.no_cpu_baseline:
            n_chk = max(1, min(2, int(a.cpu_seconds * 6e9 / (float(n) * d * 4))))
        n_chk = env.bcast_int(n_chk)
        if n_chk:
            got = searcher.search(queries[:n_chk], k)
            if env.rank == 0:
                import bench_check

                ref = bench_check.oracle_topk_stream(env.oracle_chunks(n, d, metric), queries[:n_chk], metric, k)
                out["recall_at_k"], out["ids_and_distances_bit_exact"] = bench_check.compare(got, ref)
                out["recall_queries"] = n_chk
        return out
    finally:
        if searcher is not None and hasattr(searcher, "close"):
            searcher.close()
        idx.close()


def side_shard_of_8(env, a, c2_ms_per_step, shape=None):
    """N = 1 line: the headline's PER-RANK load of an 8-GPU run -- rows [0, rows / 8) of the same corpus, same metric
    and k -- through tsh_search_sharded over real RCCL in a world of one, in the driver's own shape (timed regions of
    --steps single-query steps per call, fenced on both sides).  What one GPU of eight has to do per step, collective
    launch included; what it cannot show is the wait for seven peers.  upper_bound_speedup = the headline's
    ms_per_step / this leg's: the 8-GPU speed-up if the real collective cost no more than the one-rank one.
    shape: another configuration's share instead of the headline's -- side.C4_shard_of_8 is BASELINE.json's C4
    (10 M x 1536, inner product, k = 100, rows split over 8 GPUs) at one rank's 1.25 M x 1536 rows: the same calls,
    the oracle over that shard on sampled queries (the shard is 7.7 GB: regenerated chunk by chunk for it, no host
    copy), and a 1024-query call on the shard's matrix cores."""
    d, k, metric = a.dim, a.k, METRICS[a.metric]
    n, cfg, mname = a.rows, a.config.upper(), a.metric
    if shape:
        n, d, k, mname, cfg = shape["rows"], shape["dim"], shape["k"], shape["metric"], shape["config"]
        metric = METRICS[mname]
    per = (n + 7) // 8
    big = float(per) * d * 4 > 4e9
    # (no host copy of another configuration's shard -- C4's is 7.7 GB: the oracle sees it chunk by chunk, regenerated)
    idx, host = env.build_index(d, metric, n, 0, per, keep_host=not a.no_cpu_baseline and not shape)
    cs = None
    try:
        idx.set_batch_min_nq(0)  # every query scans the shard on its own, as in the headline
        cs = env.shard_comm(idx)
        steps, warmup = a.steps, min(max(a.warmup, 2), 50)
        repeats = auto_repeats(steps, a.repeats)
        queries = make_queries(max(64, steps * min(repeats, 8) + warmup), d, metric, seed=20260619)
        nqp = len(queries)

        def batch_of(first, count):
            return np.ascontiguousarray(queries[[(first + j) % nqp for j in range(count)]])

        def call(first, count, group=0):
            return cs.search_many(batch_of(first, count), k, None, None, group=group)

        def regions_of(group, reps):
            call(0, max(warmup, 1), group)
            call(0, steps, group)
            cs.timeline(reset=True)
            c0 = idx.counters()
            out = []
            batches = [batch_of(warmup + r * steps, steps) for r in range(reps)]  # (the inputs exist before the clock starts)
            with timed_region():
                for r in range(reps):
                    env.fence()
                    t0 = time.perf_counter()
                    cs.search_many(batches[r], k, None, None, group=group)
                    env.fence()
                    out.append(time.perf_counter() - t0)
            c1 = idx.counters()
            ns = c1["scan_us_samples"] - c0["scan_us_samples"]
            scan_us = (c1["scan_us_sum"] - c0["scan_us_sum"]) / ns if ns > 0 else float("nan")
            return out, cs.timeline(), scan_us

        regions, tl, scan_us = regions_of(0, repeats)
        med = float(np.median(regions))
        ms = med / steps * 1e3
        mine = {"rank": 0, "timeline": tl}
        shard_bytes = float(per) * d * 4
        out = {"workload": "one rank's share of %s at N = 8: %dx%d f32, %s, k=%d, %d single-query steps per "
                           "tsh_search_sharded call, RCCL in a world of one" % (cfg, per, d, mname, k, steps),
               "steps": steps, "warmup": warmup, "us_per_query": ms * 1e3, "ms_per_step": ms,
               "timed_regions": {"count": repeats, "steps_each": steps, "reported": "median",
                                 "seconds": [float(x) for x in regions]},
               "upper_bound_speedup": c2_ms_per_step / ms if ms > 0 and c2_ms_per_step else None,
               "upper_bound_note": "headline ms_per_step / this leg's: what 8 GPUs reach if the 8-rank collective costs "
                                   "no more than the 1-rank one and no rank waits for another" if c2_ms_per_step else
                                   "the whole configuration's one-GPU step is not measured in this run (61 GB of rows): none",
               "queries_per_exchange": library_schedule(steps, per, d),
               "queries_per_exchange_source": "bench.py's mirror of the library's schedule (sharded_schedule)",
               "exchange_timeline": exchange_timeline([mine], steps * repeats, ms),
               "roofline": {"bound": "hbm", "kernel": "tsh::scan_kernel", "kernel_us": scan_us,
                            "algorithmic_bytes_per_launch": shard_bytes, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "achieved": shard_bytes / (scan_us * 1e-6) / 1e9 if scan_us == scan_us else None,
                            "frac": shard_bytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBS if scan_us == scan_us else None,
                            "scans_side_by_side": 2 if (per + 63) // 64 < 6144 else 1,
                            "note": "shards below 6144 tiles alternate their scans between two streams: a launch's own "
                                    "duration is about twice its share of the HBM time"},
               "floor_us_per_query": shard_bytes / (HBM_PEAK_GBS * 1e9) * 1e6}
        # the library's group choice against its neighbours (same queries, fewer regions)
        sweep = {}
        for g in sorted(({1, 2, 4, 5, 10, steps} if not shape else {5, 10, steps}) - {0}):
            if g > steps:
                continue
            rg, _, _ = regions_of(g, max(3, min(repeats, 7)))
            sweep[str(g)] = float(np.median(rg)) / steps * 1e6
        out["group_sweep_us_per_query"] = sweep
        n_chk = min(steps, 16) if not big else min(steps, 8)
        got = call(0, n_chk)
        if shape:  # 1024 queries per call: the shard's share goes to its matrix cores, through the same entry point
            try:
                idx.set_batch_min_nq(1)
                qb = np.ascontiguousarray(queries[[j % nqp for j in range(1024)]])
                cs.search(qb, k)
                bt = []
                with timed_region():
                    for _ in range(3):
                        env.fence()
                        t0 = time.perf_counter()
                        got_b = cs.search(qb, k)
                        env.fence()
                        bt.append(time.perf_counter() - t0)
                    gemm_us, flops = idx.bench_batch(qb, k, iters=2)
                ran = idx.counters()["batch_kernel_last"]
                out["batch_1024"] = {"value": 1024 / float(np.median(bt)), "unit": "queries/s",
                                     "ms_per_call": float(np.median(bt)) * 1e3, "calls": len(bt),
                                     "batch_kernel": BATCH_KERNEL_NAME.get(ran, str(ran)),
                                     "roofline": batch_roofline(ran, gemm_us, flops),
                                     "single_and_batched_agree": bool(all(np.array_equal(got[j], got_b[j][:n_chk]) for j in range(3)))}
            except Exception as e:  # noqa: BLE001
                out["batch_1024"] = {"error": repr(e)}
            finally:
                idx.set_batch_min_nq(0)
        if not a.no_cpu_baseline:
            import bench_check

            sel = [j % nqp for j in range(n_chk)]
            chunks = [(0, host)] if host is not None else env.oracle_chunks(n, d, metric, 0, per)
            ref = bench_check.oracle_topk_stream(chunks, queries[sel], metric, k)
            out["recall_at_k"], out["ids_and_distances_bit_exact"] = bench_check.compare(got, ref)
            out["checked_queries"] = len(sel)
        return out
    finally:
        if cs is not None and hasattr(cs, "close"):
            cs.close()
        idx.close()


def run_bench(a, env=None):
    env = env or make_env(a)
    metric = METRICS[a.metric]
    world, rank = env.world, env.rank
    n, d, k = a.rows, a.dim, a.k

    # ---- resident corpus: this rank generates and holds its own row range only -------
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    # a host copy for the CPU baseline / recall check, unless the corpus is too big for one (C4: 61 GB): then the
    # oracle sees it chunk by chunk, as in a sharded run
    big = float(n) * d * 4 > 16e9
    want_host = rank == 0 and world == 1 and not a.no_cpu_baseline and not big
    # --in-process: this one process holds all --gpus devices' shards in one handle (no ranks, no collective)
    n_dev = a.gpus if getattr(a, "in_process", False) and a.gpus > 1 else 1
    if n_dev > 1:
        idx, host_rows = env.build_index(d, metric, n, lo, hi, keep_host=want_host, n_devices=n_dev)
    else:
        idx, host_rows = env.build_index(d, metric, n, lo, hi, keep_host=want_host)
    per_dev = ((hi - lo + n_dev - 1) // n_dev + 63) // 64 * 64 if n_dev > 1 else hi - lo  # rows of one device's shard

    if a.batch > 0:
        return bench_batch(a, env, idx, host_rows, metric)

    pool = query_pool_size(a.steps, a.warmup, a.recall_queries if world == 1 else 0)
    queries = make_queries(pool, d, metric)
    nqp = len(queries)
    # headline workload: every query scans the corpus on its own (no MFMA batching) -- set before the first sharded call, whose
    # agreement tells every rank whether the ranks batch (that decides the group schedule of calls of up to 128 queries)
    idx.set_batch_min_nq(0)
    searcher = env.searcher(idx)

    row_mask = None
    if a.mask_keep > 0:  # C5 as the main line
        row_mask, _ = make_mask(n, a.mask_keep, a.mask_kind)

    # the harness's own garbage collector stays out of everything that is timed (a full collection of a Python
    # heap with torch imported takes 37 ms: four timed regions' worth)
    import gc

    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    repeats = auto_repeats(a.steps, a.repeats)
    m = measure_single(env, a, idx, searcher, queries, k, row_mask, a.steps, a.warmup, repeats)
    regions, one, py_groups = m["regions"], m["one"], m["py_groups"]
    elapsed = float(np.median(regions))

    # single-query latency, one at a time (not the headline value).  Round 2's line carried p99 = 42.8 ms: one
    # call in a few hundred coincided with a full (generation 2) collection of THIS harness's Python heap
    # (37 ms with torch imported; tools/attic/lone_stall_probe.py, gc.callbacks) -- the library was not involved.  The
    # collector is held off for the duration of the leg; collections inside it would be reported.
    lat, gc_ms = [], []

    def _gc_cb(phase, info, _t=[0.0]):
        if phase == "start":
            _t[0] = time.perf_counter()
        else:
            gc_ms.append((time.perf_counter() - _t[0]) * 1e3)
    gc.callbacks.append(_gc_cb)
    try:
        with timed_region():
            for i in range(a.lat_queries if world == 1 else min(a.lat_queries, 200)):
                t1 = time.perf_counter()
                one(i)
                lat.append(time.perf_counter() - t1)
    finally:
        if gc_was:
            gc.enable()
        gc.callbacks.remove(_gc_cb)
    lat_raw = np.asarray(lat) * 1e3
    lat = np.sort(lat_raw)

    # ---- roofline of the dominant kernel (K1 scan): HIP events recorded by the library
    # around real scan launches on its pipeline stream, during the timed regions above
    ns, scan_us = m["scan_samples"], m["scan_us"]
    with timed_region():  # (the hook measures ONE shard's kernel: a single-device handle)
        scan_alone_us = idx.bench_scan(queries[0], iters=50, row_mask=row_mask) if hi > lo and n_dev == 1 else float("nan")
    if not math.isfinite(scan_us):
        scan_us = scan_alone_us
    shard_bytes = float(min(per_dev, hi - lo)) * d * 4  # algorithmic: every stored f32 read once (per device: its shard)
    if row_mask is not None:  # C5: only kept rows are read, plus the mask itself
        kept = int(np.unpackbits(row_mask, bitorder="little")[lo:hi].sum())
        shard_bytes = float(kept) * d * 4 + (hi - lo) / 8
    scan_us = env.reduce_max(scan_us)

    # ---- recall + CPU baseline: oracle on rank 0 (bench_check.py), every rank joins the GPU searches ----
    n_cpu, ref, cpu_elapsed, big_cpu = 0, None, 0.0, None
    if rank == 0 and host_rows is not None:
        import bench_check

        ref, cpu_elapsed = bench_check.cpu_baseline_single(host_rows, queries, metric, k, row_mask, a.cpu_seconds)
        n_cpu = len(ref)
    elif rank == 0 and not a.no_cpu_baseline:
        # sharded run (or a corpus too big for a host copy): no process holds the corpus.  Rank 0 regenerates it
        # chunk by chunk (counter-based generator) and runs the exhaustive oracle on a few sampled queries -- a
        # parity check; cpu_baseline is reported at N = 1 only
        import bench_check

        n_cpu = max(2, min(8, int(a.cpu_seconds * 6e9 / (float(n) * d * 4))))
        if world == 1:
            n_cpu = max(n_cpu, 8)  # (--config c4 --gpus 1: at least eight queries against the whole 10 M rows)
        r_ids, r_dist, r_cnt = bench_check.oracle_topk_stream(env.oracle_chunks(n, d, metric), queries[:n_cpu], metric, k,
                                                              row_mask)
        ref = [(r_ids[i, :r_cnt[i]], r_dist[i, :r_cnt[i]]) for i in range(n_cpu)]
        if world == 1:  # big corpus on one GPU: the single-thread baseline on a bounded sample of its rows
            big_cpu = bench_check.cpu_baseline_sample_of_rows(env.oracle_chunks(n, d, metric), n, queries, metric, k,
                                                              row_mask, a.cpu_seconds)
    n_cpu = env.bcast_int(n_cpu)
    got = [one(i) for i in range(n_cpu)]

    out = None
    if rank == 0:
        achieved = shard_bytes / (scan_us * 1e-6) / 1e9
        traffic, traffic_source = pmc_traffic(hi - lo, d) if row_mask is None and n_dev == 1 else (None, None)
        if a.inflight <= 1:
            in_flight = 1
        elif searcher is None and a.group > 0:
            in_flight = min(env.max_inflight(), a.group, max(a.steps, 1))
        elif searcher is None:
            in_flight = a.inflight
        else:
            in_flight = min(env.max_inflight(), sharded_group(a.group, a.steps) if py_groups else library_group(a.steps, hi - lo, d))
        out = {
            "metric": "kNN queries/sec + recall@k, 1Mx768 f32 brute-force" if a.config == "c2" else
                      "kNN queries/sec + recall@k, %dx%d f32 brute-force (%s)" % (n, d, a.config.upper()),
            "value": a.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world if n_dev == 1 else n_dev,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "timed_regions": {"count": repeats, "steps_each": a.steps, "reported": "median",
                              "seconds": [float(x) for x in regions],
                              "value_min": a.steps / max(regions), "value_max": a.steps / min(regions)},
            "config": {"workload": "%s: %dx%d f32, %s, k=%d, single query per step" % (a.config.upper(), n, d, a.metric, k),
                       "rows": n, "dim": d, "k": k, "metric": a.metric, "mask_keep": a.mask_keep or None,
                       "mask_kind": a.mask_kind if a.mask_keep else None,
                       "queries_in_flight": in_flight,
                       "queries_per_call": (min(a.group, a.steps) if a.group else 1) if searcher is None
                       else (sharded_group(a.group, a.steps) if py_groups else library_schedule(a.steps, hi - lo, d)),
                       "note": "every query scans the whole corpus on its own (HBM-bound kernel, no matrix-core "
                               "batching); independent queries are handed over in groups and pipelined",
                       "sharding": "row-range x%d, all-gather of top-k candidate blocks: %s" % (world, env.exchange)
                       if searcher is not None else
                       ("single GPU" if n_dev == 1 else
                        "row-range x%d IN ONE PROCESS (tsh_index_create n_devices = %d): one host thread per shard, candidate "
                        "blocks copied back, host merge -- no collective%s"
                        % (n_dev, n_dev, "; the shards SHARE cuda:0 (rehearsal, not a scaling figure)" if a.shards_share_gpu else "")),
                       "exchange_note": getattr(env, "exchange_note", None),
                       "harness": "python gc held off during the timed legs"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_measured_in_run": False,  # (counters need rocprofv3 passes of their own: a committed figure)
                         "kernel": "tsh::scan_kernel", "kernel_us": scan_us, "kernel_us_samples": int(ns),
                         "kernel_us_back_to_back_alone": scan_alone_us,
                         "algorithmic_bytes_per_launch": shard_bytes},
        }
        if (min(per_dev, hi - lo) + 63) // 64 < 6144 and a.inflight > 1 and n_dev == 1:
            # shards below 6144 tiles alternate their scans between two streams: two scans
            # run side by side, so one launch's own duration is about twice its share of the HBM time
            out["roofline"]["scans_side_by_side"] = 2
            out["roofline"]["achieved_alone"] = shard_bytes / (scan_alone_us * 1e-6) / 1e9
        if searcher is not None:  # N > 1: where every rank's time went, against the step
            out["exchange_timeline"] = exchange_timeline(m["per_rank"], a.steps * repeats, elapsed / a.steps * 1e3)
        if ref is not None:
            import bench_check

            hits, tot, exact = bench_check.compare_refs(got, ref)
            out["recall_at_k"] = hits / max(tot, 1)
            out["recall_queries"] = n_cpu
            out["ids_and_distances_bit_exact"] = exact
            if world == 1 and a.recall_queries > n_cpu and host_rows is not None:
                # recall@k over >= 1000 queries: the oracle's OpenMP form, same
                # per-(query,row) arithmetic, against the GPU answers of the same queries
                rl = bench_check.recall_leg(host_rows, queries, metric, k, row_mask,
                                            lambda qq: idx.search(qq, k, None, row_mask), min(a.recall_queries, nqp),
                                            a.recall_seconds)
                out["recall_at_k"] = rl["recall_at_k"]
                out["recall_queries"] = rl["recall_queries"]
                out["ids_and_distances_bit_exact"] = bool(rl["bit_exact"] and exact)
                out["cpu_baseline_mt_batched"] = rl["cpu_baseline_mt_batched"]
            if world == 1 and host_rows is None:
                out["cpu_baseline"] = {
                    "value": big_cpu[0], "unit": "queries/s", "cores": 1, "kind": "port",
                    "sample": "%d of the same queries over the first %d of the %d rows, oracle/vs_oracle.c single "
                              "thread, scaled by %d / %d (the scan is linear in the rows)" % (big_cpu[1], big_cpu[2], n, big_cpu[2], n)}
            elif world == 1:
                out["cpu_baseline"] = {
                    "value": n_cpu / cpu_elapsed, "unit": "queries/s", "cores": 1, "kind": "port",
                    "sample": "%d of the same queries over the full %dx%d corpus, oracle/vs_oracle.c "
                              "single thread (the reference searches on one isolate)" % (n_cpu, n, d)}
                try:
                    out["cpu_baseline_mt"] = bench_check.cpu_baseline_mt(host_rows, queries, metric, k, row_mask,
                                                                         max(2, min(n_cpu, 8)))
                except Exception:  # noqa: BLE001
                    pass
        out["latency_ms_one_at_a_time"] = {"p50": float(lat[len(lat) // 2]), "p99": float(lat[int(len(lat) * 0.99)]),
                                           "max": float(lat[-1]), "argmax": int(np.argmax(lat_raw)),
                                           "mean": float(lat.mean()), "queries": int(len(lat)),
                                           "harness_gc_collections_inside": len(gc_ms)}
        c = idx.counters()
        if m["host_cpu"]:
            out["host_cpu"] = m["host_cpu"]
            if world > 1:
                out["host_cpu"]["per_rank"] = [r.get("host_cpu") for r in m["per_rank"]]
        out["counters"] = {"fallback_searches": c["fallback_searches"],
                           "candidates_per_query": c["candidates_total"] / max(c["searches"], 1)}

    # ---- C4 on ONE GPU (--config c4 --gpus 1): the 1024-query batched figure of SURVEY section 8d beside the headline
    if world == 1 and rank == 0 and a.config == "c4" and a.mask_keep == 0 and not a.no_side:
        try:
            r = measure_batch(env, idx, None, metric, n, d, k, 1024, 3, 1, 3, 0)
            out["batch_1024"] = {key: r[key] for key in ("value", "unit", "ms_per_step", "ms_per_step_max", "batch_kernel",
                                                           "roofline", "key_passes_share_of_step")}
            if n_cpu and ref is not None:  # the oracle's sampled queries once more, through the batched path
                import bench_check

                idx.set_batch_min_nq(1)
                got_b = idx.search(queries[:n_cpu], k)
                idx.set_batch_min_nq(0)
                _, _, eb = bench_check.compare_refs([tuple(x[i:i + 1] for x in got_b) for i in range(n_cpu)], ref)
                out["batch_1024"]["checked_queries"] = n_cpu
                out["batch_1024"]["ids_and_distances_bit_exact"] = eb
        except Exception as e:  # noqa: BLE001
            out["batch_1024"] = {"error": repr(e)}

    # ---- side legs: the other BASELINE.json configurations, same process, after the headline ----
    if world == 1 and rank == 0 and not a.no_side and a.mask_keep == 0 and searcher is None and a.config == "c2" and n_dev == 1:
        side = {}
        legs = [s.strip() for s in a.side.split(",") if s.strip()]
        t_side = time.perf_counter()
        if "c5" in legs:
            try:
                side["C5"] = side_c5(env, idx, host_rows, queries, metric, n, d, k, a.c5_check)
            except Exception as e:  # noqa: BLE001
                side["C5"] = {"error": repr(e)}
        idx.close()
        idx = None
        host_rows = None
        if "s8" in legs:
            try:
                side["shard_of_8"] = side_shard_of_8(env, a, elapsed / a.steps * 1e3)
            except Exception as e:  # noqa: BLE001
                side["shard_of_8"] = {"error": repr(e)}
        if "c4s8" in legs and a.c4_rows_per_rank > 0:
            try:
                side["C4_shard_of_8"] = side_shard_of_8(env, a, None, {"config": "C4", "rows": 8 * int(a.c4_rows_per_rank),
                                                                         "dim": 1536, "k": 100, "metric": "ip"})
            except Exception as e:  # noqa: BLE001
                side["C4_shard_of_8"] = {"error": repr(e)}
        if "c1" in legs:
            try:
                side["C1"] = side_c1(env, not a.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                side["C1"] = {"error": repr(e)}
        if "c3" in legs:
            try:
                side["C3"] = side_c3(env, a, not a.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001
                side["C3"] = {"error": repr(e)}
        side["seconds"] = time.perf_counter() - t_side
        out["side"] = side
    # ---- N > 1: the weak-scaling point (C4's per-rank shape), every rank takes part ----
    if world > 1 and searcher is not None and not a.no_side and a.c4_rows_per_rank > 0 and a.mask_keep == 0 \
            and a.config == "c2":
        if hasattr(searcher, "close"):
            searcher.close()
        idx.close()
        idx = None
        t_side = time.perf_counter()
        try:
            leg = side_c4_per_rank(env, a)
        except Exception as e:  # noqa: BLE001 -- (a rank failing alone here leaves the others in a collective: the
            leg = {"error": repr(e)}  # launcher's timeout ends the job; the headline is then lost with it)
            raise
        if rank == 0:
            leg["seconds"] = time.perf_counter() - t_side
            out["side"] = {"C4_per_rank": leg}
    env.finish()
    if idx is not None:
        idx.close()
    return dumps(out) if rank == 0 else None


def cgroup_cpu_stat():
    """cgroup cpu.stat of this container (usage_usec, nr_throttled, throttled_usec ...): all ranks share it."""
    d = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for ln in open(path):
                key, v = ln.split()
                d[key] = int(v)
            break
        except (OSError, ValueError):
            pass
    return d


def host_cpus():
    """CPUs this container may use: the cgroup quota when there is one, else the CPU count."""
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(q) // int(period))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv, timeout):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, with the
    environment torch.distributed.run would give them.  -> (return code, rank 0's stdout).  A rank that fails
    takes the others down (they would wait in a collective forever); so does the timeout."""
    import subprocess

    n = a.gpus
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", TSH_BENCH_RANK_PROCESS="1")
    # torch sizes its CPU thread pool by the hardware threads it sees (256 on an MI355X box), not by the container's
    # quota: N ranks waking 256 OpenMP threads each for every small host-side copy burn the whole quota and get the
    # job throttled (tools/attic/thread_cpu_probe.py: 300 threads per rank, 2.3 CPUs per rank doing nothing).
    # torch.distributed.run sets OMP_NUM_THREADS=1 for the same reason.
    if "OMP_NUM_THREADS" not in base:
        base["OMP_NUM_THREADS"] = str(max(1, host_cpus() // n))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0, rc = time.time(), None
    while rc is None:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            rc = next(c for c in codes if c not in (None, 0))
        elif all(c == 0 for c in codes):
            rc = 0
        elif time.time() - t0 > timeout:
            sys.stderr.write("[bench] %d ranks still running after %.0f s: giving up\n" % (n, timeout))
            rc = 124
        else:
            time.sleep(0.05)
    for p in procs:  # exactly the processes started above
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    reader.join(timeout=5)
    return rc, (out0[0] if out0 else b"").decode("utf-8", "replace")


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.fake_rccl and not os.environ.get("TSH_RCCL_LIB"):  # before any rank is started / the library is loaded
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import fake_rccl

        os.environ["TSH_RCCL_LIB"] = fake_rccl.build()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.force_sharded and not a.in_process:
        # started plainly: be the launcher.  Rank 0's line is handed through only when the whole job succeeded.
        rc, line = launch_ranks(a, argv, a.launch_timeout)
        if rc != 0 and a.exchange == "auto":
            sys.stderr.write("[bench] the run failed (rc %d): once more with --exchange torch\n" % rc)
            rc, line = launch_ranks(a, argv + ["--exchange", "torch"], a.launch_timeout)
        lines = [ln for ln in line.splitlines() if ln.strip()]
        if rc == 0 and len(lines) == 1:
            sys.stdout.write(lines[0] + "\n")
            sys.stdout.flush()
            return 0
        sys.stderr.write("[bench] no result: rc %d, %d line(s) from rank 0\n" % (rc, len(lines)))
        return rc or 1
    # stdout must carry exactly ONE JSON line: park fd 1 on stderr while libraries
    # (RCCL prints a version banner) run, and write the line to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # fd 1 stays parked on stderr until the process exits: RCCL's banner sits in C stdio's buffer
    # and is only flushed at exit, i.e. after the JSON line
    line = run_bench(a)
    sys.stdout.flush()
    if line is not None:
        os.write(real_stdout, (line + "\n").encode())
    os.close(real_stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
