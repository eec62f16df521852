#!/usr/bin/env python
"""bench.py -- BASELINE.json headline metric: kNN queries/sec (+ recall@k) on
1M x 768 f32 brute force, L2, k=100, single query per step (config C2).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one single-query search through the C-ABI: K1 scan (reads every stored
row once, the query rides in the kernel arguments), K2 select, K4 f64 re-rank
(results stored straight into pinned host memory), host merge.  Independent
queries are handed to the library in groups (--group, default 64) and the
library keeps 8 of them in flight: scans run back to back, a query's
select/re-rank overlap the next scan on reserved CUs.  The corpus is resident
in HBM before the timed region.  N > 1: the SAME corpus is row-range sharded
over the ranks (strong scaling); every rank scans its shard, candidate blocks
are all-gathered over RCCL, every rank merges.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--metric", default="l2", choices=["l2", "ip", "cosine"])
    ap.add_argument("--inflight", type=int, default=8,
                    help="independent single-query searches kept in flight (1 = strictly one at a time)")
    ap.add_argument("--group", type=int, default=64,
                    help="N=1: hand the library this many independent queries per call (its own C++ pipeline "
                         "keeps 8 in flight); 0 = drive submit/wait from Python")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per step through the batched matrix-core path (config C3: --batch 1024 "
                         "--metric cosine); 0 = the headline single-query workload")
    ap.add_argument("--mask-keep", type=float, default=0.0,
                    help="config C5: row mask keeping this fraction of the rows (0 = no mask)")
    ap.add_argument("--batch-kernel", type=int, choices=[0, 1, 2, 3], default=3,
                    help="C3 pre-filter keys: 0 f32 MFMA, 1 bf16x3, 2 fp16, 3 auto = the library default (fp16 for cosine, "
                         "bf16x3 otherwise); results are identical")
    ap.add_argument("--mask-kind", choices=["bernoulli", "range"], default="bernoulli",
                    help="C5 mask shape: i.i.d. Bernoulli(keep) per row, or one contiguous id range of keep*rows rows")
    ap.add_argument("--recall-queries", type=int, default=1000,
                    help="N=1: queries whose GPU answer is compared with the exhaustive CPU oracle (all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo lets several ranks share one GPU in tests)")
    ap.add_argument("--ranks-share-gpu", action="store_true", help="testing: every rank uses cuda:0")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the N>1 code path (process group, all-gather, merge) even with one rank")
    return ap.parse_args()


def make_corpus(torch, n, d, metric, device):
    """Recipe of the reference's demo (/root/reference/example/lib/tostore_example.dart:728-747):
    i.i.d. N(0,1) components, rows L2-normalised, stored f32; for L2/IP each row is also scaled
    by U(0.5,2) so the three metrics rank differently.  Seeded: every
    rank builds identical rows."""
    g = torch.Generator(device=device)
    g.manual_seed(20260612)
    chunk = 131072
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        if metric != 2:
            x *= torch.rand((e - s, 1), generator=g, device=device) * 1.5 + 0.5
        out[s:e] = x
    return out


def make_queries(nq, d, metric):
    rng = np.random.Generator(np.random.Philox(20260613))
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    if metric == 2:  # caller-side _normalizeFloat32 (vector_index_manager.dart:516-520)
        from tostore_amd import normalize_float32
        q = np.stack([normalize_float32(x) for x in q])
    return np.ascontiguousarray(q)


def bench_batch(a, idx, host_rows, metric, world, rank):
    """Config C3: one step = one call with `--batch` queries (matrix-core path).  Side
    measurement, not the headline line; single GPU only."""
    import torch

    assert world == 1, "the batched benchmark is single-GPU"
    n, d, k, nq = a.rows, a.dim, a.k, a.batch
    steps, warm = max(1, min(a.steps, 20)), max(1, min(a.warmup, 3))
    qs = make_queries(nq * 2, d, metric)
    idx.set_batch_kernel(a.batch_kernel)
    for i in range(warm):
        idx.search(qs[(i % 2) * nq:(i % 2 + 1) * nq], k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ids, dist, cnt = idx.search(qs[(i % 2) * nq:(i % 2 + 1) * nq], k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gemm_us, flops = idx.bench_batch(qs[:nq], k, iters=3)
    a.batch_kernel = idx.counters()["batch_kernel_last"]  # what auto resolved to, for the report below
    tf = flops / (gemm_us * 1e-6) / 1e12
    out = {"metric": "kNN queries/sec, %dx%d f32 %s k=%d, %d-query batch (matrix-core path)" % (n, d, a.metric, k, nq),
           "value": nq * steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": warm,
           "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": {0: "f32", 1: "f32 as bf16 hi+lo (3 bf16 MFMAs per product), f64 rerank",
                     2: "fp16 pre-filter keys (1 f16 MFMA per product), f64 rerank"}[a.batch_kernel],
           "data": "synthetic",
           "config": {"workload": "C3: %dx%d f32, %s, k=%d, %d-query batch" % (n, d, a.metric, k, nq),
                      "batch_kernel": {0: "f32 MFMA", 1: "bf16x3", 2: "f16"}[a.batch_kernel]}}
    if a.batch_kernel == 0:
        out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                           "traffic": None, "kernel": "tsh::batch_score_kernel (sample + filtered passes)",
                           "kernel_us": gemm_us, "algorithmic_flops_per_launch": flops}
    elif a.batch_kernel == 2:
        out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                           "traffic": None, "kernel": "tsh::batch_score_bf16x3_kernel<MODE=f16> (sample + filtered passes)",
                           "kernel_us": gemm_us, "algorithmic_flops_per_launch": flops, "vs_f32_mfma_peak": tf / 157.3}
    else:  # three bf16 MFMAs per algorithmic multiply-add: the ceiling for ALGORITHMIC flops is 2500 / 3
        out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 2500.0 / 3, "unit": "TFLOP/s",
                           "frac": tf / (2500.0 / 3), "traffic": None,
                           "kernel": "tsh::batch_score_bf16x3_kernel (sample + filtered passes)",
                           "kernel_us": gemm_us, "algorithmic_flops_per_launch": flops,
                           "executed_bf16_tflops": 3 * tf, "bf16_dense_peak": 2500.0,
                           "vs_f32_mfma_peak": tf / 157.3}
    if host_rows is not None:
        import oracle
        m = 4
        ok, hits = True, 0
        for i in range(m):
            j = ((steps - 1) % 2) * nq + i
            eids, edist = oracle.search_heap_mt(host_rows, qs[j], metric, k)
            hits += len(set(ids[i, :cnt[i]].tolist()) & set(eids.tolist()))
            ok &= bool(np.array_equal(ids[i, :cnt[i]], eids) and np.array_equal(dist[i, :cnt[i]], edist))
        out["recall_at_k"] = hits / (m * k)
        out["ids_and_distances_bit_exact"] = ok
    c = idx.counters()
    out["counters"] = {k2: c[k2] for k2 in ("batch_launches", "scan_launches", "fallback_searches")}
    out["counters"]["candidates_per_query"] = c["candidates_total"] / max(c["searches"], 1)
    idx.close()
    return json.dumps(out)


def main():
    a = parse()
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


def run_bench(a):
    import torch

    from tostore_amd import HipVectorIndex, _ffi
    from tostore_amd.sharded import ShardedSearcher

    metric = {"l2": 0, "ip": 1, "cosine": 2}[a.metric]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    if a.ranks_share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or a.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    ctl = dev if a.backend == "nccl" else torch.device("cpu")  # where small control tensors live
    n, d, k = a.rows, a.dim, a.k
    assert _ffi.lib().tsh_device_count() >= 1, "libtostore_hip.so sees no device"

    # ---- resident corpus (this rank's row range) -----------------------------------
    corpus = make_corpus(torch, n, d, metric, dev)
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    idx = HipVectorIndex(d, metric, capacity_rows=hi - lo, shard_device=local_rank, row_base=lo)
    torch.cuda.synchronize()  # the library copies on its own stream: the producer must be done
    if hi > lo:
        idx.append_device(lo, hi - lo, corpus[lo:hi].data_ptr())
    torch.cuda.synchronize()
    host_rows = None
    if rank == 0 and not a.no_cpu_baseline:
        host_rows = corpus.cpu().numpy()  # for the CPU baseline / recall check only
    del corpus
    torch.cuda.empty_cache()

    if a.batch > 0:
        return bench_batch(a, idx, host_rows, metric, world, rank)

    nq_total = a.warmup + a.steps
    queries = make_queries(max(nq_total, 1), d, metric)
    searcher = ShardedSearcher(idx) if dist is not None else None
    idx.set_batch_min_nq(0)  # headline workload: every query scans the corpus on its own (no MFMA batching)

    row_mask = None
    if a.mask_keep > 0:  # C5: WHERE pre-filter as a device-side row bitmask (seed 20260614)
        if a.mask_kind == "range":  # e.g. WHERE id BETWEEN ...: one contiguous run of node ids
            keepbits = np.zeros(n, bool)
            start = int(np.random.Generator(np.random.Philox(20260614)).integers(0, max(1, n - int(n * a.mask_keep))))
            keepbits[start:start + int(n * a.mask_keep)] = True
        else:
            keepbits = np.random.Generator(np.random.Philox(20260614)).random(n) < a.mask_keep
        row_mask = np.packbits(keepbits, bitorder="little")

    def one(i):
        q = queries[i % len(queries)]
        if searcher is not None:
            return searcher.search(q, k, None, row_mask)
        return idx.search(q, k, None, row_mask)

    def sharded_group(count):
        # N > 1: up to --group queries per all-gather / merge; short runs use smaller groups so that
        # at least four of them pipeline (scan of group g+1 behind the exchange of group g)
        return max(16, min(a.group or 64, count // 4))

    def run(first, count):
        """`count` single-query searches, `--inflight` of them in flight.  Each query still
        streams the whole (shard of the) corpus on its own; only the select / re-rank /
        copy tail of one query overlaps the scan of the next."""
        if a.inflight <= 1:
            for i in range(count):
                one(first + i)
        elif searcher is not None:
            # N > 1: groups of `--group` queries share one all-gather + one merge call, and the
            # next group's shard scans run while this group is exchanged and merged
            sel = [(first + j) % len(queries) for j in range(count)]
            searcher.search_many(queries[sel], k, None, row_mask, group=sharded_group(count))
        elif a.group > 0:
            for g0 in range(0, count, a.group):
                sel = [(first + g0 + j) % len(queries) for j in range(min(a.group, count - g0))]
                idx.search(queries[sel], k, None, row_mask)
        else:
            from collections import deque
            pend = deque()
            for i in range(count):
                if len(pend) == a.inflight:
                    idx.wait(pend.popleft())
                pend.append(idx.submit(queries[(first + i) % len(queries)], k, row_mask))
            while pend:
                idx.wait(pend.popleft())

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, a.warmup)
    fence()
    c0 = idx.counters()
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    fence()
    elapsed = time.perf_counter() - t0
    c1 = idx.counters()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # single-query latency, one at a time (not the headline value)
    lat = []
    for i in range(min(200, max(20, a.steps // 5))):
        t1 = time.perf_counter()
        one(i)
        lat.append(time.perf_counter() - t1)
    lat = np.sort(np.asarray(lat)) * 1e3

    # ---- roofline of the dominant kernel (K1 scan): HIP events recorded by the library
    # around real scan launches on its pipeline stream, during the timed region above
    ns = c1["scan_us_samples"] - c0["scan_us_samples"]
    scan_us = (c1["scan_us_sum"] - c0["scan_us_sum"]) / ns if ns > 0 else float("nan")
    scan_alone_us = idx.bench_scan(queries[0], iters=50, row_mask=row_mask) if hi > lo else float("nan")
    shard_bytes = float(hi - lo) * d * 4  # algorithmic: every stored f32 read once
    if row_mask is not None:  # C5: only kept rows are read, plus the mask itself
        kept = int(np.unpackbits(row_mask, bitorder="little")[lo:hi].sum())
        shard_bytes = float(kept) * d * 4 + (hi - lo) / 8
    if dist is not None:
        tt = torch.tensor([scan_us], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        scan_us = float(tt.item())

    # ---- recall + CPU baseline: oracle on rank 0, every rank joins the GPU searches ----
    n_cpu, ref, cpu_elapsed = 0, None, 0.0
    if rank == 0 and host_rows is not None:
        import oracle

        t1 = time.perf_counter()
        oracle.search_heap(host_rows, queries[0], metric, k, None, row_mask)
        per_q = time.perf_counter() - t1
        budget = a.cpu_seconds if world == 1 else min(a.cpu_seconds, 4.0)
        n_cpu = int(max(2, min(32, budget / max(per_q, 1e-3))))
        t1 = time.perf_counter()
        ref = [oracle.search_heap(host_rows, queries[i], metric, k, None, row_mask) for i in range(n_cpu)]
        cpu_elapsed = time.perf_counter() - t1
    if dist is not None:
        tt = torch.tensor([n_cpu], dtype=torch.int64, device=ctl)
        dist.broadcast(tt, src=0)
        n_cpu = int(tt.item())
    got = [one(i) for i in range(n_cpu)]

    out = None
    if rank == 0:
        achieved = shard_bytes / (scan_us * 1e-6) / 1e9
        traffic = None  # HBM bytes per launch from committed rocprofv3 PMC passes (profiles/)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                ent = json.load(f).get("%dx%d" % (hi - lo, d))
            if ent:
                traffic = ent["traffic_bytes"]
        except (OSError, ValueError):
            pass
        out = {
            "metric": "kNN queries/sec + recall@k, 1Mx768 f32 brute-force",
            "value": a.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: %dx%d f32, %s, k=%d, single query per step" % (n, d, a.metric, k),
                       "rows": n, "dim": d, "k": k, "metric": a.metric, "mask_keep": a.mask_keep or None,
                       "mask_kind": a.mask_kind if a.mask_keep else None,
                       "queries_in_flight": 8,
                       "queries_per_call": (a.group or 1) if searcher is None else sharded_group(a.steps),
                       "note": "every query scans the whole corpus on its own (HBM-bound kernel, no matrix-core "
                               "batching); independent queries are handed over in groups and pipelined",
                       "sharding": "row-range x%d, RCCL all-gather of top-k candidates" % world
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "tsh::scan_kernel", "kernel_us": scan_us, "kernel_us_samples": int(ns),
                         "kernel_us_back_to_back_alone": scan_alone_us,
                         "algorithmic_bytes_per_launch": shard_bytes},
        }
        if (hi - lo + 63) // 64 < 6144 and a.inflight > 1 and os.environ.get("TSH_SCAN_STREAMS") != "1":
            # shards below 6144 tiles alternate their scans between two streams: two scans
            # run side by side, so one launch's own duration is about twice its share of the HBM time
            out["roofline"]["scans_side_by_side"] = 2
            out["roofline"]["achieved_alone"] = shard_bytes / (scan_alone_us * 1e-6) / 1e9
        if ref is not None:
            hits, tot, exact = 0, 0, True
            for i in range(n_cpu):
                ids, dd, cnt = got[i]
                g = ids[0, :cnt[0]]
                hits += len(set(g.tolist()) & set(ref[i][0].tolist()))
                tot += len(ref[i][0])
                exact &= bool(np.array_equal(g, ref[i][0]) and np.array_equal(dd[0, :cnt[0]], ref[i][1]))
            out["recall_at_k"] = hits / max(tot, 1)
            out["ids_and_distances_bit_exact"] = exact
            if world == 1 and a.recall_queries > n_cpu:
                # recall@k over >= 1000 queries: the oracle's OpenMP form,
                # same per-(query,row) arithmetic, against the GPU answers of the same queries
                import oracle

                nr = min(a.recall_queries, len(queries))
                t1 = time.perf_counter()
                r_ids, r_dist, r_cnt = oracle.search_heap_many_mt(host_rows, queries[:nr], metric, k, None, row_mask)
                t_or = time.perf_counter() - t1
                g_ids, g_dist, g_cnt = idx.search(queries[:nr], k, None, row_mask)
                hits = sum(len(set(g_ids[i, :g_cnt[i]].tolist()) & set(r_ids[i, :r_cnt[i]].tolist())) for i in range(nr))
                same = all(g_cnt[i] == r_cnt[i] and np.array_equal(g_ids[i, :g_cnt[i]], r_ids[i, :r_cnt[i]])
                           and np.array_equal(g_dist[i, :g_cnt[i]], r_dist[i, :r_cnt[i]]) for i in range(nr))
                out["recall_at_k"] = hits / max(int(r_cnt[:nr].sum()), 1)
                out["recall_queries"] = nr
                out["ids_and_distances_bit_exact"] = bool(same)
                out["cpu_baseline_mt_batched"] = {"value": nr / t_or, "unit": "queries/s",
                                                  "cores": oracle.mt_max_threads(), "kind": "port",
                                                  "sample": "%d queries, OpenMP over query groups" % nr}
            if world == 1:
                import oracle

                out["cpu_baseline"] = {
                    "value": n_cpu / cpu_elapsed, "unit": "queries/s", "cores": 1, "kind": "port",
                    "sample": "%d of the same queries over the full %dx%d corpus, oracle/vs_oracle.c "
                              "single thread (the reference searches on one isolate)" % (n_cpu, n, d)}
                try:
                    thr = oracle.mt_max_threads()
                    t1 = time.perf_counter()
                    m = max(2, min(n_cpu, 8))
                    for i in range(m):
                        oracle.search_heap_mt(host_rows, queries[i], metric, k, None, row_mask)
                    out["cpu_baseline_mt"] = {"value": m / (time.perf_counter() - t1), "unit": "queries/s",
                                              "cores": thr, "kind": "port", "sample": "%d queries, OpenMP" % m}
                except Exception:
                    pass
        out["latency_ms_one_at_a_time"] = {"p50": float(lat[len(lat) // 2]), "p99": float(lat[int(len(lat) * 0.99)]),
                                           "mean": float(lat.mean()), "queries": int(len(lat))}
        c = idx.counters()
        out["counters"] = {"fallback_searches": c["fallback_searches"],
                           "candidates_per_query": c["candidates_total"] / max(c["searches"], 1)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    idx.close()
    return json.dumps(out) if rank == 0 else None


if __name__ == "__main__":
    main()
