"""Where do the rare ~40 ms single-query calls come from?  

C2-sized index, the bench's own sequence (pipelined groups first, then one query at a time), every call
stamped: outliers are printed with their index, the time since the previous outlier, the submit / wait split
(tsh_search_submit = host enqueue, tsh_search_wait = GPU wait + finalise) and the cgroup cpu.stat deltas
(CFS throttling) around them.  Run on the box: gpurun -- 'python tools/attic/lone_stall_probe.py [n] [idle_ms]'."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def cpu_stat():
    d = {}
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for ln in open(p):
                k, v = ln.split()
                d[k] = int(v)
            break
        except OSError:
            pass
    return d


def main():
    n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    import torch

    import bench

    a = bench.parse(["--no-cpu-baseline", "--no-side"])
    env = bench.Env(a)
    corpus = env.corpus(a.rows, a.dim, 0)
    idx = env.make_index(a.dim, 0, corpus, 0, a.rows)
    env.release(corpus)
    del corpus
    qs = bench.make_queries(1024, a.dim, 0)
    idx.set_batch_min_nq(0)
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?",
          "cpus:", os.cpu_count(), flush=True)
    # the bench's sequence: pipelined groups ...
    for r in range(5):
        for g0 in range(0, 256, 64):
            idx.search(qs[g0:g0 + 64], a.k)
    torch.cuda.synchronize()

    def report(name, lat, parts=None, stats=None):
        lat = np.asarray(lat) * 1e3
        s = np.sort(lat)
        print("%s: n %d p50 %.3f p99 %.3f max %.3f mean %.3f ms" % (
            name, len(lat), s[len(s) // 2], s[int(len(s) * 0.99)], s[-1], lat.mean()), flush=True)
        thr = max(2.0, 3 * s[len(s) // 2])
        for i in np.nonzero(lat > thr)[0][:40]:
            extra = ""
            if parts is not None:
                extra = " submit %.3f wait %.3f ms" % (parts[i][0] * 1e3, parts[i][1] * 1e3)
            if stats is not None:
                b, e = stats[i]
                extra += " throttled +%d periods +%.1f ms, cpu +%.1f ms" % (
                    e.get("nr_throttled", 0) - b.get("nr_throttled", 0),
                    (e.get("throttled_usec", 0) - b.get("throttled_usec", 0)) / 1e3,
                    (e.get("usage_usec", 0) - b.get("usage_usec", 0)) / 1e3)
            print("   call %5d: %.3f ms%s" % (i, lat[i], extra), flush=True)

    # ... then one at a time through tsh_search
    for rep in range(2):
        lat, stats = [], []
        s0 = cpu_stat()
        for i in range(n_calls):
            if idle_ms:
                time.sleep(idle_ms * 1e-3)
            b = cpu_stat() if (i % 1 == 0) else None
            t = time.perf_counter()
            idx.search(qs[i % 1024], a.k)
            lat.append(time.perf_counter() - t)
            stats.append((b, cpu_stat()))
        s1 = cpu_stat()
        report("tsh_search one at a time (pass %d)" % rep, lat, None, stats)
        print("   whole pass: throttled +%d periods, +%.1f ms; cpu %.1f ms" % (
            s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0),
            (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3,
            (s1.get("usage_usec", 0) - s0.get("usage_usec", 0)) / 1e3), flush=True)
    # submit / wait split
    lat, parts = [], []
    for i in range(n_calls):
        t = time.perf_counter()
        tk = idx.submit(qs[i % 1024], a.k)
        t1 = time.perf_counter()
        idx.wait(tk)
        t2 = time.perf_counter()
        lat.append(t2 - t)
        parts.append((t1 - t, t2 - t1))
    report("submit + wait one at a time", lat, parts)
    # without the cpu.stat reads in between (they are file reads: do they matter?)
    lat = []
    for i in range(n_calls):
        t = time.perf_counter()
        idx.search(qs[i % 1024], a.k)
        lat.append(time.perf_counter() - t)
    report("tsh_search one at a time, no stat reads", lat)
    idx.close()


if __name__ == "__main__":
    main()
