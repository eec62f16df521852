"""Where does the container's CPU quota throttle the process?  Prints cgroup cpu.stat deltas around the phases of
a batched C3 run (tools/README.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stat():
    d = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            d[k] = int(v)
    except OSError:
        pass
    return d


last = stat()
t_last = time.perf_counter()


def mark(what):
    global last, t_last
    s = stat()
    now = time.perf_counter()
    print("%-34s wall %8.1f ms  cpu %9.1f ms  throttled periods %3d  throttled %9.1f ms" % (
        what, (now - t_last) * 1e3, (s.get("usage_usec", 0) - last.get("usage_usec", 0)) / 1e3,
        s.get("nr_throttled", 0) - last.get("nr_throttled", 0),
        (s.get("throttled_usec", 0) - last.get("throttled_usec", 0)) / 1e3), flush=True)
    last, t_last = s, now


import numpy as np
import torch
mark("import torch")
import bench
a = bench.parse(["--batch", "1024", "--metric", "cosine"])
env = bench.Env(a)
mark("env")
corpus = env.corpus(a.rows, a.dim, 2)
torch.cuda.synchronize()
mark("corpus on device")
idx = env.make_index(a.dim, 2, corpus, 0, a.rows)
mark("index")
if "--host" in sys.argv:
    host = env.corpus_host(corpus)
    mark("corpus to host")
env.release(corpus)
del corpus
mark("release")
qs = bench.make_queries(2048, a.dim, 2)
mark("queries")
idx.set_batch_min_nq(1)
idx.set_batch_kernel(3)
for i in range(30):
    idx.search(qs[(i % 2) * 1024:(i % 2 + 1) * 1024], a.k)
    mark("batch call %d" % i)
