"""Throughput of concurrent single-query callers with and without coalescing.
usage: python tools/bench_coalesce.py [rows] [dim] [threads] [queries_per_thread]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
per = int(sys.argv[4]) if len(sys.argv) > 4 else 40
idx = V.FlatIndex(dim)
idx.reserve(n)
for ci, c0 in enumerate(range(0, n, 500_000)):
    c = min(500_000, n - c0)
    g = torch.Generator(device="cuda:0"); g.manual_seed(1234 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(c0, c0 + c, dtype=np.uint64), x, validate=False)
    del x
rng = np.random.default_rng(5)
Q = rng.standard_normal((T * per, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
lone = [idx.search_arrays(Q[i], 10, 0) for i in range(0, T * per, per)]


def run(label):
    res = [None] * (T * per)
    bar = threading.Barrier(T + 1)

    def w(t):
        bar.wait()
        for i in range(t * per, (t + 1) * per):
            res[i] = idx.search_arrays(Q[i], 10, 0)
    th = [threading.Thread(target=w, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    bar.wait(); t0 = time.perf_counter()
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    same = sum(int(res[t * per][0].tolist() == lone[t][0].tolist() and res[t * per][1].tolist() == lone[t][1].tolist()) for t in range(T))
    print(f"{label}: {T} threads x {per} queries, N={n} dim={dim}: {T * per / dt:.1f} QPS, mean latency {dt / per * 1e3:.2f} ms, "
          f"identical to lone search {same}/{T}", flush=True)


run("uncoalesced")
for mb, win in ((64, 0), (64, 200)):
    idx.set_coalescing(mb, win)
    run(f"warm-up coalesced(max {mb}, window {win}us)")
    b0, q0 = idx.coalesce_stats()
    run(f"coalesced(max {mb}, window {win}us)")
    b1, q1 = idx.coalesce_stats()
    print(f"   passes {b1 - b0} for {q1 - q0} queries ({(q1 - q0) / max(b1 - b0, 1):.1f} per pass)", flush=True)
