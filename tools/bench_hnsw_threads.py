"""HNSW: single-query latency and multi-thread throughput through vl_index_search (reference ef = min(k, len)).
usage: python tools/bench_hnsw_threads.py [rows] [dim] [threads]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
A = torch.randn((16, dim), dtype=torch.float64, device=dev, generator=g)
hn = V.HNSWIndex(dim, V.SimilarityMetric.Cosine)
t0 = time.perf_counter()
for c0 in range(0, n, 250_000):
    c = min(250_000, n - c0)
    x = torch.randn((c, 16), dtype=torch.float64, device=dev, generator=g) @ A + 0.05 * torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    hn.add_rows(np.arange(c0, c0 + c, dtype=np.uint64), x)
print(f"built {n} x {dim} in {time.perf_counter() - t0:.1f}s", flush=True)
rng = np.random.default_rng(4321)
per = 200
Q = rng.standard_normal((T * per, 16)) @ A.cpu().numpy() + 0.05 * rng.standard_normal((T * per, dim))
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
for i in range(20):
    hn.search(Q[i], 10, 0)
t0 = time.perf_counter()
for i in range(per):
    hn.search(Q[i], 10, 0)
dt = time.perf_counter() - t0
print(f"1 thread: {per / dt:.0f} QPS, {dt / per * 1e6:.0f} us per query", flush=True)


def run(label):
    bar = threading.Barrier(T + 1)

    def w(t):
        bar.wait()
        for i in range(t * per, (t + 1) * per):
            hn.search(Q[i], 10, 0)
    th = [threading.Thread(target=w, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    bar.wait(); t0 = time.perf_counter()
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"{label}: {T} threads: {T * per / dt:.0f} QPS, mean latency {dt / per * 1e6:.0f} us", flush=True)


run("concurrent single-query searches")
if hasattr(hn, "set_coalescing"):
    hn.set_coalescing(256, 0)
    run("warm-up coalesced")
    run("coalesced (max 256, window 0)")
    hn.set_coalescing(256, 100)
    run("coalesced (max 256, window 100us)")
