"""Single-query latency by N and k (which pipeline answers, and how long it takes).
usage: python tools/latency_probe.py [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 384
rng = np.random.default_rng(5)
Q = rng.standard_normal((64, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
names = {0: "none", 1: "fast", 2: "exact+select", 3: "exact+sort"}
for n in (1_000, 100_000, 1_000_000, 10_000_000):
    idx = V.FlatIndex(dim)
    idx.reserve(n)
    for ci, c0 in enumerate(range(0, n, 500_000)):
        c = min(500_000, n - c0)
        g = torch.Generator(device="cuda:0"); g.manual_seed(1234 + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(c0, c0 + c, dtype=np.uint64), x, validate=False)
        del x
    for k in (10, 48, 60, 64, 100, 1000, 2000):
        if k > n:
            continue
        reps = 40 if k <= 48 else 6
        for i in range(12):
            idx.search_arrays(Q[i], k, 0)
        ts = []
        for i in range(reps):
            t = time.perf_counter()
            idx.search_arrays(Q[i % 64], k, 0)
            ts.append(time.perf_counter() - t)
        # median: a one-off host stall (allocator, Python GC finalising the previous index) inside 6 repetitions
        # once showed up as "6 ms" for N = 100000, k = 2000; per-call times there are 0.23-0.29 ms
        dt = float(np.median(ts))
        print(f"N={n:>9} k={k:>5}: {dt * 1e3:8.3f} ms/query (median of {reps}, max {max(ts) * 1e3:.3f})  ({names[V.last_path()]})", flush=True)
    del idx
