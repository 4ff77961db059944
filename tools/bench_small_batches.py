#!/usr/bin/env python3
"""search_batch latency for small and medium batches at N = 10 M x 384 (cosine): one K4r launch sequence per batch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vectorlite_amd as V
rows, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 384
dev = torch.device("cuda", 0)
idx = V.FlatIndex(dim); idx.reserve(rows)
done = ci = 0
while done < rows:
    c = min(500_000, rows - done)
    g = torch.Generator(device=dev); g.manual_seed(1234 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
    done += c; ci += 1
rng = np.random.default_rng(4321)
Q = rng.standard_normal((2048, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
idx.search_batch(Q[:64], 10, 0); idx.search_batch(Q[:64], 10, 0)
t0 = time.perf_counter()
for i in range(20): idx.search_arrays(Q[i], 10, 0)
print(f"single search(): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
for nq in (2, 4, 8, 16, 32, 64, 128, 129, 256, 384, 512, 640, 768, 896, 1024, 1280, 1408, 2048):  # 384 = 3 chunks of 128, 640 = 5, ...
    idx.search_batch(Q[:nq], 10, 0)
    idx.profile_read(); idx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(5): idx.search_batch(Q[:nq], 10, 0)
    dt = (time.perf_counter() - t0) / 5
    idx.profile_enable(False)
    n, ms, _ = idx.profile_read()
    print(f"nq={nq:5d}: {dt * 1e3:8.3f} ms per batch ({nq / dt:9.0f} QPS), filter kernels {ms / 5:7.3f} ms, passes {n // 5}")
