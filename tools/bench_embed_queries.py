#!/usr/bin/env python3
"""Config 3's shard shape (1024 queries x 1.25 M x 768, Euclidean) from f32 embeddings: vl_index_search_batch_embeddings_f32
(host array / device tensor) beside search_batch on host f64 queries and search_batch_device on device f64 queries."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vectorlite_amd as V
rows, dim, nq = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000), 768, 1024
dev = torch.device("cuda", 0)
idx = V.FlatIndex(dim); idx.reserve(rows)
done = ci = 0
while done < rows:
    c = min(250_000, rows - done)
    g = torch.Generator(device=dev); g.manual_seed(1234 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
    done += c; ci += 1
emb = np.random.default_rng(4321).standard_normal((nq, dim)).astype(np.float32)
demb = torch.from_numpy(emb).to(dev)
E = emb.astype(np.float64); Q = E / np.sqrt((E * E).sum(axis=1, keepdims=True))  # not the reference's summation order: timing only
dQ = torch.from_numpy(Q).to(dev)
def timed(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    return (time.perf_counter() - t0) / reps * 1e3, out
a, ra = timed(lambda: idx.search_batch_embeddings(emb, 10, 1))
b, rb = timed(lambda: idx.search_batch_embeddings(demb, 10, 1))
c, _ = timed(lambda: idx.search_batch(Q, 10, 1))
d, _ = timed(lambda: idx.search_batch_device(dQ, 10, 1))
print(f"host f32 embeddings {a:.3f} ms, device f32 embeddings {b:.3f} ms | host f64 queries {c:.3f} ms, device f64 queries {d:.3f} ms per {nq}-query batch; "
      f"host == device embeddings: {ra[0].tolist() == rb[0].tolist() and ra[1].tolist() == rb[1].tolist()}")
