"""Ingest rates: single add() calls (the reference's path, O(N) duplicate scan there) and bulk adds from host rows.
usage: python tools/bench_ingest.py [rows] [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
rng = np.random.default_rng(0)
rows = rng.standard_normal((n, dim)); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
idx = V.FlatIndex(dim)
t = time.perf_counter(); idx.add_rows(np.arange(n, dtype=np.uint64), rows); dt = time.perf_counter() - t
print(f"flat bulk add from host rows (validated): {n} x {dim} in {dt:.2f}s = {n / dt / 1e6:.2f} M rows/s = {n * dim * 8 / dt / 1e9:.1f} GB/s of f64 over PCIe", flush=True)
emb = rng.standard_normal((n, dim)).astype(np.float32)
idx2 = V.FlatIndex(dim)
idx2.add_embeddings(np.arange(1000, dtype=np.uint64), emb[:1000]); idx2 = V.FlatIndex(dim)  # warm
t = time.perf_counter(); idx2.add_embeddings(np.arange(n, dtype=np.uint64), emb); dt = time.perf_counter() - t
print(f"flat add_embeddings from host f32 (device widen + L2 normalise, validated): {n} x {dim} in {dt:.2f}s = {n / dt / 1e6:.2f} M rows/s = {n * dim * 4 / dt / 1e9:.1f} GB/s of f32 over PCIe", flush=True)
import torch
demb = torch.from_numpy(emb).cuda(); torch.cuda.synchronize()
idx3 = V.FlatIndex(dim)
t = time.perf_counter(); idx3.add_embeddings(np.arange(n, dtype=np.uint64), demb); dt = time.perf_counter() - t
print(f"flat add_embeddings from a device f32 tensor: {n} x {dim} in {dt:.3f}s = {n / dt / 1e6:.2f} M rows/s", flush=True)
del idx2, idx3, demb
extra = rng.standard_normal((2000, dim))
t = time.perf_counter()
for i in range(2000):
    idx.add(V.Vector(n + i, extra[i]))
dt = time.perf_counter() - t
print(f"flat single add() at N = {n}: {2000 / dt:.0f} adds/s ({dt / 2000 * 1e6:.0f} us each)", flush=True)
t = time.perf_counter()
for i in range(200):
    idx.delete(int(i * 1000))
dt = time.perf_counter() - t
print(f"flat delete() at N = {n} (order-preserving compaction): {200 / dt:.0f} deletes/s ({dt / 200 * 1e3:.2f} ms each)", flush=True)
hn = V.HNSWIndex(dim, V.SimilarityMetric.Cosine)
m = min(n, 200_000)
t = time.perf_counter(); hn.add_rows(np.arange(m, dtype=np.uint64), rows[:m]); dt = time.perf_counter() - t
print(f"hnsw bulk add: {m} x {dim} in {dt:.2f}s = {m / dt:.0f} inserts/s", flush=True)
t = time.perf_counter()
for i in range(500):
    hn.add(V.Vector(m + i, extra[i]))
dt = time.perf_counter() - t
print(f"hnsw single add() at N = {m}: {500 / dt:.0f} adds/s ({dt / 500 * 1e6:.0f} us each)", flush=True)
