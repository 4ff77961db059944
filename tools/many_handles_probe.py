#!/usr/bin/env python3
"""Many small collections in one process (the reference keeps a HashMap of collections, src/client.rs:243-247): thousands of
handles alive at once, each with its own streams and scratch; every one answers; everything is returned on destroy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V
H = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(4)
dim = 32
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
hs = []
rows = rng.standard_normal((100, dim))
for i in range(H):
    h = V.FlatIndex(dim) if i % 10 else V.HNSWIndex(dim, 1)
    h.add_rows(np.arange(100, dtype=np.uint64) + np.uint64(i), rows + i * 1e-3)
    hs.append(h)
t1 = time.time()
for i, h in enumerate(hs):
    r = h.search(rows[7] + i * 1e-3, 1, 1)
    assert r[0].id == 7 + i, (i, r[0].id)
t2 = time.time()
used = free0 - torch.cuda.mem_get_info()[0]
del hs, h
import gc; gc.collect(); torch.cuda.synchronize()
back = free0 - torch.cuda.mem_get_info()[0]
print(f"{H} handles (every tenth an HNSW index): created and filled in {t1 - t0:.1f}s, one search each in {t2 - t1:.1f}s "
      f"({(t2 - t1) / H * 1e3:.2f} ms per first search), {used / 2**30:.1f} GiB held, {back / 2**20:.0f} MiB not returned after destroy")
assert back < 512 * 2**20
print("many handles ok")
