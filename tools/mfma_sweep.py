"""Sweep k_mfma_scan launch shapes / grids through the library's env knobs (one process).
usage: python tools/mfma_sweep.py [rows] [dim] [nq] [metric]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
metric = int(sys.argv[4]) if len(sys.argv) > 4 else 0
idx = V.FlatIndex(dim); idx.reserve(n)
for ci, c0 in enumerate(range(0, n, 500_000)):
    c = min(500_000, n - c0)
    g = torch.Generator(device="cuda:0"); g.manual_seed(1234 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(c0, c0 + c, dtype=np.uint64), x, validate=False)
    del x
rng = np.random.default_rng(4321)
Q = rng.standard_normal((nq, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
base = None
configs = [("81", "0"), ("81", "128"), ("41", "0"), ("41", "64"), ("41", "96"), ("42", "0"), ("42", "128"), ("42", "96")]
if len(sys.argv) > 5:
    configs = [tuple(c.split(":")) for c in sys.argv[5].split(",")]
for shape, grid in configs:
    os.environ["VL_MFMA_SHAPE"] = shape
    os.environ["VL_MFMA_GRID"] = grid
    for _ in range(2):
        out = idx.search_batch(Q, 10, metric)
    idx.profile_read(); idx.profile_enable(True)
    t = time.perf_counter()
    for _ in range(3):
        out = idx.search_batch(Q, 10, metric)
    dt = (time.perf_counter() - t) / 3
    idx.profile_enable(False)
    npass, ms, _ = idx.profile_read()
    if base is None:
        base = out
    same = out[0].tolist() == base[0].tolist() and out[1].tolist() == base[1].tolist()
    pf = 2.0 * nq * n * dim / (ms / max(npass, 1) * (max(npass, 1) / 3) * 1e-3) / 1e15 if npass else 0
    print(f"shape {shape} grid {grid:>4}: batch {dt * 1e3:7.2f} ms, candidate pipeline {ms / 3:7.2f} ms per batch "
          f"-> {pf:.3f} PFLOP/s, same results {same}", flush=True)
