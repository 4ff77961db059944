"""A few single-query searches at N rows for a rocprofv3 --kernel-trace timeline.  usage: trace_single.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vectorlite_amd as V
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = 384
idx = V.FlatIndex(dim); idx.reserve(n)
for ci, c0 in enumerate(range(0, n, 500_000)):
    c = min(500_000, n - c0)
    g = torch.Generator(device="cuda:0"); g.manual_seed(ci)
    x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(c0, c0 + c, dtype=np.uint64), x, validate=False); del x
Q = np.random.default_rng(1).standard_normal((40, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
for i in range(40):
    idx.search_arrays(Q[i], 10, 0)
t = time.perf_counter()
for i in range(40):
    idx.search_arrays(Q[i], 10, 0)
print("ms/query", (time.perf_counter() - t) / 40 * 1e3)
