import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
rng = np.random.default_rng(1)
n, dim = 50_000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows = rng.standard_normal((n, dim)); idx = V.FlatIndex(dim); idx.add_rows(np.arange(n, dtype=np.uint64), rows)
Q = rng.standard_normal((800, dim))
idx.search_batch(Q[:64], 10, 2)
t0 = time.perf_counter(); idx.search_batch(Q, 10, 2); dt = time.perf_counter() - t0
print(f"800 manhattan queries: {dt*1e3:.1f} ms = {dt/100*1e6:.0f} us per 8-query pass")
