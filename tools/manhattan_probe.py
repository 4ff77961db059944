import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
rng = np.random.default_rng(1)
n, dim, nq = int(os.environ.get("ROWS", 50_000)), int(os.environ.get("DIM", 64)), int(os.environ.get("NQ", 102_400))
rows = rng.standard_normal((n, dim)); idx = V.FlatIndex(dim); idx.add_rows(np.arange(n, dtype=np.uint64), rows)
Q = rng.standard_normal((nq, dim))
idx.search_batch(Q[:1024], 10, 2)
for rep in range(3):
    t0 = time.perf_counter(); bi, bs, bn = idx.search_batch(Q, 10, 2); dt = time.perf_counter() - t0
    print(f"manhattan {nq} queries on {n} x {dim}: {dt * 1e3:.1f} ms = {nq / dt / 1e3:.0f} k QPS", flush=True)
for qi in (() if os.environ.get("NO_CHECK") else (0, 511, 512, nq - 1)):
    si, ss = idx.search_arrays(Q[qi], 10, 2)
    assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), qi
print("sampled rows == single searches")
