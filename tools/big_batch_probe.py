import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
rng = np.random.default_rng(1)
n, dim, nq = 50_000, 64, 200_000
rows = rng.standard_normal((n, dim)); idx = V.FlatIndex(dim); idx.add_rows(np.arange(n, dtype=np.uint64), rows)
Q = rng.standard_normal((nq, dim))
for metric in (0, 2):
    t0 = time.perf_counter(); bi, bs, bn = idx.search_batch(Q, 10, metric); dt = time.perf_counter() - t0
    assert bn.tolist() == [10] * nq
    for qi in (0, 2047, 2048, 99_999, nq - 1):
        si, ss = idx.search_arrays(Q[qi], 10, metric)
        assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), (metric, qi)
    print(f"metric {metric}: {nq} queries in one search_batch call: {dt:.2f}s = {nq / dt / 1e3:.0f} k QPS; sampled rows == single searches")
print("big nq ok")
