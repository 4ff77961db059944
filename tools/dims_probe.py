import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
from oracle import oracle as O
rng = np.random.default_rng(3)
for dim in (769, 1000, 1536, 3072, 4096, 8191, 20000):
    n = 3000 if dim <= 4096 else 600
    rows = rng.standard_normal((n, dim)); rows[5] = rows[9]
    ids = np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(1)
    g = V.FlatIndex(dim); g.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = rng.standard_normal((5, dim)); Q[0] = rows[9]
    ok = 0
    for m in range(4):
        for qi in range(3):
            gi, gs = g.search_arrays(Q[qi], 10, m); wi, ws = ref.search(Q[qi], 10, m)
            assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist(), (dim, m, qi)
            ok += 1
        bi, bs, bn = g.search_batch(Q, 10, m)
        for qi in range(5):
            wi, ws = ref.search(Q[qi], 10, m)
            assert bi[qi].tolist() == wi.tolist() and bs[qi].tolist() == ws.tolist(), (dim, m, qi, "batch")
    print("dim", dim, "rows", n, "ok", ok, "paths", V.last_path())
print("dims probe ok")
