import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vectorlite_amd as V
from oracle import oracle as O
M = {"cosine": 0, "euclidean": 1, "manhattan": 2, "dotproduct": 3}
def unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)); x /= np.linalg.norm(x, axis=1, keepdims=True); return x
bad = 0
for dim in [1, 3, 4, 5, 33, 128, 384]:
    rng = np.random.default_rng(1234 + dim)
    for n in (1, 2, 63, 64, 65, 257, 1000):
        rows = unit_rows(rng, n, dim)
        ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(2 ** 40)
        gpu = V.FlatIndex(dim); gpu.add_rows(ids, rows, validate=False)
        ref = O.FlatOracle(dim, ids, rows)
        for qi in range(3):
          q = unit_rows(rng, 1, dim)[0]
          for name, m in M.items():
            for k in (1, 10, 32):
                try:
                    gi, gs = gpu.search_arrays(q, k, m)
                except Exception as e:
                    print("EXC", dim, n, qi, name, k, e, flush=True); bad += 1; continue
                ri, rs = ref.search(q, k, m)
                if gi.tolist() != ri.tolist() or gs.tolist() != rs.tolist():
                    bad += 1
                    if bad < 12:
                        print("MISMATCH", dim, n, name, k, "path", V.last_path(), gi.tolist()[:6], ri.tolist()[:6], flush=True)
print("bad", bad)
