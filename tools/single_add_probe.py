import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
rng = np.random.default_rng(1)
dim = 384
rows = rng.standard_normal((102_000, dim)); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
for kind in ("flat", "hnsw"):
    h = V.FlatIndex(dim) if kind == "flat" else V.HNSWIndex(dim, 0)
    h.add_rows(np.arange(100_000, dtype=np.uint64), rows[:100_000])
    t0 = time.perf_counter()
    for i in range(100_000, 102_000):
        h.add(V.Vector(i, rows[i]))
    dt = (time.perf_counter() - t0) / 2000
    r = h.search(rows[101_999], 1, 0)
    found10 = sum(int(h.search(rows[i], 10, 0)[0].id == i) for i in range(100_000, 102_000, 10))
    print(f"{kind}: single add() on a 100 k x 384 index: {dt * 1e6:.0f} us per add ({1 / dt:.0f} adds/s); the last added row is found at k = 1: {r[0].id == 101_999}; "
          f"of 200 singly added rows {found10} find themselves first at k = 10 (the reference's beam ef = k; these are i.i.d. gaussian rows, the hard case for any graph)")
