#!/usr/bin/env python3
"""One bulk add of more than 4 GiB of HOST rows (byte counts past 2^32 in the host -> device path), then a host batch of more
than 2^31 bytes of queries is not attempted (6 MB is the realistic size); rows found at the end of the block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
n, dim = 1_500_000, 384            # 4.6 GB of f64 rows in ONE call
rng = np.random.default_rng(5)
rows = rng.standard_normal((n, dim), dtype=np.float32).astype(np.float64)
ids = np.arange(n, dtype=np.uint64) + np.uint64(10)
idx = V.FlatIndex(dim)
t0 = time.perf_counter(); idx.add_rows(ids, rows, validate=False); dt = time.perf_counter() - t0
assert len(idx) == n
for p in (0, n // 2, n - 1):
    assert np.array_equal(np.asarray(idx.get_vector(int(ids[p])).values), rows[p])
    assert idx.search(rows[p], 1, 3)[0].id == int(ids[p])
e_ids, e_vals = idx.export()
assert np.array_equal(e_ids, ids) and np.array_equal(e_vals[-1], rows[-1]) and np.array_equal(e_vals[n // 3], rows[n // 3])
emb = rows[:700_000].astype(np.float32)       # 1.07 GB of f32 embeddings in one call
j = V.FlatIndex(dim); j.add_embeddings(np.arange(700_000, dtype=np.uint64), emb)
assert len(j) == 700_000
print(f"host bulk probe: {rows.nbytes / 2**30:.1f} GiB of host rows in one add ({dt:.1f}s = {n / dt / 1e6:.1f} M rows/s), export round trip equal, ok")
