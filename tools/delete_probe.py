#!/usr/bin/env python3
"""Order-preserving delete on a big index (Vec::retain, src/index/flat.rs:94): gigabytes of rows slide through the bounce
buffer; afterwards every remaining row must still be where its id says, duplicates of a deleted id must all be gone, and
the fast path must still equal the exact one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 384
dev = torch.device("cuda", 0)
idx = V.FlatIndex(dim); idx.reserve(n)
keep = {}
for ci, c0 in enumerate(range(0, n, 500_000)):
    c = min(500_000, n - c0)
    g = torch.Generator(device=dev); g.manual_seed(99 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    ids = np.arange(c0, c0 + c, dtype=np.uint64) * np.uint64(5) + np.uint64(2)
    if ci == 2:
        ids[1000:1004] = 777          # four MORE rows share the id of position 155 (5 * 155 + 2 = 777): FlatIndex::new keeps them
    idx.add_rows(ids, x, validate=False)
    for p in (0, 17, c - 1):
        keep[int(ids[p])] = x[p].cpu().numpy()
    del x
assert len(idx) == n
victims = [int(5 * 10 + 2), int(5 * (n // 2) + 2), 777, int(5 * (n - 2) + 2), 123456789]   # early, middle, the duplicated id, late, absent
t0 = time.perf_counter()
for v in victims:
    idx.delete(v)
dt = time.perf_counter() - t0
assert len(idx) == n - 3 - 5, len(idx)
assert idx.get_vector(777) is None
bad = 0
for id_, row in keep.items():
    if id_ in victims:
        continue
    got = idx.get_vector(id_)
    assert got is not None and np.array_equal(np.asarray(got.values), row), id_
    r = idx.search(row, 1, 0)
    assert r[0].id == id_, (id_, r[0].id)
rng = np.random.default_rng(1)
for m in range(4):
    q = rng.standard_normal(dim)
    fi, fs = idx.search_arrays(q, 10, m)
    idx.force_path(V.PATH_EXACT_SELECT)
    try:
        ei, es = idx.search_arrays(q, 10, m)
    finally:
        idx.force_path(0)
    assert fi.tolist() == ei.tolist() and fs.tolist() == es.tolist(), m
print(f"delete probe: {n} x {dim} rows, 5 deletes (one id held by five rows, one absent) in {dt * 1e3:.0f} ms; {len(keep)} kept rows found where their ids say; fast == exact; ok")
