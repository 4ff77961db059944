#!/usr/bin/env python3
"""Capacity growth until the card is full: bulk adds without reserve() grow the buffers geometrically; the growth that no
longer fits must fail with the out-of-memory status and leave the index exactly as it was (rows, answers), and a smaller
add must still work afterwards."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

dim = 384
dev = torch.device("cuda", 0)
idx = V.FlatIndex(dim)
g = torch.Generator(device=dev); g.manual_seed(1)
probe_rows = {}
n = 0
failed_at = None
t0 = time.time()
while n < 80_000_000:
    c = 1_000_000
    x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    ids = np.arange(n, n + c, dtype=np.uint64)
    try:
        idx.add_rows(ids, x, validate=False)
    except V.VectorLiteError as e:
        failed_at = (n, str(e)[:120])
        del x
        break
    probe_rows[n + 5] = x[5].cpu().numpy()
    n += c
    del x
    if n % 8_000_000 == 0:
        print(f"  {n} rows, free {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB, {time.time() - t0:.0f}s", flush=True)
print("growth stopped at", failed_at, "len", len(idx), f"free {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB", flush=True)
assert failed_at is not None and len(idx) == n and "memory" in failed_at[1].lower()
for id_, row in list(probe_rows.items())[:: max(1, len(probe_rows) // 8)]:
    got = idx.get_vector(id_)
    assert got is not None and np.array_equal(np.asarray(got.values), row)
    assert idx.search(row, 1, 0)[0].id == id_
# the card is full and the index holds exactly its capacity: a delete makes room for exactly one add
victim = int(list(probe_rows)[0])
idx.delete(victim)
assert len(idx) == n - 1 and idx.get_vector(victim) is None
x = torch.randn((1, dim), dtype=torch.float64, device=dev, generator=g)
idx.add_rows(np.array([10**12], dtype=np.uint64), x, validate=False)
assert len(idx) == n and idx.search(x[0].cpu().numpy(), 1, 0)[0].id == 10**12
print("growth probe ok")
