#!/usr/bin/env python3
"""Handles created, filled, searched and destroyed from eight threads at once (collections come and go in a server,
src/client.rs:212-260): no shared global state may break, every answer is right, memory comes back."""
import os, sys, threading, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V
rng = np.random.default_rng(6)
rows = rng.standard_normal((3000, 40))
torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
errors = []
def worker(t):
    try:
        for r in range(25):
            kind = (t + r) % 4
            if kind == 0:
                h = V.FlatIndex(40)
            elif kind == 1:
                h = V.HNSWIndex(40, 0)
            else:
                h = V.MultiFlatIndex(40, [0, 0], "replicas" if kind == 2 else "row_shards")
            n = 200 + 100 * ((t * 7 + r) % 20)
            h.add_rows(np.arange(n, dtype=np.uint64) + np.uint64(t * 10**6), rows[:n])
            res = h.search(rows[(t + r) % n], 3, 0)
            assert res[0].id == (t + r) % n + t * 10**6, (t, r, res[0].id)
            if kind != 1 and r % 5 == 0:
                b = h.search_batch(rows[:70], 5, 1)       # 70 queries: the MFMA path's scratch on a small index
                assert b[0][:, 0].tolist() == (np.arange(70) + t * 10**6).tolist()
            c = h.clone(); del h
            assert len(c) == n
            del c
    except Exception as e:  # noqa: BLE001
        errors.append((t, repr(e)))
th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
t0 = time.time(); [x.start() for x in th]; [x.join(timeout=600) for x in th]
hung = sum(x.is_alive() for x in th)
gc.collect(); torch.cuda.synchronize()
back = free0 - torch.cuda.mem_get_info()[0]
print(f"8 threads x 25 handles (flat / HNSW / replicas / row shards) in {time.time() - t0:.1f}s; errors {errors[:2]}; hung {hung}; {back / 2**20:.0f} MiB not returned")
assert not errors and not hung and back < 768 * 2**20
print("handle churn ok")
