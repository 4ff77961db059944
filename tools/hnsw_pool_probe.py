#!/usr/bin/env python3
"""Walk-scratch pool under pressure: a graph large enough that one 'max' scratch is 2 GiB of visited sets, eight threads
submitting large batches at once.  The pool is bounded by bytes (8 GiB per index): device memory in use must stay under
that bound (+ slack), nobody may hang, and every answer must equal the single-threaded one (walks are deterministic).
usage: python tools/hnsw_pool_probe.py [rows]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dim, latent = 64, 12
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
A = torch.randn((latent, dim), dtype=torch.float64, device=dev, generator=g)
hn = V.HNSWIndex(dim, 0)
t0 = time.time()
done = 0
while done < n:
    c = min(500_000, n - done)
    x = torch.randn((c, latent), dtype=torch.float64, device=dev, generator=g) @ A
    x += 0.05 * torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    hn.add_rows(np.arange(done, done + c, dtype=np.uint64), x)
    done += c
print(f"built {n} nodes x {dim} in {time.time() - t0:.0f}s", flush=True)
rng = np.random.default_rng(3)
Q = rng.standard_normal((8, 700, latent)) @ A.cpu().numpy() + 0.05 * rng.standard_normal((8, 700, dim))
Q /= np.linalg.norm(Q, axis=2, keepdims=True)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
want = [hn.search_batch(Q[t], 10, 0, ef=64) for t in range(8)]      # one at a time: the reference answers
free1 = torch.cuda.mem_get_info()[0]
print(f"one 700-query batch at a time: pool holds {(free0 - free1) / 2**30:.2f} GiB", flush=True)
low = [free1]
errors = []
stop = False
def watch():
    while not stop:
        low[0] = min(low[0], torch.cuda.mem_get_info()[0]); time.sleep(0.002)
def worker(t):
    try:
        for r in range(6):
            got = hn.search_batch(Q[(t + r) % 8], 10, 0, ef=64)
            w = want[(t + r) % 8]
            assert got[0].tolist() == w[0].tolist() and got[1].tolist() == w[1].tolist() and got[2].tolist() == w[2].tolist()
            one = hn.search_arrays(Q[t][r], 10, 0, ef=64)      # small class beside the large ones
            assert one[0].tolist() == want[t][0][r][: len(one[0])].tolist()
    except Exception as e:  # noqa: BLE001
        errors.append(repr(e))
wt = threading.Thread(target=watch); wt.start()
t1 = time.time()
th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
[x.start() for x in th]; [x.join(timeout=300) for x in th]
stop = True; wt.join()
alive = [x.is_alive() for x in th]
print(f"8 threads x 6 batches of 700 in {time.time() - t1:.1f}s; errors {errors}; hung threads {sum(alive)}; "
      f"pool high-water {(free0 - low[0]) / 2**30:.2f} GiB (budget 8 GiB of visited sets)", flush=True)
assert not errors and not any(alive) and (free0 - low[0]) < 10 * 2**30
print("pool probe ok")
