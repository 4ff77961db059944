#!/usr/bin/env python3
"""Scale probe: one index far beyond the benchmark size on one 288 GB GPU -- every byte offset past 2^32 and 2^36,
fast path == exact path, batch == single, rows found where they were put.  usage: python tools/scale_probe.py [rows] [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dev = torch.device("cuda", 0)
t0 = time.time()
idx = V.FlatIndex(dim); idx.reserve(n)
done = ci = 0
while done < n:
    c = min(500_000, n - done)
    g = torch.Generator(device=dev); g.manual_seed(4321 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(done, done + c, dtype=np.uint64) * np.uint64(3) + np.uint64(1), x, validate=False)
    done += c; ci += 1
    del x
    if ci % 20 == 0:
        print(f"  {done} rows, {time.time() - t0:.0f}s, free {torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB", flush=True)
torch.cuda.synchronize()
print(f"built {n} x {dim} in {time.time() - t0:.0f}s; device memory free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
rng = np.random.default_rng(7)
Q = rng.standard_normal((40, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
# rows near the end of the slab are found where they were put (ids = 3 pos + 1)
for probe in (0, n // 2 + 12345, n - 1):
    v = idx.get_vector(3 * probe + 1).values
    r = idx.search(v, 1, 0)
    assert r[0].id == 3 * probe + 1 and abs(r[0].score - 1.0) < 1e-12, (probe, r[0].id, r[0].score)
ok = 0
for metric in range(4):
    for qi in range(3):
        fi, fs = idx.search_arrays(Q[qi], 10, metric)
        assert V.last_path() == V.PATH_FAST
        idx.force_path(V.PATH_EXACT_SELECT)
        try:
            ei, es = idx.search_arrays(Q[qi], 10, metric)
        finally:
            idx.force_path(0)
        assert fi.tolist() == ei.tolist() and fs.tolist() == es.tolist(), (metric, qi)
        ok += 1
print(f"fast == exact on {ok} (metric, query) pairs", flush=True)
idx.profile_read(); idx.profile_enable(True)
t1 = time.perf_counter()
for i in range(20):
    idx.search_arrays(Q[i], 10, 0)
dt = (time.perf_counter() - t1) / 20
idx.profile_enable(False)
nl, ms, by = idx.profile_read()
print(f"single query: {dt * 1e3:.3f} ms = {1 / dt:.1f} QPS; k_scan {ms / nl:.3f} ms = {by / nl / (ms / nl * 1e-3) / 1e12:.2f} TB/s", flush=True)
bi, bs, bn = idx.search_batch(Q[:32], 10, 0)   # builds the bf16 copy of the slab (MFMA filter)
for qi in (0, 7, 31):
    si, ss = idx.search_arrays(Q[qi], 10, 0)
    assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), qi
t2 = time.perf_counter(); idx.search_batch(Q[:32], 10, 0); print(f"batch of 32: {(time.perf_counter() - t2) * 1e3:.2f} ms; device memory free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
print("scale probe ok")
