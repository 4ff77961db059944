#!/usr/bin/env python3
"""Throughput of search_batch (k_scan_batch: 8 queries per slab pass) vs single-query search."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--nq", type=int, default=256)
    a = ap.parse_args()
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    idx = V.FlatIndex(a.dim); idx.reserve(a.rows)
    done = ci = 0
    while done < a.rows:
        c = min(500_000, a.rows - done)
        g = torch.Generator(device=dev); g.manual_seed(1234 + ci)
        x = torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c; ci += 1
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((a.nq, a.dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx.search_batch(Q[:16], 10, a.metric)
    idx.search_batch(Q[:min(a.nq, 128)], 10, a.metric)
    idx.search_batch(Q, 10, a.metric)
    idx.profile_read(); idx.profile_enable(True)
    t0 = time.perf_counter(); bi, bs, bn = idx.search_batch(Q, 10, a.metric); tb = time.perf_counter() - t0
    n, ms, b = idx.profile_read()
    print(f"batch: {a.nq} queries in {tb*1e3:.1f} ms -> {a.nq/tb:.1f} QPS; scan passes {n}, avg {ms/max(n,1):.3f} ms/pass = {b/max(n,1)/(ms/max(n,1))/1e6:.0f} GB/s slab stream")
    ns = min(a.nq, 64)
    t0 = time.perf_counter()
    singles = [idx.search_arrays(Q[i], 10, a.metric) for i in range(ns)]
    ts = time.perf_counter() - t0
    print(f"single: {ns} queries in {ts*1e3:.1f} ms -> {ns/ts:.1f} QPS")
    ok = all(bi[i].tolist() == singles[i][0].tolist() and bs[i].tolist() == singles[i][1].tolist() for i in range(ns))
    print("batch == single (ids and scores):", ok)

if __name__ == "__main__":
    main()
