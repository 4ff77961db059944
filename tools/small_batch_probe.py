#!/usr/bin/env python3
"""What a SMALL batch costs on the headline index (10 M x 384, cosine): the coalescer hands the library batches of 2..256 queries
(16 host threads -> ~16 per pass), so the per-batch time at those sizes is the many-readers throughput.  Per batch size: median
wall time of search_batch, the path taken (last_filter), queries per second, and that the answers equal the lone searches.
usage: python tools/small_batch_probe.py [--rows 10000000] [--dim 384] [--sizes 1,2,4,...]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--sizes", default="1,2,4,8,12,16,24,32,48,64,96,128,256")
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    idx = V.FlatIndex(a.dim)
    idx.reserve(a.rows)
    done = 0
    while done < a.rows:
        c = min(500_000, a.rows - done)
        x = torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c
        del x
    Q = np.random.default_rng(3).standard_normal((512, a.dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    lone = [idx.search_arrays(Q[i], 10, a.metric) for i in range(8)]
    for nq in [int(s) for s in a.sizes.split(",")]:
        for _ in range(3):
            idx.search_batch(Q[:nq], 10, a.metric)
        ts = []
        for r in range(a.reps):
            q = Q[(r * 7) % 64:(r * 7) % 64 + nq]
            t0 = time.perf_counter()
            bi, bs, bn = idx.search_batch(q, 10, a.metric)
            ts.append(time.perf_counter() - t0)
        bi, bs, bn = idx.search_batch(Q[:nq], 10, a.metric)
        same = sum(1 for i in range(min(nq, 8)) if bi[i].tolist() == lone[i][0].tolist() and bs[i].tolist() == lone[i][1].tolist())
        ms = float(np.median(ts)) * 1e3
        try:
            plan = idx.last_filter()
        except Exception as e:  # noqa
            plan = str(e)
        print(json.dumps({"queries": nq, "ms_per_batch": round(ms, 4), "queries_per_s": round(nq / ms * 1e3, 0),
                          "ms_min": round(min(ts) * 1e3, 4), "identical_to_lone": f"{same}/{min(nq, 8)}", "last_filter": plan}), flush=True)


if __name__ == "__main__":
    main()
