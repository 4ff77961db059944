#!/usr/bin/env python3
"""Sweep k_scan launch shapes (lanes per row, row groups in flight, grid) on one GPU.
Interleaved rounds in ONE process (cdna guide rule 24); prints median/min kernel ms and GB/s."""
import argparse
import itertools
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--per-round", type=int, default=10)
    ap.add_argument("--g", default="32,16,8,4")
    ap.add_argument("--u", default="1,2,4")
    ap.add_argument("--grid", default="0,768,1024,1280,1536,1792")
    a = ap.parse_args()
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    idx = V.FlatIndex(a.dim)
    idx.reserve(a.rows)
    done = 0
    ci = 0
    while done < a.rows:
        c = min(500_000, a.rows - done)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + ci)
        x = torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c
        ci += 1
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((64, a.dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    configs = []
    for g_, u_, grid in itertools.product(a.g.split(","), a.u.split(","), a.grid.split(",")):
        configs.append((int(g_), int(u_), int(grid)))
    res = {c: [] for c in configs}
    ref = None
    valid = {}
    idx.profile_enable(True)
    for r in range(a.rounds):
        for c in configs:
            os.environ["VL_SCAN_G"], os.environ["VL_SCAN_U"], os.environ["VL_SCAN_GRID"] = map(str, c)
            idx.profile_read()
            for i in range(a.per_round):
                ids, sc = idx.search_arrays(Q[i % 64], 10, a.metric)
                if i == 0:
                    if ref is None:
                        ref = (ids.tolist(), sc.tolist())
                    valid[c] = (ids.tolist(), sc.tolist()) == ref
            n, ms, b = idx.profile_read()
            res[c].append((ms / max(n, 1), b / max(n, 1)))
    print(f"rows={a.rows} dim={a.dim} metric={a.metric}")
    print("   G   U  grid   med_ms   min_ms   med_GB/s  ok")
    rows = []
    for c, v in res.items():
        ms = [x[0] for x in v]
        by = v[0][1]
        med = statistics.median(ms)
        rows.append((med, c, min(ms), by / med / 1e6, valid.get(c)))
    for med, c, mn, gbs, ok in sorted(rows):
        print(f"{c[0]:4d} {c[1]:3d} {c[2]:5d}  {med:7.4f}  {mn:7.4f}  {gbs:9.1f}  {ok}")


if __name__ == "__main__":
    main()
