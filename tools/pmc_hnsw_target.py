#!/usr/bin/env python3
"""The program rocprofv3 --pmc FETCH_SIZE runs for config 4 (tools/r4_traffic_hnsw.sh): bench.py's own latent-16 corpus and
queries (the same generator calls, so the graph and the walks are the bench's), then ONLY walks at one beam width, so that
every k_hnsw_search dispatch of the pass belongs to it.  Prints one JSON line: workload, beam, batches, evaluations per query."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ef", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=int, default=1000)
    a = ap.parse_args()
    import torch
    import bench
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    name = "latent16"
    g = torch.Generator(device=dev)
    g.manual_seed(99 + len(name))
    state = {"A": torch.randn((16, a.dim), dtype=torch.float64, device=dev, generator=g)}
    hn = V.HNSWIndex(a.dim, 0)
    done = 0
    while done < a.rows:
        c = min(250_000, a.rows - done)
        x = bench.gen_c4_rows(torch, dev, name, c, a.dim, 31337 + done, state)
        hn.add_rows(np.arange(done, done + c, dtype=np.uint64), x)
        done += c
        del x
    Q = bench.gen_c4_rows(torch, dev, name, a.nq, a.dim, 4321, state).cpu().numpy()
    q0, e0 = hn.walk_stats()
    for _ in range(a.batches):
        hn.search_batch(Q, 10, 0, ef=(0 if a.ef == 10 else a.ef))
    q1, e1 = hn.walk_stats()
    print(json.dumps({"config": "c4", "data": name, "rows": a.rows, "dim": a.dim, "queries": a.nq, "ef": a.ef, "batches": a.batches,
                      "ef_construction": 400, "distance_evals_per_query": round((e1 - e0) / max(q1 - q0, 1), 1),
                      "list_slots": 1 if a.ef <= 64 else (2 if a.ef <= 128 else (4 if a.ef <= 256 else 8))}), flush=True)


if __name__ == "__main__":
    main()
