#!/usr/bin/env python3
"""The program rocprofv3 --pmc FETCH_SIZE runs for configs 3 and 5 (tools/r4_traffic.sh): ONLY whole batches of one shape
go through the batch filter, so that (dispatches of k_mfma_rows<K, 1, metric>) / batches = stages x sequences exactly.
Prints one JSON line: the workload, the batches run and the filter's launch plan (vl_index_last_filter)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c5"])
    ap.add_argument("--batches", type=int, default=4)
    a = ap.parse_args()
    rows, dim, nq, metric = (1_250_000, 768, 1024, 1) if a.config == "c3" else (10_000_000, 384, 4096, 0)
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    idx = V.FlatIndex(dim)
    idx.reserve(rows)
    done = ci = 0
    while done < rows:
        c = min(250_000, rows - done)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c
        ci += 1
        del x
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((nq, dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    dQ = torch.from_numpy(Q).to(dev)
    for _ in range(a.batches):
        idx.search_batch_device(dQ, 10, metric)
    torch.cuda.synchronize()
    seq = 2048 if dim <= 512 else 1536
    print(json.dumps({"config": a.config, "rows": rows, "dim": dim, "queries": nq, "metric": metric, "batches": a.batches,
                      "sequences_per_batch": (nq + seq - 1) // seq, "plan": idx.last_filter()}), flush=True)


if __name__ == "__main__":
    main()
