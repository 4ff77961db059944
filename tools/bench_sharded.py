#!/usr/bin/env python3
"""Config 3 of BASELINE.json: flat L2, N = 10M, dim = 768, batch = 1024 queries, rows sharded across
the ranks, ONE RCCL all-gather of the per-shard exact top-k.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      tools/bench_sharded.py --rows 10000000 --dim 768 --batch 1024

Runs with a single rank too (no collective).  Prints one JSON line on rank 0."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    import vectorlite_amd as V
    from vectorlite_amd.sharded import ShardedFlatIndex, shard_ranges
    starts = shard_ranges(a.rows, world)
    lo, hi = starts[rank], starts[rank + 1]
    idx = V.FlatIndex(a.dim, device=local)
    idx.reserve(hi - lo)
    pos = lo
    while pos < hi:
        c = min(250_000, hi - pos)
        g = torch.Generator(device=dev); g.manual_seed(1234 + pos)  # a function of the global row range
        x = torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(pos, pos + c, dtype=np.uint64), x, validate=False)
        pos += c
    sh = ShardedFlatIndex(idx, offset=lo, device=dev if world > 1 else None)
    rng = np.random.Generator(np.random.PCG64(4321))  # the same queries on every rank
    Q = rng.standard_normal((a.batch, a.dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    sh.search_batch(Q[: min(a.batch, 128)], a.k, a.metric)
    sh.search_batch(Q, a.k, a.metric)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ids, scores, n = sh.search_batch(Q, a.k, a.metric)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        dt = float(t.item())
        print(json.dumps({"metric": "row-sharded batched flat search", "value": round(a.steps * a.batch / dt, 1), "unit": "queries/s",
                          "n_gpus": world, "ms_per_batch": round(dt / a.steps * 1e3, 3),
                          "config": {"rows": a.rows, "dim": a.dim, "batch": a.batch, "k": a.k, "metric": a.metric,
                                     "rows_per_rank": hi - lo, "collective": "1 all_gather_into_tensor of %d B per rank" % (a.batch * (a.k + 1) * 24)}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
