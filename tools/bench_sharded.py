#!/usr/bin/env python3
"""Config 3 of BASELINE.json: flat L2, N = 10M, dim = 768, batch = 1024 queries, rows sharded across
the ranks, ONE RCCL all-gather of the per-shard exact top-k inside libvectorlite_amd.so
(vl_comm_create + vl_shard_sync + vl_shard_search_batch), device merge.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      tools/bench_sharded.py --rows 10000000 --dim 768 --batch 1024

A single rank runs the same code (world-1 communicator).  torch.distributed is used only to hand rank 0's
ncclUniqueId to the other ranks and for the timing barrier.  Prints one JSON line on rank 0; `--rows-per-rank`
sizes one rank's shard directly (1 250 000 = config 3's shard)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_PEAK_TFLOPS = 2500.0  # bf16 dense, MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--rows-per-rank", type=int, default=0)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--check", type=int, default=8, help="queries re-answered by single search() on rank 0's shard-local path")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl" if world > 1 else "gloo", rank=rank, world_size=world,
                                **({"device_id": dev} if world > 1 else {}))
    import vectorlite_amd as V
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex, shard_ranges
    total = a.rows_per_rank * world if a.rows_per_rank else a.rows
    starts = shard_ranges(total, world)
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
    comm = Comm.from_torch_distributed(device=local)
    sh = ShardedFlatIndex(idx, comm=comm)
    assert (sh.offset, sh.total) == (lo, total)
    rng = np.random.Generator(np.random.PCG64(4321))  # the same queries on every rank
    Q = rng.standard_normal((a.batch, a.dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    sh.search_batch(Q[: min(a.batch, 128)], a.k, a.metric)
    sh.search_batch(Q, a.k, a.metric)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ids, scores, n, gpos = sh.search_batch(Q, a.k, a.metric, with_positions=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev if world > 1 else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the same with the batch already in each rank's GPU memory (vl_shard_search_batch_dev: what a broadcast leaves behind)
    dQ = torch.from_numpy(Q).to(dev)
    sh.search_batch(dQ, a.k, a.metric)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        d_ids, d_scores, d_n = sh.search_batch(dQ, a.k, a.metric)
    torch.cuda.synchronize()
    t_dev = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if world > 1 else "cpu")
    dist.all_reduce(t_dev, op=dist.ReduceOp.MAX)
    dev_same = bool(d_ids.tolist() == ids.tolist() and d_scores.tolist() == scores.tolist())
    # every rank must hold the same merged answer: compare a digest
    dig = torch.tensor([int(np.bitwise_xor.reduce(ids.reshape(-1).view(np.int64))),
                        int(np.bitwise_xor.reduce(np.ascontiguousarray(scores).reshape(-1).view(np.int64)))], dtype=torch.int64,
                       device=dev if world > 1 else "cpu")
    lo_d, hi_d = dig.clone(), dig.clone()
    dist.all_reduce(lo_d, op=dist.ReduceOp.MIN); dist.all_reduce(hi_d, op=dist.ReduceOp.MAX)
    same = bool((lo_d == hi_d).all().item())
    # this rank's rows that made the global top-k must be what its own single search() returns for them
    agree = 0
    for qi in range(min(a.check, a.batch)):
        li, ls = idx.search_arrays(Q[qi], a.k, a.metric)
        mine = [(int(i), float(s)) for i, s, p in zip(ids[qi], scores[qi], gpos[qi]) if lo <= int(p) < hi]
        agree += int(mine == list(zip(li.tolist(), ls.tolist()))[: len(mine)])
    if rank == 0:
        dt = float(t.item())
        flops = 2.0 * a.batch * total * a.dim
        print(json.dumps({"metric": "row-sharded batched flat search (config 3 shape)", "value": round(a.steps * a.batch / dt, 1), "unit": "queries/s",
                          "n_gpus": world, "ms_per_batch": round(dt / a.steps * 1e3, 3),
                          "roofline": {"bound": "mfma", "achieved": round(flops * a.steps / dt / 1e12 / world, 1), "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(flops * a.steps / dt / 1e12 / world / MFMA_PEAK_TFLOPS, 4),
                                       "note": "whole call per GPU (host staging, MFMA filter, f64 rescoring, all-gather, merge), flops = 2*Q*N*dim"},
                          "device_queries": {"ms_per_batch": round(float(t_dev.item()) / a.steps * 1e3, 3),
                                             "queries_per_s": round(a.steps * a.batch / float(t_dev.item()), 1),
                                             "identical_to_host_queries": dev_same},
                          "identical_on_every_rank": same,
                          "own_rows_match_single_search": f"{agree}/{min(a.check, a.batch)}",
                          "config": {"rows": total, "dim": a.dim, "batch": a.batch, "k": a.k, "metric": a.metric,
                                     "rows_per_rank": hi - lo,
                                     "collective": "1 ncclAllGather of %d B per rank (vl_shard_search_batch)" % (8 * (4 + a.batch + 3 * a.batch * a.k))}}))
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
