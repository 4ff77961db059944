#!/usr/bin/env python3
"""Batched flat search as a bf16 MFMA GEMM (K4): BASELINE config 5 (Q = 4096, N = 10 M, dim 384, cosine) and one
rank's shard of config 3 (1024 queries, 1.25 M x 768, Euclidean).  One JSON line with a `roofline` object:
flops = 2 * Q * N * dim against the 2.5 PFLOP/s dense bf16 peak (MI355X_MICROARCH.md), once over the whole
search_batch call (host staging, H2D, filter kernels, exact f64 rescoring, D2H -- what a caller sees) and once over
the filter kernels alone (HIP events around launch_mfma_candidates, summed over the passes of <= 1024 queries).

  python tools/bench_mfma.py --config c5      python tools/bench_mfma.py --config c3
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_TFLOPS = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5", choices=["c5", "c3"])
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--nq", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", type=int, default=32)
    a = ap.parse_args()
    rows, dim, nq, metric, name = ((10_000_000, 384, 4096, 0, "config 5: Q=4096 x N=10M x dim 384, cosine, k=10, 1 GPU") if a.config == "c5"
                                   else (1_250_000, 768, 1024, 1, "config 3, one rank's shard: 1024 queries x 1.25M x dim 768, Euclidean, k=10"))
    rows, nq, dim = a.rows or rows, a.nq or nq, a.dim or dim
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    idx = V.FlatIndex(dim); idx.reserve(rows)
    done = ci = 0
    while done < rows:
        c = min(250_000, rows - done)
        g = torch.Generator(device=dev); g.manual_seed(1234 + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c; ci += 1
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((nq, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx.search_batch(Q[:64], 10, metric)       # builds the bf16 slab, scratch
    idx.search_batch(Q, 10, metric)            # warm
    torch.cuda.synchronize()
    idx.profile_read(); idx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        bi, bs, bn = idx.search_batch(Q, 10, metric)
    wall = (time.perf_counter() - t0) / a.reps
    idx.profile_enable(False)
    n_pass, ms, _ = idx.profile_read()
    # the same batch with the queries already on the GPU (vl_index_search_batch_dev: staged by a kernel, no PCIe copy)
    dQ = torch.from_numpy(Q).to(dev)
    idx.search_batch_device(dQ, 10, metric)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        di, ds, dn = idx.search_batch_device(dQ, 10, metric)
    wall_dev = (time.perf_counter() - t0) / a.reps
    dev_same = bool(di.tolist() == bi.tolist() and ds.tolist() == bs.tolist())
    kern = ms / a.reps * 1e-3
    flops = 2.0 * nq * rows * dim
    ns = min(a.check, nq)
    pick = np.linspace(0, nq - 1, ns).astype(int)
    ok = 0
    for qi in pick:
        si, ss = idx.search_arrays(Q[qi], 10, metric)
        ok += int(bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist())
    out = {
        "metric": "batched flat search as a bf16 MFMA GEMM", "value": round(nq / wall, 1), "unit": "queries/s",
        "ms_per_batch": round(wall * 1e3, 3), "filter_kernels_ms_per_batch": round(kern * 1e3, 3),
        "passes_per_batch": n_pass // max(a.reps, 1),
        "device_queries": {"ms_per_batch": round(wall_dev * 1e3, 3), "queries_per_s": round(nq / wall_dev, 1),
                           "identical_to_host_queries": dev_same},
        "roofline": {"bound": "mfma", "achieved": round(flops / kern / 1e12, 1), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(flops / kern / 1e12 / PEAK_TFLOPS, 4), "kernel": "k_mfma_rows: sampling pass + pass-1 stages (+ thresholds, refines, candidate select)",
                     "flops_per_batch": flops, "traffic": None,
                     "whole_call": {"achieved": round(flops / wall / 1e12, 1), "frac": round(flops / wall / 1e12 / PEAK_TFLOPS, 4)}},
        "parity": f"{ok}/{ns} sampled rows identical to single search() (ids and f64 scores)",
        "dtype": "bf16 filter, f64 scores", "config": {"workload": name, "rows": rows, "dim": dim, "nq": nq, "metric": metric, "k": 10},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
