#!/usr/bin/env python3
"""A/B of the two shapes of k_mfma_rows' pass 1 in ONE process, interleaved rounds (guide rule 24): VL_MFMA_ROWS_SHAPE=8x32
(two waves per SIMD, 32-row wave blocks) against 4x64 (one wave per SIMD, 64-row wave blocks).  Filter time per batch from
HIP events around the filter's launch sequence (vl_index_profile_*); answers of the two shapes must be identical.
usage: python tools/rows_shape_ab.py [--rows N] [--dim D] [--nq Q] [--metric 0|1|3] [--rounds R]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_250_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--metric", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    idx = V.FlatIndex(a.dim)
    idx.reserve(a.rows)
    done = ci = 0
    while done < a.rows:
        c = min(250_000, a.rows - done)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + ci)
        x = torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c
        ci += 1
        del x
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((a.nq, a.dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    dQ = torch.from_numpy(Q).to(dev)
    shapes = ("8x32", "4x64")
    ans = {}
    for sh in shapes:
        os.environ["VL_MFMA_ROWS_SHAPE"] = sh
        ans[sh] = idx.search_batch_device(dQ, 10, a.metric)
    same = bool(np.array_equal(ans["8x32"][0], ans["4x64"][0]) and np.array_equal(ans["8x32"][1], ans["4x64"][1]))
    filt = {sh: [] for sh in shapes}
    wall = {sh: [] for sh in shapes}
    for r in range(a.rounds):
        for sh in shapes:
            os.environ["VL_MFMA_ROWS_SHAPE"] = sh
            idx.profile_read()
            idx.profile_enable(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                idx.search_batch_device(dQ, 10, a.metric)
            w = (time.perf_counter() - t0) / a.reps
            idx.profile_enable(False)
            n_pass, ms, _ = idx.profile_read()
            filt[sh].append(ms / a.reps)
            wall[sh].append(w * 1e3)
    flops = 2.0 * a.nq * a.rows * a.dim
    out = {"workload": f"{a.nq} queries x {a.rows} rows x dim {a.dim}, metric {a.metric}", "answers_identical": same}
    for sh in shapes:
        f = np.asarray(filt[sh])
        out[sh] = {"filter_ms_median": round(float(np.median(f)), 4), "filter_ms_min": round(float(f.min()), 4),
                   "frac_of_bf16_peak_median": round(flops / (float(np.median(f)) * 1e-3) / 2.5e15, 4),
                   "whole_call_ms_median": round(float(np.median(wall[sh])), 4), "filter_ms_all": [round(float(v), 4) for v in f]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
