#!/usr/bin/env python3
"""HNSW at the reference's parameter space (round 4): ef_construction in {128, 200, 400} -- 400 is what SURVEY 9.5 recalls as
crate hnsw 0.11.0's Params::default(), unverifiable here -- at the reference's strict beam ef = min(k, len) = 10
(src/index/hnsw.rs:437,454) and at ef 32 / 128, on two embedding-like distributions:
  latent16     rows = A z + 0.05 noise, z in R^16 (low intrinsic dimension; rounds 1-3's distribution)
  clustered    a mixture of 2000 von-Mises-like clusters on the sphere (centre + 0.35 gaussian, renormalised): topical
               clusters, the other shape real sentence embeddings take
Per (distribution, ef_construction): build time, recall@10 against the exact flat order, batched QPS, distance evaluations
per query, the GPU's lone-query latency at ef 10, and the CPU walk of the SAME graph (oracle/vl_hnsw_cpu.c, 1 core) at ef 10:
its recall and its time per query -- SURVEY H5's question (is a lone strict-beam query better left on the CPU?) in like-for-like
numbers.  One JSON line per cell.   usage: python tools/hnsw_efc_sweep.py [--rows N] [--dim D] [--efcs 128,200,400]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gen(torch, dev, g, kind, n, dim, state):
    if kind == "latent16":
        if "A" not in state:
            state["A"] = torch.randn((16, dim), dtype=torch.float64, device=dev, generator=g)
        x = torch.randn((n, 16), dtype=torch.float64, device=dev, generator=g) @ state["A"]
        x += 0.05 * torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    else:
        if "C" not in state:
            c = torch.randn((2000, dim), dtype=torch.float64, device=dev, generator=g)
            state["C"] = c / torch.linalg.vector_norm(c, dim=1, keepdim=True)
        which = torch.randint(0, 2000, (n,), device=dev, generator=g)
        x = state["C"][which] + (0.35 / dim ** 0.5) * torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--efcs", default="128,200,400")
    ap.add_argument("--kinds", default="latent16,clustered")
    ap.add_argument("--cpu-queries", type=int, default=24)
    a = ap.parse_args()
    import torch
    import vectorlite_amd as V
    from oracle import oracle as O
    O.build()
    dev = torch.device("cuda", 0)
    k = 10
    for kind in a.kinds.split(","):
        for efc in [int(e) for e in a.efcs.split(",")]:
            g = torch.Generator(device=dev)
            g.manual_seed(1234)
            state = {}
            flat = V.FlatIndex(a.dim)
            flat.reserve(a.rows)
            hn = V.HNSWIndex(a.dim, 0, ef_construction=efc)
            t_build, done = 0.0, 0
            while done < a.rows:
                c = min(250_000, a.rows - done)
                x = gen(torch, dev, g, kind, c, a.dim, state)
                ids = np.arange(done, done + c, dtype=np.uint64)
                flat.add_rows(ids, x, validate=False)
                t0 = time.perf_counter()
                hn.add_rows(ids, x)
                t_build += time.perf_counter() - t0
                done += c
                del x
            Q = gen(torch, dev, g, kind, a.nq, a.dim, state).cpu().numpy()
            ti, _, _ = flat.search_batch(Q, k, 0)
            cell = {"data": kind, "rows": a.rows, "dim": a.dim, "ef_construction": efc, "build_s": round(t_build, 2),
                    "inserts_per_s": round(a.rows / t_build, 0)}
            for ef in (10, 32, 128):
                hn.search_batch(Q[:8], k, 0, ef=ef)
                q0, e0 = hn.walk_stats()
                t0 = time.perf_counter()
                hi, hs, hnn = hn.search_batch(Q, k, 0, ef=(0 if ef == 10 else ef))   # ef 10 = the trait's own search: ef = min(k, len)
                dt = time.perf_counter() - t0
                q1, e1 = hn.walk_stats()
                rec = float(np.mean([len(set(hi[i, :int(hnn[i])].tolist()) & set(ti[i].tolist())) / float(k) for i in range(a.nq)]))
                cell[f"ef{ef}"] = {"recall_at_10": round(rec, 4), "queries_per_s": round(a.nq / dt, 0),
                                   "distance_evals_per_query": round((e1 - e0) / max(q1 - q0, 1), 1)}
            lat = []
            for i in range(30):
                t0 = time.perf_counter()
                hn.search_arrays(Q[i], k, 0)
                lat.append(time.perf_counter() - t0)
            cell["gpu_lone_query_ms_ef10"] = round(float(np.median(lat)) * 1e3, 4)
            # the CPU walk of the same graph at the SAME beam (ef 10), one core
            graph = hn.graph(with_rows=True)
            walker = O.HnswCpuWalker(graph, O.COSINE)
            n_cpu = min(a.cpu_queries, a.nq)
            walker.search(Q[0], 10, k)
            walker.evals.value = 0
            t0 = time.perf_counter()
            cw = [walker.search(Q[i], 10, k) for i in range(n_cpu)]
            t_cpu = (time.perf_counter() - t0) / n_cpu
            gi, _, gn = hn.search_batch(Q[:n_cpu], k, 0)
            cell["cpu_walk_same_graph_ef10"] = {
                "queries": n_cpu, "cores": 1, "ms_per_query": round(t_cpu * 1e3, 4),
                "distance_evals_per_query": round(walker.evals.value / n_cpu, 1),
                "recall_at_10": round(float(np.mean([len(set(np.asarray(cw[i][0]).tolist()) & set(ti[i].tolist())) / float(k) for i in range(n_cpu)])), 4),
                "gpu_recall_at_10_same_queries": round(float(np.mean([len(set(gi[i, :int(gn[i])].tolist()) & set(ti[i].tolist())) / float(k) for i in range(n_cpu)])), 4)}
            del walker, graph, hn, flat
            torch.cuda.empty_cache()
            print(json.dumps(cell), flush=True)


if __name__ == "__main__":
    main()
