#!/usr/bin/env python3
"""The many-readers figure without Python's GIL: T NATIVE threads (tools/native_loadgen.c, pthreads) in a closed loop of single
searches on one handle as created (coalescing on, window 0, adaptive gather), and the same with the gather off.  bench.py's
`concurrent_16_threads` runs Python threads: every return-and-call-again passes through the GIL, one thread at a time, which is what
the leader's wait (~240 us per pass there) is spent on; a Rust or C caller is back in microseconds.
usage: python tools/concurrent_native.py [--rows 10000000] [--dim 384] [--threads 16,64] [--per-thread 60]"""
import argparse, ctypes as C, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--threads", default="16,64")
    ap.add_argument("--per-thread", type=int, default=60)
    ap.add_argument("--index", default="flat", choices=["flat", "hnsw"], help="hnsw: config 4's index (latent-16 rows, default profile), strict beam")
    a = ap.parse_args()
    so = os.path.join(tempfile.mkdtemp(), "native_loadgen.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(ROOT, "tools", "native_loadgen.c")], check=True)
    G = C.CDLL(so)
    import torch
    import vectorlite_amd as V
    from vectorlite_amd import _lib
    L = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    hnsw = a.index == "hnsw"
    idx = V.HNSWIndex(a.dim, 0) if hnsw else V.FlatIndex(a.dim)
    if not hnsw:
        idx.reserve(a.rows)
    A = torch.randn((16, a.dim), dtype=torch.float64, device=dev, generator=g)

    def gen(c):
        if hnsw:  # rows = A z + 0.05 noise, z in R^16: bench.py's latent-16 corpus
            x = torch.randn((c, 16), dtype=torch.float64, device=dev, generator=g) @ A
            x += 0.05 * torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        else:
            x = torch.randn((c, a.dim), dtype=torch.float64, device=dev, generator=g)
        return x / torch.linalg.vector_norm(x, dim=1, keepdim=True)
    done = 0
    while done < a.rows:
        c = min(250_000 if hnsw else 500_000, a.rows - done)
        x = gen(c)
        if hnsw:
            idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x)
        else:
            idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c
        del x
    k = 10
    fn = C.cast(L.vl_index_search_cap, C.c_void_p)
    for T in [int(t) for t in a.threads.split(",")]:
        per = a.per_thread
        if hnsw:
            Q = np.ascontiguousarray(gen(T * per).cpu().numpy())
        else:
            Q = np.random.default_rng(11 + T).standard_normal((T * per, a.dim))
            Q /= np.linalg.norm(Q, axis=1, keepdims=True)
            Q = np.ascontiguousarray(Q)
        lone = [idx.search_arrays(Q[t * per], k, 0) for t in range(T)]
        idx.search_batch(Q[:16], k, 0)      # (flat: the bf16 copy of the rows exists before anything is timed)
        out = {"index": a.index, "threads": T, "queries": T * per, "rows": a.rows, "dim": a.dim}
        for label, adaptive in (("settle", True), ("adaptive_gather", True), ("window0_without_gather", False)):
            idx.coalesce_gather(adaptive)
            b0, q0 = idx.coalesce_stats()
            w0, us0 = idx.coalesce_gather()
            lat = np.zeros(T * per)
            fid = np.zeros(T * k, dtype=np.uint64)
            fsc = np.zeros(T * k)
            el = C.c_double(0.0)
            rc = G.vl_loadgen(fn, idx._h, Q.ctypes.data_as(C.c_void_p), C.c_uint64(a.dim), C.c_uint64(k), C.c_int(0),
                              C.c_int(T), C.c_int(per), lat.ctypes.data_as(C.c_void_p), fid.ctypes.data_as(C.c_void_p),
                              fsc.ctypes.data_as(C.c_void_p), C.byref(el))
            assert rc == 0, rc
            b1, q1 = idx.coalesce_stats()
            w1, us1 = idx.coalesce_gather()
            same = sum(int(fid[t * k:(t + 1) * k].tolist() == lone[t][0].tolist() and fsc[t * k:(t + 1) * k].tolist() == lone[t][1].tolist())
                       for t in range(T))
            la = np.sort(lat) * 1e3
            if label != "settle":
                out[label] = {"queries_per_s": round(T * per / el.value, 1),
                              "latency_ms": {"mean": round(float(la.mean()), 3), "p50": round(float(la[len(la) // 2]), 3),
                                             "p99": round(float(la[int(len(la) * 0.99)]), 3)},
                              "passes": int(b1 - b0), "queries_per_pass": round((q1 - q0) / max(b1 - b0, 1), 2),
                              "leader_waits": int(w1 - w0), "leader_wait_us_mean": round((us1 - us0) / max(w1 - w0, 1), 1),
                              "identical_to_lone_search": f"{same}/{T}"}
        idx.coalesce_gather(True)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
