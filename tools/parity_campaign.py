"""A larger parity run than the test suite affords: every metric against the CPU oracle at N = 1 M, then
the batch / bf16 / coalesced pipelines against the single-query pipeline at N = 10 M (bit-identical or not).
usage: python tools/parity_campaign.py [queries_per_metric]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vectorlite_amd as V
from oracle import oracle as O

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dim = 384
dev = torch.device("cuda", 0)


def build(n, keep_host):
    idx = V.FlatIndex(dim); idx.reserve(n)
    parts = []
    for ci, c0 in enumerate(range(0, n, 500_000)):
        c = min(500_000, n - c0)
        g = torch.Generator(device=dev); g.manual_seed(1234 + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(c0, c0 + c, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(97), x, validate=False)
        if keep_host:
            parts.append(x.cpu().numpy())
        del x
    return idx, (np.concatenate(parts) if keep_host else None)


rng = np.random.default_rng(4321)
O.build()
n1 = 1_000_000
idx, rows = build(n1, True)
ids = np.arange(n1, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(97)
ref = O.FlatOracle(dim, ids, rows)
print(f"N = {n1}, dim = {dim}: GPU vs CPU oracle (reference-faithful f64), k = 10, {nq} queries per metric", flush=True)
for name, m in (("cosine", 0), ("euclidean", 1), ("manhattan", 2), ("dotproduct", 3)):
    Q = rng.standard_normal((nq, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[0] = rows[12345]                      # an exact hit
    Q[1] = rows[777] * 0.5 + rows[778] * 0.5
    same_ids = same_scores = 0
    worst = 0.0
    t = time.perf_counter()
    for i in range(nq):
        gi, gs = idx.search_arrays(Q[i], 10, m)
        ri, rs = ref.search(Q[i], 10, m)
        same_ids += int(gi.tolist() == ri.tolist())
        same_scores += int(gs.tolist() == rs.tolist())
        worst = max(worst, float(np.max(np.abs(gs - rs))))
    bi, bs, bn = idx.search_batch(Q, 10, m)
    batch_same = sum(int(bi[i].tolist() == ref.search(Q[i], 10, m)[0].tolist()) for i in range(min(nq, 16)))
    print(f"  {name:10s}: ids identical {same_ids}/{nq}, scores bit-identical {same_scores}/{nq}, max |score diff| {worst:.1e}; "
          f"batch ids identical {batch_same}/{min(nq, 16)}  ({time.perf_counter() - t:.0f}s)", flush=True)
del idx, ref, rows

n2 = 10_000_000
idx, _ = build(n2, False)
Q = rng.standard_normal((512, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
print(f"N = {n2}: other pipelines vs the f32 single-query pipeline (fast path), k = 10, cosine", flush=True)
single = [idx.search_arrays(Q[i], 10, 0) for i in range(512)]
bi, bs, bn = idx.search_batch(Q, 10, 0)
print("  bf16 MFMA batch of 512 :", sum(int(bi[i].tolist() == single[i][0].tolist() and bs[i].tolist() == single[i][1].tolist()) for i in range(512)), "/ 512 bit-identical", flush=True)
idx.set_single_filter("bf16")
print("  bf16-first single query:", sum(int((lambda r: r[0].tolist() == single[i][0].tolist() and r[1].tolist() == single[i][1].tolist())(idx.search_arrays(Q[i], 10, 0))) for i in range(128)), "/ 128 bit-identical", flush=True)
idx.set_single_filter("f32")
idx.force_path(V.PATH_EXACT_SELECT)
print("  exact f64 scan + select:", sum(int((lambda r: r[0].tolist() == single[i][0].tolist() and r[1].tolist() == single[i][1].tolist())(idx.search_arrays(Q[i], 10, 0))) for i in range(32)), "/ 32 bit-identical", flush=True)
idx.force_path(0)
idx.set_coalescing(64, 200)
out = [None] * 512


def w(t):
    for i in range(t, 512, 16):
        out[i] = idx.search_arrays(Q[i], 10, 0)


th = [threading.Thread(target=w, args=(t,)) for t in range(16)]
[x.start() for x in th]; [x.join() for x in th]
print("  coalesced, 16 threads  :", sum(int(out[i][0].tolist() == single[i][0].tolist() and out[i][1].tolist() == single[i][1].tolist()) for i in range(512)), "/ 512 bit-identical", flush=True)
