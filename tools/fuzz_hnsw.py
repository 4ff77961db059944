"""Randomised property hunt for the HNSW wrapper: every returned id is live and unique, every score is the reference
conversion of the exact u64 callback distance (CPU oracle), results are sorted by score, counts obey the reference's
rules (min(k, live) without tombstones, never more with them).  usage: python tools/fuzz_hnsw.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
O.build()
t_end = time.time() + budget
t_report = time.time() + 30
cases = checks = 0
recalls = []
while time.time() < t_end:
    if time.time() > t_report:
        print(f"  ... {cases} cases, {checks} result checks", flush=True); t_report = time.time() + 30
    seed += 1
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([2, 3, 8, 17, 32, 64, 100, 256, 384, 1000]))
    n = int(rng.choice([1, 5, 40, 300, 3000, 12000]))
    m = int(rng.integers(0, 4))
    lat = max(1, min(dim, int(rng.choice([2, 4, 16]))))
    rows = rng.standard_normal((n, lat)) @ rng.standard_normal((lat, dim)) + 0.02 * rng.standard_normal((n, dim))
    if rng.random() < 0.3:
        rows = np.round(rows)  # many exact ties
    ids = rng.permutation(n).astype(np.uint64) * np.uint64(7) + np.uint64(3)
    idx = V.HNSWIndex(dim, m)
    # ingest in 1-4 bulk calls, the tail one vector at a time, sometimes through a clone
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 4)))]))
    tail = int(rng.integers(0, min(n, 6) + 1))
    for a, b in zip(cuts[:-1], cuts[1:]):
        b2 = min(b, n - tail)
        if b2 > a:
            idx.add_rows(ids[a:b2], rows[a:b2])
        if rng.random() < 0.15:
            idx = idx.clone()
    for j in range(n - tail, n):
        idx.add(V.Vector(int(ids[j]), rows[j]))
    assert len(idx) == n
    dead = set()
    if n > 10 and rng.random() < 0.5:
        for j in rng.choice(n, size=max(1, n // 20), replace=False):
            idx.delete(int(ids[j])); dead.add(int(ids[j]))
    live = [int(i) for i in ids if int(i) not in dead]
    id2row = {int(ids[i]): i for i in range(n)}
    nq = int(rng.choice([1, 4, 16]))
    Q = rows[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, dim))
    k = int(rng.choice([1, 5, 10, 50, 128, 200]))
    ef = int(rng.choice([0, 0, 32, 128]))
    bi, bs, bn = idx.search_batch(Q, k, m, ef=ef)
    for qi in range(nq):
        c = int(bn[qi]); got = bi[qi, :c].tolist(); sc = bs[qi, :c].tolist()
        assert len(set(got)) == c and all(g in id2row and g not in dead for g in got), (seed, "ids")
        assert all(sc[i] >= sc[i + 1] for i in range(c - 1)), (seed, "order")
        assert c <= min(k, len(live)), (seed, "count")
        # the dot-product "distance" 1000 - clamp(a.b) is not a metric: on unnormalised rows a few large-norm hubs absorb the
        # edges and the reachable set can be smaller than k (any HNSW shows this; the reference's embeddings are unit vectors)
        if not dead and m != 3:
            assert c == min(k, len(live)), (seed, "count without tombstones", c, k, len(live))
        for g, s in zip(got, sc):
            d = O.hnsw_distance(m, Q[qi], rows[id2row[g]])
            assert s == O.hnsw_score(d, m), (seed, "score", g, s, O.hnsw_score(d, m))
            checks += 1
        if c and n >= 300 and k == 10:
            dall = np.array([O.hnsw_distance(m, Q[qi], rows[id2row[g]]) for g in live], dtype=np.uint64)
            kth = np.partition(dall, min(9, len(dall) - 1))[min(9, len(dall) - 1)]
            recalls.append(np.mean([O.hnsw_distance(m, Q[qi], rows[id2row[g]]) <= kth for g in got]))
    cases += 1
    del idx
print(f"hnsw fuzz: {cases} cases, {checks} returned (id, score) pairs checked in {budget:.0f}s: ids live and unique, scores = reference "
      f"conversion of the exact callback distance, sorted, counts right; mean recall@10 vs the u64 order on the n >= 300 cases {np.mean(recalls) if recalls else float('nan'):.3f}")
