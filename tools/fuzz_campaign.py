"""Randomised parity hunt: random shapes / data kinds / k / metric / pipeline against the CPU oracle until the time
budget is spent.  Prints a one-line summary; the first mismatch is printed with its seed and aborts.
usage: python tools/fuzz_campaign.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
O.build()
t_end = time.time() + budget
cases = checks = n_multi = 0
paths = {}
case_seed = seed0
t_report = time.time() + 30
while time.time() < t_end:
    if time.time() > t_report:
        print(f"  ... {cases} cases, {checks} checks so far", flush=True)
        t_report = time.time() + 30
    case_seed += 1
    rng = np.random.default_rng(case_seed)
    dim = int(rng.choice([1, 2, 5, 8, 16, 24, 32, 48, 64, 100, 128, 200, 256, 384, 512, 768]))
    n = int(rng.choice([1, 3, 64, 65, 300, 2000, 9000, 30000, 120000]))
    if n * dim > 40_000_000:
        n = 40_000_000 // dim
    kind = rng.choice(["gauss", "unit", "grid", "dups", "scaled", "clustered"])
    if kind == "gauss":
        rows = rng.standard_normal((n, dim))
    elif kind == "unit":
        rows = rng.standard_normal((n, dim)); rows /= np.maximum(np.linalg.norm(rows, axis=1, keepdims=True), 1e-300)
    elif kind == "grid":
        rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float64)
    elif kind == "dups":
        base = rng.standard_normal((max(1, n // 50), dim)); rows = base[rng.integers(0, base.shape[0], size=n)]
    elif kind == "scaled":
        rows = rng.standard_normal((n, dim)) * np.exp2(rng.integers(-20, 21, size=(n, 1)).astype(np.float64))
    else:
        c = rng.standard_normal((8, dim)); rows = c[rng.integers(0, 8, size=n)] + 1e-4 * rng.standard_normal((n, dim))
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(11)) % np.uint64(2 ** 50)
    multi = None
    if rng.random() < 0.25:  # one handle over several parts (vl_flat_create_multi; the parts share the one card here)
        multi = (["replicas", "row_shards"][int(rng.integers(0, 2))], [0] * int(rng.integers(2, 4)))
        gpu = V.MultiFlatIndex(dim, multi[1], multi[0])
        cut = int(rng.integers(0, n + 1))  # two bulk pieces: the shards level out, insertion order stays global
        gpu.add_rows(ids[:cut], rows[:cut], validate=False); gpu.add_rows(ids[cut:], rows[cut:], validate=False)
        n_multi += 1
    else:
        gpu = V.FlatIndex(dim); gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    nq = int(rng.choice([1, 2, 5, 9, 33]))
    if n >= 9000 and dim <= 128 and rng.random() < 0.5:
        nq = int(rng.choice([64, 150, 300]))  # big batches on the MFMA filter, with hostile queries mixed in
    Q = rng.standard_normal((nq, dim))
    if rng.random() < 0.5:
        Q[0] = rows[rng.integers(0, n)]
    if nq >= 64:
        for j in rng.integers(0, nq, size=6):
            Q[j] = [rows[rng.integers(0, n)], np.zeros(dim), Q[j] * 1e30, Q[j] * 1e-30, -rows[rng.integers(0, n)]][int(rng.integers(0, 5))]
    m = int(rng.integers(0, 4))
    k = int(rng.choice([1, 10, 48, 60, 61, 100, 220, 221, 500]))
    mode = "batch" if nq >= 64 else rng.choice(["single", "batch", "bf16", "positions"])
    if multi is not None and mode in ("bf16", "positions"):
        mode = "single"
    if mode == "bf16":
        gpu.set_single_filter("bf16")
    if mode == "batch":
        if multi is None and rng.integers(0, 2):  # half of the batches come from device memory (vl_index_search_batch_dev)
            import torch
            bi, bs, bn = gpu.search_batch_device(torch.from_numpy(np.ascontiguousarray(Q)).to('cuda:0'), k, m)
        else:
            bi, bs, bn = gpu.search_batch(Q, k, m)
        got = [(bi[i, : bn[i]], bs[i, : bn[i]]) for i in range(nq)]
    elif mode == "positions":
        got = []
        for i in range(nq):
            pos, gi, gs = gpu.search_positions(Q[i], k, m)
            assert [int(ids[p]) for p in pos] == gi.tolist()
            got.append((gi, gs))
    else:
        got = [gpu.search_arrays(Q[i], k, m) for i in range(nq)]
        paths[V.last_path()] = paths.get(V.last_path(), 0) + 1
    for i in range(nq):
        ri, rs = ref.search(Q[i], k, m)
        if got[i][0].tolist() != ri.tolist() or got[i][1].tolist() != rs.tolist():
            print(f"MISMATCH seed {case_seed}: dim {dim} n {n} kind {kind} metric {m} k {k} mode {mode} multi {multi} query {i}", flush=True)
            print(" got ", got[i][0][:8].tolist(), got[i][1][:4].tolist()); print(" want", ri[:8].tolist(), rs[:4].tolist())
            sys.exit(1)
        checks += 1
    cases += 1
    del gpu, ref
print(f"fuzz campaign: {cases} cases, {checks} query checks in {budget:.0f}s from seed {seed0}: all bit-identical; {n_multi} cases on multi-part handles (replicas / row shards over 2-3 parts); single-query paths seen {paths}")
