"""Targeted fuzz of the dim-768 MFMA shape (K split over wave pairs): random dims 513-768, hostile rows and queries;
search_batch (bf16 MFMA filter + exact finalize) must equal the single-query pipeline (f32 scan + exact finalize) on
every row, and a sample of rows is checked against the CPU oracle.
usage: python tools/fuzz_splitk.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
O.build()
t_end = time.time() + budget
t_report = time.time() + 30
cases = checks = oracle_checks = mfma_passes = 0
while time.time() < t_end:
    if time.time() > t_report:
        print(f"  ... {cases} cases, {checks} rows compared", flush=True)
        t_report = time.time() + 30
    seed += 1
    rng = np.random.default_rng(seed)
    dim = int(rng.integers(513, 769))
    n = int(rng.choice([8192, 8193, 9000, 20011, 60000]))
    kind = rng.choice(["gauss", "unit", "grid", "dups", "scaled", "clustered"])
    if kind == "gauss":
        rows = rng.standard_normal((n, dim))
    elif kind == "unit":
        rows = rng.standard_normal((n, dim)); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    elif kind == "grid":
        rows = rng.integers(-3, 4, size=(n, dim)).astype(np.float64)
    elif kind == "dups":
        base = rng.standard_normal((max(1, n // 50), dim)); rows = base[rng.integers(0, base.shape[0], size=n)]
    elif kind == "scaled":
        rows = rng.standard_normal((n, dim)) * np.exp2(rng.integers(-20, 21, size=(n, 1)).astype(np.float64))
    else:
        c = rng.standard_normal((8, dim)); rows = c[rng.integers(0, 8, size=n)] + 1e-4 * rng.standard_normal((n, dim))
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(11)) % np.uint64(2 ** 50)
    gpu = V.FlatIndex(dim); gpu.add_rows(ids, rows, validate=False)
    nq = int(rng.choice([2, 7, 33, 128, 129, 300]))
    Q = rng.standard_normal((nq, dim))
    for j in rng.integers(0, nq, size=min(nq, 6)):
        Q[j] = [rows[rng.integers(0, n)], np.zeros(dim), Q[j] * 1e30, Q[j] * 1e-30, -rows[rng.integers(0, n)]][int(rng.integers(0, 5))]
    m = int(rng.choice([0, 1, 3]))
    k = int(rng.choice([1, 10, 48, 60]))
    gpu.profile_read(); gpu.profile_enable(True)
    if rng.integers(0, 2):  # half of the batches come from device memory (vl_index_search_batch_dev)
        import torch
        bi, bs, bn = gpu.search_batch_device(torch.from_numpy(np.ascontiguousarray(Q)).to('cuda:0'), k, m)
    else:
        bi, bs, bn = gpu.search_batch(Q, k, m)
    gpu.profile_enable(False)
    mfma_passes += 1 if gpu.profile_read()[0] < nq else 0   # fewer passes than queries: the batch filter served it
    for i in range(nq):
        si, ss = gpu.search_arrays(Q[i], k, m)
        if bi[i, : bn[i]].tolist() != si.tolist() or bs[i, : bn[i]].tolist() != ss.tolist():
            print(f"MISMATCH seed {seed}: dim {dim} n {n} kind {kind} metric {m} k {k} nq {nq} query {i}", flush=True)
            sys.exit(1)
        checks += 1
    ref = O.FlatOracle(dim, ids, rows)
    for i in rng.integers(0, nq, size=2):
        ri, rs = ref.search(Q[i], k, m)
        if bi[i, : bn[i]].tolist() != ri.tolist() or bs[i, : bn[i]].tolist() != rs.tolist():
            print(f"ORACLE MISMATCH seed {seed}: dim {dim} n {n} kind {kind} metric {m} k {k} nq {nq} query {i}", flush=True)
            sys.exit(1)
        oracle_checks += 1
    cases += 1
    del gpu, ref
print(f"split-K fuzz: {cases} cases (dims 513-768, n 8192-60000, 2-300 queries, cosine/Euclidean/dot), {checks} batch rows == single-query "
      f"rows, {oracle_checks} rows == CPU oracle, bit for bit; {mfma_passes} cases served by the MFMA filter; seed {seed}")
