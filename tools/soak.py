"""Soak: create / fill / search (every pipeline) / clone / mutate / destroy in a loop; device memory must return to
its starting level and every answer must stay self-consistent.  usage: python tools/soak.py [seconds]"""
import gc, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vectorlite_amd as V

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(1)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
t_end = time.time() + budget
rounds = searches = 0
low = free0
while time.time() < t_end:
    dim = int(rng.choice([64, 128, 384, 768]))
    n = int(rng.choice([5000, 20000, 120000]))
    rows = rng.standard_normal((n, dim)); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(n, dtype=np.uint64), rows, validate=False)
    Q = rows[rng.integers(0, n, 40)] + 0.01
    for m in range(4):
        s = idx.search_arrays(Q[0], 10, m); searches += 1
        assert len(s[0]) == 10 and all(s[1][i] >= s[1][i + 1] for i in range(9))
    b = idx.search_batch(Q, 10, 0); searches += 40
    for i in (0, 7, 39):
        one = idx.search_arrays(Q[i], 10, 0)
        assert b[0][i].tolist() == one[0].tolist() and b[1][i].tolist() == one[1].tolist()
    dq = torch.from_numpy(np.ascontiguousarray(Q)).to("cuda:0")  # the same batch from device memory
    bd = idx.search_batch_device(dq, 10, 0); searches += 40
    assert bd[0].tolist() == b[0].tolist() and bd[1].tolist() == b[1].tolist()
    del dq
    eb = idx.search_batch_embeddings(Q.astype(np.float32), 10, 0); searches += 40  # f32 embeddings: device-side post-processing, pooled scratch
    assert eb[2].tolist() == [10] * 40
    idx.set_single_filter("bf16"); idx.search_arrays(Q[1], 10, 0); idx.set_single_filter("f32")
    idx.search_arrays(Q[2], 100, 1); idx.search_arrays(Q[2], 700, 1)
    idx.set_coalescing(32, 100)
    th = [threading.Thread(target=lambda t=t: [idx.search_arrays(Q[(t * 5 + j) % 40], 10, 0) for j in range(5)]) for t in range(8)]
    [x.start() for x in th]; [x.join() for x in th]; searches += 40
    e = V.FlatIndex(dim); e.add_embeddings(np.arange(3000, dtype=np.uint64), rows[:3000].astype(np.float32)); del e
    c = idx.clone()
    for j in range(20):
        c.delete(int(j * 3))
    c.add(V.Vector(10 ** 9, rows[0]))
    assert len(c) == n - 20 + 1 and len(idx) == n
    if n <= 20000:
        h = V.HNSWIndex(dim, V.SimilarityMetric.Cosine)
        h.add_rows(np.arange(n, dtype=np.uint64), rows)
        r = h.search(Q[3], 10, 0); searches += 1
        assert len(r) == 10
        # concurrent walks, each borrowing its own scratch from the pool, while a batch runs too
        hw = [h.search_arrays(Q[j], 10, 0, ef=48) for j in range(8)]
        ht = [threading.Thread(target=lambda t=t: [h.search_arrays(Q[(t + j) % 8], 10, 0, ef=48) for j in range(6)]) for t in range(8)]
        [x.start() for x in ht]; hb = h.search_batch(Q[:8], 10, 0, ef=48); [x.join() for x in ht]; searches += 56
        assert all(hb[0][j].tolist() == hw[j][0].tolist() for j in range(8))
        g = h.graph(); assert g["n"] == n and int(g["cnt0"].max()) <= g["m0"]
        h2 = h.clone(); h2.delete(5); del h2, h
    # row-sharded search through the library's own RCCL communicator (world 1): create / sync / search / destroy
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex
    if not os.environ.get("SOAK_NO_COMM"):
        comm = Comm(Comm.unique_id(), 1, 0, 0)
        sh = ShardedFlatIndex(idx, comm=comm)
        si, ss, sn = sh.search_batch(Q[:16], 10, 0); searches += 16
        assert si[0].tolist() == b[0][0].tolist() and ss[0].tolist() == b[1][0].tolist()
        comm.close(); del sh, comm
    # one handle over three parts (vl_flat_create_multi; the parts share this card): worker threads, exchange slots and
    # mergers are created and torn down every round; concurrent callers make it grow to several slots
    for mode in ("replicas", "row_shards"):
        mh = V.MultiFlatIndex(dim, [0, 0, 0], mode)
        nm = min(n, 6000)
        mh.add_rows(np.arange(nm, dtype=np.uint64), rows[:nm], validate=False)
        mb = mh.search_batch(Q, 10, 0); searches += 40
        mt = [threading.Thread(target=lambda t=t: [mh.search_arrays(Q[(t * 3 + j) % 40], 10, 0) for j in range(4)]) for t in range(6)]
        [x.start() for x in mt]; mh.search_batch(Q[:12], 10, 1); [x.join() for x in mt]; searches += 36
        one = mh.search_arrays(Q[4], 10, 0)
        assert mb[0][4].tolist() == one[0].tolist() and mb[1][4].tolist() == one[1].tolist()
        mc = mh.clone(); mc.delete(5); assert len(mc) == nm - 1 and len(mh) == nm
        del mh, mc
    del idx, c
    gc.collect()
    torch.cuda.synchronize()
    low = min(low, torch.cuda.mem_get_info()[0])
    rounds += 1
gc.collect(); torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"{rounds} rounds, {searches} searches in {budget:.0f}s; device memory free: start {free0 / 2**30:.2f} GiB, "
      f"lowest {low / 2**30:.2f} GiB, end {free1 / 2**30:.2f} GiB (delta {(free0 - free1) / 2**20:.1f} MiB)")
assert free0 - free1 < 256 * 2**20, "device memory did not come back"
