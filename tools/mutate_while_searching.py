#!/usr/bin/env python3
"""Writers and readers at once on one handle (the reference guards this with RwLock in the caller, src/client.rs:333,383,398;
the library has its own reader/writer lock): four searcher threads, one adder, one deleter.  Every answer must be well-formed
(sorted, unique live-or-just-deleted ids, right count bounds), nothing may crash or hang, and the final state must equal an
oracle that applied the same mutations.  usage: python tools/mutate_while_searching.py [flat|hnsw|replicas|row_shards]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "flat"
rng = np.random.default_rng(17)
dim, n0 = 48, 20000
rows = rng.standard_normal((n0 + 600, dim)); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
ids0 = np.arange(n0, dtype=np.uint64)
if kind == "flat":
    idx = V.FlatIndex(dim)
elif kind == "hnsw":
    idx = V.HNSWIndex(dim, 1)
else:
    idx = V.MultiFlatIndex(dim, [0, 0, 0], kind)
idx.add_rows(ids0, rows[:n0]) if kind == "hnsw" else idx.add_rows(ids0, rows[:n0], validate=False)
universe = set(range(n0 + 600))
errors, stop = [], False
metric = 1

def searcher(t):
    r = np.random.default_rng(100 + t)
    n_s = 0
    try:
        while not stop:
            q = rows[int(r.integers(0, n0))] + 0.01 * r.standard_normal(dim)
            if n_s % 5 == 0:
                bi, bs, bn = idx.search_batch(np.stack([q, -q, q * 2]), 10, metric)
                outs = [(bi[j, : bn[j]], bs[j, : bn[j]]) for j in range(3)]
            else:
                outs = [idx.search_arrays(q, 10, metric)]
            for gi, gs in outs:
                assert len(gi) <= 10 and len(set(gi.tolist())) == len(gi), gi
                assert all(int(x) in universe for x in gi)
                assert all(gs[j - 1] >= gs[j] for j in range(1, len(gs))), gs
            n_s += 1
    except Exception as e:  # noqa: BLE001
        errors.append(("searcher", t, repr(e)))
    counts[t] = n_s

counts = {}
added, deleted = [], []
def adder():
    try:
        for i in range(600):
            idx.add(V.Vector(n0 + i, rows[n0 + i])); added.append(n0 + i)
    except Exception as e:  # noqa: BLE001
        errors.append(("adder", repr(e)))
def deleter():
    r = np.random.default_rng(9)
    try:
        for v in r.choice(n0, size=300, replace=False):
            idx.delete(int(v)); deleted.append(int(v))
    except Exception as e:  # noqa: BLE001
        errors.append(("deleter", repr(e)))

def cloner():   # Clone (src/persistence.rs:118 clones the index to save it), export and point lookups beside the writers
    r = np.random.default_rng(77)
    n_c = 0
    try:
        while not stop:
            c = idx.clone()
            q = rows[int(r.integers(0, n0))]
            gi, gs = c.search_arrays(q, 5, metric)
            assert len(gi) <= 5 and all(int(x) in universe for x in gi)
            e_ids, e_vals = c.export()
            assert len(e_ids) == len(c) and e_vals.shape[0] == len(e_ids)
            v = idx.get_vector(int(r.integers(0, n0)))       # may have been deleted meanwhile: None is fine
            assert v is None or len(v.values) == dim
            del c
            n_c += 1
    except Exception as e:  # noqa: BLE001
        errors.append(("cloner", repr(e)))
    counts["clones"] = n_c
tc = threading.Thread(target=cloner)
th = [threading.Thread(target=searcher, args=(t,)) for t in range(4)] + [tc]
wa, wd = threading.Thread(target=adder), threading.Thread(target=deleter)
t0 = time.time()
[x.start() for x in th]; wa.start(); wd.start()
wa.join(timeout=300); wd.join(timeout=300)
stop = True
[x.join(timeout=60) for x in th]
hung = [x.is_alive() for x in th + [wa, wd]]
print(f"{kind}: {sum(v for k2, v in counts.items() if k2 != 'clones')} searches and {counts.get('clones', 0)} clone + export rounds beside {len(added)} adds and {len(deleted)} deletes in {time.time() - t0:.1f}s; errors {errors[:3]}; hung {sum(hung)}")
assert not errors and not any(hung)
assert len(idx) == n0 + 600 - 300
if kind != "hnsw":   # final state against the oracle (flat semantics: bit for bit)
    keep = np.array([i for i in range(n0) if i not in set(deleted)], dtype=np.uint64)
    ref = O.FlatOracle(dim, keep, rows[keep.astype(np.int64)])
    # deletes and adds interleaved: the storage order is (surviving old rows in order) then (added rows in order) only if every
    # add came after ... no: adds append, deletes close gaps -- relative order of survivors is insertion order either way
    for i in added:
        ref.add(i, rows[i])
    for j in range(12):
        q = rows[int(rng.integers(0, n0 + 600))]
        wi, ws = ref.search(q, 10, metric)
        gi, gs = idx.search_arrays(q, 10, metric)
        assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist(), j
    print("final state == oracle")
print("ok")
