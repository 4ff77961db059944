#!/usr/bin/env python3
"""Ids are arbitrary u64 in the reference (src/lib.rs:164-174): 0, 2^63, 2^64 - 1 and friends must behave like any other id on
every handle kind -- add, duplicate refusal, search, get_vector, delete, max_id, export."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
from oracle import oracle as O

rng = np.random.default_rng(8)
dim = 12
edge = [0, 1, 2**32 - 1, 2**32, 2**63 - 1, 2**63, 2**64 - 2, 2**64 - 1]
rows = rng.standard_normal((len(edge) + 40, dim))
ids = np.array(edge + list(range(1000, 1040)), dtype=np.uint64)
for kind in ("flat", "hnsw", "replicas", "row_shards"):
    if kind == "flat":
        h = V.FlatIndex(dim)
    elif kind == "hnsw":
        h = V.HNSWIndex(dim, 1)
    else:
        h = V.MultiFlatIndex(dim, [0, 0, 0], kind)
    for i, r in zip(ids, rows):
        h.add(V.Vector(int(i), r, f"t{int(i)}"))
    assert len(h) == len(ids)
    for e, r in zip(edge, rows):
        got = h.get_vector(e)
        assert got is not None and np.array_equal(np.asarray(got.values), r), (kind, e)
        res = h.search(r, 1, 1)
        assert res[0].id == e, (kind, e, res[0].id)
        try:
            h.add(V.Vector(e, r))
            raise SystemExit(f"{kind}: duplicate id {e} accepted")
        except V.IndexOpError as ex:
            assert "already exists" in str(ex)
    if kind != "hnsw":
        ref = O.FlatOracle(dim, ids, rows)
        for m in range(4):
            wi, ws = ref.search(rows[3] * 0.9, 8, m)
            gi, gs = h.search_arrays(rows[3] * 0.9, 8, m)
            assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist(), (kind, m)
        e_ids, _ = h.export()
        assert e_ids.tolist() == ids.tolist()
    assert h.max_id() == 2**64 - 1, h.max_id()
    h.delete(2**64 - 1); h.delete(0)
    assert len(h) == len(ids) - 2 and h.get_vector(2**64 - 1) is None and h.get_vector(0) is None
    assert h.max_id() == 2**64 - 2
    after = h.search(rows[7], 1, 1)   # HNSW: the walk's one candidate is the tombstoned node itself -> dropped -> empty, like src/index/hnsw.rs:475
    assert (kind == "hnsw" and after == []) or after[0].id != 2**64 - 1
    print(kind, "ok")
print("id edge probe ok")
