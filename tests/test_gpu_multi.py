"""ONE flat-index handle over several GPUs in one process (vl_flat_create_multi; the reference is one process with
its collections behind Arc<RwLock<..>>, src/client.rs:243-247,398).  The test box has one card, so the parts are
made on the same device -- device_ids = {0}, {0, 0}, {0, 0, 0} -- which runs the same host code (per-part worker
threads, replica dealing, per-shard search + device merge) that eight cards would.  Every answer is compared with
ONE oracle holding all rows in insertion order: ids and f64 scores ==."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, dim):
    x = rng.standard_normal((n, dim))
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _same(got, want):
    gi, gs = got
    wi, ws = want
    assert gi.tolist() == wi.tolist()
    assert gs.tolist() == ws.tolist()


@pytest.mark.parametrize("mode", ["replicas", "row_shards"])
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])  # eight parts: the shape of one 8-GPU node
def test_trait_surface_matches_one_oracle(mode, devices):
    import vectorlite_amd as V
    from oracle import oracle as O
    rng = np.random.default_rng(7 + len(devices))
    dim, n = 48, 3000
    rows = _unit(rng, n, dim)
    rows[100] = rows[5]          # equal rows: ties broken by insertion order, across shards too
    rows[2000] = rows[5]
    rows[2999] = rows[5]
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(17)) % np.uint64(2 ** 48)
    m = V.MultiFlatIndex(dim, devices, mode)
    ref = O.FlatOracle(dim)
    # bulk load in uneven pieces (the shards level out), then single adds
    for a, b in ((0, 1000), (1000, 1001), (1001, 2500)):
        m.add_rows(ids[a:b], rows[a:b], validate=False)
        ref.extend(ids[a:b], rows[a:b])
    for i in range(2500, n):
        m.add(V.Vector(int(ids[i]), rows[i]))
        ref.add(int(ids[i]), rows[i])
    assert len(m) == n == len(ref)
    parts = m.parts()
    assert parts["n_parts"] == len(devices)
    if mode == "row_shards":
        assert sum(parts["rows"]) == n and max(parts["rows"]) - min(parts["rows"]) <= 1
    else:
        assert parts["rows"] == [n] * len(devices)
    Q = _unit(rng, 24, dim)
    Q[3] = rows[5]               # the query that hits the four equal rows
    for metric in range(4):
        for k in (1, 10, 60):
            for qi in (0, 3, 7):
                _same(m.search_arrays(Q[qi], k, metric), ref.search(Q[qi], k, metric))
        bi, bs, bn = m.search_batch(Q, 10, metric)
        for qi in range(len(Q)):
            wi, ws = ref.search(Q[qi], 10, metric)
            assert bn[qi] == len(wi) and bi[qi, :bn[qi]].tolist() == wi.tolist() and bs[qi, :bn[qi]].tolist() == ws.tolist()
    # k larger than any one shard, and larger than the index
    for k in (n // len(devices) + 5, n + 10):
        _same(m.search_arrays(Q[1], k, 0), ref.search(Q[1], k, 0))
    # duplicate id is refused wherever the first copy lives; dimension mismatch; delete of present and absent ids
    with pytest.raises(V.IndexOpError, match=f"Vector ID {int(ids[2000])} already exists"):
        m.add(V.Vector(int(ids[2000]), rows[1]))
    with pytest.raises(V.IndexOpError, match="Vector dimension mismatch"):
        m.add(V.Vector(123456789, rows[1][:5]))
    with pytest.raises(V.DimensionMismatch) as e:
        m.search_arrays(Q[0][:7], 3, 0)
    assert (e.value.expected, e.value.actual) == (dim, 7)
    for victim in (int(ids[5]), int(ids[1700]), int(ids[2999]), 999999999999):
        m.delete(victim)
        ref.delete(victim)
    assert len(m) == len(ref) == n - 3
    for metric in (0, 1):
        _same(m.search_arrays(Q[3], 10, metric), ref.search(Q[3], 10, metric))
    # point lookups and export follow insertion order of the WHOLE index
    assert m.max_id() == int(max(int(i) for i in ids if int(i) not in (int(ids[5]), int(ids[1700]), int(ids[2999]))))
    got = m.get_vector(int(ids[100]))
    assert got is not None and np.array_equal(np.asarray(got.values), rows[100])
    assert m.get_vector(int(ids[5])) is None
    e_ids, e_vals = m.export()
    keep = [i for i in range(n) if i not in (5, 1700, 2999)]
    assert e_ids.tolist() == ids[keep].tolist() and np.array_equal(e_vals, rows[keep])
    # a validated bulk add stops at the first duplicate, rows before it are kept (n sequential adds)
    new_ids = np.array([7000001, 7000002, int(ids[10]), 7000003], dtype=np.uint64)
    new_rows = _unit(rng, 4, dim)
    with pytest.raises(V.IndexOpError, match=f"Vector ID {int(ids[10])} already exists"):
        m.add_rows(new_ids, new_rows, validate=True)
    ref.add(7000001, new_rows[0])
    ref.add(7000002, new_rows[1])
    assert len(m) == len(ref)
    _same(m.search_arrays(new_rows[1], 5, 0), ref.search(new_rows[1], 5, 0))
    # Clone answers like the original and is independent of it
    c = m.clone()
    m.delete(7000002)
    _same(c.search_arrays(new_rows[1], 5, 0), ref.search(new_rows[1], 5, 0))
    ref.delete(7000002)
    _same(m.search_arrays(new_rows[1], 5, 0), ref.search(new_rows[1], 5, 0))


def test_empty_and_tiny_sharded_indexes():
    import vectorlite_amd as V
    from oracle import oracle as O
    m = V.MultiFlatIndex(4, [0, 0, 0], "row_shards")
    assert len(m) == 0 and m.max_id() is None
    i, s = m.search_arrays(np.ones(9), 3, 0)   # an empty index accepts any query length (src/index/flat.rs:99)
    assert len(i) == 0
    ref = O.FlatOracle(4)
    rows = np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0]])
    for j in range(2):                         # fewer rows than shards: one shard stays empty
        m.add(V.Vector(10 + j, rows[j]))
        ref.add(10 + j, rows[j])
    for metric in range(4):
        _same(m.search_arrays(np.array([0.6, 0.8, 0, 0]), 5, metric), ref.search(np.array([0.6, 0.8, 0, 0]), 5, metric))
    with pytest.raises(V.DimensionMismatch):
        m.search_arrays(np.ones(3), 1, 0)
    # a NaN score with two or more rows in the index is the reference's panic, whichever shard holds the row
    # (each shard here holds ONE row, whose own 1-element sort would never compare)
    with pytest.raises(V.NaNScore):
        m.search_arrays(np.array([np.nan, 0, 0, 0]), 1, 3)
    one = V.MultiFlatIndex(4, [0, 0], "row_shards")
    one.add(V.Vector(1, rows[0]))
    i, s = one.search_arrays(np.array([np.nan, 0, 0, 0]), 1, 3)   # a 1-row index returns the NaN (no comparison happens)
    assert i.tolist() == [1] and np.isnan(s[0])


def test_concurrent_searches_are_dealt_over_the_replicas():
    import vectorlite_amd as V
    from oracle import oracle as O
    rng = np.random.default_rng(99)
    dim, n = 64, 40000
    rows = _unit(rng, n, dim)
    ids = np.arange(n, dtype=np.uint64) + np.uint64(5)
    m = V.MultiFlatIndex(dim, [0, 0, 0], "replicas")
    m.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = _unit(rng, 48, dim)
    want = [ref.search(q, 10, 0) for q in Q]
    errors = []

    def worker(t):
        try:
            for r in range(3):
                for qi in range(t, len(Q), 6):
                    _same(m.search_arrays(Q[qi], 10, 0), want[qi])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errors == []
    p = m.parts()
    assert sum(p["searches"]) == 3 * len(Q) and min(p["searches"]) > 0   # every replica took part
    # a batch is cut into one run per replica; the rows come back in the caller's order
    bi, bs, bn = m.search_batch(Q, 10, 0)
    for qi in range(len(Q)):
        assert bi[qi].tolist() == want[qi][0].tolist() and bs[qi].tolist() == want[qi][1].tolist()


def test_sharded_batch_on_the_mfma_path_matches_one_index():
    """Shards large enough for the bf16 MFMA filter (>= 8192 rows each), config 3's kind of batch: the merged answer is
    the single index's, bit for bit, and device-side queries / f32 embeddings go through the same handle."""
    import torch
    import vectorlite_amd as V
    rng = np.random.default_rng(3)
    dim, n, nq = 128, 60000, 200
    rows = _unit(rng, n, dim)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(1)
    one = V.FlatIndex(dim)
    one.add_rows(ids, rows, validate=False)
    m = V.MultiFlatIndex(dim, [0, 0, 0], "row_shards")
    m.add_rows(ids, rows, validate=False)
    Q = _unit(rng, nq, dim)
    for metric in (0, 1, 3):
        wi, ws, wn = one.search_batch(Q, 10, metric)
        gi, gs, gn = m.search_batch(Q, 10, metric)
        assert gn.tolist() == wn.tolist() and gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist()
    dQ = torch.from_numpy(Q).to("cuda:0")
    di, ds, dn = m.search_batch_device(dQ, 10, 0)
    wi, ws, wn = one.search_batch(Q, 10, 0)
    assert di.tolist() == wi.tolist() and ds.tolist() == ws.tolist()


@pytest.mark.parametrize("mode", ["replicas", "row_shards"])
def test_stateful_random_stream_on_a_three_part_handle(mode):
    """A random stream of adds, unvalidated bulk adds that DUPLICATE live ids (FlatIndex::new keeps them,
    src/index/flat.rs:68-73), deletes (present -- all copies go --, absent), refused adds and clones on a handle of three
    parts; after every step the whole surface is compared with ONE oracle that received the same stream.  Row shards:
    rows of one id end up on different GPUs, deletes close gaps on each, and the answer must still rank ties by the
    insertion order of the whole index (src/index/flat.rs:94,116)."""
    import vectorlite_amd as V
    from oracle import oracle as O
    rng = np.random.default_rng(20261004 + (mode == "row_shards"))
    dim, n0 = 24, 700
    base = _unit(rng, 40, dim)                      # few distinct directions: equal rows, ties everywhere
    rows = base[rng.integers(0, 40, size=n0)]
    ids = (np.arange(n0, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(5)) % np.uint64(2 ** 40)
    m = V.MultiFlatIndex(dim, [0, 0, 0], mode)
    m.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    live = ids.tolist()
    next_id = 10 ** 12
    for step in range(60):
        op = int(rng.integers(0, 7))
        if op == 0:
            v = base[int(rng.integers(0, 40))] * float(rng.choice([1.0, 2.0]))
            m.add(V.Vector(next_id, v)); ref.add(next_id, v); live.append(next_id); next_id += 1
        elif op == 1:  # bulk add, validated
            c = int(rng.integers(1, 40))
            vs = base[rng.integers(0, 40, size=c)]
            new = np.arange(next_id, next_id + c, dtype=np.uint64)
            m.add_rows(new, vs)
            for i in range(c):
                ref.add(int(new[i]), vs[i])
            live += new.tolist(); next_id += c
        elif op == 2 and live:  # unvalidated bulk add that repeats live ids: duplicates are kept
            c = int(rng.integers(1, 12))
            dup = np.array([live[int(rng.integers(len(live)))] for _ in range(c)], dtype=np.uint64)
            vs = _unit(rng, c, dim)
            m.add_rows(dup, vs, validate=False)
            ref.extend(dup, vs)
            live += dup.tolist()
        elif op == 3 and live:  # delete: every copy of the id goes, wherever it is stored
            victim = live[int(rng.integers(len(live)))]
            m.delete(victim); ref.delete(victim)
            live = [x for x in live if x != victim]
        elif op == 4:
            m.delete(3); ref.delete(3)              # absent: Ok(())
        elif op == 5 and live:
            with pytest.raises(V.IndexOpError, match="already exists"):
                m.add(V.Vector(live[0], base[0]))
        else:
            m = m.clone()
        assert len(m) == len(ref) == len(live), (step, op)
        if not live:
            continue
        metric = int(rng.integers(0, 4))
        k = int(rng.choice([1, 7, 64, 100]))
        q = base[int(rng.integers(0, 40))] + (0.0 if rng.random() < 0.5 else 1e-3) * _unit(rng, 1, dim)[0]
        _same(m.search_arrays(q, k, metric), ref.search(q, k, metric))
        probe = live[int(rng.integers(len(live)))]
        assert np.array_equal(np.asarray(m.get_vector(probe).values), np.asarray(ref.get_vector(probe)))
    e_ids, e_vals = m.export()
    assert e_ids.tolist() == live
    Q = base[:9] + 1e-3 * _unit(rng, 9, dim)
    for metric in range(4):
        bi, bs, bn = m.search_batch(Q, 10, metric)
        for j in range(9):
            wi, ws = ref.search(Q[j], 10, metric)
            assert bi[j, : bn[j]].tolist() == wi.tolist() and bs[j, : bn[j]].tolist() == ws.tolist(), (metric, j)


@pytest.mark.parametrize("mode", ["row_shards", "replicas"])
def test_concurrent_callers_of_single_and_batch_searches_on_a_three_part_handle(mode):
    """Many readers at once (RwLock::read, src/client.rs:398): eight threads mix single searches and batches on one handle.
    A sharded search borrows its own exchange slot (record block + merger) and, when the part workers are taken by another
    call, walks its parts on the calling thread -- nobody queues behind one set of buffers, and every answer is the oracle's."""
    import vectorlite_amd as V
    from oracle import oracle as O
    rng = np.random.default_rng(4242 + (mode == "replicas"))
    dim, n = 48, 30000
    rows = _unit(rng, n, dim)
    rows[11000] = rows[3]
    rows[29999] = rows[3]
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(9)) % np.uint64(2 ** 44)
    m = V.MultiFlatIndex(dim, [0, 0, 0], mode)
    m.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = _unit(rng, 40, dim)
    Q[5] = rows[3]
    want = {(qi, metric): ref.search(Q[qi], 10, metric) for qi in range(len(Q)) for metric in (0, 1)}
    errors = []

    def worker(t):
        try:
            for r in range(4):
                metric = (t + r) % 2
                if (t + r) % 3 == 0:  # a batch of 10
                    q0 = (t * 5 + r * 7) % 30
                    bi, bs, bn = m.search_batch(Q[q0: q0 + 10], 10, metric)
                    for j in range(10):
                        wi, ws = want[(q0 + j, metric)]
                        assert bi[j, : bn[j]].tolist() == wi.tolist() and bs[j, : bn[j]].tolist() == ws.tolist(), (t, r, j)
                else:
                    for qi in range(t, len(Q), 8):
                        _same(m.search_arrays(Q[qi], 10, metric), want[(qi, metric)])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    for coalesce in (False, True):   # second round: every part coalesces its concurrent single searches into shared slab passes
        m.set_coalescing(32, 100) if coalesce else m.set_coalescing(0)   # (on by default since round 4: the first round turns it off)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert errors == [], (coalesce, errors)
    assert len(m) == n


@pytest.mark.parametrize("mode", ["replicas", "row_shards"])
def test_a_part_that_fails_a_bulk_add_leaves_every_part_as_it_was(mode, monkeypatch):
    """Advisor, round 3: a fan-out that failed on ONE part used to leave the replicas diverged for good (and a sharded index
    holding a non-prefix subset of the batch).  Now every part makes room first and, if a part fails anyway, every part
    forgets the rows of that call: the handle answers exactly as before the call, and the retry goes through."""
    import vectorlite_amd as V
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    dim, n0, n1 = 32, 900, 600
    rows = _unit(rng, n0 + n1, dim)
    ids = np.arange(n0 + n1, dtype=np.uint64) + np.uint64(1000)
    m = V.MultiFlatIndex(dim, [0, 0, 0], mode)
    ref = O.FlatOracle(dim)
    m.add_rows(ids[:n0], rows[:n0], validate=True)
    ref.extend(ids[:n0], rows[:n0])
    Q = _unit(rng, 12, dim)
    before = [m.search_arrays(Q[i], 10, i % 4) for i in range(12)]
    for bad_part in (0, 1, 2):
        monkeypatch.setenv("VL_MULTI_INJECT_ADD_FAIL", str(bad_part))
        with pytest.raises(V.VectorLiteError, match="injected add failure"):
            m.add_rows(ids[n0:], rows[n0:], validate=True)
        monkeypatch.delenv("VL_MULTI_INJECT_ADD_FAIL")
        assert len(m) == n0
        parts = m.parts()
        assert (parts["rows"] == [n0] * 3) if mode == "replicas" else (sum(parts["rows"]) == n0)
        for rep in range(3):                  # replicas: every replica is asked in turn (round robin) -- all say the same
            for i in range(12):
                _same(m.search_arrays(Q[i], 10, i % 4), before[i])
        assert m.get_vector(int(ids[n0 + 5])) is None
    # the retry is a plain add: no "already exists" from parts that had kept their share
    m.add_rows(ids[n0:], rows[n0:], validate=True)
    ref.extend(ids[n0:], rows[n0:])
    assert len(m) == n0 + n1
    for rep in range(3):
        for i in range(12):
            _same(m.search_arrays(Q[i], 10, i % 4), ref.search(Q[i], 10, i % 4))
    bi, bs, bn = m.search_batch(Q, 7, 1)
    for i in range(12):
        wi, ws = ref.search(Q[i], 7, 1)
        assert bi[i, :bn[i]].tolist() == wi.tolist() and bs[i, :bn[i]].tolist() == ws.tolist()


def test_a_delete_that_fails_on_one_part_retires_the_handle(monkeypatch):
    """A compaction that died half way cannot be rolled back: the parts disagree, so the handle refuses every later call
    instead of serving answers that depend on which replica is asked."""
    import vectorlite_amd as V
    rng = np.random.default_rng(78)
    dim, n = 16, 300
    rows = _unit(rng, n, dim)
    m = V.MultiFlatIndex(dim, [0, 0], "replicas")
    m.add_rows(np.arange(n, dtype=np.uint64), rows, validate=False)
    monkeypatch.setenv("VL_MULTI_INJECT_DELETE_FAIL", "1")
    with pytest.raises(V.VectorLiteError, match="injected delete failure"):
        m.delete(7)
    monkeypatch.delenv("VL_MULTI_INJECT_DELETE_FAIL")
    for call in (lambda: m.search_arrays(rows[0], 3, 0), lambda: m.search_batch(rows[:4], 3, 0),
                 lambda: m.add(V.Vector(5000, rows[1])), lambda: m.delete(8)):
        with pytest.raises(V.VectorLiteError, match="no longer answers"):
            call()
