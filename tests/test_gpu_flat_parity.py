"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden fixtures.  Bar: ids bit-exact, scores bit-exact (==), far inside north_star's 1e-5.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M = {"cosine": 0, "euclidean": 1, "manhattan": 2, "dotproduct": 3}


@pytest.fixture(scope="module")
def V():
    import vectorlite_amd as V
    n_dev, _ = V.runtime_info()
    assert n_dev > 0, "GPU tests need a HIP device"
    return V


@pytest.fixture(scope="module")
def O():
    from oracle import oracle as O
    O.build()
    return O


def unit_rows(rng, n, dim):
    """i.i.d. N(0,1), L2-normalised in f64 (mirrors src/embeddings.rs:173-179)."""
    x = rng.standard_normal((n, dim))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def permuted_ids(n):
    return (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(2 ** 40)


def assert_same(V, gpu, ref, ctx):
    gi, gs = gpu
    ri, rs = ref
    assert gi.tolist() == ri.tolist(), ("ids", ctx, V.last_path())
    # bit-exact scores, NaN-free; +-0.0 compared by value like partial_cmp
    assert gs.tolist() == rs.tolist(), ("scores", ctx, V.last_path())


# ---------------------------------------------------------------------------------------------
# golden fixtures (the reference's own tests)
# ---------------------------------------------------------------------------------------------
def test_reference_flat_kats(V, kats):
    seen = {}
    for kat in kats["flat_kats"]:
        data = [V.Vector(id=i, values=r, text="test") for i, r in zip(kat["ids"], kat["rows"])]
        idx = V.FlatIndex(kat["dim"], data)
        res = idx.search(kat["query"], kat["k"], M[kat["metric"]])
        seen[kat["src"]] = res
        assert len(res) == kat["len"], kat["src"]
        assert res[0].id == kat["first_id"], kat["src"]
        assert res[0].text == "test"
        if "first_score" in kat:
            assert abs(res[0].score - kat["first_score"]) <= kat["tol"], kat["src"]
        if "first_score_gt" in kat:
            assert res[0].score > kat["first_score_gt"]
        if kat.get("sorted_desc"):
            assert all(res[i - 1].score >= res[i].score for i in range(1, len(res)))
    for pair in kats["flat_pairs_differ"]:
        assert seen[pair["a"]][0].score != seen[pair["b"]][0].score


def test_reference_metric_kats_via_single_row_index(V, kats):
    """calculate(a, b) is reachable as the score of a 1-row index holding `a` queried with `b`."""
    for kat in kats["metric_kats"]:
        idx = V.FlatIndex(len(kat["a"]), [V.Vector(id=7, values=kat["a"])])
        res = idx.search(kat["b"], 1, M[kat["metric"]])
        assert res[0].id == 7
        assert abs(res[0].score - kat["expect"]) <= kat["tol"], kat["src"]


def test_survey_9_6_values_bit_exact(V):
    idx = V.FlatIndex(3, [V.Vector(1, [1, 0, 0]), V.Vector(2, [0, 1, 0]), V.Vector(3, [0, 0, 1])])
    ids, sc = idx.search_arrays([1, 0, 0], 2, 0)
    assert ids.tolist() == [1, 2] and sc.tolist() == [1.0, 0.0]
    ids, sc = idx.search_arrays([1.1, 0.1, 0.1], 2, 0)
    assert ids.tolist() == [1, 2]
    assert sc[0] == pytest.approx(0.99183659813417546, abs=1e-15)
    assert sc[1] == pytest.approx(0.090166963466743216, abs=1e-15)
    idx = V.FlatIndex(2, [V.Vector(1, [1, 2]), V.Vector(2, [2, 1])])
    assert idx.search_arrays([1, 2], 1, 0)[1][0] == 0.99999999999999978
    assert idx.search_arrays([1, 2], 1, 3)[1][0] == 5.0


# ---------------------------------------------------------------------------------------------
# seeded random parity against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [1, 3, 4, 5, 33, 128, 384, 768, 100])
def test_random_parity_all_metrics(V, O, dim):
    rng = np.random.default_rng(1234 + dim)
    for n in (1, 2, 63, 64, 65, 257, 1000):
        rows = unit_rows(rng, n, dim)
        ids = permuted_ids(n)
        gpu = V.FlatIndex(dim)
        gpu.add_rows(ids, rows, validate=False)
        ref = O.FlatOracle(dim, ids, rows)
        for qi in range(3):
            q = unit_rows(rng, 1, dim)[0]
            for name, m in M.items():
                for k in (1, 10, 32, 48, 49):
                    assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (dim, n, name, k))


@pytest.mark.parametrize("dim", [1000, 1024, 1536, 2048, 3072, 4096, 5000])
def test_large_dims_parity(V, O, dim):
    """Embedding sizes beyond the benchmark's 384 / 768 (specialised and generic scan shapes, the
    finalize kernel's chunked f64 rescoring): singles, an 8-query batch, and the forced exact path."""
    rng = np.random.default_rng(dim)
    n = 1500
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, 9, dim)
    for name, m in M.items():
        assert_same(V, gpu.search_arrays(Q[0], 10, m), ref.search(Q[0], 10, m), (dim, name))
        assert V.last_path() == V.PATH_FAST
    bi, bs, bn = gpu.search_batch(Q, 10, M["cosine"])
    for j in range(9):
        want = ref.search(Q[j], 10, M["cosine"])
        assert bi[j, : bn[j]].tolist() == want[0].tolist() and bs[j, : bn[j]].tolist() == want[1].tolist()
    gpu.force_path(V.PATH_EXACT_SELECT)
    assert_same(V, gpu.search_arrays(Q[1], 10, M["euclidean"]), ref.search(Q[1], 10, M["euclidean"]), (dim, "exact"))


def test_fast_path_is_the_one_running(V, O):
    rng = np.random.default_rng(99)
    n, dim = 20000, 384
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    for qi in range(8):
        q = unit_rows(rng, 1, dim)[0]
        for name, m in M.items():
            assert_same(V, gpu.search_arrays(q, 10, m), ref.search(q, 10, m), (name, qi))
            assert V.last_path() == V.PATH_FAST, name


def test_unnormalised_and_scaled_data(V, O):
    rng = np.random.default_rng(5)
    dim, n = 96, 3000
    for scale in (1e-6, 1.0, 37.5, 1e6):
        rows = rng.standard_normal((n, dim)) * scale * rng.uniform(0.1, 10.0, size=(n, 1))
        ids = permuted_ids(n)
        gpu = V.FlatIndex(dim)
        gpu.add_rows(ids, rows, validate=False)
        ref = O.FlatOracle(dim, ids, rows)
        q = rng.standard_normal(dim) * scale
        for name, m in M.items():
            assert_same(V, gpu.search_arrays(q, 10, m), ref.search(q, 10, m), (scale, name))


# ---------------------------------------------------------------------------------------------
# ties, duplicates, zero rows: the stable (insertion-order) tie-break
# ---------------------------------------------------------------------------------------------
def test_all_rows_identical_mock_embedding_case(V, O):
    """MockEmbeddingFunction stores vec![1.0; dim] for every text (src/client.rs:504-523)."""
    dim, n = 384, 500
    rows = np.ones((n, dim))
    ids = np.arange(n, dtype=np.uint64)[::-1].copy()  # ids descending: position, not id, breaks ties
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows)
    ref = O.FlatOracle(dim, ids, rows)
    for name, m in M.items():
        assert_same(V, gpu.search_arrays(np.ones(dim), 10, m), ref.search(np.ones(dim), 10, m), name)
        assert V.last_path() in (V.PATH_EXACT_SELECT, V.PATH_EXACT_SORT)
    assert gpu.search_arrays(np.ones(dim), 1, 0)[0].tolist() == [n - 1]  # first inserted row


def test_duplicates_and_zero_rows(V, O):
    rng = np.random.default_rng(11)
    dim, n = 64, 4000
    rows = unit_rows(rng, n, dim)
    dup = rng.choice(n, size=n // 100, replace=False)
    rows[dup] = rows[(dup + 7) % n]  # 1 % exact duplicates
    rows[17] = 0.0  # zero row: cosine 0.0 branch (src/lib.rs:439-440)
    rows[n - 1] = 0.0
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    for qi in range(6):
        q = rows[dup[qi]] if qi < 3 else unit_rows(rng, 1, dim)[0]  # query equal to a duplicated row
        for name, m in M.items():
            for k in (1, 2, 10, 32):
                assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (qi, name, k))
    zq = np.zeros(dim)  # zero query: every cosine score is 0.0 -> pure insertion order
    for name, m in M.items():
        assert_same(V, gpu.search_arrays(zq, 10, m), ref.search(zq, 10, m), ("zero query", name))


# ---------------------------------------------------------------------------------------------
# k handling and the exact pipelines
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim", [(50, 8), (700, 48), (5000, 384)])
def test_large_k_and_forced_exact_paths(V, O, n, dim):
    rng = np.random.default_rng(n)
    rows = unit_rows(rng, n, dim)
    rows[n // 2] = rows[n // 3]
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    q = unit_rows(rng, 1, dim)[0]
    for name, m in M.items():
        for k in (0, 1, 33, 64, 65, 200, n, n + 1, 10 * n):
            assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (name, k))
        for path in (V.PATH_EXACT_SELECT, V.PATH_EXACT_SORT):
            gpu.force_path(path)
            for k in (1, 10, 64):
                assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (name, k, path))
                assert V.last_path() == path
        gpu.force_path(0)


def test_out_of_domain_rows_use_exact_path(V, O):
    rng = np.random.default_rng(3)
    dim, n = 32, 300
    rows = unit_rows(rng, n, dim)
    rows[5] *= 1e30   # beyond the f32 fast-path domain
    rows[9] *= 1e-30
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    q = unit_rows(rng, 1, dim)[0]
    for name, m in M.items():
        assert_same(V, gpu.search_arrays(q, 10, m), ref.search(q, 10, m), name)
        assert V.last_path() == V.PATH_EXACT_SELECT
    gpu.delete(int(ids[5]))
    gpu.delete(int(ids[9]))
    ref.delete(int(ids[5]))
    ref.delete(int(ids[9]))
    assert_same(V, gpu.search_arrays(q, 10, 0), ref.search(q, 10, 0), "after delete")
    assert V.last_path() == V.PATH_FAST
    # a huge query also leaves the domain
    assert_same(V, gpu.search_arrays(q * 1e200, 10, 0), ref.search(q * 1e200, 10, 0), "huge query")


def test_nan_scores_map_to_the_panic(V, O):
    idx = V.FlatIndex(2, [V.Vector(1, [1.0, 2.0]), V.Vector(2, [3.0, 4.0])])
    with pytest.raises(V.NaNScore):
        idx.search([float("nan"), 0.0], 1, 3)
    with pytest.raises(V.NaNScore):
        idx.search([float("inf"), 1.0], 1, 0)  # inf/inf
    one = V.FlatIndex(2, [V.Vector(1, [1.0, 2.0])])
    res = one.search([float("nan"), 0.0], 1, 3)  # 1 row: the comparator never runs
    assert len(res) == 1 and res[0].score != res[0].score


# ---------------------------------------------------------------------------------------------
# trait surface: add / delete / get / errors (src/index/flat.rs:82-135)
# ---------------------------------------------------------------------------------------------
def test_trait_surface_and_errors(V, O):
    idx = V.FlatIndex(3)
    assert idx.is_empty() and len(idx) == 0 and idx.dimension() == 3 and idx.max_id() is None
    assert idx.search([1, 2], 5, 0) == []  # empty index accepts any query length (:99)
    idx.add(V.Vector(7, [1, 0, 0], "seven", {"k": 1}))
    with pytest.raises(V.IndexOpError, match="Vector ID 7 already exists"):
        idx.add(V.Vector(7, [0, 1, 0]))
    with pytest.raises(V.IndexOpError, match="Vector dimension mismatch"):
        idx.add(V.Vector(8, [0, 1]))
    with pytest.raises(V.DimensionMismatch) as e:
        idx.search([1, 2], 1, 0)
    assert (e.value.expected, e.value.actual) == (3, 2)
    idx.add(V.Vector(9, [0, 0, 0]))
    idx.add(V.Vector(3, [-1, 0, 0]))
    res = idx.search([1, 0, 0], 10, V.SimilarityMetric.Cosine)
    assert [r.id for r in res] == [7, 9, 3] and [r.score for r in res] == [1.0, 0.0, -1.0]
    assert res[0].text == "seven" and res[0].metadata == {"k": 1}
    assert idx.search([1, 0, 0], 0, 0) == []
    idx.delete(9)
    idx.delete(12345)  # absent id: Ok
    assert len(idx) == 2 and idx.max_id() == 7
    assert [r.id for r in idx.search([1, 0, 0], 10, 0)] == [7, 3]
    assert idx.get_vector(3).values == [-1.0, 0.0, 0.0] and idx.get_vector(9) is None
    c = idx.clone()
    idx.delete(7)
    assert [r.id for r in c.search([1, 0, 0], 10, 0)] == [7, 3] and len(idx) == 1
    ids, vals = c.export()
    assert ids.tolist() == [7, 3] and vals.tolist() == [[1, 0, 0], [-1, 0, 0]]


def test_add_delete_sequence_matches_oracle(V, O):
    rng = np.random.default_rng(21)
    dim = 40
    gpu, ref = V.FlatIndex(dim), O.FlatOracle(dim)
    live = []
    next_id = 0
    for step in range(400):
        if live and rng.random() < 0.3:
            victim = live.pop(int(rng.integers(len(live))))
            gpu.delete(victim)
            ref.delete(victim)
        else:
            v = unit_rows(rng, 1, dim)[0]
            if rng.random() < 0.1 and live:
                v = gpu.get_vector(live[0]).values  # duplicate content, new id
            gpu.add(V.Vector(next_id, v))
            ref.add(next_id, v)
            live.append(next_id)
            next_id += 1
        if step % 40 == 39:
            q = unit_rows(rng, 1, dim)[0]
            for name, m in M.items():
                assert_same(V, gpu.search_arrays(q, 7, m), ref.search(q, 7, m), (step, name))
    assert len(gpu) == len(ref)


def test_bulk_add_validation_stops_at_first_duplicate(V):
    idx = V.FlatIndex(2)
    idx.add_rows([1, 2, 3], np.eye(3, 2))
    with pytest.raises(V.IndexOpError, match="Vector ID 2 already exists"):
        idx.add_rows([4, 2, 5], np.ones((3, 2)))
    assert len(idx) == 4 and idx.get_vector(4) is not None and idx.get_vector(5) is None
    # FlatIndex::new keeps duplicate ids; delete removes all of them, get returns the first
    dup = V.FlatIndex(2, [V.Vector(1, [1, 0]), V.Vector(1, [0, 1]), V.Vector(2, [1, 1])])
    assert len(dup) == 3 and dup.get_vector(1).values == [1.0, 0.0]
    dup.delete(1)
    assert len(dup) == 1 and [r.id for r in dup.search([1, 1], 5, 0)] == [2]


def test_search_batch_equals_single_searches(V, O):
    rng = np.random.default_rng(8)
    n, dim, nq, k = 3000, 128, 17, 10
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, nq, dim)
    for name, m in M.items():
        bi, bs, bn = gpu.search_batch(Q, k, m)
        for i in range(nq):
            ri, rs = ref.search(Q[i], k, m)
            assert bn[i] == k and bi[i].tolist() == ri.tolist() and bs[i].tolist() == rs.tolist(), (name, i)


@pytest.mark.parametrize("dim", [128, 384, 768, 100])
def test_batch_scan_kernel_parity(V, O, dim):
    """k_scan_batch (8 queries per slab pass) vs the oracle, including queries that must fall back:
    duplicates at the cut, an out-of-domain query, a zero query; dim = 100 has no batch shape."""
    rng = np.random.default_rng(100 + dim)
    n, nq = 6000, 7   # 7 queries: the f32 batch kernel (8 and more take the bf16 MFMA filter)
    rows = unit_rows(rng, n, dim)
    rows[4000:4040] = rows[17]  # 41 identical rows: a tie group wider than any k below
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, nq, dim)
    Q[3] = rows[17]       # hits the tie group -> exact fallback for this query only
    Q[5] = Q[5] * 1e100   # out of the f32 domain
    Q[6] = 0.0            # zero query
    for name, m in M.items():
        for k in (1, 10, 32, 48):
            bi, bs, bn = gpu.search_batch(Q, k, m)
            pos, pid, psc, pn = gpu.search_batch_positions(Q, k, m)
            for i in range(nq):
                ri, rs = ref.search(Q[i], k, m)
                assert bn[i] == len(ri) and pn[i] == len(ri)
                assert bi[i].tolist() == ri.tolist(), (dim, name, k, i)
                assert bs[i].tolist() == rs.tolist(), (dim, name, k, i)
                assert pid[i].tolist() == ri.tolist() and psc[i].tolist() == rs.tolist()
                assert [int(ids[p]) for p in pos[i]] == ri.tolist()


@pytest.mark.parametrize("dim,n", [(128, 9000), (384, 12000), (768, 8500), (700, 9001), (100, 4000), (384, 1500), (128, 8191), (128, 100000), (100, 9000), (512, 9000), (450, 8300), (256, 20000)])
def test_mfma_large_batch_parity(V, O, dim, n):
    """>= 2 queries per call on an index of >= 8192 rows take the bf16 MFMA candidate filter; results must still be
    the oracle's bit for bit, including queries the filter cannot certify (ties, out-of-domain)."""
    rng = np.random.default_rng(7 * dim + n)
    nq = 150
    rows = unit_rows(rng, n, dim)
    rows[1000:1040] = rows[17]
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, nq, dim)
    Q[3] = rows[17]
    Q[5] = Q[5] * 1e100
    Q[9] = 0.0
    Q[11] = rows[500] * 0.999 + Q[11] * 0.001
    # which pipeline answers: one bf16 MFMA pass for the whole batch once the index has MFMA_MIN_ROWS = 8192
    # rows (below that the sampling pass cannot produce thresholds), else 19 f32 passes of 8 queries
    gpu.profile_read()
    gpu.profile_enable(True)
    gpu.search_batch(Q, 10, 0)
    gpu.profile_enable(False)
    passes = gpu.profile_read()[0]
    if n >= 8192:  # dim 100 included: its bf16 rows are padded to the 128-column MFMA shape
        assert passes <= 1 + 24, passes  # the MFMA pass + the few queries it cannot certify (ties, out-of-domain), one by one
    else:
        assert passes == 19 if dim != 100 else nq - 2 <= passes <= nq, passes  # dim 100 has no 8-query f32 shape: one by one (the out-of-domain query skips the scan)
    for name in ("cosine", "dotproduct", "euclidean"):
        m = M[name]
        for k in (1, 10, 48):
            bi, bs, bn = gpu.search_batch(Q, k, m)
            for i in range(nq):
                ri, rs = ref.search(Q[i], k, m)
                assert bn[i] == len(ri)
                assert bi[i].tolist() == ri.tolist(), (dim, name, k, i)
                assert bs[i].tolist() == rs.tolist(), (dim, name, k, i)
    # rows added and deleted after the bf16 slab exists are seen by the next batch
    extra = unit_rows(rng, 3, dim)
    for j in range(3):
        gpu.add(V.Vector(10 ** 9 + j, extra[j]))
        ref.add(10 ** 9 + j, extra[j])
    gpu.delete(int(ids[500]))
    ref.delete(int(ids[500]))
    Q[0] = extra[1]
    bi, bs, bn = gpu.search_batch(Q, 10, 0)
    for i in range(nq):
        ri, rs = ref.search(Q[i], 10, 0)
        assert bi[i].tolist() == ri.tolist() and bs[i].tolist() == rs.tolist(), i


def test_sharded_index_single_rank_equals_flat(V, O):
    from vectorlite_amd.sharded import ShardedFlatIndex
    rng = np.random.default_rng(31)
    n, dim = 5000, 384
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    sh = ShardedFlatIndex(gpu, transport="torch")  # no process group: a world of one, device merge all the same
    assert (sh.offset, sh.total, sh.world) == (0, n, 1)
    Q = unit_rows(rng, 11, dim)
    for m in range(4):
        bi, bs, bn = sh.search_batch(Q, 10, m)
        for i in range(11):
            ri, rs = ref.search(Q[i], 10, m)
            assert bi[i].tolist() == ri.tolist() and bs[i].tolist() == rs.tolist()
    assert sh.global_len() == n


def test_device_resident_ingest_matches_host_ingest(V, O):
    import torch
    rng = np.random.default_rng(2)
    n, dim = 2000, 384
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    a = V.FlatIndex(dim)
    a.add_rows(ids, torch.from_numpy(rows).to("cuda:0"), validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    q = unit_rows(rng, 1, dim)[0]
    for name, m in M.items():
        assert_same(V, a.search_arrays(q, 10, m), ref.search(q, 10, m), name)
    assert np.array_equal(a.export()[1], rows)


# ---------------------------------------------------------------------------------------------
# HNSW distance callbacks (src/index/hnsw.rs:113-174)
# ---------------------------------------------------------------------------------------------
def test_hnsw_distance_callbacks_bit_exact(V, O):
    rng = np.random.default_rng(4)
    for dim in (3, 96, 384, 500):
        n = 300
        rows = unit_rows(rng, n, dim) * rng.uniform(0.5, 30.0, size=(n, 1))
        rows[3] = 0.0
        gpu = V.FlatIndex(dim)
        gpu.add_rows(np.arange(n, dtype=np.uint64), rows)
        q = unit_rows(rng, 1, dim)[0] * 3.0
        pos = rng.integers(0, n, size=150)
        for name, m in M.items():
            got = gpu.hnsw_distances(q, pos, m)
            want = [O.hnsw_distance(m, q, rows[p]) for p in pos]
            assert got.tolist() == want, (dim, name)
    # reference fixture (src/index/hnsw.rs:605-634): d_u64 = 173, 1424, 1424, 911
    gpu = V.FlatIndex(3, [V.Vector(100, [1, 0, 0]), V.Vector(200, [0, 1, 0]), V.Vector(300, [0, 0, 1]),
                          V.Vector(400, [1, 1, 0])])
    assert gpu.hnsw_distances([1.1, 0.1, 0.1], [0, 1, 2, 3], 1).tolist() == [173, 1424, 1424, 911]
    # saturation: inf -> u64::MAX, NaN -> 0, zero norm -> 1000
    big = V.FlatIndex(1, [V.Vector(1, [1e300]), V.Vector(2, [0.0])])
    assert big.hnsw_distances([-1e300], [0], 1).tolist() == [2 ** 64 - 1]
    assert big.hnsw_distances([1.0], [1], 0).tolist() == [1000]
    assert big.hnsw_distances([float("nan")], [0], 3).tolist() == [0]


# ---------------------------------------------------------------------------------------------
# concurrency (searches under RwLock::read run concurrently in the reference, src/client.rs:398)
# and size-independent properties at a larger N
# ---------------------------------------------------------------------------------------------
def test_concurrent_searches_from_many_threads(V, O):
    import threading
    rng = np.random.default_rng(77)
    n, dim = 50000, 128
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, 64, dim)
    want = [ref.search(Q[i], 10, i % 4) for i in range(64)]
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for i in range(t, 64, 8):
                    gi, gs = gpu.search_arrays(Q[i], 10, i % 4)
                    if gi.tolist() != want[i][0].tolist() or gs.tolist() != want[i][1].tolist():
                        errors.append((t, i))
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert errors == []


def test_coalesced_concurrent_searches_match_lone_searches(V, O):
    """vl_index_set_coalescing: callers arriving together share slab passes (f32 batch kernel for 2-7,
    bf16 MFMA filter for >= 8) and still get exactly the lone-search answer; mixed metrics / k values
    split into separate passes; a NaN query fails alone (src/index/flat.rs:116 panics that search only)."""
    import threading
    rng = np.random.default_rng(78)
    n, dim = 60000, 128
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    nq = 96
    Q = unit_rows(rng, nq, dim)
    spec = [(0, 10) if i % 3 else ((i // 3) % 4, 5 + (i % 2) * 5) for i in range(nq)]  # mostly (cosine, 10)
    want = [ref.search(Q[i], k, m) for i, (m, k) in enumerate(spec)]
    gpu.set_coalescing(64, 200)
    errors = []
    nan_seen = []
    barrier = threading.Barrier(16)

    def worker(t):
        try:
            barrier.wait()
            for rep in range(2):
                for i in range(t, nq, 16):
                    m, k = spec[i]
                    gi, gs = gpu.search_arrays(Q[i], k, m)
                    if gi.tolist() != want[i][0].tolist() or gs.tolist() != want[i][1].tolist():
                        errors.append((t, i))
                if t == 3:
                    bad = Q[0].copy()
                    bad[5] = np.nan
                    try:
                        gpu.search_arrays(bad, 10, 0)
                        errors.append((t, "NaN query did not fail"))
                    except V.NaNScore:
                        nan_seen.append(rep)
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert errors == []
    assert nan_seen == [0, 1]
    batches, queries = gpu.coalesce_stats()
    assert queries == 2 * nq + 2
    assert batches < queries  # passes were shared
    gpu.set_coalescing(0)
    gi, gs = gpu.search_arrays(Q[1], 10, 0)
    assert gi.tolist() == want[1][0].tolist()
    assert gpu.coalesce_stats() == (batches, queries)


def test_coalescing_is_on_by_default_and_a_lone_caller_takes_the_single_search_path(V, O, monkeypatch):
    """Round 4: handles start with coalescing on (max 256, window 0) -- the reference's many-readers usage
    (src/client.rs:398, src/server.rs:269).  A lone caller leads a pass of one = the plain single search (fast path, no
    batch kernels); concurrent callers share passes and get the lone answers; VL_COALESCE=0 starts handles with it off."""
    import threading
    rng = np.random.default_rng(404)
    n, dim, k = 40000, 96, 10
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, 64, dim)
    assert gpu.coalesce_stats() == (0, 0)
    for i in range(4):
        gi, gs = gpu.search_arrays(Q[i], k, i)
        assert V.last_path() == V.PATH_FAST
        ri, rs = ref.search(Q[i], k, i)
        assert gi.tolist() == ri.tolist() and gs.tolist() == rs.tolist()
    assert gpu.coalesce_stats() == (4, 4)          # four passes of one query each: nobody waited for anybody
    want = [ref.search(Q[i], k, 0) for i in range(64)]
    errors = []
    bar = threading.Barrier(8)

    def worker(t):
        try:
            bar.wait()
            for rep in range(3):
                for i in range(t, 64, 8):
                    gi, gs = gpu.search_arrays(Q[i], k, 0)
                    if gi.tolist() != want[i][0].tolist() or gs.tolist() != want[i][1].tolist():
                        errors.append((t, i))
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert errors == []
    batches, queries = gpu.coalesce_stats()
    assert queries == 4 + 3 * 64 and batches < queries   # passes were shared without anybody asking for it
    c = gpu.clone()
    c.search_arrays(Q[0], k, 0)
    assert c.coalesce_stats() == (1, 1)            # a clone starts the same way
    monkeypatch.setenv("VL_COALESCE", "0")
    off = V.FlatIndex(dim)
    off.add_rows(ids[:1000], rows[:1000], validate=False)
    off.search_arrays(Q[0], k, 0)
    assert off.coalesce_stats() == (0, 0)


def test_adaptive_gather_changes_how_passes_fill_not_what_they_answer(V, O):
    """vl_index_coalesce_gather: a lone caller never waits (estimates 1, 1); eight callers in a closed loop get the lone
    answers with the gather on and off; with it on, leaders wait for their peers (counted) and the passes are fuller;
    switching it off stops the waiting.  Flat and HNSW handles answer the same entry point."""
    import threading
    rng = np.random.default_rng(77)
    n, dim, k = 300_000, 64, 10
    rows = unit_rows(rng, n, dim)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(np.arange(n, dtype=np.uint64), rows, validate=False)
    Q = unit_rows(rng, 96, dim)
    want = [gpu.search_arrays(Q[i], k, 0) for i in range(96)]
    assert gpu.coalesce_gather() == (0, 0)               # 96 lone searches: nobody ever waited

    def closed_loop():
        errors = []
        bar = threading.Barrier(8)

        def worker(t):
            try:
                bar.wait()
                for rep in range(6):
                    for i in range(t, 96, 8):
                        gi, gs = gpu.search_arrays(Q[i], k, 0)
                        if gi.tolist() != want[i][0].tolist() or gs.tolist() != want[i][1].tolist():
                            errors.append((t, i))
            except Exception as e:  # pragma: no cover
                errors.append((t, repr(e)))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        b0, q0 = gpu.coalesce_stats()
        [x.start() for x in th]
        [x.join() for x in th]
        b1, q1 = gpu.coalesce_stats()
        assert errors == [] and q1 - q0 == 6 * 96
        return (q1 - q0) / (b1 - b0)

    fill_on = closed_loop()
    w_on, us_on = gpu.coalesce_gather()
    assert w_on > 0 and us_on >= 0                        # leaders did wait for peers on their way back
    assert gpu.coalesce_gather(False) == (w_on, us_on)   # the switch; the counters stay
    fill_off = closed_loop()
    assert gpu.coalesce_gather() == (w_on, us_on)         # off: nobody waits
    assert fill_on > 1.0 and fill_off > 1.0               # passes were shared either way
    gpu.coalesce_gather(True)
    hn = V.HNSWIndex(dim, 0)
    hn.add_rows(np.arange(5000, dtype=np.uint64), rows[:5000])
    hn.search_arrays(Q[0], k, 0)
    assert hn.coalesce_gather() == (0, 0) and hn.coalesce_gather(False) == (0, 0)


def test_large_index_properties(V):
    """N = 2M x 128 (too big for the oracle in a test): fast path == exact path bit for bit,
    sortedness, idempotence, self-query returns the row itself with score 1, delete removes it."""
    import torch
    n, dim = 2_000_000, 128
    g = torch.Generator(device="cuda:0")
    g.manual_seed(7)
    idx = V.FlatIndex(dim)
    idx.reserve(n)
    for c0 in range(0, n, 500_000):
        x = torch.randn((500_000, dim), dtype=torch.float64, device="cuda:0", generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(c0, c0 + 500_000, dtype=np.uint64) * 3 + 1, x, validate=False)
    assert len(idx) == n
    rng = np.random.default_rng(1)
    Q = unit_rows(rng, 6, dim)
    for m in range(4):
        for qi in range(6):
            fi, fs = idx.search_arrays(Q[qi], 10, m)
            assert V.last_path() == V.PATH_FAST
            fi2, fs2 = idx.search_arrays(Q[qi], 10, m)
            assert fi.tolist() == fi2.tolist() and fs.tolist() == fs2.tolist()  # idempotent
            assert all(fs[i - 1] >= fs[i] for i in range(1, 10))               # sorted
            idx.force_path(V.PATH_EXACT_SELECT)
            ei, es = idx.search_arrays(Q[qi], 10, m)
            idx.force_path(0)
            assert fi.tolist() == ei.tolist() and fs.tolist() == es.tolist(), (m, qi)
    bi, bs, bn = idx.search_batch(Q, 10, 0)
    for qi in range(6):
        fi, fs = idx.search_arrays(Q[qi], 10, 0)
        assert bi[qi].tolist() == fi.tolist() and bs[qi].tolist() == fs.tolist()
    probe_id = 1_234_567 * 3 + 1
    v = idx.get_vector(probe_id).values
    r = idx.search(v, 3, 0)
    assert r[0].id == probe_id and abs(r[0].score - 1.0) < 1e-12
    idx.delete(probe_id)
    assert len(idx) == n - 1 and idx.get_vector(probe_id) is None
    r2 = idx.search(v, 3, 0)
    assert r2[0].id == r[1].id and r2[0].score == r[1].score


@pytest.mark.parametrize("copies", [700, 3000])
def test_staged_filter_with_thousands_of_equal_keys_in_a_candidate_buffer(V, copies):
    """The staged bf16 filter on an index long enough for refinements between stages (600 k x 64, 2048 queries: 16 query
    chunks), with `copies` identical rows spread over the index and queries next to them: their keys are EQUAL, so they all
    sit at or above every threshold -- the candidate buffers of those queries hold hundreds to thousands of entries that no
    refinement can thin out (k_refine_thresholds' and k_select_candidates' long-buffer paths, beyond what their LDS
    compaction keeps).  Every row of the batch must equal the lone search, whose ties resolve in insertion order."""
    import torch
    n, dim, nq, k = 600_000, 64, 2048, 10
    g = torch.Generator(device="cuda:0")
    g.manual_seed(11)
    x = torch.randn((n, dim), dtype=torch.float64, device="cuda:0", generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    rng = np.random.default_rng(5)
    where = np.sort(rng.choice(n, copies, replace=False))
    twin = rng.standard_normal(dim)
    twin /= np.linalg.norm(twin)
    x[torch.from_numpy(where).to("cuda:0")] = torch.from_numpy(twin).to("cuda:0")
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(n, dtype=np.uint64) + 5, x, validate=False)
    del x
    Q = unit_rows(rng, nq, dim)
    planted = [0, 95, 96, 1023, 1024, 2047]
    for qi in planted:
        v = twin + 0.02 * rng.standard_normal(dim)
        Q[qi] = v / np.linalg.norm(v)
    for metric in (0, 1, 3):
        bi, bs, bn = idx.search_batch(Q, k, metric)
        assert idx.last_filter()["stages"] >= 2          # refinements ran
        assert bn.tolist() == [k] * nq
        for qi in planted + [1, 500, 1500]:
            si, ss = idx.search_arrays(Q[qi], k, metric)
            assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), (metric, qi)
        for qi in planted:                               # the twins win, first inserted first
            assert bi[qi].tolist() == (where[:k] + 5).tolist() and len(set(bs[qi].tolist())) == 1, (metric, qi)


# ---------------------------------------------------------------------------------------------
# adversarial near-ties: the f32 / bf16 filters cannot order these rows, the bound check must notice
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("eps", [1e-3, 1e-6, 1e-9, 1e-13])
def test_near_duplicate_rows_below_filter_precision(V, O, eps):
    """A cluster of 200 rows that differ from one base row by ~eps per component: their f64 scores
    are distinct, their f32 (eps <= 1e-9) or bf16 (eps <= 1e-3) keys are not.  Whatever path answers,
    ids and scores must be the oracle's."""
    rng = np.random.default_rng(int(-np.log10(eps)))
    n, dim = 6000, 128
    rows = unit_rows(rng, n, dim)
    base = rows[77].copy()
    cluster = rng.choice(np.arange(100, n), size=200, replace=False)
    rows[cluster] = base + eps * rng.standard_normal((200, dim))
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    queries = [base, base + 0.01 * rng.standard_normal(dim), unit_rows(rng, 1, dim)[0]]
    for q in queries:
        for name, m in M.items():
            for k in (1, 10, 32):
                assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (eps, name, k))
    Q = np.stack([queries[i % 3] * (1.0 + 0.001 * i) for i in range(96)])  # 96 queries -> MFMA filter
    for name in ("cosine", "euclidean", "dotproduct", "manhattan"):
        bi, bs, bn = gpu.search_batch(Q, 10, M[name])
        for i in range(96):
            ri, rs = ref.search(Q[i], 10, M[name])
            assert bi[i].tolist() == ri.tolist() and bs[i].tolist() == rs.tolist(), (eps, name, i)


def test_bound_check_survives_random_scales(V, O):
    """Random row norms over 12 decades and random query scales: the error bound uses the largest row
    norm and the query norm, so mixed magnitudes are where a wrong bound would show."""
    rng = np.random.default_rng(2024)
    dim, n = 64, 3000
    for trial in range(6):
        scales = 10.0 ** rng.uniform(-6, 6, size=(n, 1))
        rows = rng.standard_normal((n, dim)) * scales
        ids = permuted_ids(n)
        gpu = V.FlatIndex(dim)
        gpu.add_rows(ids, rows, validate=False)
        ref = O.FlatOracle(dim, ids, rows)
        q = rng.standard_normal(dim) * 10.0 ** rng.uniform(-6, 6)
        for name, m in M.items():
            assert_same(V, gpu.search_arrays(q, 10, m), ref.search(q, 10, m), (trial, name))


@pytest.mark.parametrize("dim", [128, 384, 768])
def test_bf16_single_query_filter_is_exact(V, O, dim):
    """Opt-in first stage on the bf16 slab: same ids and scores as the oracle, on easy queries
    (certified by the bf16 bound) and on near-ties (handed on to the f32 scan / exact kernels)."""
    rng = np.random.default_rng(dim)
    n = 20000
    rows = unit_rows(rng, n, dim)
    rows[5000:5030] = rows[9] + 1e-4 * rng.standard_normal((30, dim))  # closer than bf16 can resolve
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    gpu.set_single_filter("bf16")
    ref = O.FlatOracle(dim, ids, rows)
    for qi in range(10):
        q = rows[9] if qi == 0 else unit_rows(rng, 1, dim)[0]
        for name, m in M.items():
            for k in (1, 10, 40):
                assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (dim, qi, name, k))
    gpu.add(V.Vector(10 ** 12, rows[3] * 1.0000001))
    ref.add(10 ** 12, rows[3] * 1.0000001)
    assert_same(V, gpu.search_arrays(rows[3], 5, 0), ref.search(rows[3], 5, 0), "after add")


def test_seeded_fuzz_adversarial_values(V, O):
    """240 seeded cases with random shapes and unfriendly values: mixed magnitudes up to the domain
    edge (2^40, 2^-40), denormals, signed zeros, integer grids (exact ties in every metric), repeated
    rows, zero rows, and queries drawn from the rows themselves.  Singles, and a batch per case."""
    rng = np.random.default_rng(20251003)
    kinds = ("gauss", "grid", "mixed_mag", "tiny", "huge", "dups")
    for case in range(240):
        kind = kinds[case % len(kinds)]
        dim = int(rng.choice([1, 2, 3, 7, 8, 16, 31, 64, 100, 128, 200]))
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 130, 700]))
        if kind == "gauss":
            rows = rng.standard_normal((n, dim))
        elif kind == "grid":
            rows = rng.integers(-2, 3, size=(n, dim)).astype(np.float64)
        elif kind == "mixed_mag":
            rows = rng.standard_normal((n, dim)) * np.exp2(rng.integers(-30, 31, size=(n, 1)).astype(np.float64))
        elif kind == "tiny":
            rows = rng.standard_normal((n, dim)) * 2.0 ** -36
            rows[rng.integers(0, n)] = 5e-324 * rng.integers(0, 3, size=dim)  # denormals (norm < 2^-40: out of the fast domain)
        elif kind == "huge":
            rows = rng.uniform(-1, 1, size=(n, dim)) * 2.0 ** 39
        else:
            base = rng.standard_normal((max(1, n // 4), dim))
            rows = base[rng.integers(0, base.shape[0], size=n)]
        if n > 3:
            rows[rng.integers(0, n)] = 0.0
            rows[rng.integers(0, n)] *= -0.0 if kind == "grid" else 1.0
        ids = permuted_ids(n)
        gpu = V.FlatIndex(dim)
        gpu.add_rows(ids, rows, validate=False)
        ref = O.FlatOracle(dim, ids, rows)
        Q = [rows[rng.integers(0, n)].copy(), rng.standard_normal(dim) * (2.0 ** 20 if kind == "huge" else 1.0),
             np.zeros(dim)]
        if kind == "grid":
            Q.append(rng.integers(-2, 3, size=dim).astype(np.float64))
        for q in Q:
            for m in range(4):
                for k in (1, 10, n) + ((100, 200) if n >= 700 else ()):  # 100 / 200: the multi-list fast path
                    assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (case, kind, dim, n, m, k))
        Qb = np.stack(Q + [rows[i % n] * 0.5 for i in range(9 - len(Q))])
        m = case % 4
        bi, bs, bn = gpu.search_batch(Qb, 10, m)
        for j in range(Qb.shape[0]):
            want = ref.search(Qb[j], 10, m)
            assert bi[j, : bn[j]].tolist() == want[0].tolist() and bs[j, : bn[j]].tolist() == want[1].tolist(), (case, kind, j)


def test_stateful_fuzz_large_index_all_filters(V, O):
    """An index big enough for every candidate filter (f32 scan, bf16 single-query filter, bf16 MFMA batch
    filter) under a random stream of adds, bulk adds, deletes (present, absent, duplicated ids), failed adds
    and clones; every search is compared with the oracle that received the same stream."""
    rng = np.random.default_rng(424242)
    dim, n0 = 128, 9000
    rows = unit_rows(rng, n0, dim)
    ids = permuted_ids(n0)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    live = ids.tolist()
    next_id = 10 ** 12
    for step in range(80):
        op = rng.integers(0, 8)
        if op == 0:  # single add
            v = unit_rows(rng, 1, dim)[0] * float(rng.choice([1.0, 3.0, 0.25]))
            gpu.add(V.Vector(next_id, v)); ref.add(next_id, v); live.append(next_id); next_id += 1
        elif op == 1:  # bulk add
            c = int(rng.integers(1, 60))
            vs = unit_rows(rng, c, dim)
            new = np.arange(next_id, next_id + c, dtype=np.uint64)
            gpu.add_rows(new, vs)
            for i in range(c):
                ref.add(int(new[i]), vs[i])
            live += new.tolist(); next_id += c
        elif op == 2 and live:  # delete a live id
            victim = live.pop(int(rng.integers(len(live))))
            gpu.delete(victim); ref.delete(victim)
        elif op == 3:  # delete an absent id: Ok, nothing happens (src/index/flat.rs:93-96)
            gpu.delete(7); ref.delete(7)
        elif op == 4 and live:  # duplicate id: refused, index unchanged
            with pytest.raises(V.IndexOpError, match="already exists"):
                gpu.add(V.Vector(live[0], unit_rows(rng, 1, dim)[0]))
            with pytest.raises(O.OracleError):
                ref.add(live[0], unit_rows(rng, 1, dim)[0])
        elif op == 5:  # wrong dimension: refused
            with pytest.raises(V.IndexOpError, match="dimension mismatch"):
                gpu.add(V.Vector(next_id + 10 ** 6, np.ones(dim + 1)))
        elif op == 6:  # the clone takes over (src/persistence.rs:118 clones the index)
            gpu = gpu.clone()
        else:
            gpu.set_single_filter("bf16" if rng.random() < 0.5 else "f32")
        assert len(gpu) == len(ref) == len(live)
        m = int(rng.integers(0, 4))
        k = int(rng.choice([1, 10, 48, 70]))
        Q = unit_rows(rng, 12, dim)
        Q[0] = np.asarray(gpu.get_vector(live[int(rng.integers(len(live)))]).values)  # an exact hit
        assert_same(V, gpu.search_arrays(Q[0], k, m), ref.search(Q[0], k, m), (step, "single", m, k))
        bi, bs, bn = gpu.search_batch(Q, k, m)
        for j in range(12):
            want = ref.search(Q[j], k, m)
            assert bi[j, : bn[j]].tolist() == want[0].tolist() and bs[j, : bn[j]].tolist() == want[1].tolist(), (step, j, m, k)


def test_exact_select_rounds_for_k_above_64(V, O):
    """64 < k <= 1024 on a large enough index: rounds of 64, each restricted to the rows behind the previous
    round's last entry; duplicated rows straddle the round boundaries (ties resolved by position)."""
    rng = np.random.default_rng(5150)
    n, dim = 600_000, 16
    rows = unit_rows(rng, n, dim)
    rows[1000:1200] = rows[999]      # 201 identical rows: equal scores across several round boundaries
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    q = rows[999] + 0.001 * rng.standard_normal(dim)
    gpu.force_path(V.PATH_EXACT_SELECT)  # keep the multi-list fast path (k <= 220) out of this test
    for m in (0, 1, 3):
        for k in (65, 128, 129, 200):
            assert_same(V, gpu.search_arrays(q, k, m), ref.search(q, k, m), (m, k))
            # 600 k rows allow 2 rounds (a round is cheaper than the sort only on big indexes, flat_index.cpp)
            assert V.last_path() == (V.PATH_EXACT_SELECT if k <= 128 else V.PATH_EXACT_SORT)
    gpu.force_path(0)
    assert_same(V, gpu.search_arrays(q, 1000, 0), ref.search(q, 1000, 0), "k=1000")
    assert V.last_path() == V.PATH_EXACT_SORT
    # unforced, the same queries sit on 201 tied rows: the multi-list fast path cannot certify them and hands on
    for k in (65, 200):
        assert_same(V, gpu.search_arrays(q, k, 0), ref.search(q, k, 0), ("unforced", k))
        assert V.last_path() != V.PATH_FAST


def test_multi_list_fast_path_for_k_up_to_220(V, O):
    """60 < k <= 220: 2-4 partitions of 64 candidates rescored and ranked together, still ONE f32 scan; bit-identical to
    the oracle, fast path on well-separated data, exact path when the bound cannot hold (ties at the cut)."""
    rng = np.random.default_rng(6161)
    n, dim = 200_000, 32
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, 3, dim)
    fast = 0
    for qi in range(3):
        for m in range(4):
            for k in (61, 64, 92, 93, 100, 156, 157, 220, 221):
                assert_same(V, gpu.search_arrays(Q[qi], k, m), ref.search(Q[qi], k, m), (qi, m, k))
                if k <= 220:
                    fast += int(V.last_path() == V.PATH_FAST)
                else:
                    assert V.last_path() != V.PATH_FAST
    assert fast >= 3 * 4 * 8 * 0.9, fast  # the bound holds on this data almost always
    # ties across the cut: 100 copies of one row, k = 80 cuts through them
    rows2 = rows.copy()
    rows2[500:600] = rows2[499]
    gpu2 = V.FlatIndex(dim)
    gpu2.add_rows(ids, rows2, validate=False)
    ref2 = O.FlatOracle(dim, ids, rows2)
    assert_same(V, gpu2.search_arrays(rows2[499], 80, 0), ref2.search(rows2[499], 80, 0), "ties")
    assert V.last_path() != V.PATH_FAST
    # small index: fewer rows than candidates
    gpu3 = V.FlatIndex(dim)
    gpu3.add_rows(ids[:300], rows[:300], validate=False)
    ref3 = O.FlatOracle(dim, ids[:300], rows[:300])
    for k in (61, 100, 220, 300, 301):
        assert_same(V, gpu3.search_arrays(Q[0], k, 1), ref3.search(Q[0], k, 1), ("small", k))


def test_concurrent_writers_and_coalesced_readers_do_not_deadlock_or_corrupt(V, O):
    """Readers (coalesced and not) run against a writer that adds, deletes and bulk-adds (RwLock::write in the
    reference, src/client.rs:333,383).  Every answer must be a valid answer for SOME state the index went through:
    checked here through invariants (sorted scores, ids that existed, no duplicates), then exactly once quiescent."""
    import threading
    rng = np.random.default_rng(99)
    dim, n0 = 64, 20000
    rows = unit_rows(rng, n0, dim)
    ids = np.arange(n0, dtype=np.uint64)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    gpu.set_coalescing(32, 50)
    extra = unit_rows(rng, 400, dim)
    stop = threading.Event()
    errors = []
    known = set(range(n0 + 400))

    def reader(t):
        r = np.random.default_rng(1000 + t)
        try:
            while not stop.is_set():
                q = unit_rows(r, 1, dim)[0]
                k = int(r.choice([1, 10, 48]))
                gi, gs = gpu.search_arrays(q, k, int(r.integers(0, 4)))
                if len(gi) != k or len(set(gi.tolist())) != k or not set(gi.tolist()) <= known:
                    errors.append(("ids", t, gi.tolist()))
                if any(gs[i] < gs[i + 1] for i in range(len(gs) - 1)):
                    errors.append(("order", t))
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    def writer():
        try:
            for j in range(200):
                gpu.add(V.Vector(n0 + j, extra[j]))
                if j % 3 == 0:
                    gpu.delete(int(j * 7))
            gpu.add_rows(np.arange(n0 + 200, n0 + 400, dtype=np.uint64), extra[200:])
        except Exception as e:  # pragma: no cover
            errors.append(("writer", repr(e)))

    readers = [threading.Thread(target=reader, args=(t,)) for t in range(8)]
    [t.start() for t in readers]
    w = threading.Thread(target=writer)
    w.start()
    w.join(timeout=120)
    stop.set()
    [t.join(timeout=60) for t in readers]
    assert not w.is_alive() and not any(t.is_alive() for t in readers), "deadlock"
    assert errors == []
    # quiescent: exact comparison with an oracle that received the same writes
    ref = O.FlatOracle(dim, ids, rows)
    for j in range(200):
        ref.add(n0 + j, extra[j])
        if j % 3 == 0:
            ref.delete(int(j * 7))
    for j in range(200, 400):
        ref.add(n0 + j, extra[j])
    assert len(gpu) == len(ref)
    q = unit_rows(rng, 1, dim)[0]
    for m in range(4):
        assert_same(V, gpu.search_arrays(q, 10, m), ref.search(q, 10, m), m)


def _embedding_cases(rng):
    # up to 16384 rows take the wave-per-row kernel (k_embed_f32_rows: a batch of queries, a small add), more the
    # lane-per-row one (k_embed_f32: bulk ingest); rows longer than 2048 values always the latter
    for n, dim in [(1, 1), (5, 3), (63, 100), (64, 64), (65, 65), (257, 384), (130, 1000), (17000, 48), (9, 2500)]:
        e = rng.standard_normal((n, dim)).astype(np.float32)
        if n > 4:
            e[2] = 0.0                       # norm == 0: left as it is (src/embeddings.rs:176-180)
            e[3] *= np.float32(1e-30)        # squares far below f32 range, fine in f64
            e[4] *= np.float32(1e30)
        yield n, dim, e


@pytest.mark.parametrize("normalize", [True, False])
def test_embedding_ingest_is_bit_identical_to_host_postprocessing(V, O, kats, normalize):
    """vl_index_add_embeddings_f32: widening + L2 normalisation on the device equals
    src/embeddings.rs:169-181 bit for bit, from host arrays and from device tensors."""
    import torch
    rng = np.random.default_rng(77)
    for n, dim, e in _embedding_cases(rng):
        want = O.embed_f32(e, normalize=normalize).reshape(n, dim)
        ids = permuted_ids(n)
        for src in ("host", "device"):
            idx = V.FlatIndex(dim)
            idx.add_embeddings(ids, e if src == "host" else torch.from_numpy(e).cuda(), normalize=normalize)
            got_ids, got = idx.export()
            assert got_ids.tolist() == ids.tolist()
            assert np.array_equal(got.reshape(n, dim).view(np.uint64), want.view(np.uint64)), (n, dim, src)
            if normalize:  # the reference's own assertion about generate_embedding's output (src/embeddings.rs:374-383)
                tol = kats["embedding_kats"][0]["tol"]
                for r in range(n):
                    if e[r].any():
                        acc = 0.0
                        for x in got.reshape(n, dim)[r].tolist():
                            acc += x * x
                        assert abs(acc ** 0.5 - 1.0) < tol, (n, dim, r)
        # searching the ingested rows == searching an index built from the host-normalised rows
        ref = V.FlatIndex(dim)
        ref.add_rows(ids, want)
        q = rng.standard_normal(dim)
        for m in M.values():
            a, b = idx.search_arrays(q, 7, m), ref.search_arrays(q, 7, m)
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_embedding_ingest_chunks_duplicates_and_hnsw(V, O):
    rng = np.random.default_rng(78)
    # more rows than one 512 MB chunk of f64 rows holds at this width: 2 chunks
    dim, n = 3072, 25000
    e = rng.standard_normal((n, dim)).astype(np.float32)
    idx = V.FlatIndex(dim)
    idx.add_embeddings(np.arange(n, dtype=np.uint64), e)
    assert len(idx) == n
    for r in (0, 21844, 21845, 21846, n - 1):
        assert idx.get_vector(r).values == O.embed_f32(e[r]).tolist()
    # n sequential add() calls: rows in front of the first duplicate id stay (src/index/flat.rs:86-88)
    small = V.FlatIndex(4)
    em = rng.standard_normal((6, 4)).astype(np.float32)
    with pytest.raises(V.IndexOpError, match="already exists"):
        small.add_embeddings([1, 2, 3, 2, 5, 6], em)
    assert len(small) == 3 and small.get_vector(3).values == O.embed_f32(em[2]).tolist()
    # HNSW handles take the same entry point (always validating)
    h = V.HNSWIndex(16, V.SimilarityMetric.Cosine)
    eh = rng.standard_normal((300, 16)).astype(np.float32)
    h.add_embeddings(np.arange(300, dtype=np.uint64), eh)
    assert len(h) == 300 and h.get_vector(17).values == O.embed_f32(eh[17]).tolist()
    res = h.search(O.embed_f32(eh[17]).tolist(), 1, V.SimilarityMetric.Cosine)
    assert res[0].id == 17
    with pytest.raises(V.IndexOpError, match="already exists"):
        h.add_embeddings([1000, 17], eh[:2])
    assert len(h) == 301


def test_coalesced_searches_with_a_huge_k_do_not_wedge_the_queue(V, O):
    """a coalesced pass sized its scratch with the caller's raw k -- k = 2^64 - 1 threw length_error with the
    leader flag set and every later coalesced search blocked forever; k = 2^63 with two queries wrapped the size to 0.
    Every caller must get min(k, len) results, and the queue must keep working afterwards."""
    import threading
    rng = np.random.default_rng(404)
    n, dim = 9000, 16
    rows = unit_rows(rng, n, dim)
    ids = permuted_ids(n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(rng, 8, dim)
    want = [ref.search(Q[i], n, 0) for i in range(8)]
    gpu.set_coalescing(16, 5000)  # a 5 ms window: the threads below really share passes
    out, errors = {}, []
    bar = threading.Barrier(8)

    def worker(t, k):
        try:
            bar.wait()
            out[t] = gpu.search_arrays(Q[t], k, 0)
        except Exception as e:  # pragma: no cover
            errors.append((t, repr(e)))

    for k in (2 ** 64 - 1, 2 ** 63, n + 5):
        out.clear()
        th = [threading.Thread(target=worker, args=(t, k)) for t in range(8)]
        [x.start() for x in th]
        [x.join(timeout=120) for x in th]
        assert not any(x.is_alive() for x in th), "a coalesced search never returned"
        assert errors == []
        for t in range(8):
            gi, gs = out[t]
            assert gi.tolist() == want[t][0].tolist() and gs.tolist() == want[t][1].tolist(), (k, t)
    gi, gs = gpu.search_arrays(Q[0], 10, 0)  # and an ordinary search afterwards
    assert gi.tolist() == want[0][0][:10].tolist()


@pytest.mark.parametrize("dim,metric,nq", [(384, "cosine", 640), (384, "dotproduct", 1408), (768, "euclidean", 1024),
                                            (768, "cosine", 480), (200, "euclidean", 896)])
def test_mfma_batches_whose_chunk_count_8_does_not_divide(V, O, dim, metric, nq):
    """5 / 11 chunks of 128 queries, 11 / 5 chunks of 96 at stride 768, 7 chunks at stride 256: the grids (51, 23, 36 ...
    row-block lanes per chunk) on which the XCD-aware workgroup map (csrc/xcd_map.hpp) is not the identity.  Every row
    block must still be worked on exactly once per chunk: each batch row equals the single-query pipeline, a sample the oracle."""
    rng = np.random.default_rng(dim + nq)
    n = 150_000
    rows = unit_rows(rng, n, dim) * (1.0 + rng.random((n, 1)))
    ids = permuted_ids(n)
    idx = V.FlatIndex(dim)
    idx.add_rows(ids, rows, validate=False)
    Q = unit_rows(rng, nq, dim)
    idx.profile_read()
    idx.profile_enable(True)
    bi, bs, bn = idx.search_batch(Q, 10, M[metric])
    idx.profile_enable(False)
    n_seq, _, _ = idx.profile_read()
    # the MFMA filter's launch sequences served it (a few queries it cannot certify are redone one by one: the GEMM-form
    # Euclidean bound is loose when the row norms vary), not one f32 pass per 8 queries
    assert 1 <= n_seq <= 2 + nq // 20
    assert bn.tolist() == [10] * nq
    for qi in list(range(0, nq, 37)) + [nq - 1]:
        si, ss = idx.search_arrays(Q[qi], 10, M[metric])
        assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), (qi, V.last_path())
    ref = O.FlatOracle(dim, ids, rows)
    for qi in (0, nq // 2, nq - 1):
        assert_same(V, (bi[qi], bs[qi]), ref.search(Q[qi], 10, M[metric]), (dim, metric, nq, qi))


@pytest.mark.parametrize("dim", [32, 64, 50, 96])
def test_f32_batch_path_many_groups_per_launch_matches_the_oracle(V, O, dim):
    """Round 4's K3: up to 512 queries per scan launch (groups of 8 as blockIdx.y), one finalize launch, lazy list insertion,
    four lanes per row at strides 32 / 64.  Small indexes (below the MFMA filter's 8192 rows) and Manhattan take this path:
    batches that straddle the group (8) and launch (512) boundaries against the oracle, all four metrics, ties included."""
    rng = np.random.default_rng(1000 + dim)
    for n in (70, 3000):
        rows = rng.standard_normal((n, dim))
        rows[n // 2] = rows[3]                      # an exact duplicate: insertion order decides
        ids = permuted_ids(n)
        gpu = V.FlatIndex(dim)
        gpu.add_rows(ids, rows, validate=False)
        ref = O.FlatOracle(dim, ids, rows)
        Q = rng.standard_normal((1030, dim))
        Q[5] = rows[3]
        Q[600] = rows[3]
        for metric in range(4):
            for nq in (7, 9, 513, 1030):
                bi, bs, bn = gpu.search_batch(Q[:nq], 10, metric)
                assert bn.tolist() == [10] * nq
                for qi in sorted({min(q, nq - 1) for q in (0, 5, 6, 7, 8, 511, 512, 600, nq - 1)}):
                    ri, rs = ref.search(Q[qi], 10, metric)
                    assert bi[qi].tolist() == ri.tolist() and bs[qi].tolist() == rs.tolist(), (n, metric, nq, qi)
