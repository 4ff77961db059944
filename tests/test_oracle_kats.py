"""Pins the CPU oracle (oracle/vl_oracle.c + the pure-Python restatement) against
the reference's own known-answer tests (tests/golden/reference_kats.json) and
against SURVEY.md section 9.6's derived values.  CPU only."""
import math

import numpy as np
import pytest

from oracle import oracle as O

M = O.METRICS


def _check_scalar(val, kat):
    if "expect" in kat:
        assert abs(val - kat["expect"]) <= kat["tol"], (kat["src"], val)
    if "gt" in kat:
        assert val > kat["gt"], kat["src"]
    if "ge" in kat:
        assert val >= kat["ge"], kat["src"]
    if "lt" in kat:
        assert val < kat["lt"], kat["src"]
    if "le" in kat:
        assert val <= kat["le"], kat["src"]


def test_metric_kats(kats):
    for kat in kats["metric_kats"]:
        m = M[kat["metric"]]
        c = O.calculate(m, kat["a"], kat["b"])
        p = O.py_calculate(m, kat["a"], kat["b"])
        assert c == p, kat["src"]  # two independent restatements agree bit for bit
        _check_scalar(c, kat)


def _run_flat(kat, impl):
    m = M[kat["metric"]]
    if impl == "c":
        idx = O.FlatOracle(kat["dim"], kat["ids"], kat["rows"])
        ids, scores = idx.search(kat["query"], kat["k"], m)
        return list(map(int, ids)), list(map(float, scores))
    return O.py_flat_search(list(zip(kat["ids"], kat["rows"])), kat["query"], kat["k"], m)


@pytest.mark.parametrize("impl", ["c", "py"])
def test_flat_kats(kats, impl):
    by_src = {}
    for kat in kats["flat_kats"]:
        ids, scores = _run_flat(kat, impl)
        by_src[kat["src"]] = (ids, scores)
        assert len(ids) == kat["len"], kat["src"]
        assert ids[0] == kat["first_id"], kat["src"]
        if "first_score" in kat:
            assert abs(scores[0] - kat["first_score"]) <= kat["tol"], kat["src"]
        if "first_score_gt" in kat:
            assert scores[0] > kat["first_score_gt"], kat["src"]
        if kat.get("sorted_desc"):
            assert all(scores[i - 1] >= scores[i] for i in range(1, len(scores))), kat["src"]
    for pair in kats["flat_pairs_differ"]:
        assert by_src[pair["a"]][1][0] != by_src[pair["b"]][1][0], pair["src"]


def test_conversion_kats(kats):
    for kat in kats["conversion_kats"]:
        _check_scalar(O.convert_distance_to_similarity(kat["distance"], M[kat["metric"]]), kat)
    mono = kats["conversion_monotone"]
    for name in mono["metrics"]:
        prev = mono["start"]
        for d in mono["distances"]:
            s = O.convert_distance_to_similarity(d, M[name])
            assert s <= prev, (mono["src"], name, d)
            prev = s


def test_survey_9_6_derived_values():
    """full-precision values derived from the reference's test inputs."""
    idx = O.FlatOracle(3, [1, 2, 3], [[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    ids, sc = idx.search([1, 0, 0], 2, O.COSINE)
    assert list(ids) == [1, 2] and list(sc) == [1.0, 0.0]  # 2 beats 3 by position on the 0.0 tie
    ids, sc = idx.search([1.1, 0.1, 0.1], 2, O.COSINE)
    assert list(ids) == [1, 2]
    assert sc[0] == pytest.approx(0.99183659813417546, abs=1e-15)
    assert sc[1] == pytest.approx(0.090166963466743216, abs=1e-15)
    idx = O.FlatOracle(2, [1, 2, 3], [[0, 0], [3, 4], [6, 8]])
    ids, sc = idx.search([0, 0], 3, O.EUCLIDEAN)
    assert list(ids) == [1, 2, 3] and list(sc) == [1.0, 1.0 / 6.0, 1.0 / 11.0]
    ids, sc = idx.search([0, 0], 3, O.MANHATTAN)
    assert list(ids) == [1, 2, 3] and list(sc) == [1.0, 0.125, 1.0 / 15.0]
    idx = O.FlatOracle(2, [1, 2, 3], [[1, 2], [2, 1], [0, 0]])
    ids, sc = idx.search([1, 2], 2, O.DOT)
    assert list(ids) == [1, 2] and list(sc) == [5.0, 4.0]
    assert O.calculate(O.COSINE, [1, 2], [1, 2]) == 0.99999999999999978
    idx = O.FlatOracle(3, [0, 1], [[1, 2, 3], [4, 5, 6]])
    ids, sc = idx.search([1.1, 2.1, 3.1], 2, O.COSINE)
    assert list(ids) == [0, 1]
    assert sc[0] == pytest.approx(0.99985929035365739, abs=1e-15)
    assert sc[1] == pytest.approx(0.97824918056165788, abs=1e-15)


def test_hnsw_distance_quantisation():
    """src/index/hnsw.rs:113-174 + SURVEY 9.6 rows for hnsw.rs:605-634 / 679-749."""
    q = [1.1, 0.1, 0.1]
    rows = {100: [1, 0, 0], 200: [0, 1, 0], 300: [0, 0, 1], 400: [1, 1, 0]}
    d = {i: O.hnsw_distance(O.EUCLIDEAN, q, v) for i, v in rows.items()}
    assert d == {100: 173, 200: 1424, 300: 1424, 400: 911}
    assert O.hnsw_score(173, O.EUCLIDEAN) == pytest.approx(0.85251491901108267, abs=1e-15)
    assert O.hnsw_score(911, O.EUCLIDEAN) == pytest.approx(0.52328623757195181, abs=1e-15)
    assert O.hnsw_distance(O.EUCLIDEAN, q, [1, 1, 1]) == 1276
    # cosine: zero norm -> 1000 (:139-141); identical -> 0; rounding slightly above 1 saturates to 0
    assert O.hnsw_distance(O.COSINE, [0, 0, 0], [1, 2, 3]) == 1000
    assert O.hnsw_distance(O.COSINE, [1, 2, 3], [1, 2, 3]) == 0
    assert O.hnsw_distance(O.COSINE, [1, 0], [0, 1]) == 1000
    assert O.hnsw_distance(O.COSINE, [1, 2, 3], [-1, -2, -3]) == 2000
    # true cosine 0.8 -> (1-0.8)*1000 = 199.99.. -> 199 -> score 1 - 199/1e6  
    a, b = [1.0, 0.0], [0.8, 0.6]
    assert O.hnsw_distance(O.COSINE, a, b) == 199
    assert O.hnsw_score(199, O.COSINE) == pytest.approx(0.999801, abs=1e-12)
    # dot: clamp to +-1000 before 1000 - dot (:172); NaN -> 0 (Rust `as u64`)
    assert O.hnsw_distance(O.DOT, [1e6], [1.0]) == 0
    assert O.hnsw_distance(O.DOT, [-1e6], [1.0]) == 2000
    assert O.hnsw_distance(O.DOT, [1, 2, 3], [1, 2, 3]) == 986
    assert O.hnsw_distance(O.DOT, [float("nan")], [1.0]) == 0
    assert O.hnsw_distance(O.MANHATTAN, [0, 0], [3, 4]) == 7000
    assert O.hnsw_distance(O.EUCLIDEAN, [1e300], [-1e300]) == 2 ** 64 - 1  # inf saturates


def test_flat_edge_cases():
    idx = O.FlatOracle(3)
    # empty index accepts any query length (src/index/flat.rs:99)
    ids, sc = idx.search([1, 2], 5, O.COSINE)
    assert len(ids) == 0
    idx.add(7, [1, 0, 0])
    with pytest.raises(O.OracleError) as e:
        idx.add(7, [0, 1, 0])
    assert e.value.code == O.DUP_ID
    with pytest.raises(O.OracleError) as e:
        idx.add(8, [0, 1])
    assert e.value.code == O.DIM_MISMATCH
    with pytest.raises(O.OracleError) as e:
        idx.search([1, 2], 1, O.COSINE)
    assert e.value.code == O.DIM_MISMATCH and e.value.detail == {"expected": 3, "actual": 2}
    idx.add(9, [0, 0, 0])  # zero row: cosine 0.0 branch (src/lib.rs:439-440)
    idx.add(3, [-1, 0, 0])
    ids, sc = idx.search([1, 0, 0], 10, O.COSINE)  # k > len
    assert list(ids) == [7, 9, 3] and list(sc) == [1.0, 0.0, -1.0]
    assert len(idx.search([1, 0, 0], 0, O.COSINE)[0]) == 0  # k = 0
    idx.delete(9)
    idx.delete(12345)  # missing id: Ok (src/index/flat.rs:93-96)
    assert len(idx) == 2
    assert list(idx.search([1, 0, 0], 10, O.COSINE)[0]) == [7, 3]
    assert idx.get_vector(3).tolist() == [-1, 0, 0] and idx.get_vector(9) is None
    # NaN score with >= 2 rows panics in the comparator (src/index/flat.rs:116)
    with pytest.raises(O.OracleError) as e:
        idx.search([float("nan"), 0, 0], 1, O.DOT)
    assert e.value.code == O.NAN_PANIC


def test_c_vs_python_restatement_random():
    rng = np.random.default_rng(7)
    for dim in (1, 2, 5, 33, 384):
        rows = rng.standard_normal((40, dim))
        rows[3] = rows[11]  # exact duplicate -> tie
        rows[5] = 0.0
        q = rng.standard_normal(dim)
        ids = (np.arange(40, dtype=np.uint64) * 2654435761) % (2 ** 40)
        idx = O.FlatOracle(dim, ids, rows)
        for name, m in M.items():
            ci, cs = idx.search(q, 12, m)
            pi, ps = O.py_flat_search(list(zip(ids.tolist(), rows.tolist())), q.tolist(), 12, m)
            assert ci.tolist() == pi, (dim, name)
            assert cs.tolist() == ps, (dim, name)


def test_sum_identity_sign():
    # `.sum::<f64>()` starts from -0.0: an all-(-0.0) dot product keeps its sign.
    s = O.calculate(O.DOT, [0.0, 0.0], [-1.0, -2.0])
    assert s == 0.0 and math.copysign(1.0, s) == -1.0
    assert math.copysign(1.0, O.calculate(O.DOT, [0.0], [1.0])) == 1.0


def test_embedding_postprocessing_restatement(kats):
    # src/embeddings.rs:169-181; the reference's own check is |norm - 1| < 1e-10 (src/embeddings.rs:374-383)
    assert O.embed_f32(np.array([3, 4], dtype=np.float32)).tolist() == [0.6, 0.8]
    assert O.embed_f32(np.array([3, 4], dtype=np.float32), normalize=False).tolist() == [3.0, 4.0]
    z = O.embed_f32(np.zeros(5, dtype=np.float32))
    assert z.tolist() == [0.0] * 5  # norm == 0: the row is left as it is
    rng = np.random.default_rng(3)
    for dim in (1, 7, 64, 384, 1000):
        e = rng.standard_normal((6, dim)).astype(np.float32)
        out = O.embed_f32(e)
        for r in range(6):
            w = [float(x) for x in e[r]]  # `x as f64`
            s = -0.0
            for x in w:
                s += x * x
            norm = math.sqrt(s)
            assert out[r].tolist() == [x / norm for x in w]
            chk = -0.0
            for x in out[r].tolist():
                chk += x * x
            for kat in kats["embedding_kats"]:  # src/embeddings.rs:374-383
                assert kat["property"] == "l2_norm_is_one" and abs(math.sqrt(chk) - 1.0) < kat["tol"]
