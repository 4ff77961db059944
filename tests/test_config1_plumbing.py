"""BASELINE config 1: flat index, cosine, N = 1 000, dim = 384, k = 10 on the CPU path -- the plumbing gate
(corpus seed 1234, query seed 4321, i.i.d. N(0,1) rows L2-normalised in f64 like src/embeddings.rs:173-179,
ids a bijection of the position, plus the adversarial add-ons: 1 % exact duplicates, one all-zero row,
k in {0, 1, 10, N + 1}, a query of the wrong length, an empty index).
CPU half: the C restatement against the independent pure-Python restatement, and against numpy in f64 to 1e-12.
GPU half (`-m gpu`): the HIP path through the C ABI equals the oracle on the same inputs, ids and scores ==."""
import numpy as np
import pytest

M = {"cosine": 0, "euclidean": 1, "manhattan": 2, "dotproduct": 3}


def config1():
    n, dim = 1000, 384
    rows = np.random.Generator(np.random.PCG64(1234)).standard_normal((n, dim))
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    for j in range(10):                       # 1 % exact duplicates: the earlier row must win every tie
        rows[900 + j] = rows[37 + 11 * j]
    rows[500] = 0.0                           # zero norm: cosine is 0.0 by definition (src/lib.rs:439-440)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(97)
    Q = np.random.Generator(np.random.PCG64(4321)).standard_normal((40, dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q[3] = rows[37]                           # a query that IS a duplicated row
    return rows, ids, Q


def test_config1_c_restatement_equals_python_restatement_and_numpy():
    from oracle import oracle as O
    O.build()
    rows, ids, Q = config1()
    n, dim = rows.shape
    ref = O.FlatOracle(dim, ids, rows)
    pairs = list(zip(ids.tolist(), rows.tolist()))
    for qi in range(6):
        for name, m in M.items():
            ci, cs = ref.search(Q[qi], 10, m)
            pi, ps = O.py_flat_search(pairs, Q[qi].tolist(), 10, m)
            assert ci.tolist() == pi and cs.tolist() == ps, (qi, name)
    # cosine against numpy in f64: same ranking up to exact ties, scores to 1e-12 (different summation order)
    for qi in range(40):
        ci, cs = ref.search(Q[qi], 10, 0)
        nrm = np.linalg.norm(rows, axis=1)
        cos = np.where(nrm > 0, rows @ Q[qi] / np.maximum(nrm, 1e-300) / np.linalg.norm(Q[qi]), 0.0)
        order = np.lexsort((np.arange(n), -cos))[:10]
        assert np.allclose(cs, cos[order], rtol=0, atol=1e-12)
        assert all(abs(cos[int(np.nonzero(ids == i)[0][0])] - s) < 1e-12 for i, s in zip(ci, cs))
    ci, cs = ref.search(Q[3], 2, 0)
    assert ci.tolist() == [int(ids[37]), int(ids[900])] and cs[0] == cs[1]      # insertion order on the exact tie
    # k in {0, 1, N + 1}
    assert len(ref.search(Q[0], 0, 0)[0]) == 0 and len(ref.search(Q[0], 1, 0)[0]) == 1
    fi, fs = ref.search(Q[0], n + 1, 0)
    assert len(fi) == n and all(fs[i - 1] >= fs[i] for i in range(1, n)) and sorted(fi.tolist()) == sorted(ids.tolist())
    zero_at = fi.tolist().index(int(ids[500]))
    assert fs[zero_at] == 0.0
    # wrong length -> DimensionMismatch{expected, actual}; empty index accepts anything (src/index/flat.rs:99-104)
    with pytest.raises(O.OracleError) as e:
        ref.search(Q[0][:-1], 10, 0)
    assert e.value.code == O.DIM_MISMATCH
    assert len(O.FlatOracle(dim).search(Q[0][:5], 10, 0)[0]) == 0


@pytest.mark.gpu
def test_config1_gpu_path_equals_the_oracle():
    import vectorlite_amd as V
    from oracle import oracle as O
    O.build()
    rows, ids, Q = config1()
    n, dim = rows.shape
    ref = O.FlatOracle(dim, ids, rows)
    gpu = V.FlatIndex(dim, [V.Vector(int(i), r) for i, r in zip(ids, rows)])      # FlatIndex::new(dim, data)
    for qi in range(40):
        for m in range(4):
            for k in (1, 10):
                gi, gs = gpu.search_arrays(Q[qi], k, m)
                ri, rs = ref.search(Q[qi], k, m)
                assert gi.tolist() == ri.tolist() and gs.tolist() == rs.tolist(), (qi, m, k)
    for k in (0, n + 1):
        gi, gs = gpu.search_arrays(Q[1], k, 0)
        ri, rs = ref.search(Q[1], k, 0)
        assert gi.tolist() == ri.tolist() and gs.tolist() == rs.tolist()
    bi, bs, bn = gpu.search_batch(Q, 10, 0)
    for qi in range(40):
        ri, rs = ref.search(Q[qi], 10, 0)
        assert bi[qi].tolist() == ri.tolist() and bs[qi].tolist() == rs.tolist()
    with pytest.raises(V.DimensionMismatch) as e:
        gpu.search(Q[0][:-1], 10, 0)
    assert (e.value.expected, e.value.actual) == (dim, dim - 1)
    assert V.FlatIndex(dim).search(Q[0][:5], 10, 0) == []
