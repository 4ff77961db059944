"""vl_index_search_batch_dev: queries already in device memory.  Every row must equal what the host form
(vl_index_search_batch / _positions) returns for the same queries -- ids, positions, scores bit for bit -- whichever path
served it: the kernel-staged MFMA batch path, the host path for what that path does not take, the exact path for queries
outside the fast-path domain, and the oracle on a sample."""
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
def torch():
    import torch
    return torch


def rows_and_queries(seed, n, dim, nq):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = rng.standard_normal((nq, dim))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return x, q


def both(V, torch, idx, Q, k, metric):
    dq = torch.from_numpy(np.ascontiguousarray(Q)).to("cuda:0")
    got = idx.search_batch_device(dq, k, metric, with_positions=True)
    paths = V.last_path()
    ref = idx.search_batch_positions(Q, k, metric)
    for name, a, b in zip(("positions", "ids", "scores", "n"), got, ref):
        assert a.tolist() == b.tolist(), (name, metric, paths)
    ids3 = idx.search_batch_device(dq, k, metric)
    assert ids3[0].tolist() == ref[1].tolist() and ids3[1].tolist() == ref[2].tolist()
    return got


@pytest.mark.parametrize("metric,dim,n,nq", [("cosine", 384, 20000, 37), ("euclidean", 768, 12000, 130),
                                               ("dotproduct", 100, 9000, 5), ("cosine", 384, 30000, 300)])
def test_device_queries_equal_host_queries_on_the_mfma_batch_path(V, torch, metric, dim, n, nq):
    x, Q = rows_and_queries(11 + dim + nq, n, dim, nq)
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(n, dtype=np.uint64) * 3 + 1, x)
    got = both(V, torch, idx, Q, 10, M[metric])
    assert (got[3] == 10).all()
    # it was the MFMA filter that streamed the slab for the device queries (profile counters count its launch sequences)
    idx.profile_read()
    idx.profile_enable(True)
    idx.search_batch_device(torch.from_numpy(Q).to("cuda:0"), 10, M[metric])
    idx.profile_enable(False)
    n_seq, _, n_bytes = idx.profile_read()
    assert n_seq >= 1 and n_bytes >= n * dim * 2  # bf16 bytes of one pass at least
    # and the answer is the reference's: oracle on a few queries
    from oracle import oracle as O
    O.build()
    ref = O.FlatOracle(dim, np.arange(n, dtype=np.uint64) * 3 + 1, x)
    for qi in (0, nq - 1):
        ri, rs = ref.search(Q[qi], 10, M[metric])
        assert got[1][qi].tolist() == ri.tolist() and got[2][qi].tolist() == rs.tolist()


def test_device_queries_outside_the_fast_path_domain(V, torch):
    """zero, huge, NaN-free-but-tiny queries in the batch: staged as zeros by the kernel, answered on the exact path."""
    dim, n = 384, 16000
    x, Q = rows_and_queries(5, n, dim, 24)
    Q[3] = 0.0                      # zero query: cosine 0.0 for every row (src/lib.rs:439-440), insertion order
    Q[7] *= 1e15                    # |v| > 2^40
    Q[11] *= 1e-14                  # norm < 2^-40
    Q[12, 5] = 3e300                # overflows the squared norm
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(n, dtype=np.uint64), x)
    for metric in ("cosine", "euclidean", "dotproduct"):
        both(V, torch, idx, Q, 10, M[metric])


def test_device_queries_on_paths_the_filter_does_not_take(V, torch):
    dim = 96
    x, Q = rows_and_queries(9, 12000, dim, 9)
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(12000, dtype=np.uint64), x)
    both(V, torch, idx, Q, 10, M["manhattan"])          # no MFMA form
    both(V, torch, idx, Q[:1], 10, M["cosine"])         # one query
    both(V, torch, idx, Q, 200, M["cosine"])            # k beyond the fast paths
    small = V.FlatIndex(dim)
    small.add_rows(np.arange(50, dtype=np.uint64), x[:50])
    both(V, torch, small, Q, 10, M["cosine"])           # small index
    both(V, torch, small, Q, 1000, M["euclidean"])      # k > len: len results
    empty = V.FlatIndex(dim)
    got = empty.search_batch_device(torch.from_numpy(Q).to("cuda:0"), 10, 0)
    assert got[2].tolist() == [0] * 9


def test_device_queries_with_ties_the_filter_cannot_certify(V, torch):
    """every row five times: the bound check cannot separate equal scores, those queries are redone exactly."""
    dim = 128
    x, Q = rows_and_queries(21, 2000, dim, 16)
    x = np.repeat(x, 5, axis=0)
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(x.shape[0], dtype=np.uint64), x)
    got = both(V, torch, idx, Q, 10, M["cosine"])
    assert got[0][0, 0] + 1 == got[0][0, 1]  # equal scores in insertion order


def test_device_queries_errors_match_the_host_form(V, torch):
    dim = 64
    x, Q = rows_and_queries(2, 9000, dim, 4)
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(9000, dtype=np.uint64), x)
    wrong = torch.zeros((4, dim + 1), dtype=torch.float64, device="cuda:0")
    with pytest.raises(V.DimensionMismatch):
        idx.search_batch_device(wrong, 10, 0)
    with pytest.raises(ValueError):
        idx.search_batch_device(torch.zeros((4, dim), dtype=torch.float32, device="cuda:0"), 10, 0)
    with pytest.raises(ValueError):
        idx.search_batch_device(torch.zeros((4, dim), dtype=torch.float64), 10, 0)  # host tensor


def test_device_queries_on_an_hnsw_handle_go_through_the_host(V, torch):
    import ctypes as C
    dim, n = 48, 3000
    x, Q = rows_and_queries(4, n, dim, 6)
    h = V.HNSWIndex(dim, M["euclidean"])
    h.add_rows(np.arange(n, dtype=np.uint64), x)
    ref = h.search_batch(Q, 10, M["euclidean"])
    dq = torch.from_numpy(Q).to("cuda:0")
    ids = np.zeros((6, 10), dtype=np.uint64)
    sc = np.zeros((6, 10), dtype=np.float64)
    nn = np.zeros(6, dtype=np.uint64)
    L = h._L
    rc = L.vl_index_search_batch_dev(h._h, C.c_void_p(dq.data_ptr()), 6, dim, 10, M["euclidean"], None,
                                     ids.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(C.POINTER(C.c_double)),
                                     nn.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0
    assert ids.tolist() == ref[0].tolist() and sc.tolist() == ref[1].tolist() and nn.tolist() == ref[2].tolist()
    # a wrong query length is the reference's DimensionMismatch, reported before any query is read
    rc = L.vl_index_search_batch_dev(h._h, C.c_void_p(dq.data_ptr()), 6, dim + 3, 10, M["euclidean"], None,
                                     ids.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(C.POINTER(C.c_double)),
                                     nn.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 1  # VL_ERR_DIM_MISMATCH


@pytest.mark.parametrize("normalize", [True, False])
def test_embedding_queries_equal_host_postprocessing_then_search(V, torch, normalize):
    """vl_index_search_batch_embeddings_f32: f32 embeddings (host array or device tensor) -> widened and normalised on the
    device with the arithmetic of src/embeddings.rs:169-181 -> searched.  Every row equals search_batch on the oracle's
    host post-processing of the same embeddings (MFMA batch path and small-index path; HNSW: the next test)."""
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(31)
    dim, n, nq = 384, 15000, 65
    rows = O.embed_f32(rng.standard_normal((n, dim)).astype(np.float32), True)
    idx = V.FlatIndex(dim)
    idx.add_rows(np.arange(n, dtype=np.uint64), rows)
    emb = (rng.standard_normal((nq, dim)) * 3.0).astype(np.float32)
    emb[5] = 0.0  # a zero embedding stays zero (norm == 0: left unchanged, :176)
    Q = O.embed_f32(emb, normalize)
    for metric in (0, 1, 3):
        ref = idx.search_batch(Q, 10, metric)
        for e in (emb, torch.from_numpy(emb).to("cuda:0"), torch.from_numpy(emb)):
            got = idx.search_batch_embeddings(e, 10, metric, normalize=normalize)
            assert got[0].tolist() == ref[0].tolist() and got[1].tolist() == ref[1].tolist() and got[2].tolist() == ref[2].tolist()
    small = V.FlatIndex(dim)
    small.add_rows(np.arange(40, dtype=np.uint64), rows[:40])
    ref = small.search_batch(Q, 10, 0)
    got = small.search_batch_embeddings(torch.from_numpy(emb).to("cuda:0"), 10, 0, normalize=normalize)
    assert got[0].tolist() == ref[0].tolist() and got[1].tolist() == ref[1].tolist()
    with pytest.raises(V.DimensionMismatch):
        idx.search_batch_embeddings(emb[:, :100], 10, 0)
    empty = V.FlatIndex(dim)
    assert empty.search_batch_embeddings(emb[:, :100], 10, 0)[2].tolist() == [0] * nq  # an empty flat index accepts any length


def test_embedding_queries_on_an_hnsw_handle(V, torch):
    import ctypes as C
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(32)
    dim, n, nq = 64, 4000, 7
    rows = O.embed_f32(rng.standard_normal((n, dim)).astype(np.float32), True)
    h = V.HNSWIndex(dim, M["cosine"])
    h.add_rows(np.arange(n, dtype=np.uint64), rows)
    emb = rng.standard_normal((nq, dim)).astype(np.float32)
    ref = h.search_batch(O.embed_f32(emb, True), 10, M["cosine"])
    demb = torch.from_numpy(emb).to("cuda:0")
    for ptr, on_dev in ((C.c_void_p(emb.ctypes.data), 0), (C.c_void_p(demb.data_ptr()), 1)):
        ids = np.zeros((nq, 10), dtype=np.uint64)
        sc = np.zeros((nq, 10), dtype=np.float64)
        nn = np.zeros(nq, dtype=np.uint64)
        rc = h._L.vl_index_search_batch_embeddings_f32(h._h, ptr, nq, dim, 1, on_dev, 10, M["cosine"],
                                                       ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       sc.ctypes.data_as(C.POINTER(C.c_double)),
                                                       nn.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0
        assert ids.tolist() == ref[0].tolist() and sc.tolist() == ref[1].tolist() and nn.tolist() == ref[2].tolist()
    rc = h._L.vl_index_search_batch_embeddings_f32(h._h, C.c_void_p(emb.ctypes.data), nq, dim - 1, 1, 0, 10, M["cosine"],
                                                   ids.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(C.POINTER(C.c_double)),
                                                   nn.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 1  # VL_ERR_DIM_MISMATCH
