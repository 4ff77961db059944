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
