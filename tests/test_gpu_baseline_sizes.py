"""Parity at BASELINE.json's own sizes, inside the driver-run `-m gpu` suite.

  C2  flat cosine, N = 1 000 000, dim = 384, k = 10: against the CPU oracle (16 cosine queries + 4 per other metric)
  C5  batched flat as a bf16 MFMA GEMM, Q = 4 096 queries, dim = 384 (N = 250 000 here; N = 10 M below): every row
      against single search(), 64 sampled rows against the oracle -- crosses the 256-query chunk boundary
      (blockIdx.y > 0) and the 1 024-query host pass boundary of the MFMA filter
  C3  one rank's shard of the row-sharded config: 1 250 000 x 768, Euclidean, 1 024 queries in one batch: against
      single search() on 32 sampled queries
  headline  N = 10 000 000 x 384: the f32 fast path against the exact f64 path (property; the oracle would need
      minutes per query), and a 300-query MFMA batch against single search().

Bar everywhere: ids ==, f64 scores == (bit-exact), far inside north_star's 1e-5."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    x = rng.standard_normal((n, dim))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def ids_for(start, n):
    return np.arange(start, start + n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(97)


def device_index(V, n, dim, seed=1234, chunk=500_000):
    """bench.py's generator: rows drawn and normalised on the device in f64, ingested device to device."""
    import torch
    idx = V.FlatIndex(dim)
    idx.reserve(n)
    done = ci = 0
    while done < n:
        c = min(chunk, n - done)
        g = torch.Generator(device="cuda:0")
        g.manual_seed(seed + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(ids_for(done, c), x, validate=False)
        done += c
        ci += 1
        del x
    torch.cuda.synchronize()
    return idx


def test_config2_1m_x_384_against_the_oracle(V, O):
    rng = np.random.default_rng(1234)
    n, dim, k = 1_000_000, 384, 10
    rows = unit_rows(rng, n, dim)
    rows[777_777] = rows[5]          # an exact duplicate far away: the position tie-break at full size
    ids = ids_for(0, n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    ref = O.FlatOracle(dim, ids, rows)
    Q = unit_rows(np.random.default_rng(4321), 28, dim)
    Q[0] = rows[5]
    plan = [(0, qi) for qi in range(16)] + [(m, 16 + 4 * (m - 1) + j) for m in (1, 2, 3) for j in range(4)]
    for metric, qi in plan:
        gi, gs = gpu.search_arrays(Q[qi], k, metric)
        assert V.last_path() == V.PATH_FAST, (metric, qi)
        ri, rs = ref.search(Q[qi], k, metric)
        assert gi.tolist() == ri.tolist(), (metric, qi)
        assert gs.tolist() == rs.tolist(), (metric, qi)
    # the duplicate pair comes back in insertion order with equal scores
    gi, gs = gpu.search_arrays(Q[0], 2, 0)
    assert gi.tolist() == [int(ids[5]), int(ids[777_777])] and gs[0] == gs[1]


def test_config5_shape_4096_queries_cross_chunk_and_pass_boundaries(V, O):
    rng = np.random.default_rng(55)
    n, dim, nq, k = 250_000, 384, 4096, 10
    rows = unit_rows(rng, n, dim)
    ids = ids_for(0, n)
    gpu = V.FlatIndex(dim)
    gpu.add_rows(ids, rows, validate=False)
    Q = unit_rows(rng, nq, dim)
    Q[300] = rows[123]               # a self-query in the second chunk, one in the last pass
    Q[4000] = rows[200_000]
    for metric in (0, 1, 3):         # the three metrics the MFMA filter serves
        bi, bs, bn = gpu.search_batch(Q, k, metric)
        assert bn.tolist() == [k] * nq
        for qi in range(nq):
            si, ss = gpu.search_arrays(Q[qi], k, metric)
            assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), (metric, qi)
    ref = O.FlatOracle(dim, ids, rows)
    bi, bs, bn = gpu.search_batch(Q, k, 0)
    sample = sorted(set(range(0, nq, 65)) | {255, 256, 257, 300, 1023, 1024, 1025, 4000, 4095})
    assert len(sample) >= 64
    for qi in sample:
        ri, rs = ref.search(Q[qi], k, 0)
        assert bi[qi].tolist() == ri.tolist() and bs[qi].tolist() == rs.tolist(), qi
    assert bi[300, 0] == ids[123] and bi[4000, 0] == ids[200_000]


def test_config3_shard_1p25m_x_768_euclidean_batch_1024(V):
    n, dim, nq, k = 1_250_000, 768, 1024, 10
    idx = device_index(V, n, dim, seed=99, chunk=250_000)
    assert len(idx) == n
    Q = unit_rows(np.random.default_rng(4321), nq, dim)
    bp, bi, bs, bn = idx.search_batch_positions(Q, k, 1)
    assert bn.tolist() == [k] * nq
    assert all(bs[q, j - 1] >= bs[q, j] for q in range(0, nq, 37) for j in range(1, k))
    for qi in range(0, nq, 32):
        sp, si, ss = idx.search_positions(Q[qi], k, 1)
        assert V.last_path() == V.PATH_FAST
        assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist() and bp[qi].tolist() == sp.tolist(), qi
    # and the exact f64 pipeline agrees on a few (fast == exact at the shard's size)
    for qi in (0, 511, 1023):
        idx.force_path(V.PATH_EXACT_SELECT)
        try:
            ei, es = idx.search_arrays(Q[qi], k, 1)
        finally:
            idx.force_path(0)
        assert bi[qi].tolist() == ei.tolist() and bs[qi].tolist() == es.tolist(), qi


def test_headline_10m_x_384_fast_equals_exact_and_batch_equals_single(V):
    n, dim, k = 10_000_000, 384, 10
    idx = device_index(V, n, dim)          # exactly bench.py's corpus
    assert len(idx) == n
    Q = unit_rows(np.random.default_rng(9876), 300, dim)
    for metric, qs in ((0, (0, 1, 2, 3)), (1, (4,)), (2, (5,)), (3, (6,))):
        for qi in qs:
            fi, fs = idx.search_arrays(Q[qi], k, metric)
            assert V.last_path() == V.PATH_FAST
            idx.force_path(V.PATH_EXACT_SELECT)
            try:
                ei, es = idx.search_arrays(Q[qi], k, metric)
                assert V.last_path() == V.PATH_EXACT_SELECT
            finally:
                idx.force_path(0)
            assert fi.tolist() == ei.tolist() and fs.tolist() == es.tolist(), (metric, qi)
            assert all(fs[j - 1] >= fs[j] for j in range(1, k))
    # C5's N: a 300-query cosine batch (two MFMA query chunks) against single searches
    bi, bs, bn = idx.search_batch(Q, k, 0)
    assert bn.tolist() == [k] * 300
    for qi in list(range(0, 300, 23)) + [255, 256, 299]:
        si, ss = idx.search_arrays(Q[qi], k, 0)
        assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), qi
    # C5's FULL product: 4096 queries x 10 M rows, cosine -- two launch sequences of 2048 queries on the MFMA filter;
    # 64 sampled rows (both sequences, their first / last queries and the chunk edges) against single searches
    Q5 = unit_rows(np.random.default_rng(5555), 4096, dim)
    idx.search_batch(Q5[:256], k, 0)
    idx.profile_read()
    idx.profile_enable(True)
    bi5, bs5, bn5 = idx.search_batch(Q5, k, 0)
    idx.profile_enable(False)
    passes5 = idx.profile_read()[0]
    assert passes5 <= 2 + 8, passes5          # two sequences (+ a few queries redone one by one at most)
    assert bn5.tolist() == [k] * 4096
    sample = sorted(set([0, 1, 127, 128, 2047, 2048, 2049, 2175, 2176, 4094, 4095] + list(range(13, 4096, 77))))[:64]
    assert any(q < 2048 for q in sample) and any(q >= 2048 for q in sample)
    for qi in sample:
        si, ss = idx.search_arrays(Q5[qi], k, 0)
        assert bi5[qi].tolist() == si.tolist() and bs5[qi].tolist() == ss.tolist(), qi
    # a stored row finds itself first with score 1 (to rounding), its id is the bijection of its position
    probe = 7_654_321
    v = idx.get_vector(int(ids_for(probe, 1)[0])).values
    r = idx.search(v, 1, 0)
    assert r[0].id == int(ids_for(probe, 1)[0]) and abs(r[0].score - 1.0) < 1e-12


def test_dim768_batch_of_2048_runs_as_two_sequences_on_the_mfma_filter(V):
    """At stride 768 a launch sequence takes 1024 queries (16 chunks of 64: the sampling pass needs >= 128 groups);
    2048 queries must therefore be TWO filter sequences -- not one with degenerate thresholds and 2048 fallbacks."""
    n, dim, nq, k = 300_000, 768, 2048, 10
    idx = device_index(V, n, dim, seed=5, chunk=100_000)
    Q = unit_rows(np.random.default_rng(77), nq, dim)
    idx.search_batch(Q[:64], k, 1)
    idx.profile_read()
    idx.profile_enable(True)
    bi, bs, bn = idx.search_batch(Q, k, 1)
    idx.profile_enable(False)
    passes = idx.profile_read()[0]
    assert passes <= 2 + 8, passes          # two sequences (+ a few queries redone one by one at most)
    assert bn.tolist() == [k] * nq
    for qi in list(range(0, nq, 97)) + [1023, 1024, 2047]:
        si, ss = idx.search_arrays(Q[qi], k, 1)
        assert bi[qi].tolist() == si.tolist() and bs[qi].tolist() == ss.tolist(), qi
