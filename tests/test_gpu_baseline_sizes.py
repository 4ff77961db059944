"""Parity at BASELINE.json's own sizes, inside the driver-run `-m gpu` suite.

  C2  flat cosine, N = 1 000 000, dim = 384, k = 10: against the CPU oracle (16 cosine queries + 4 per other metric)
  C5  batched flat as a bf16 MFMA GEMM, Q = 4 096 queries, dim = 384 (N = 250 000 here; N = 10 M below): every row
      against single search(), 64 sampled rows against the oracle -- crosses the 256-query chunk boundary
      (blockIdx.y > 0) and the 1 024-query host pass boundary of the MFMA filter
  C3  one rank's shard of the row-sharded config: 1 250 000 x 768, Euclidean, 1 024 queries in one batch: against
      single search() on 32 sampled queries
  C3 at its own size (round 4): 10 000 000 x 768 as 8 contiguous row shards on ONE card (107 GB), the 1 024-query
      Euclidean batch through vl_shard_search_local_dev x 8 + vl_shard_merge on the 8 records -- the calls an 8-rank run
      makes, minus the wire -- against ONE unsharded index of the same 10 M rows (ids, global positions, f64 scores ==),
      with ties planted across shard boundaries; the same pipeline at 8 x 100 000 rows against the oracle on the union
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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 3 at its own size: N = 10 M, dim = 768, 1024-query Euclidean batch, 8 row shards (BASELINE.json configs[2];
# semantics: src/index/flat.rs:98-119 on the union of the shards)
# ---------------------------------------------------------------------------------------------------------------------
def _device_rows(n, dim, seed, chunk, plant=None, first_row=0):
    """bench.py's generator, chunk by chunk: yields (global start row, f64 device tensor).  plant: {global row: vector}."""
    import torch
    done = ci = 0
    while done < n:
        c = min(chunk, n - done)
        g = torch.Generator(device="cuda:0")
        g.manual_seed(seed + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        for row, vec in (plant or {}).items():
            if first_row + done <= row < first_row + done + c:
                x[row - first_row - done] = torch.from_numpy(np.ascontiguousarray(vec)).to("cuda:0")
        yield first_row + done, x
        done += c
        ci += 1


def _build_unsharded_and_shards(V, n, dim, world, seed, chunk, plant):
    """(factory for the unsharded index, factory for the list of shard handles): the SAME rows in the same order -- chunk
    boundaries coincide with shard boundaries, so both forms ingest identical tensors."""
    import torch
    per = n // world
    assert per * world == n and per % chunk == 0

    def unsharded():
        idx = V.FlatIndex(dim)
        idx.reserve(n)
        for r in range(world):
            for start, x in _device_rows(per, dim, seed + 1000 * r, chunk, plant, first_row=r * per):
                idx.add_rows(ids_for(start, x.shape[0]), x, validate=False)
                del x
        torch.cuda.synchronize()
        return idx

    def shards():
        out = []
        for r in range(world):
            s = V.FlatIndex(dim)
            s.reserve(per)
            for start, x in _device_rows(per, dim, seed + 1000 * r, chunk, plant, first_row=r * per):
                s.add_rows(ids_for(start, x.shape[0]), x, validate=False)
                del x
            out.append(s)
        torch.cuda.synchronize()
        return out

    return unsharded, shards


def _tie_plan(rng, n, world, dim):
    """One vector stored in four places that straddle shard boundaries (last row of shard 0, first row of shard 1, a row in
    the middle of shard 5, the very last row) and a second one in shards 2 and 3: the reference's stable sort ranks equal
    scores by storage position (src/index/flat.rs:116), i.e. by shard, then by position inside the shard."""
    per = n // world
    t1, t2 = unit_rows(rng, 2, dim)
    return {per - 1: t1, per: t1, 5 * per + per // 2: t1, n - 1: t1, 3 * per - 1: t2, 3 * per: t2}, t1, t2


def test_config3_pipeline_at_8_x_100k_against_the_oracle_on_the_union(V, O):
    import torch
    from vectorlite_amd.sharded import OneProcessShards
    n, dim, nq, k, world = 800_000, 768, 1024, 10, 8
    per = n // world
    plant, t1, t2 = _tie_plan(np.random.default_rng(31), n, world, dim)
    _, make_shards = _build_unsharded_and_shards(V, n, dim, world, seed=700, chunk=per, plant=plant)
    shards = make_shards()
    sh = OneProcessShards(shards)
    assert sh.total == n and sh.offsets == [r * per for r in range(world)]
    Q = unit_rows(np.random.default_rng(4321), nq, dim)
    Q[7], Q[500] = t1, t2                     # the queries that hit the planted ties (Euclidean score 1.0 on every copy)
    dQ = torch.from_numpy(Q).to("cuda:0")
    ids, scores, cnt, gpos = sh.search_batch(dQ, k, 1, with_positions=True)      # device queries: vl_shard_search_local_dev
    assert cnt.tolist() == [k] * nq
    hi, hs, hn, hp = sh.search_batch(Q, k, 1, with_positions=True)               # host queries: the same records
    assert hi.tolist() == ids.tolist() and hs.tolist() == scores.tolist() and hp.tolist() == gpos.tolist()
    # ties across shard boundaries come back in GLOBAL storage order
    assert gpos[7, :4].tolist() == [per - 1, per, 5 * per + per // 2, n - 1] and scores[7, 0] == scores[7, 3] == 1.0
    assert gpos[500, :2].tolist() == [3 * per - 1, 3 * per] and scores[500, 0] == scores[500, 1] == 1.0
    assert ids[7, :4].tolist() == [int(ids_for(p, 1)[0]) for p in (per - 1, per, 5 * per + per // 2, n - 1)]
    # the oracle on the union (rows copied back from the shards: exactly what they hold), 16 sampled queries + the two tie queries
    rows = np.empty((n, dim))
    for r, s in enumerate(shards):
        _, vals = s.export()
        rows[r * per:(r + 1) * per] = vals
    ref = O.FlatOracle(dim, ids_for(0, n), rows)
    for qi in [7, 500] + list(range(0, nq, 73)):
        ri, rs = ref.search(Q[qi], k, 1)
        assert ids[qi].tolist() == ri.tolist() and scores[qi].tolist() == rs.tolist(), qi


def test_config3_10m_x_768_as_8_row_shards_on_one_card_equals_one_unsharded_index(V):
    import gc
    import torch
    from vectorlite_amd.sharded import OneProcessShards
    n, dim, nq, k, world = 10_000_000, 768, 1024, 10, 8
    per = n // world
    plant, t1, t2 = _tie_plan(np.random.default_rng(32), n, world, dim)
    make_unsharded, make_shards = _build_unsharded_and_shards(V, n, dim, world, seed=900, chunk=250_000, plant=plant)
    Q = unit_rows(np.random.default_rng(4321), nq, dim)
    Q[7], Q[500] = t1, t2
    dQ = torch.from_numpy(Q).to("cuda:0")
    # (1) ONE index holding all 10 M rows: the answer the shards must reproduce
    one = make_unsharded()
    assert len(one) == n
    wp, wi, ws, wn = one.search_batch_device(dQ, k, 1, with_positions=True)
    assert wn.tolist() == [k] * nq
    for qi in (0, 7, 500, 1023):              # ... which is itself the single-search answer
        sp, si, ss = one.search_positions(Q[qi], k, 1)
        assert wi[qi].tolist() == si.tolist() and ws[qi].tolist() == ss.tolist() and wp[qi].tolist() == sp.tolist(), qi
    want = (wi.copy(), ws.copy(), wp.copy())
    del one
    gc.collect()
    torch.cuda.empty_cache()
    # (2) the same rows as 8 shards of 1.25 M on the same card, through the two halves of vl_shard_search_batch
    shards = make_shards()
    sh = OneProcessShards(shards)
    assert sh.total == n and sh.max_len == per
    t = {}
    ids, scores, cnt, gpos = sh.search_batch(dQ, k, 1, with_positions=True, timings=t)
    assert cnt.tolist() == [k] * nq
    assert ids.tolist() == want[0].tolist()
    assert scores.tolist() == want[1].tolist()          # f64, bit for bit
    assert gpos.tolist() == want[2].tolist()            # global position = shard offset + local position
    assert gpos[7, :4].tolist() == [per - 1, per, 5 * per + per // 2, n - 1]
    assert gpos[500, :2].tolist() == [3 * per - 1, 3 * per]
    ids2, scores2, cnt2 = sh.search_batch(dQ, k, 1, timings=t)   # a second, warm pass: the timing worth printing
    assert ids2.tolist() == ids.tolist() and scores2.tolist() == scores.tolist()
    print("config 3 at full size on one card: per-shard ms", [round(x, 2) for x in t["local_ms"]], "merge ms", round(t["merge_ms"], 3))
