"""Row-sharded batched search through the C ABI on the GPU:
  * the device merge kernel (vl_shard_merge) on hand-made exchange records: ties across shards, short and empty
    shards, a failing shard's status;
  * world 2 and 3 with REAL GPU shards (ranks share the one card, records travel by gloo, merge on the device);
  * world 1 over RCCL: vl_comm_create + vl_shard_sync + vl_shard_search_batch (ncclAllGather inside the library);
  * config 3's shard shape (d = 768, Euclidean, 1024 queries) through vl_shard_search_batch.
Every answer is compared with ONE index / ONE oracle holding all rows: ids, f64 scores and global positions ==."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _pu64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _pf64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _merge(records, nq, ks, k):
    import vectorlite_amd as V
    from vectorlite_amd import _lib
    L = _lib.load()
    world = len(records)
    g = np.ascontiguousarray(np.stack(records).astype(np.uint64))
    kk = max(k, 1)
    gpos, ids = np.zeros((nq, kk), np.uint64), np.zeros((nq, kk), np.uint64)
    scores, n = np.zeros((nq, kk), np.float64), np.zeros(nq, np.uint64)
    rc = L.vl_shard_merge(0, _pu64(g), world, nq, ks, k, _pu64(gpos), _pu64(ids), _pf64(scores), _pu64(n))
    return rc, gpos, ids, scores, n


def _record(nq, ks, lists, status=0):
    """lists[q] = [(score, gpos, id), ...] already in (score desc, gpos asc) order."""
    from vectorlite_amd.sharded import pack_words, SHARD_HDR_WORDS
    rec = np.zeros(pack_words(nq, ks), np.uint64)
    rec[0] = status
    body = rec[SHARD_HDR_WORDS + nq:].reshape(3, nq, ks)
    for q, lst in enumerate(lists):
        rec[SHARD_HDR_WORDS + q] = len(lst)
        for j, (s, p, i) in enumerate(lst):
            body[0, q, j] = np.float64(s).view(np.uint64)
            body[1, q, j], body[2, q, j] = p, i
    return rec


def test_merge_kernel_orders_ties_by_global_position_and_handles_short_shards():
    nq, ks = 3, 4
    a = _record(nq, ks, [[(0.9, 7, 70), (0.5, 2, 20), (0.5, 9, 90)], [], [(1.0, 1, 11), (-0.0, 2, 12)]])
    b = _record(nq, ks, [[(0.9, 3, 30), (0.5, 1, 10), (0.1, 4, 40)], [(0.25, 50, 500)], [(0.0, 0, 10), (-1.0, 9, 19)]])
    c = _record(nq, ks, [[], [], []])
    rc, gpos, ids, scores, n = _merge([a, b, c], nq, ks, 4)
    assert rc == 0 and n.tolist() == [4, 1, 4]
    assert gpos[0].tolist() == [3, 7, 1, 2] and ids[0].tolist() == [30, 70, 10, 20]
    assert scores[0].tolist() == [0.9, 0.9, 0.5, 0.5]
    assert (gpos[1, 0], ids[1, 0], scores[1, 0]) == (50, 500, 0.25)
    # -0.0 and +0.0 tie like partial_cmp says: position decides (0 before 2)
    assert gpos[2].tolist() == [1, 0, 2, 9]
    assert np.signbit(scores[2, 2]) and not np.signbit(scores[2, 1])
    # k larger than everything on offer: all 6 entries of query 0, in order
    rc, gpos, ids, scores, n = _merge([a, b, c], nq, ks, 12)
    assert rc == 0 and n.tolist() == [6, 1, 4]
    assert gpos[0, :6].tolist() == [3, 7, 1, 2, 9, 4]
    # k = 1
    rc, gpos, ids, scores, n = _merge([a, b, c], nq, ks, 1)
    assert rc == 0 and n.tolist() == [1, 1, 1] and gpos[:, 0].tolist() == [3, 50, 1]


def test_merge_kernel_reports_the_first_failing_shard():
    import vectorlite_amd as V
    nq, ks = 2, 2
    ok = _record(nq, ks, [[(0.5, 1, 1)], [(0.5, 1, 1)]])
    bad = _record(nq, ks, [[], []], status=V.VL_ERR_NAN_SCORE)
    worse = _record(nq, ks, [[], []], status=V.VL_ERR_DEVICE)
    rc, *_ = _merge([ok, bad, worse], nq, ks, 2)
    assert rc == V.VL_ERR_NAN_SCORE
    assert b"rank 1" in V._lib.load().vl_last_error()


def test_merge_kernel_random_lists_match_numpy(seed=3):
    rng = np.random.default_rng(seed)
    for world, nq, ks, k in [(8, 37, 10, 10), (2, 5, 64, 100), (5, 3, 300, 700), (64, 2, 3, 50)]:
        recs, per_q = [], [[] for _ in range(nq)]
        gp = 0
        for r in range(world):
            lists = []
            for q in range(nq):
                c = int(rng.integers(0, ks + 1))
                s = np.round(rng.standard_normal(c), 1)  # coarse: plenty of exact ties
                p = np.sort(rng.choice(10_000, size=c, replace=False)) + r * 10_000
                order = np.lexsort((p, -s))
                lst = [(float(s[o]), int(p[o]), int(p[o]) * 3 + 1) for o in order]
                lists.append(lst)
                per_q[q].extend(lst)
            recs.append(_record(nq, ks, lists))
        rc, gpos, ids, scores, n = _merge(recs, nq, ks, k)
        assert rc == 0
        for q in range(nq):
            want = sorted(per_q[q], key=lambda t: (-t[0], t[1]))[:k]
            m = int(n[q])
            assert m == len(want)
            assert gpos[q, :m].tolist() == [w[1] for w in want], (world, q)
            assert ids[q, :m].tolist() == [w[2] for w in want]
            assert scores[q, :m].tolist() == [w[0] for w in want]


@pytest.mark.parametrize("world", [2, 3])
def test_real_gpu_shards_gloo_exchange_device_merge(tmp_path, world):
    from test_sharded_gloo import check_against_single_oracle, run_ranks, sharded_case
    rows, ids, Q = sharded_case(900 + world, n=30_011, dim=96, nq=33)
    ks = [1, 10, 70]
    np.savez(tmp_path / "data.npz", rows=rows, ids=ids, Q=Q, ks=np.array(ks))
    outs = run_ranks(tmp_path, world, "gpu", timeout=600)
    check_against_single_oracle(outs, rows, ids, Q, ks)


def test_uneven_and_empty_gpu_shards(tmp_path):
    from test_sharded_gloo import check_against_single_oracle, run_ranks, sharded_case
    rows, ids, Q = sharded_case(31, n=300, dim=16, nq=6)
    ks = [1, 7, 100, 1000]
    np.savez(tmp_path / "data.npz", rows=rows, ids=ids, Q=Q, ks=np.array(ks), starts=np.array([0, 0, 4, 300]))
    outs = run_ranks(tmp_path, 3, "gpu", timeout=600)
    check_against_single_oracle(outs, rows, ids, Q, ks)


def test_world1_over_rccl(tmp_path):
    """The library's own ncclAllGather path (communicator from a ncclUniqueId), in a fresh process."""
    from test_sharded_gloo import check_against_single_oracle, run_ranks, sharded_case
    rows, ids, Q = sharded_case(4242, n=20_000, dim=64, nq=17)
    ks = [1, 10, 200]
    np.savez(tmp_path / "data.npz", rows=rows, ids=ids, Q=Q, ks=np.array(ks))
    outs = run_ranks(tmp_path, 1, "rccl", timeout=600)
    check_against_single_oracle(outs, rows, ids, Q, ks)


def test_config3_shard_shape_through_vl_shard_search_batch():
    """Config 3 as one rank sees it (d = 768, Euclidean, 1024 queries, k = 10) over RCCL at world 1:
    every row of the sharded answer equals the plain batch search and, on a sample, single search()."""
    import vectorlite_amd as V
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex
    rng = np.random.default_rng(8)
    n, dim, nq, k = 60_000, 768, 1024, 10
    rows = rng.standard_normal((n, dim))
    ids = np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(3)
    shard = V.FlatIndex(dim, device=0)
    shard.add_rows(ids, rows, validate=False)
    Q = rng.standard_normal((nq, dim))
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        sh = ShardedFlatIndex(shard, comm=comm)
        assert (sh.offset, sh.total) == (0, n)
        i, s, cnt, p = sh.search_batch(Q, k, V.SimilarityMetric.Euclidean, with_positions=True)
        bp, bi, bs, bn = shard.search_batch_positions(Q, k, V.SimilarityMetric.Euclidean)
        assert cnt.tolist() == bn.tolist() == [k] * nq
        assert i.tolist() == bi.tolist() and s.tolist() == bs.tolist() and p.tolist() == bp.tolist()
        for qi in range(0, nq, 64):
            si, ss = shard.search_arrays(Q[qi], k, V.SimilarityMetric.Euclidean)
            assert si.tolist() == i[qi].tolist() and ss.tolist() == s[qi].tolist()
        # the same batch already in GPU memory (vl_shard_search_batch_dev; vl_shard_search_local_dev + vl_shard_merge)
        import torch
        dQ = torch.from_numpy(Q).to("cuda:0")
        di, ds, dcnt, dp = sh.search_batch(dQ, k, V.SimilarityMetric.Euclidean, with_positions=True)
        assert di.tolist() == i.tolist() and ds.tolist() == s.tolist() and dp.tolist() == p.tolist() and dcnt.tolist() == cnt.tolist()
        sh_t = ShardedFlatIndex(shard, transport="torch")  # world 1 without torch.distributed: the record goes straight to the merge
        ti, ts, tcnt, tp = sh_t.search_batch(dQ, k, V.SimilarityMetric.Euclidean, with_positions=True)
        assert ti.tolist() == i.tolist() and ts.tolist() == s.tolist() and tp.tolist() == p.tolist()
        # a mutation without a re-sync is reported, not silently merged
        shard.add(V.Vector(id=10 ** 12, values=rows[0].tolist()))
        with pytest.raises(V.VectorLiteError):
            sh.search_batch(Q[:4], k, 1)
        sh.sync()
        assert sh.total == n + 1
        i2, s2, n2 = sh.search_batch(Q[:4], k, 1)
        assert n2.tolist() == [k] * 4
        # k = 0 and k > N
        i3, s3, n3 = sh.search_batch(Q[:2], 0, 1)
        assert n3.tolist() == [0, 0]
    finally:
        comm.close()


@pytest.mark.parametrize("metric", [0, 1, 3])
def test_device_written_exchange_record_over_several_launch_sequences_and_with_patched_queries(metric):
    """The finalize kernel writes this rank's exchange record in device memory (ShardRecordSink): a batch longer than one
    launch sequence (2500 queries at d = 128: 2048 + 452), with queries whose cut falls inside a block of 70 identical rows
    (the bf16 filter cannot certify those: they are re-answered exactly and patched into the record).  The answer
    must equal the plain batch search row for row, and the record must have been written on the device."""
    import vectorlite_amd as V
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex
    rng = np.random.default_rng(100 + metric)
    n, dim, nq, k = 20_000, 128, 2500, 10
    rows = rng.standard_normal((n, dim))
    twin = rng.standard_normal(dim)
    where = rng.choice(n, 70, replace=False)
    rows[where] = twin
    ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(11)
    shard = V.FlatIndex(dim, device=0)
    shard.add_rows(ids, rows, validate=False)
    Q = rng.standard_normal((nq, dim))
    planted = [0, 7, 2047, 2048, 2049, 2499]          # both sides of the sequence boundary, first and last query
    Q[planted] = twin + 1e-3 * rng.standard_normal((len(planted), dim))
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        sh = ShardedFlatIndex(shard, comm=comm)
        before = comm.record_paths()
        i, s, cnt, p = sh.search_batch(Q, k, metric, with_positions=True)
        after = comm.record_paths()
        assert (after["on_device"] - before["on_device"], after["via_host"] - before["via_host"]) == (1, 0)
        bp, bi, bs, bn = shard.search_batch_positions(Q, k, metric)
        assert cnt.tolist() == bn.tolist() == [k] * nq
        assert i.tolist() == bi.tolist() and s.tolist() == bs.tolist() and p.tolist() == bp.tolist()
        for qi in planted + [1, 1000, 2046]:
            si, ss = shard.search_arrays(Q[qi], k, metric)
            assert si.tolist() == i[qi].tolist() and ss.tolist() == s[qi].tolist()
        for qi in planted:                              # ties inside the block: insertion order decides
            assert p[qi].tolist() == np.sort(where)[:k].tolist() and len(set(s[qi].tolist())) == 1
    finally:
        comm.close()


def test_a_nan_score_on_one_shard_is_the_whole_calls_panic():
    """FlatIndex::search panics on a NaN score (partial_cmp().unwrap(), src/index/flat.rs:116) -> VL_ERR_NAN_SCORE; in the
    sharded form the failing shard's status travels inside the exchange and every rank reports it."""
    import vectorlite_amd as V
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex
    rng = np.random.default_rng(2)
    rows = rng.standard_normal((500, 8))
    rows[77, 3] = np.inf                      # inf - inf in the Euclidean difference of a query with +inf there
    shard = V.FlatIndex(8, device=0)
    shard.add_rows(np.arange(500, dtype=np.uint64), rows, validate=False)
    q = rng.standard_normal((3, 8))
    q[1, 3] = np.inf
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        sh = ShardedFlatIndex(shard, comm=comm)
        with pytest.raises(V.NaNScore):
            sh.search_batch(q, 5, V.SimilarityMetric.Euclidean)
        with pytest.raises(V.NaNScore):   # the plain batch entry says the same
            shard.search_batch(q, 5, V.SimilarityMetric.Euclidean)
        ids, scores, n = sh.search_batch(q[[0, 2]], 5, V.SimilarityMetric.Euclidean)   # the other queries are fine
        assert n.tolist() == [5, 5]
    finally:
        comm.close()


def test_a_rank_that_cannot_size_its_exchange_buffers_fails_the_call_on_every_rank_and_the_comm_survives():
    """Advisor round 2: a rank-local failure before the big all-gather must not leave the peers waiting in it.  Buffer growth
    is decided by (nq, ks, world) alone, so all ranks take a pre-flight status all-gather together; a rank whose allocation
    fails (injected here) says so there, every rank returns VL_ERR_OOM, and the communicator stays usable."""
    import vectorlite_amd as V
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex
    rng = np.random.default_rng(12)
    rows = rng.standard_normal((4000, 32))
    shard = V.FlatIndex(32, device=0)
    shard.add_rows(np.arange(4000, dtype=np.uint64), rows, validate=False)
    Q = rng.standard_normal((9, 32))
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        sh = ShardedFlatIndex(shard, comm=comm)
        os.environ["VL_SHARD_INJECT_OOM"] = "0"
        try:
            with pytest.raises(V.VectorLiteError) as ei:
                sh.search_batch(Q, 10, V.SimilarityMetric.Cosine)
            assert "could not size its exchange buffers" in str(ei.value), str(ei.value)
        finally:
            del os.environ["VL_SHARD_INJECT_OOM"]
        i, s, n = sh.search_batch(Q, 10, V.SimilarityMetric.Cosine)      # the communicator was not aborted
        bi, bs, bn = shard.search_batch(Q, 10, V.SimilarityMetric.Cosine)
        assert i.tolist() == bi.tolist() and s.tolist() == bs.tolist() and n.tolist() == bn.tolist()
    finally:
        comm.close()
