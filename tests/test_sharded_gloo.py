"""N > 1 path on CPU: world_size-2 and -3 gloo runs of the row-sharded search harness
(vectorlite_amd/sharded.py: sync, offsets, exchange record, the one all-gather, k clamping), checked against
ONE oracle index holding all rows.  No GPU: the two C-ABI halves are stood in for by the test itself
(tests/_sharded_worker.py); the real ones run in tests/test_gpu_sharded.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def sharded_case(seed, n=1501, dim=24, nq=9):
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, dim))
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    # duplicates that straddle shard boundaries: the global-position tie-break must hold across ranks
    a, b, c = (n * 4) // 5, n // 2, (n * 14) // 15
    rows[a] = rows[10]
    rows[b] = rows[10]
    rows[3] = rows[c]
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(5)) % np.uint64(2 ** 40)
    Q = rng.standard_normal((nq, dim))
    Q[0] = rows[10]
    Q[1] = rows[c]
    return rows, ids, Q


def run_ranks(tmp_path, world, mode, timeout=300):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"), str(tmp_path), mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, out.decode()[-3000:]
    return [np.load(tmp_path / f"out_rank{r}.npz") for r in range(world)]


def check_against_single_oracle(outs, rows, ids, Q, ks):
    from oracle import oracle as O
    ref = O.FlatOracle(rows.shape[1], ids, rows)
    pos_of = {int(i): p for p, i in enumerate(ids)}
    for m in range(4):
        for k in ks:
            for qi in range(Q.shape[0]):
                ri, rs = ref.search(Q[qi], k, m)
                for r, o in enumerate(outs):  # every rank holds the same merged answer
                    c = int(o[f"n_{m}_{k}"][qi])
                    assert c == len(ri)
                    assert o[f"ids_{m}_{k}"][qi, :c].tolist() == ri.tolist(), (m, k, qi, r)
                    assert o[f"scores_{m}_{k}"][qi, :c].tolist() == rs.tolist(), (m, k, qi, r)
                    if len(pos_of) == len(ids):  # unique ids: the global positions are checkable too
                        assert o[f"gpos_{m}_{k}"][qi, :c].tolist() == [pos_of[int(i)] for i in ri], (m, k, qi, r)


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_search_matches_single_index(tmp_path, world):
    rows, ids, Q = sharded_case(77 + world)
    ks = [1, 10, 50, 5000]  # 5000 > N: every row comes back, in the reference's order
    np.savez(tmp_path / "data.npz", rows=rows, ids=ids, Q=Q, ks=np.array(ks))
    outs = run_ranks(tmp_path, world, "cpu")
    check_against_single_oracle(outs, rows, ids, Q, ks)


def test_uneven_and_empty_shards(tmp_path):
    """One rank holds nothing, one holds fewer rows than k."""
    rows, ids, Q = sharded_case(5, n=40, dim=8, nq=4)
    ks = [1, 7, 30, 100]
    np.savez(tmp_path / "data.npz", rows=rows[:40], ids=ids[:40], Q=Q, ks=np.array(ks), starts=np.array([0, 0, 3, 40]))
    outs = run_ranks(tmp_path, 3, "cpu")
    check_against_single_oracle(outs, rows[:40], ids[:40], Q, ks)


def test_record_layout_helpers():
    from vectorlite_amd.sharded import pack_words, shard_ranges, unpack_record
    assert shard_ranges(10, 3) == [0, 4, 7, 10]
    assert pack_words(3, 5) == 4 + 3 + 45
    rec = np.arange(pack_words(2, 3), dtype=np.uint64)
    st, ln, dim, cnt, sc, gp, idv = unpack_record(rec, 2, 3)
    assert (st, ln, dim) == (0, 1, 2) and cnt.tolist() == [4, 5]
    assert gp.tolist() == [[12, 13, 14], [15, 16, 17]] and idv.tolist() == [[18, 19, 20], [21, 22, 23]]
    assert sc.view(np.uint64).tolist() == [[6, 7, 8], [9, 10, 11]]
