"""N > 1 path on CPU: world_size-2 and -3 gloo runs of the row-sharded search (collective + merge),
checked against ONE oracle index holding all rows.  No GPU needed."""
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


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_search_matches_single_index(tmp_path, world):
    from oracle import oracle as O
    rng = np.random.default_rng(77 + world)
    n, dim, nq = 1501, 24, 9
    rows = rng.standard_normal((n, dim))
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    # duplicates that straddle shard boundaries: the global-position tie-break must hold across ranks
    rows[1200] = rows[10]
    rows[760] = rows[10]
    rows[3] = rows[1400]
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(5)) % np.uint64(2 ** 40)
    Q = rng.standard_normal((nq, dim))
    Q[0] = rows[10]
    Q[1] = rows[1400]
    np.savez(tmp_path / "data.npz", rows=rows, ids=ids, Q=Q)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-3000:]
    ref = O.FlatOracle(dim, ids, rows)
    outs = [np.load(tmp_path / f"out_rank{r}.npz") for r in range(world)]
    for m in range(4):
        for k in (1, 10, 50):
            for qi in range(nq):
                ri, rs = ref.search(Q[qi], k, m)
                for r in range(world):  # every rank holds the same merged answer
                    o = outs[r]
                    c = int(o[f"n_{m}_{k}"][qi])
                    assert c == len(ri)
                    assert o[f"ids_{m}_{k}"][qi, :c].tolist() == ri.tolist(), (m, k, qi, r)
                    assert o[f"scores_{m}_{k}"][qi, :c].tolist() == rs.tolist(), (m, k, qi, r)


def test_merge_orders_ties_by_global_position():
    from vectorlite_amd.sharded import merge_shard_results, shard_ranges
    assert shard_ranges(10, 3) == [0, 4, 7, 10]
    scores = np.array([[0.9, 0.5, 0.5], [0.9, 0.5, 0.1]])
    gpos = np.array([[7, 2, 9], [3, 1, 4]], dtype=np.int64)
    ids = np.array([[70, 20, 90], [30, 10, 40]], dtype=np.uint64)
    i, s, p = merge_shard_results(scores, gpos, ids, np.array([3, 2]), 4)
    assert p.tolist() == [3, 7, 1, 2] and i.tolist() == [30, 70, 10, 20] and s.tolist() == [0.9, 0.9, 0.5, 0.5]
