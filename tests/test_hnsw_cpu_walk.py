"""The checker's CPU HNSW walk (oracle/vl_hnsw_cpu.c) on hand-made graphs: no GPU needed.
It is what the GPU walk is compared with at N = 1 M (tests/test_gpu_hnsw_1m.py)."""
import numpy as np


def _graph(rows, nbr0, levels=None, upper=None, entry=0):
    n = len(rows)
    m0 = max(len(x) for x in nbr0)
    g = {"n": n, "entry": entry, "m0": m0, "m": 4, "rows": np.asarray(rows, np.float64)}
    g["cnt0"] = np.array([len(x) for x in nbr0], np.uint32)
    g["nbr0"] = np.zeros((n, m0), np.uint32)
    for i, x in enumerate(nbr0):
        g["nbr0"][i, :len(x)] = x
    g["level"] = np.array(levels if levels is not None else [0] * n, np.uint8)
    g["max_level"] = int(g["level"][entry])
    offs, lists = [], []
    for i in range(n):
        offs.append(len(lists))
        for layer in range(1, int(g["level"][i]) + 1):
            lists.append(upper[i][layer - 1])
    g["upper_off"] = np.array(offs, np.uint32)
    g["cntU"] = np.array([len(x) for x in lists] or [0], np.uint32)
    g["nbrU"] = np.zeros((max(len(lists), 1), 4), np.uint32)
    for i, x in enumerate(lists):
        g["nbrU"][i, :len(x)] = x
    return g


def test_walk_on_a_line_graph_reaches_the_far_end_and_orders_by_u64_distance():
    from oracle import oracle as O
    O.build()
    rows = [[float(i), 0.0] for i in range(30)]
    nbr0 = [[j for j in (i - 1, i + 1) if 0 <= j < 30] for i in range(30)]
    w = O.HnswCpuWalker(_graph(rows, nbr0), O.EUCLIDEAN)
    nodes, d = w.search([27.25, 0.0], ef=4, k=3)
    assert nodes.tolist() == [27, 28, 26] and d.tolist() == [250, 750, 1250]   # trunc(distance * 1000), src/index/hnsw.rs:121
    assert [int(x) for x in d] == [O.hnsw_distance(O.EUCLIDEAN, [27.25, 0.0], rows[i]) for i in nodes]
    nodes, d = w.search([27.2, 0.0], ef=4, k=1)
    assert nodes.tolist() == [27] and d.tolist() == [199]                       # sqrt(0.04) * 1000 = 199.99.. truncates
    # a beam of 1 is a greedy walk: it still arrives (the line has no local minima)
    nodes, d = w.search([27.2, 0.0], ef=1, k=1)
    assert nodes.tolist() == [27]
    # k larger than the beam returns the beam
    nodes, _ = w.search([0.1, 0.0], ef=2, k=10)
    assert nodes.tolist() == [0, 1]


def test_upper_layers_are_descended_greedily_and_ties_keep_first_seen_order():
    from oracle import oracle as O
    # 8 points on a circle; node 0 and 4 also live on layer 1 and see each other there
    ang = np.arange(8) * np.pi / 4
    rows = np.stack([np.cos(ang), np.sin(ang)], 1)
    nbr0 = [[(i - 1) % 8, (i + 1) % 8] for i in range(8)]
    levels = [1, 0, 0, 0, 1, 0, 0, 0]
    upper = {0: [[4]], 4: [[0]]}
    w = O.HnswCpuWalker(_graph(rows, nbr0, levels, upper, entry=0), O.COSINE)
    q = rows[4] * 3.0
    nodes, d = w.search(q, ef=3, k=3)
    assert nodes[0] == 4 and d[0] == 0
    assert sorted(nodes[1:].tolist()) == [3, 5] and d[1] == d[2]      # symmetric neighbours tie in u64
    assert nodes[1:].tolist() == [3, 5]                                # ... and the first seen stays in front
    assert w.evals.value >= 4
