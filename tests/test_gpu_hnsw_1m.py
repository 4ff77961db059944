"""BASELINE config 4 at its own size inside the `-m gpu` suite: HNSW, default profile (M = 16, M0 = 32), cosine,
N = 1 000 000, dim = 384.  The graph is built on the GPU (3-4 s), then
  * recall@10 of the GPU walk against the exact order of the reference's own u64 distances (ties at the 10th accepted):
    strict reference rule ef = min(k, len) = 10, the default beam floor (32), and BASELINE's ef = 128;
  * the "CPU HNSW": oracle/vl_hnsw_cpu.c walks the SAME exported graph with the reference's f64 -> u64 callbacks,
    single-threaded, timed beside the GPU walk -- a second recall reference (the GPU walk navigates by f32 distances);
  * every returned score is the reference's conversion of the exact u64 distance of the returned row.
The crate's own walk (hnsw 0.11.0) is not in the reference tree: parity unpinned, recall is the yardstick."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config4_hnsw_1m_x_384_recall_against_exact_and_cpu_walk():
    import torch
    import vectorlite_amd as V
    from oracle import oracle as O
    O.build()
    n, dim, latent, k = 1_000_000, 384, 16, 10
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    A = torch.randn((latent, dim), dtype=torch.float64, device=dev, generator=g)
    flat = V.FlatIndex(dim)
    flat.reserve(n)
    hn = V.HNSWIndex(dim, V.SimilarityMetric.Cosine)
    t_build = 0.0
    for c0 in range(0, n, 250_000):
        x = torch.randn((250_000, latent), dtype=torch.float64, device=dev, generator=g) @ A
        x += 0.05 * torch.randn((250_000, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        ids = np.arange(c0, c0 + 250_000, dtype=np.uint64)
        flat.add_rows(ids, x, validate=False)
        t0 = time.perf_counter()
        hn.add_rows(ids, x)
        t_build += time.perf_counter() - t0
    assert len(hn) == n
    rng = np.random.default_rng(4321)
    nq, n_cpu = 200, 24
    Q = rng.standard_normal((nq, latent)) @ A.cpu().numpy() + 0.05 * rng.standard_normal((nq, dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    allpos = np.arange(n, dtype=np.uint64)
    n_truth = 60
    D = [flat.hnsw_distances(Q[i], allpos, 0) for i in range(n_truth)]   # the reference's u64 distance of every row
    kth = [np.partition(d, k - 1)[k - 1] for d in D]

    def recall(ids_rows):
        return float(np.mean([sum(1 for x in ids_rows[i] if D[i][int(x)] <= kth[i]) / float(k) for i in range(n_truth)]))

    out = {}
    for name, beam, ef in (("strict_ef10", 0, 0), ("floor32_optin", 32, 0), ("ef128", 32, 128)):
        hn.set_min_beam(beam)
        hn.search_batch(Q[:8], k, 0, ef=ef)
        t0 = time.perf_counter()
        bi, bs, bn = hn.search_batch(Q, k, 0, ef=ef)
        dt = time.perf_counter() - t0
        assert bn.tolist() == [k] * nq
        out[name] = (recall(bi), nq / dt)
        # scores = convert_distance_to_similarity(exact u64 distance of the returned row) (src/index/hnsw.rs:478-479)
        for i in range(0, n_truth, 7):
            assert bs[i].tolist() == [V.hnsw_score(int(D[i][int(x)]), 0) for x in bi[i]]
            assert all(bs[i][j - 1] >= bs[i][j] for j in range(1, k))
    hn.set_min_beam(0)
    # ---- the CPU HNSW on the same graph ----
    graph = hn.graph(with_rows=True)
    assert graph["n"] == n and graph["m0"] == 32 and graph["m"] == 16 and int(graph["cnt0"].max()) <= 32
    walker = O.HnswCpuWalker(graph, O.COSINE)
    t0 = time.perf_counter()
    cpu = [walker.search(Q[i], 128, k) for i in range(n_cpu)]
    t_cpu = (time.perf_counter() - t0) / n_cpu
    cpu_recall = float(np.mean([sum(1 for x in cpu[i][0] if D[i][int(x)] <= kth[i]) / float(k) for i in range(n_cpu)]))
    for i in range(n_cpu):  # the walker's distances are the reference's callback values of the nodes it returns
        assert cpu[i][1].tolist() == [int(D[i][int(x)]) for x in cpu[i][0]]
    t0 = time.perf_counter()
    for i in range(n_cpu):
        hn.search_arrays(Q[i], k, 0, ef=128)
    t_gpu_single = (time.perf_counter() - t0) / n_cpu
    gi, _, _ = hn.search_batch(Q[:n_cpu], k, 0, ef=128)
    gpu_recall_same = float(np.mean([sum(1 for x in gi[i] if D[i][int(x)] <= kth[i]) / float(k) for i in range(n_cpu)]))
    print(f"\\n[config 4] build {t_build:.1f}s; recall@10 / batch QPS: " +
          ", ".join(f"{k2} {v[0]:.3f} / {v[1]:.0f}" for k2, v in out.items()) +
          f"; CPU walk (1 thread, ef 128): recall {cpu_recall:.3f}, {t_cpu * 1e3:.2f} ms/query, "
          f"{walker.evals.value / n_cpu:.0f} evals/query; GPU walk same queries: recall {gpu_recall_same:.3f}, "
          f"{t_gpu_single * 1e3:.3f} ms/query one at a time")
    assert t_build < 30.0
    assert out["ef128"][0] >= 0.99, out
    assert out["floor32_optin"][0] >= 0.93, out
    assert out["strict_ef10"][0] >= 0.72, out
    assert out["floor32_optin"][0] >= out["strict_ef10"][0]
    assert cpu_recall >= 0.98, cpu_recall
    assert abs(cpu_recall - gpu_recall_same) <= 0.03, (cpu_recall, gpu_recall_same)
