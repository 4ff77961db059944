#!/usr/bin/env python3
"""bench.py -- flat-cosine QPS and achieved HBM GB/s of the MI355X distance-scan path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run with one rank per GPU.  ONE JSON line is printed on stdout (rank 0).

Workload (BASELINE.json `metric`): flat index, cosine, N = 10 000 000 rows, dim = 384, k = 10,
single-query searches.  A "step" is one `FlatIndex.search` = one pass of the hot path over the whole
f32 slab (15.36 GB algorithmic bytes) + exact f64 rescoring of the 64 candidates.  Rows are synthetic:
i.i.d. N(0,1) drawn on the device in f64 (seed 1234), L2-normalised in f64 (the reference's
embedder does the same, src/embeddings.rs:173-179), ids a fixed bijection of the position.
The corpus is resident in HBM before the timed region; queries come from the host as f64 like the
reference's `search(&[f64])`.

Process model
  * `python bench.py --gpus N ...` WITHOUT a launcher (WORLD_SIZE unset): this process is a SUPERVISOR.
    It never touches the GPU.  It starts the rank processes as children (N = 1: one worker; N > 1:
    `python -m torch.distributed.run --nproc-per-node N ... bench.py --worker`), waits, and prints the
    one JSON line the rank-0 worker left in a result file.  The worker rewrites that file right after
    the timed region and again after every later block, so a failure in a later block (or a dead
    worker) cannot discard a finished measurement: it is reported under "errors" instead.
  * under torch.distributed.run (WORLD_SIZE set) or with `--inline`: the process is a rank itself and
    rank 0 prints the line (`--inline` is what rocprofv3 runs: no child process under the profiler).

What is timed: K steps, serial, one host thread per rank, bracketed by barrier + synchronize.  The
queries of the timed region, of the pre-warm and of every later block (CPU baseline, parity, extras)
come from independent generators: no block depends on --steps/--warmup (plan_queries()).

Multi-GPU (`--gpus N`): the flat index is REPLICATED (each rank holds the full corpus) and ranks
answer disjoint query streams -- no data-path collective; value = all ranks' queries / max time
(weak scaling).

After the published timed region (never inside `value`) the same run measures the other BASELINE.json
configurations, each with its own `roofline` object under `config.other_configs`:
  N = 1   c2 (1 M x 384 single query, HBM), c5 (4096-query batch on the resident corpus, MFMA), c3_shard (one
          rank's 1.25 M x 768 shard of config 3 through vl_shard_search_batch over a world-1 RCCL communicator,
          HIP events around the all-gather and the merge), c4_hnsw (build, recall@10, QPS at ef 10 / 32 / 128
          on embedding-like AND on i.i.d. gaussian rows), plus `value_sustained` (the headline search kept up
          for 10 s) and one reference-faithful CPU query at full size (`cpu_baseline.full_size_check`).
  N > 1   c3_row_sharded: config 3 at its own size -- 10 M x 768 rows cut into N contiguous shards, 1024
          Euclidean queries per batch, ONE ncclAllGather per batch inside the library, device merge.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s copy-achievable)
METRIC_NAME = "flat-cosine QPS @ N=10M dim=384 k=10; achieved HBM GB/s vs peak"

PREWARM_MIN_QUERIES = 10   # untimed, before --warmup: clocks, TLBs and the pinned result block are hot
PREWARM_MIN_SECONDS = 0.25
CHECK_QUERIES_MIN = 16     # the check/extras query set never has fewer rows than this


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--dim", type=int, default=384)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--metric", default="cosine", choices=["cosine", "euclidean", "manhattan", "dotproduct"])
    p.add_argument("--chunk", type=int, default=500_000, help="rows generated per device chunk")
    p.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    p.add_argument("--cpu-queries", type=int, default=32)
    p.add_argument("--cpu-budget-s", type=float, default=30.0, help="stop the CPU baseline after this much CPU time")
    p.add_argument("--no-cpu-baseline", action="store_true")
    # the other BASELINE.json configurations, run after the published timed region (config.other_configs)
    p.add_argument("--sustained-s", type=float, default=10.0, help="value_sustained: keep searching this long (>= 4000 queries at full size)")
    p.add_argument("--c2-rows", type=int, default=1_000_000)
    p.add_argument("--c5-queries", type=int, default=4096)
    p.add_argument("--c3-rows", type=int, default=10_000_000, help="config 3's corpus; N = 1 times ONE shard of c3-shards")
    p.add_argument("--c3-shards", type=int, default=8)
    p.add_argument("--c3-dim", type=int, default=768)
    p.add_argument("--c3-batch", type=int, default=1024)
    p.add_argument("--no-c4-sweep", action="store_true", help="skip config 4's ef_construction sweep (two more builds per embedding-like distribution)")
    p.add_argument("--no-c3-full", action="store_true", help="skip config 3 at its own size on one card (8 shards of the 10 M x 768 corpus)")
    p.add_argument("--c4-rows", type=int, default=1_000_000)
    p.add_argument("--c4-queries", type=int, default=1000)
    p.add_argument("--block-cap-s", type=float, default=60.0, help="soft cap per later block: steps still to run are skipped")
    p.add_argument("--no-other-configs", action="store_true")
    p.add_argument("--no-full-size-cpu-check", action="store_true")
    p.add_argument("--no-checks", action="store_true")
    p.add_argument("--inline", action="store_true", help="be the (single) rank in this process: no supervisor")
    p.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--result-file", default=None, help=argparse.SUPPRESS)
    p.add_argument("--deadline-s", type=float, default=1500.0, help="supervisor: give the ranks this long")
    a = p.parse_args(argv)
    if a.gpus < 1 or a.steps < 1 or a.warmup < 0 or a.rows < 1 or a.dim < 1 or a.cpu_queries < 1:
        p.error("--gpus/--steps/--rows/--dim/--cpu-queries must be >= 1 and --warmup >= 0")
    return a


def plan_queries(steps: int, warmup: int, cpu_queries: int) -> dict:
    """Which query sets a run needs and how long each is.  Pure (unit-tested on the CPU): the timed
    set is the only one whose length depends on --steps/--warmup; every later block indexes `check`
    (modulo its length), so no (steps, warmup) pair can run a block out of queries."""
    n_check = max(int(cpu_queries), CHECK_QUERIES_MIN)
    return {
        "timed": {"seed": 4321, "n": int(warmup) + int(steps), "warmup": int(warmup), "steps": int(steps)},
        "prewarm": {"seed": 555, "n": PREWARM_MIN_QUERIES},
        "check": {"seed": 9876, "n": n_check},
        "cpu": {"n": min(int(cpu_queries), n_check)},   # first rows of `check`
        "bf16": {"n": min(16, n_check)},                # first rows of `check`
        "exact": {"n": min(4, n_check)},                # first rows of `check`
    }


def unit_queries(seed: int, n: int, dim: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    q = rng.standard_normal((max(n, 1), dim))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q[:n]


def ids_for(start: int, n: int) -> np.ndarray:
    # fixed bijection of the position (odd multiplier mod 2^64): ids are NOT positions, so the
    # insertion-order tie-break is exercised on position, as in the reference
    pos = np.arange(start, start + n, dtype=np.uint64)
    return pos * np.uint64(2654435761) + np.uint64(97)


def host_cores() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (cpu.max)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


# =====================================================================================================
# supervisor: starts the ranks, never initialises the GPU
# =====================================================================================================
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker_argv(argv) -> list:
    """The supervisor's own arguments minus the ones that only it reads."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("--result-file", "--deadline-s"):
            skip = True
            continue
        if a.startswith("--result-file=") or a.startswith("--deadline-s=") or a in ("--worker", "--inline"):
            continue
        out.append(a)
    return out


def launch_command(args, argv, result_file: str, port: int) -> list:
    me = os.path.abspath(__file__)
    tail = worker_argv(argv) + ["--worker", "--result-file", result_file]
    if args.gpus == 1:
        return [sys.executable, me] + tail
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), me] + tail


def read_result(path: str):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def supervise(args, argv) -> int:
    fd, result_file = tempfile.mkstemp(prefix="vl_bench_", suffix=".json")
    os.close(fd)
    os.unlink(result_file)
    cmd = launch_command(args, argv, result_file, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    log(f"[bench] supervisor: starting {args.gpus} rank(s): {' '.join(cmd)}")
    # own session: if the deadline passes, exactly this process group is ended, nothing matched by name
    child = subprocess.Popen(cmd, stdout=sys.stderr, stderr=sys.stderr, env=env, start_new_session=True)
    note = None
    try:
        rc = child.wait(timeout=args.deadline_s)
    except subprocess.TimeoutExpired:
        note = f"ranks still running after --deadline-s {args.deadline_s:g}: ended"
        log(f"[bench] supervisor: {note}")
        try:
            os.killpg(child.pid, 15)
            rc = child.wait(timeout=20)
        except Exception:
            try:
                os.killpg(child.pid, 9)
            except Exception:
                pass
            rc = child.wait()
    out = read_result(result_file)
    try:
        os.unlink(result_file)
    except OSError:
        pass
    if out is None:
        log(f"[bench] supervisor: the ranks left no result (exit code {rc}): nothing was measured")
        return rc if rc else 1
    if rc != 0 or note:
        out.setdefault("errors", []).append(
            {"block": "worker_exit", "error": note or f"rank process exit code {rc} after the timed region"})
    print(json.dumps(out), flush=True)
    return 0


# =====================================================================================================
# rank process
# =====================================================================================================
def fair_cpu_baseline(rows64, queries, k, n_full, exact_ids, ids):
    """SURVEY 8(d) mode (ii): contiguous f32 slab + cached norms + OpenMP over every host core
    (oracle/vl_fair.c, built here with -march=native).  Reported beside cpu_baseline, never instead of it."""
    import ctypes as C
    so = os.path.join(tempfile.mkdtemp(prefix="vl_fair_"), "libvl_fair.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "oracle", "vl_fair.c"), "-lm"])
    L = C.CDLL(so)
    fp, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.vlf_search_cosine.restype = C.c_size_t
    L.vlf_search_cosine.argtypes = [fp, fp, C.c_size_t, C.c_size_t, fp, C.c_size_t, u32p, fp]
    L.vlf_inv_norms.argtypes = [fp, C.c_size_t, C.c_size_t, fp]
    L.vlf_threads.restype = C.c_int
    slab = np.ascontiguousarray(rows64, dtype=np.float32)
    ns, dim = slab.shape
    inv = np.empty(ns, np.float32)
    L.vlf_inv_norms(slab.ctypes.data_as(fp), ns, dim, inv.ctypes.data_as(fp))
    q32 = np.ascontiguousarray(queries, dtype=np.float32)
    nq = len(q32)
    pos, sc = np.empty(k, np.uint32), np.empty(k, np.float32)

    def one(i):
        g = L.vlf_search_cosine(slab.ctypes.data_as(fp), inv.ctypes.data_as(fp), ns, dim, q32[i].ctypes.data_as(fp), k,
                                pos.ctypes.data_as(u32p), sc.ctypes.data_as(fp))
        return ids[pos[:g]].tolist()
    # thread count: the CPU share may be smaller than the visible core count; keep the fastest
    best = (None, 0)
    nc = host_cores()
    for nt in sorted({nc, max(1, nc // 2), max(1, nc // 4), min(nc, 16)}):
        L.vlf_set_threads(nt)
        one(0)  # threads up, pages touched
        t = time.perf_counter()
        one(0)
        one(1 % nq)
        t = time.perf_counter() - t
        if best[0] is None or t < best[0]:
            best = (t, nt)
    L.vlf_set_threads(best[1])
    one(0)
    reps = 4
    t = time.perf_counter()
    got = [one(i) for _ in range(reps) for i in range(nq)][-nq:]
    t = time.perf_counter() - t
    qps_s = reps * nq / t
    hits = sum(len(set(a) & set(b.tolist())) for a, b in zip(got, exact_ids))
    return {
        "value": round(qps_s * ns / n_full, 4), "unit": "queries/s", "cores": int(L.vlf_threads()), "kind": "fair-cpu",
        "scan_GBps": round(ns * dim * 4 * qps_s / 1e9, 1),
        "recall_at_k_vs_oracle": round(hits / float(nq * min(k, ns)), 6),
        "sample": (f"{reps}x{nq} cosine queries on the first {ns} rows as a contiguous f32 slab with cached norms, "
                   f"per-thread top-k, OpenMP on {int(L.vlf_threads())} threads (oracle/vl_fair.c, gcc -O3 -march=native); "
                   f"{t:.1f}s wall; {qps_s:.2f} q/s at N={ns}, scaled x{ns / n_full:g}; f32 scores, not the reference's f64"),
    }


def device_copy_ceiling(torch, dev, nbytes=4 << 30, reps=10):
    """Measured device-to-device copy rate (read + write bytes / time): the practical HBM ceiling
    SURVEY 8(d) asks to be reported beside the 8 TB/s vendor peak."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return round(2 * nbytes / (ms * 1e-3) / 1e9, 1)


def step_rates(stamps, t_start: float) -> dict:
    """QPS of the first five timed steps and of the rest (is a short run a cold-start measurement?)."""
    n = len(stamps)
    if n == 0:
        return {}
    head = min(5, n)
    out = {"value_first_5_steps": round(head / (stamps[head - 1] - t_start), 3)}
    if n > head:
        out["value_after_first_5_steps"] = round((n - head) / (stamps[-1] - stamps[head - 1]), 3)
    return out



MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md (AMD's 2:1-sparsity headline figure is not used)


def gen_unit_rows(torch, dev, n, dim, seed, latent_basis=None, noise=0.05):
    """n unit rows on the device: i.i.d. N(0,1) (what src/embeddings.rs:173-179 leaves behind for
    uninformative text), or rows of low intrinsic dimension z A + noise (embedding-like) when a basis is given."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if latent_basis is None:
        x = torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    else:
        x = torch.randn((n, latent_basis.shape[0]), dtype=torch.float64, device=dev, generator=g) @ latent_basis
        x += noise * torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    return x


def digest64(*arrays) -> int:
    d = 0
    for a in arrays:
        d ^= int(np.bitwise_xor.reduce(np.ascontiguousarray(a).reshape(-1).view(np.int64)))
    return d


def filter_traffic(config: str, workload: dict, plan: dict):
    """(traffic bytes per batch, traffic / algorithmic) of the batch filter's pass-1 kernel from the committed PMC pass
    (profiles/traffic.json, rocprofv3 --pmc FETCH_SIZE with the gfx950 corrections) -- only when it was measured on exactly
    this workload AND this launch plan (vl_index_last_filter): a changed kernel or plan prints null, not stale bytes."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for e in json.load(f).get("k_mfma_rows", []):
                if e["config"] == config and e["workload"] == workload and e["plan"] == plan:
                    return e["traffic_bytes_per_batch"], e["traffic_over_algorithmic"]
    except Exception:
        pass
    return None, None


def hnsw_traffic(workload: dict, evals_per_query: float):
    """FETCH_SIZE bytes per batch of the HNSW walk kernel from the committed PMC pass (profiles/traffic.json), attached only when it
    was measured on exactly this workload and the walks did the same work (distance evaluations per query within 0.5 %: the
    graph build is deterministic, so they normally agree to the digit)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for e in json.load(f).get("k_hnsw_search", []):
                if e["workload"] == workload and abs(e["distance_evals_per_query"] - evals_per_query) <= 0.005 * evals_per_query:
                    return e["traffic_bytes_per_batch"], e["traffic_over_algorithmic"]
    except Exception:
        pass
    return None, None


def c3_rows_for(rank: int, world: int, total: int, shards_at_n1: int):
    """Config 3's row range of this rank.  world > 1: the corpus cut into `world` contiguous ranges.  world == 1: the
    FIRST of `shards_at_n1` ranges -- one GPU times one rank's shard of the 8-GPU configuration."""
    parts = world if world > 1 else max(1, shards_at_n1)
    base, rem = divmod(total, parts)
    starts = [0]
    for r in range(parts):
        starts.append(starts[-1] + base + (1 if r < rem else 0))
    r = rank if world > 1 else 0
    return starts[r], starts[r + 1], (total if world > 1 else starts[1])


def build_c3_shard(V, torch, dev, dev_index, dim, lo, hi):
    """Rows [lo, hi) of config 3's corpus as one flat handle: a function of the GLOBAL row range, so the same rows come out
    whichever rank (or one-card shard list) builds them."""
    shard = V.FlatIndex(dim, device=dev_index)
    shard.reserve(hi - lo)
    pos = lo
    while pos < hi:
        c = min(250_000, hi - pos)
        x = gen_unit_rows(torch, dev, c, dim, 424242 + pos)
        shard.add_rows(ids_for(pos, c), x, validate=False)
        pos += c
        del x
    return shard


def run_c3(V, torch, dist, args, dev, dev_index, rank, world, rehearse, k, cap_s, prebuilt=None):
    """BASELINE config 3 (flat L2, dim 768, 1024-query batches, rows sharded over the ranks): every rank answers the
    batch on its own rows with the single-GPU pipeline, ONE ncclAllGather inside libvectorlite_amd.so
    (vl_shard_search_batch) exchanges the per-shard top-k, a device kernel merges.  Collective: every rank calls it."""
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex
    t_block = time.perf_counter()
    dim, nq, metric = args.c3_dim, args.c3_batch, 1
    lo, hi, corpus = c3_rows_for(rank, world, args.c3_rows, args.c3_shards)
    shard = prebuilt if prebuilt is not None else build_c3_shard(V, torch, dev, dev_index, dim, lo, hi)
    Qs = unit_queries(2468, nq, dim)  # the same batch on every rank
    comm = None
    if rehearse:  # ranks share one card and RCCL refuses that: the records travel by gloo, the merge is the same kernel
        sh = ShardedFlatIndex(shard, transport="torch")
    elif world > 1:
        comm = Comm.from_torch_distributed(device=dev_index)
        sh = ShardedFlatIndex(shard, comm=comm)
    else:
        comm = Comm(Comm.unique_id(), 1, 0, dev_index)  # a world of one: the same RCCL calls, no peer
        sh = ShardedFlatIndex(shard, comm=comm)
    assert (sh.offset, sh.total) == (lo if world > 1 else 0, corpus), (sh.offset, sh.total, lo, corpus)
    sh.search_batch(Qs[: min(nq, 128)], k, metric)  # bf16 slab, scratch, exchange buffers
    sh.search_batch(Qs, k, metric)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    shard.profile_read()
    shard.profile_enable(True)
    if comm is not None:
        comm.profile_read()
        comm.profile_enable(True)
    steps = 0
    ts = time.perf_counter()
    for _ in range(5):
        si, ss, sn, sp = sh.search_batch(Qs, k, metric, with_positions=True)
        steps += 1
        if world == 1 and time.perf_counter() - t_block > cap_s:
            break
    torch.cuda.synchronize()
    dt = (time.perf_counter() - ts) / steps
    shard.profile_enable(False)
    n_pass, filt_ms, _ = shard.profile_read()
    plan3 = shard.last_filter()
    traffic3, ratio3 = filter_traffic("c3", {"rows": hi - lo, "dim": dim, "queries": nq, "metric": metric}, plan3)
    prof = comm.profile_read() if comm is not None else None
    if comm is not None:
        comm.profile_enable(False)
    # the same batch already in each rank's GPU memory (the embedding model ran there, or one rank broadcast it):
    # vl_shard_search_batch_dev -- no host staging, no 6.3 MB PCIe copy in front of the kernels
    dev_q = None
    if not rehearse:
        dQ = torch.from_numpy(np.ascontiguousarray(Qs)).to(dev)
        di, ds, dn_ = sh.search_batch(dQ, k, metric)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t_d = time.perf_counter()
        for _ in range(3):
            di, ds, dn_ = sh.search_batch(dQ, k, metric)
        torch.cuda.synchronize()
        dt_d = (time.perf_counter() - t_d) / 3
        if dist is not None:
            tmd = torch.tensor([dt_d], dtype=torch.float64, device=dev)
            dist.all_reduce(tmd, op=dist.ReduceOp.MAX)
            dt_d = float(tmd.item())
        dev_q = {"ms_per_batch": round(dt_d * 1e3, 3), "value": round(nq / dt_d, 1), "unit": "queries/s",
                 "identical_to_host_queries": bool(np.array_equal(di, si) and np.array_equal(ds, ss))}
        del dQ
    same, own_ok, n_own = True, 0, 4
    if dist is not None:
        dig = torch.tensor([digest64(si), digest64(ss)], dtype=torch.int64, device="cpu" if rehearse else dev)
        lo_d, hi_d = dig.clone(), dig.clone()
        dist.all_reduce(lo_d, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_d, op=dist.ReduceOp.MAX)
        same = bool((lo_d == hi_d).all().item())
        tm = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
    for qi in range(n_own):  # this rank's rows in the global answer are what its own single search() returns for them
        li, ls = shard.search_arrays(Qs[qi], k, metric)
        mine = [(int(i), float(sc)) for i, sc, pp in zip(si[qi], ss[qi], sp[qi]) if sh.offset <= int(pp) < sh.offset + (hi - lo)]
        own_ok += int(mine == list(zip(li.tolist(), ls.tolist()))[: len(mine)])
    nccl_ranks_seen = comm.world if comm is not None else None  # vl_comm_world: what RCCL's communicator itself reports
    rec_paths = comm.record_paths() if comm is not None else None
    if comm is not None:
        comm.close()
    rows_rank = hi - lo
    flops_rank = 2.0 * nq * rows_rank * dim
    filt = (filt_ms / steps) * 1e-3 if n_pass else 0.0
    out = {
        "workload": (f"flat euclidean batched search, corpus {corpus} x {dim}, {nq} queries per batch, k={k}, "
                     f"{world} row shard(s) of {rows_rank} rows" +
                     ("" if world > 1 else f" (one rank's shard of the {args.c3_shards}-GPU configuration, world-1 RCCL)")),
        "n_gpus": world, "rows_per_rank": rows_rank, "queries": nq, "dim": dim,
        "value": round(nq / dt, 1), "unit": "queries/s", "ms_per_batch": round(dt * 1e3, 3), "batches_timed": steps,
        "transport": ("gloo records + device merge (rehearsal on one card)" if rehearse else
                      "RCCL: one ncclAllGather per batch inside the library (vl_shard_search_batch), device merge"),
        "collective_bytes_per_rank": 8 * (4 + nq + 3 * nq * min(k, rows_rank)),
        "nccl_ranks_seen": nccl_ranks_seen,
        "identical_on_every_rank": same, "own_rows_match_single_search": f"{own_ok}/{n_own}",
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS,
                     "kernel": "k_mfma_rows (sampling pass + pass-1 stages + thresholds / candidate select), per GPU",
                     "flops_per_batch_per_gpu": flops_rank,
                     "achieved": round(flops_rank / filt / 1e12, 1) if filt > 0 else None,
                     "frac": round(flops_rank / filt / 1e12 / MFMA_PEAK_TFLOPS, 4) if filt > 0 else None,
                     "filter_kernels_ms_per_batch": round(filt * 1e3, 3),
                     "whole_call": {"achieved": round(flops_rank / dt / 1e12, 1),
                                    "frac": round(flops_rank / dt / 1e12 / MFMA_PEAK_TFLOPS, 4)},
                     "traffic": traffic3, "traffic_over_algorithmic": ratio3,
                     "algorithmic_bytes_per_batch": rows_rank * (plan3["ksteps"] * 16) * 2 if plan3["ksteps"] else None,
                     "kernel_plan": plan3},
    }
    if dev_q is not None:
        # the contract's `value` takes its inputs from HBM: the device-query form is the figure, the host form rides beside it
        dev_q["whole_call_frac_of_mfma_peak"] = round(flops_rank / (dev_q["ms_per_batch"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
        out["device_queries"] = dev_q
        out["host_queries_pcie_inclusive"] = {"value": out["value"], "unit": "queries/s", "ms_per_batch": out["ms_per_batch"],
                                              "note": "the same batch handed over in host memory: pinned staging + a 6.3 MB PCIe copy inside the call"}
        out["value"], out["ms_per_batch"] = dev_q["value"], dev_q["ms_per_batch"]
        out["inputs"] = "the 1024 x 768 f64 query batch resident in each rank's HBM (vl_shard_search_batch_dev)"
        out["roofline"]["whole_call"] = {"achieved": round(flops_rank / (dev_q["ms_per_batch"] * 1e-3) / 1e12, 1),
                                         "frac": dev_q["whole_call_frac_of_mfma_peak"]}
    if prof and prof["calls"]:
        c = prof["calls"]
        on_dev = bool(rec_paths and rec_paths["via_host"] == 0 and rec_paths["on_device"] > 0)
        # round 4: the finalize kernel writes the record into the all-gather's send buffer and its 32-byte header travels
        # while the search runs: nothing is left between the two events in front of the collective (they read ~0.006 ms apart)
        out["exchange_record"] = dict(rec_paths or {}, written_by="finalize kernel, in device memory" if on_dev else "host (pinned) + H2D copy")
        out["exchange_ms_per_batch"] = {"local_search_host_clock": round(prof["local_ms"] / c, 3),
                                        ("nothing_before_the_collective_two_events_apart" if on_dev else "record_h2d"): round(prof["h2d_ms"] / c, 4),
                                        "ncclAllGather": round(prof["allgather_ms"] / c, 4),
                                        "merge_kernel_and_d2h": round(prof["merge_ms"] / c, 4),
                                        "note": "HIP events on the exchange stream (vl_comm_profile_read), rank 0"}
    del sh
    if prebuilt is None:
        del shard
    torch.cuda.empty_cache()
    return out


def run_c3_full_one_card(V, torch, args, dev, dev_index, k, shards, starts):
    """BASELINE config 3 AT ITS OWN SIZE on one card: the 10 M x 768 corpus as `--c3-shards` contiguous row shards (separate
    flat handles on this GPU), the 1024-query Euclidean batch through vl_shard_search_local_dev on every shard and ONE
    vl_shard_merge over the records -- the calls an N-rank run makes (csrc/shard_comm.cpp), minus the wire.  The shards run one
    after the other here (one card), so per-shard time is what one rank of the N-GPU run spends; the merge is the real
    N-record merge."""
    from vectorlite_amd.sharded import OneProcessShards
    dim, nq, metric = args.c3_dim, args.c3_batch, 1
    world = len(shards)
    total = starts[-1]
    sh = OneProcessShards(shards)
    assert sh.total == total and sh.offsets == starts[:-1]
    Qs = unit_queries(2468, nq, dim)
    dQ = torch.from_numpy(np.ascontiguousarray(Qs)).to(dev)
    sh.search_batch(dQ[:128].contiguous(), k, metric)   # every shard's bf16 copy and scratch
    ids, scores, cnt, gpos = sh.search_batch(dQ, k, metric, with_positions=True)
    torch.cuda.synchronize()
    reps, local, merge = 3, [], []
    tw = time.perf_counter()
    for _ in range(reps):
        t = {}
        i2, s2, n2 = sh.search_batch(dQ, k, metric, timings=t)
        local.append(t["local_ms"])
        merge.append(t["merge_ms"])
    wall = (time.perf_counter() - tw) / reps
    local = np.asarray(local)
    same = bool(np.array_equal(i2, ids) and np.array_equal(s2, scores))
    # every shard's rows in the merged answer are what that shard's own single search returns for them, in order
    own_ok, n_own = 0, 4
    for qi in range(n_own):
        ok = True
        for r, shd in enumerate(shards):
            li, ls = shd.search_arrays(Qs[qi], k, metric)
            mine = [(int(i), float(sc)) for i, sc, pp in zip(ids[qi], scores[qi], gpos[qi]) if starts[r] <= int(pp) < starts[r + 1]]
            ok = ok and mine == list(zip(li.tolist(), ls.tolist()))[: len(mine)]
        own_ok += int(ok)
    sorted_ok = bool(all(scores[q, j - 1] >= scores[q, j] for q in range(nq) for j in range(1, scores.shape[1])))
    flops = 2.0 * nq * total * dim
    per_shard = float(local.mean())
    out = {"workload": (f"flat euclidean batched search, corpus {total} x {dim} as {world} contiguous row shards on ONE card, "
                        f"{nq} queries per batch (resident in HBM), k={k}: vl_shard_search_local_dev x {world} + vl_shard_merge"),
           "shards": world, "rows_per_shard": [starts[r + 1] - starts[r] for r in range(world)],
           "ms_per_batch_per_shard": round(per_shard, 3),
           "ms_per_batch_per_shard_min_max": [round(float(local.min()), 3), round(float(local.max()), 3)],
           "merge_ms_for_all_records": round(float(np.mean(merge)), 3),
           "ms_per_batch_all_shards_one_after_the_other": round(wall * 1e3, 3),
           "value": round(nq / wall, 1), "unit": "queries/s",
           "second_pass_identical": same, "scores_sorted": sorted_ok,
           "every_shards_rows_match_its_own_single_search": f"{own_ok}/{n_own}",
           "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS,
                        "whole_call_one_card": {"achieved": round(flops / wall / 1e12, 1), "frac": round(flops / wall / 1e12 / MFMA_PEAK_TFLOPS, 4)},
                        "traffic": None},
           "n_gpu_estimate_from_these_parts": {
               "ms_per_batch": round(float(local.max(axis=1).mean()) + float(np.mean(merge)), 3),
               "note": f"slowest shard + the {world}-record merge; the ncclAllGather of {world} x {8 * (4 + nq + 3 * nq * k)} bytes is NOT in it "
                       "(no multi-GPU node): arithmetic from single-card parts, not a scaling measurement"}}
    del dQ
    return out


def gen_c4_rows(torch, dev, kind, n, dim, seed, state):
    """Config 4's row distributions.  latent16: A z + 0.05 noise, z in R^16 (low intrinsic dimension); clustered: 2000 topical
    clusters on the sphere (centre + 0.35 / sqrt(dim) gaussian per coordinate) -- the other shape sentence embeddings take;
    iid_gaussian: SURVEY 8(d)'s i.i.d. N(0, 1) unit rows (near-equidistant: no graph index answers them)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if kind == "latent16":
        x = torch.randn((n, 16), dtype=torch.float64, device=dev, generator=g) @ state["A"]
        x += 0.05 * torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    elif kind == "clustered":
        which = torch.randint(0, state["C"].shape[0], (n,), device=dev, generator=g)
        x = state["C"][which] + (0.35 / dim ** 0.5) * torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    else:
        x = torch.randn((n, dim), dtype=torch.float64, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    return x


def run_c4(V, torch, args, dev, dev_index, k, cap_s, log_fn):
    """BASELINE config 4 (HNSW, default profile M 16 / M0 32, cosine, dim of the run): build on the GPU with the library's
    default ef_construction (400 = what the crate's Params::default() is recalled to be), recall@10 against the
    exhaustive order, batched QPS and distance evaluations per query at ef 10 (the reference's ef = min(k, len),
    src/index/hnsw.rs:437), 32 and 128, a lone query's latency at the strict beam with the CPU walk of the same graph at the
    SAME beam beside it, the CPU walk at ef 128 ("recall@10 vs CPU HNSW"), and the ef_construction sweep
    {128, 200, 400} at the strict beam -- on two embedding-like distributions and on SURVEY 8(d)'s i.i.d. rows.  The walk is
    this repository's own (crate hnsw 0.11.0 is not in the reference tree): parity with the crate's walk is UNPINNED, the
    numbers are recall, not parity."""
    t_block = time.perf_counter()
    n, dim, nq = args.c4_rows, args.dim, args.c4_queries
    efc_default = 400
    res = {"workload": f"HNSW cosine, N={n}, dim={dim}, M=16 M0=32 ef_construction={efc_default} (library default), {nq}-query batches, k={k}",
           "parity": "unpinned (own walk; the crate's is not in the reference tree)", "data": {}}

    def build(kind, state, efc, with_flat):
        flat = None
        if with_flat:
            flat = V.FlatIndex(dim, device=dev_index)
            flat.reserve(n)
        hn = V.HNSWIndex(dim, 0, device=dev_index, ef_construction=efc)
        t_build, done = 0.0, 0
        while done < n:
            c = min(250_000, n - done)
            x = gen_c4_rows(torch, dev, kind, c, dim, 31337 + done, state)
            ids = np.arange(done, done + c, dtype=np.uint64)
            if flat is not None:
                flat.add_rows(ids, x, validate=False)
            tb = time.perf_counter()
            hn.add_rows(ids, x)
            t_build += time.perf_counter() - tb
            done += c
            del x
        return flat, hn, t_build

    for name in ("latent16", "clustered", "iid_gaussian"):
        if time.perf_counter() - t_block > cap_s:
            res["data"][name] = {"skipped": "block time cap"}
            continue
        g = torch.Generator(device=dev)
        g.manual_seed(99 + len(name))
        state = {}
        if name == "latent16":
            state["A"] = torch.randn((16, dim), dtype=torch.float64, device=dev, generator=g)
        elif name == "clustered":
            cc = torch.randn((2000, dim), dtype=torch.float64, device=dev, generator=g)
            state["C"] = cc / torch.linalg.vector_norm(cc, dim=1, keepdim=True)
        flat, hn, t_build = build(name, state, efc_default, True)
        Q = gen_c4_rows(torch, dev, name, nq, dim, 4321, state).cpu().numpy()
        ti, _, _ = flat.search_batch(Q, k, 0)  # exhaustive f64 order
        # the graph only sees the reference's quantised u64 distances (src/index/hnsw.rs:113-174): recall is also
        # counted against THAT order (ties at the k-th distance accepted), on a subset
        nchk = min(nq, 32)
        allpos = np.arange(n, dtype=np.uint64)
        D = [flat.hnsw_distances(Q[i], allpos, 0) for i in range(nchk)]
        kth = [np.partition(d, k - 1)[k - 1] for d in D]
        rec_of = lambda hi_, hnn, m: float(np.mean([len(set(hi_[i, :int(hnn[i])].tolist()) & set(ti[i].tolist())) / float(k) for i in range(m)]))  # noqa: E731
        per_ef = {}
        for ef in (10, 32, 128):
            strict = ef == 10
            hn.search_batch(Q[:8], k, 0, ef=(0 if strict else ef))
            q0, e0 = hn.walk_stats()
            tq = time.perf_counter()
            hi_, hs_, hnn = hn.search_batch(Q, k, 0, ef=(0 if strict else ef))
            dtq = time.perf_counter() - tq
            q1, e1 = hn.walk_stats()
            evq = (e1 - e0) / max(q1 - q0, 1)
            rec = rec_of(hi_, hnn, nq)
            recq = float(np.mean([sum(1 for x in hi_[i, :int(hnn[i])] if D[i][int(x)] <= kth[i]) / float(k) for i in range(nchk)]))
            row_bytes = (evq - max(ef, k)) * dim * 4 + max(ef, k) * dim * 8
            gbps = (nq / dtq) * row_bytes / 1e9
            tr4, ratio4 = hnsw_traffic({"data": name, "rows": n, "dim": dim, "queries": nq, "ef": ef, "ef_construction": efc_default}, evq)
            per_ef[f"ef{ef}"] = {"recall_at_10_vs_exact_f64_order": round(rec, 4), "recall_at_10_vs_u64_distance_order": round(recq, 4),
                                "queries_per_s": round(nq / dtq, 1), "distance_evals_per_query": round(evq, 1),
                                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": round(gbps, 1),
                                             "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": tr4, "traffic_over_algorithmic": ratio4,
                                             "algorithmic_bytes_per_batch": int(nq * row_bytes),
                                             "note": "random 1.5-3 KB row reads of a latency-bound walk: rows read x row bytes / time"},
                                "beam": "strict reference rule ef = min(k, len)" if strict else f"ef = {ef}"}
        # a lone query at the reference's strict beam
        lat = []
        for i in range(20):
            tl = time.perf_counter()
            hn.search_arrays(Q[i % nq], k, 0)
            lat.append(time.perf_counter() - tl)
        # BASELINE config 4's own words: "recall@10 vs CPU HNSW".  The graph is exported (vl_index_hnsw_graph_export) and
        # oracle/vl_hnsw_cpu.c -- the checker's single-threaded walk with the reference's f64 -> u64 callbacks
        # (src/index/hnsw.rs:113-174) -- walks the SAME graph: at ef 128 (recall against the GPU walk's) and at the reference's
        # strict beam ef 10 (its time per query beside the GPU's lone-query latency above: SURVEY H5, like for like).
        # (A CPU baseline leg: the oracle is the checker here.)
        cpu_walk, cpu_walk10 = None, None
        try:
            from oracle import oracle as O
            n_cpu = min(nchk, 12)
            graph = hn.graph(with_rows=True)
            walker = O.HnswCpuWalker(graph, O.COSINE)
            tw = time.perf_counter()
            cw = [walker.search(Q[i], 128, k) for i in range(n_cpu)]
            t_cpu = (time.perf_counter() - tw) / n_cpu
            ev128 = walker.evals.value
            gi, _, gn = hn.search_batch(Q[:n_cpu], k, 0, ef=128)
            in_top = lambda i, ids_: sum(1 for x in ids_ if D[i][int(x)] <= kth[i]) / float(k)  # noqa: E731
            cpu_walk = {"queries": n_cpu, "ef": 128, "cores": 1,
                        "cpu_recall_at_10_vs_u64_distance_order": round(float(np.mean([in_top(i, cw[i][0]) for i in range(n_cpu)])), 4),
                        "gpu_recall_at_10_same_queries": round(float(np.mean([in_top(i, gi[i, :int(gn[i])]) for i in range(n_cpu)])), 4),
                        "cpu_ms_per_query": round(t_cpu * 1e3, 3),
                        "cpu_distance_evals_per_query": round(ev128 / n_cpu, 1),
                        "walker": "oracle/vl_hnsw_cpu.c on the graph this index built (parity with crate hnsw 0.11.0's walk: unpinned)"}
            n10 = min(nchk, 24)
            walker.evals.value = 0
            tw = time.perf_counter()
            cw10 = [walker.search(Q[i], 10, k) for i in range(n10)]
            t_cpu10 = (time.perf_counter() - tw) / n10
            g10, _, gn10 = hn.search_batch(Q[:n10], k, 0)
            cpu_walk10 = {"queries": n10, "ef": 10, "cores": 1, "cpu_ms_per_query": round(t_cpu10 * 1e3, 4),
                          "gpu_lone_query_ms_same_beam": round(float(np.median(lat)) * 1e3, 4),
                          "cpu_recall_at_10_vs_u64_distance_order": round(float(np.mean([in_top(i, cw10[i][0]) for i in range(n10)])), 4),
                          "gpu_recall_at_10_same_queries": round(float(np.mean([in_top(i, g10[i, :int(gn10[i])]) for i in range(n10)])), 4),
                          "cpu_distance_evals_per_query": round(walker.evals.value / n10, 1)}
            del walker, graph
        except Exception as e:  # noqa: BLE001 -- the recall / QPS figures above stand without it
            cpu_walk = {"skipped": f"{type(e).__name__}: {e}"[:200]}
        entry = {"ef_construction": efc_default, "build_s": round(t_build, 2), "inserts_per_s": round(n / t_build, 0),
                 "single_query_ms_strict_beam": round(float(np.median(lat)) * 1e3, 3), **per_ef,
                 "cpu_hnsw_walk_same_graph": cpu_walk, "cpu_hnsw_walk_same_graph_strict_beam": cpu_walk10}
        log_fn(f"[bench] config 4 / {name}: build {t_build:.1f}s, " + ", ".join(f"{e}: {v['recall_at_10_vs_exact_f64_order']:.3f} @ {v['queries_per_s']:.0f} q/s" for e, v in per_ef.items()))
        del hn
        # the construction beam's share: the same rows built at 128 (rounds 1-3's value) and 200, strict beam and ef 128
        if name != "iid_gaussian" and not args.no_c4_sweep:
            sweep = {str(efc_default): {"build_s": entry["build_s"], "recall_at_10_strict_beam": per_ef["ef10"]["recall_at_10_vs_exact_f64_order"],
                                        "recall_at_10_ef128": per_ef["ef128"]["recall_at_10_vs_exact_f64_order"]}}
            for efc in (128, 200):
                if time.perf_counter() - t_block > cap_s:
                    sweep[str(efc)] = {"skipped": "block time cap"}
                    continue
                _, h2, tb2 = build(name, state, efc, False)
                a_i, _, a_n = h2.search_batch(Q, k, 0)
                b_i, _, b_n = h2.search_batch(Q, k, 0, ef=128)
                sweep[str(efc)] = {"build_s": round(tb2, 2), "recall_at_10_strict_beam": round(rec_of(a_i, a_n, nq), 4),
                                   "recall_at_10_ef128": round(rec_of(b_i, b_n, nq), 4)}
                del h2
            entry["ef_construction_sweep"] = sweep
        res["data"][name] = entry
        del flat, state
        torch.cuda.empty_cache()
    return res


def run_rank(args) -> int:
    # ONE JSON line on stdout means nothing else may land there: RCCL prints its version banner to STDOUT when
    # the first communicator is made (seen with torch's nccl backend and with vl_comm_create).  Everything this
    # process and its libraries write to fd 1 goes to stderr; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        log(f"[bench] note: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # VL_BENCH_REHEARSE=1 (one-GPU rehearsal of the multi-rank path): ranks share the visible cards and
    # rendezvous over gloo, because RCCL refuses two ranks on one device.  Never set by the driver.
    rehearse = os.environ.get("VL_BENCH_REHEARSE") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # under a launcher (WORLD_SIZE set) the process group is made even for one rank, so that the RCCL branch
    # (init, barrier, device all-reduce) is the same code at N = 1 as at N = 8
    use_dist = "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from vectorlite_amd import build as vbuild
    if not os.path.exists(vbuild.SO):  # normally prebuilt by __graft_entry__.build(); never build concurrently
        if rank == 0:
            vbuild.build()
        if use_dist:
            dist.barrier()
    import vectorlite_amd as V

    metric = {"cosine": 0, "euclidean": 1, "manhattan": 2, "dotproduct": 3}[args.metric]
    n, dim, k = args.rows, args.dim, args.k
    plan = plan_queries(args.steps, args.warmup, args.cpu_queries)

    # ---- build the replica: rows generated on the device, ingested device-to-device -------------
    t0 = time.time()
    idx = V.FlatIndex(dim, device=dev_index)
    idx.reserve(n)
    sample_rows = None
    want_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    n_sample = min(args.cpu_sample_rows, n)
    sample_parts = []
    done = 0
    ci = 0
    while done < n:
        c = min(args.chunk, n - done)
        x = gen_unit_rows(torch, dev, c, dim, 1234 + ci)
        idx.add_rows(ids_for(done, c), x, validate=False)
        if want_cpu and done < n_sample:
            take = min(c, n_sample - done)
            sample_parts.append(x[:take].cpu().numpy())
        done += c
        ci += 1
        del x
    torch.cuda.synchronize()
    if want_cpu:
        sample_rows = np.ascontiguousarray(np.concatenate(sample_parts, axis=0))
        del sample_parts
    build_s = time.time() - t0
    if rank == 0:
        log(f"[bench] index built: {n} x {dim} in {build_s:.1f}s")

    # ---- queries (host f64, unit norm): timed set disjoint per rank, check set shared ------------
    Q = unit_queries(plan["timed"]["seed"] + rank, plan["timed"]["n"], dim)
    Qp = unit_queries(plan["prewarm"]["seed"] + rank, plan["prewarm"]["n"], dim)
    Qc = unit_queries(plan["check"]["seed"], plan["check"]["n"], dim)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # fixed pre-warm, outside the timed region and independent of --warmup
    tp = time.perf_counter()
    n_pre = 0
    while n_pre < PREWARM_MIN_QUERIES or time.perf_counter() - tp < PREWARM_MIN_SECONDS:
        idx.search_arrays(Qp[n_pre % len(Qp)], k, metric)
        n_pre += 1
    for i in range(args.warmup):
        idx.search_arrays(Q[i], k, metric)
    idx.profile_read()
    idx.profile_enable(True)  # HIP events around the scan kernel, on the stream it is launched on
    barrier()
    t1 = time.perf_counter()
    paths = set()
    stamps = []
    for i in range(args.warmup, args.warmup + args.steps):
        idx.search_arrays(Q[i], k, metric)
        stamps.append(time.perf_counter())
        paths.add(V.last_path())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    barrier()
    idx.profile_enable(False)
    n_launch, scan_ms, scan_bytes = idx.profile_read()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())

    run_sharded = use_dist and world > 1 and not args.no_checks and not args.no_other_configs
    STUCK_S = 420.0  # a collective that never returns: give up, non-zero (a rank that touched the GPU never exits 0 on a hang)

    def c3_all_ranks():
        return run_c3(V, torch, dist, args, dev, dev_index, rank, world, rehearse, k, args.block_cap_s)

    if rank != 0:
        dist.barrier()  # rank 0 has published the line
        if run_sharded:
            wd = threading.Timer(STUCK_S, lambda: os._exit(3))
            wd.daemon = True
            wd.start()
            try:
                c3_all_ranks()
            except BaseException as e:  # noqa: BLE001
                log(f"[bench] rank {rank}: config 3 (row-sharded) failed: {type(e).__name__}: {e}")
            wd.cancel()
        dist.destroy_process_group()
        return 0

    qps = world * args.steps / elapsed_max
    ld = (dim + 3) // 4 * 4
    alg_bytes = n * ld * 4  # N_scanned * dim * sizeof(f32) per slab pass
    avg_scan_ms = scan_ms / max(n_launch, 1)
    achieved = alg_bytes / (avg_scan_ms * 1e-3) / 1e9 if n_launch and avg_scan_ms > 0 else 0.0
    # HBM bytes per k_scan launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, gfx950 x2
    # correction; profiles/traffic.json).  PMC cannot be collected from inside this process, so the
    # figure is attached only when it was measured on exactly this workload.
    traffic = None
    scan_used = idx.last_scan()
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)["k_scan"]
        # ... and by exactly this kernel: the instantiation and grid the PMC pass saw (a changed k_scan drops the figure)
        if (tj["workload"] == {"rows": n, "dim": dim, "metric": args.metric} and tj.get("variant") == scan_used["variant"]
                and tj.get("grid") == scan_used["grid"] and tj.get("query_in_kernarg") == scan_used["query_in_kernarg"]):
            traffic = tj["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    errors = []
    out = {
        "metric": METRIC_NAME,
        "value": round(qps, 3),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"flat {args.metric} single-query search, N={n}, dim={dim}, k={k}",
            "rows": n, "dim": dim, "k": k, "metric": args.metric,
            "parallelism": "1 GPU" if world == 1 else f"{world} replicas (queries dealt across ranks, no collective)",
            "index_build_s": round(build_s, 1),
            "search_paths_seen": sorted(paths),
            "prewarm_queries_untimed": n_pre,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "kernel": "k_scan",
            "kernel_variant": scan_used,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": round(avg_scan_ms, 4),
            "launches_timed": n_launch,
        },
    }
    out.update(step_rates(stamps, t1))

    def publish():
        """The line as it stands: to the supervisor's result file now, to stdout at the end."""
        if errors:
            out["errors"] = errors
        if args.result_file:
            tmp = args.result_file + ".tmp"
            with open(tmp, "w") as f:
                json.dump(out, f)
            os.replace(tmp, args.result_file)

    def block(name, fn):
        """A later block can fail; the timed region's numbers cannot be taken down with it."""
        try:
            fn()
        except BaseException as e:  # noqa: BLE001 -- includes SystemExit from helpers; KeyboardInterrupt re-raised
            if isinstance(e, KeyboardInterrupt):
                raise
            errors.append({"block": name, "error": f"{type(e).__name__}: {e}"[:400]})
            log(f"[bench] block '{name}' failed: {type(e).__name__}: {e}")
        publish()

    publish()  # the contract line exists from here on
    log(f"[bench] timed region: {qps:.1f} q/s, {out['ms_per_step']} ms/step, k_scan {achieved:.0f} GB/s "
        f"({out['roofline']['frac']:.3f} of peak)")
    if use_dist:
        dist.barrier()

    extras = world == 1 and not args.no_checks  # N>1: every rank leaves together, nothing runs on rank 0 alone

    other = out["config"].setdefault("other_configs", {})

    if run_sharded:
        def on_stuck():  # a collective that never returns must not take the measured line with it
            errors.append({"block": "c3_row_sharded", "error": f"no answer within {STUCK_S:g} s: abandoned"})
            publish()
            if not args.result_file:
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(3)
        wd = threading.Timer(STUCK_S, on_stuck)
        wd.daemon = True
        wd.start()
        block("c3_row_sharded", lambda: other.__setitem__("c3_row_sharded", c3_all_ranks()))
        out["nccl_ranks_seen"] = (other.get("c3_row_sharded") or {}).get("nccl_ranks_seen")  # did RCCL itself see N ranks?
        publish()
        wd.cancel()

    # ---- CPU baseline: the oracle (reference-faithful restatement), bounded sample ---------------
    def cpu_block():
        from oracle import oracle as O
        O.build()
        ns = sample_rows.shape[0]
        ref = O.FlatOracle(dim, ids_for(0, ns), sample_rows)
        sub = V.FlatIndex(dim, device=dev_index)
        sub.add_rows(ids_for(0, ns), sample_rows, validate=False)
        ref_out = []
        tc = time.perf_counter()
        for i in range(plan["cpu"]["n"]):
            ref_out.append(ref.search(Qc[i], k, metric))
            if time.perf_counter() - tc > args.cpu_budget_s and len(ref_out) >= 4:
                break
        cpu_s = time.perf_counter() - tc
        nqc = len(ref_out)
        cpu_qps_sample = nqc / cpu_s
        scale = ns / n
        out["cpu_baseline"] = {
            "value": round(cpu_qps_sample * scale, 5),
            "unit": "queries/s",
            "cores": 1,
            "kind": "port",
            "sample": (f"{nqc} queries on the first {ns} of {n} rows with oracle/vl_oracle.c "
                       f"(reference-faithful: AoS rows, f64 sequential sums, N result records, stable sort); "
                       f"{cpu_s:.1f}s CPU; measured {cpu_qps_sample:.3f} q/s at N={ns}, scaled x{scale:g} "
                       f"to N={n} (the scan is linear in N)"),
        }
        publish()
        ids_equal = 0
        max_diff = 0.0
        recall_hits = 0
        for i in range(nqc):
            gi, gs = sub.search_arrays(Qc[i], k, metric)
            ri, rs = ref_out[i]
            ids_equal += int(gi.tolist() == ri.tolist())
            recall_hits += len(set(gi.tolist()) & set(ri.tolist()))
            if len(gs) == len(rs) and len(gs):
                max_diff = max(max_diff, float(np.max(np.abs(gs - rs))))
        out["parity"] = {
            "checked_queries": nqc,
            "rows": ns,
            "ids_bit_exact": f"{ids_equal}/{nqc}",
            "max_abs_score_diff": max_diff,
            "recall_at_10": round(recall_hits / float(nqc * min(k, ns)), 6),
        }
        publish()
        if args.metric == "cosine":
            try:
                out["cpu_baseline_fair"] = fair_cpu_baseline(sample_rows, Qc[:nqc], k, n, [r[0] for r in ref_out],
                                                             ids_for(0, ns))
            except Exception as e:  # no compiler on the box: say so instead of inventing a number
                out["cpu_baseline_fair"] = {"value": None, "note": f"oracle/vl_fair.c could not be run here: {e}"}

    if want_cpu:
        block("cpu_baseline", cpu_block)

    # ---- correctness property at full size: fast path == exact path -----------------------------
    def exact_block():
        n_chk = plan["exact"]["n"]
        agree = 0
        try:
            for i in range(n_chk):
                fi, fs = idx.search_arrays(Qc[i], k, metric)
                idx.force_path(V.PATH_EXACT_SELECT)
                ei, es = idx.search_arrays(Qc[i], k, metric)
                idx.force_path(0)
                agree += int(fi.tolist() == ei.tolist() and fs.tolist() == es.tolist())
        finally:
            idx.force_path(0)
        out["config"]["fast_vs_exact_full_size"] = f"{agree}/{n_chk} queries bit-identical (ids and f64 scores)"

    # ---- informational: measured copy ceiling, and the opt-in bf16-first filter (NOT the headline) ----
    def d2d_block():
        out["roofline"]["measured_d2d_copy_GBps"] = device_copy_ceiling(torch, dev)

    def bf16_block():
        nb = plan["bf16"]["n"]
        try:
            idx.set_single_filter("bf16")
            for i in range(min(10, nb)):
                idx.search_arrays(Qc[i], k, metric)
            idx.profile_read()
            idx.profile_enable(True)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            outs = [idx.search_arrays(Qc[i], k, metric) for i in range(nb)]
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            idx.profile_enable(False)
            nl, ms16, by16 = idx.profile_read()
        finally:
            idx.profile_enable(False)
            idx.set_single_filter("f32")
        same = 0
        n_cmp = min(nb, 8)
        for i in range(n_cmp):
            fi, fs = idx.search_arrays(Qc[i], k, metric)
            same += int(fi.tolist() == outs[i][0].tolist() and fs.tolist() == outs[i][1].tolist())
        out["config"]["bf16_first_filter_optin"] = {
            "qps": round(nb / tb, 1), "ms_per_step": round(tb / nb * 1e3, 4),
            "scan_GBps_on_bf16_bytes": round(by16 / max(nl, 1) / (ms16 / max(nl, 1) * 1e-3) / 1e9, 1) if nl and ms16 > 0 else None,
            "identical_to_f32_path": f"{same}/{n_cmp} queries (ids and f64 scores)",
            "note": "vl_index_set_single_filter(h, 1): scan a bf16 copy of the slab first, same exact f64 "
                    "rescoring and bound check, fall back to the f32 scan when not certified",
        }

    # ---- sustained rate: the same serial searches for --sustained-s seconds (thousands of steps, not a 20-step burst) ----
    def sustained_block():
        Qs = unit_queries(777, 512, dim)
        idx.profile_read()
        idx.profile_enable(True)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        cnt = 0
        while True:
            idx.search_arrays(Qs[cnt % 512], k, metric)
            cnt += 1
            if (cnt & 63) == 0 and time.perf_counter() - ts >= args.sustained_s:
                break
        torch.cuda.synchronize()
        el = time.perf_counter() - ts
        idx.profile_enable(False)
        nl, ms_s, _ = idx.profile_read()
        out["value_sustained"] = {"value": round(cnt / el, 3), "unit": "queries/s", "queries": cnt, "seconds": round(el, 2),
                                  "ms_per_step": round(el / cnt * 1e3, 4),
                                  "k_scan_avg_launch_ms": round(ms_s / max(nl, 1), 4),
                                  "k_scan_frac_of_hbm_peak": round(alg_bytes / (ms_s / max(nl, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if nl else None}

    # ---- BASELINE config 2: the same search on the first --c2-rows rows (1 M): the scan no longer hides the fixed cost ----
    def c2_block():
        rows2 = min(args.c2_rows, n)
        sub = V.FlatIndex(dim, device=dev_index)
        sub.reserve(rows2)
        d2, ci2 = 0, 0
        while d2 < rows2:  # the first rows of the SAME corpus (same generator seeds as the build above)
            c = min(args.chunk, n - d2)
            x = gen_unit_rows(torch, dev, c, dim, 1234 + ci2)
            take = min(c, rows2 - d2)
            sub.add_rows(ids_for(d2, take), x[:take], validate=False)
            d2 += take
            ci2 += 1
            del x
        Q2 = unit_queries(2222, 64, dim)
        for i in range(20):
            sub.search_arrays(Q2[i], k, metric)
        sub.profile_read()
        sub.profile_enable(True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        reps2 = 400
        for i in range(reps2):
            sub.search_arrays(Q2[i % 64], k, metric)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        sub.profile_enable(False)
        nl2, ms2, by2 = sub.profile_read()
        same = 0
        for i in range(4):  # the answer is a prefix-consistent function of the rows: the big index restricted to them
            a_i, a_s = sub.search_arrays(Q2[i], k, metric)
            sub.force_path(V.PATH_EXACT_SELECT)
            b_i, b_s = sub.search_arrays(Q2[i], k, metric)
            sub.force_path(0)
            same += int(a_i.tolist() == b_i.tolist() and a_s.tolist() == b_s.tolist())
        ach2 = (by2 / max(nl2, 1)) / (ms2 / max(nl2, 1) * 1e-3) / 1e9 if nl2 and ms2 > 0 else 0.0
        other["c2"] = {"workload": f"flat {args.metric} single-query search, N={rows2}, dim={dim}, k={k}, 1 GPU",
                       "value": round(reps2 / e2, 1), "unit": "queries/s", "ms_per_step": round(e2 / reps2 * 1e3, 4),
                       "fast_vs_exact": f"{same}/4 queries bit-identical (ids and f64 scores)",
                       "parity_vs_oracle": "the `parity` object of this line is measured on this row set" if want_cpu and rows2 == n_sample else None,
                       "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": round(ach2, 1),
                                    "frac": round(ach2 / HBM_PEAK_GBPS, 4), "kernel": "k_scan", "kernel_variant": sub.last_scan(),
                                    "algorithmic_bytes_per_launch": rows2 * ld * 4, "avg_launch_ms": round(ms2 / max(nl2, 1), 4),
                                    "launches_timed": nl2, "traffic": None,
                                    "whole_call_frac": round(rows2 * ld * 4 / (e2 / reps2) / 1e9 / HBM_PEAK_GBPS, 4)}}

    # ---- BASELINE config 5: 4096 queries in one search_batch on the resident corpus (bf16 MFMA filter + exact f64 finalize) ----
    def c5_block():
        nq5 = args.c5_queries
        Q5 = unit_queries(5555, nq5, dim)
        idx.search_batch(Q5[:64], k, metric)  # builds the fragment-major bf16 slab and the scratch
        idx.search_batch(Q5, k, metric)
        torch.cuda.synchronize()
        idx.profile_read()
        idx.profile_enable(True)
        t5 = time.perf_counter()
        reps5 = 3
        for _ in range(reps5):
            bi, bs, bn = idx.search_batch(Q5, k, metric)
        w5 = (time.perf_counter() - t5) / reps5
        idx.profile_enable(False)
        n_pass, ms5, _ = idx.profile_read()
        plan5 = idx.last_filter()
        traffic5, ratio5 = filter_traffic("c5", {"rows": n, "dim": dim, "queries": nq5, "metric": metric}, plan5)
        kern = ms5 / reps5 * 1e-3
        flops = 2.0 * nq5 * n * dim
        # the contract's `value` takes its inputs from HBM: the same batch resident in device memory
        # (vl_index_search_batch_dev: a kernel stages it); the host form above is the PCIe-inclusive figure beside it
        dQ5 = torch.from_numpy(np.ascontiguousarray(Q5)).to(dev)
        idx.search_batch_device(dQ5, k, metric)
        torch.cuda.synchronize()
        t5d = time.perf_counter()
        for _ in range(reps5):
            di, ds, dn5 = idx.search_batch_device(dQ5, k, metric)
        w5d = (time.perf_counter() - t5d) / reps5
        same5 = bool(np.array_equal(di, bi) and np.array_equal(ds, bs))
        del dQ5
        pick = np.linspace(0, nq5 - 1, 16).astype(int)
        ok5 = 0
        for qi in pick:
            s_i, s_s = idx.search_arrays(Q5[qi], k, metric)
            ok5 += int(bi[qi].tolist() == s_i.tolist() and bs[qi].tolist() == s_s.tolist())
        other["c5"] = {"workload": f"batched flat {args.metric} search as a bf16 MFMA GEMM: Q={nq5}, N={n}, dim={dim}, k={k}, 1 GPU",
                       "value": round(nq5 / w5d, 1), "unit": "queries/s", "ms_per_batch": round(w5d * 1e3, 3),
                       "inputs": "the 4096 x 384 f64 query batch resident in HBM (vl_index_search_batch_dev)",
                       "host_queries_pcie_inclusive": {"value": round(nq5 / w5, 1), "unit": "queries/s", "ms_per_batch": round(w5 * 1e3, 3),
                                                       "identical_to_device_queries": same5,
                                                       "note": "the same batch handed over in host memory: pinned staging + a 12.6 MB PCIe copy inside the call"},
                       "launch_sequences_per_batch": n_pass // max(reps5, 1),
                       "rows_identical_to_single_search": f"{ok5}/16 sampled (ids and f64 scores)",
                       "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS,
                                    "kernel": "k_mfma_rows (sampling pass + pass-1 stages + thresholds / candidate select)",
                                    "flops_per_batch": flops, "achieved": round(flops / kern / 1e12, 1) if kern > 0 else None,
                                    "frac": round(flops / kern / 1e12 / MFMA_PEAK_TFLOPS, 4) if kern > 0 else None,
                                    "filter_kernels_ms_per_batch": round(kern * 1e3, 3),
                                    "whole_call": {"achieved": round(flops / w5d / 1e12, 1), "frac": round(flops / w5d / 1e12 / MFMA_PEAK_TFLOPS, 4)},
                                    "traffic": traffic5, "traffic_over_algorithmic": ratio5, "kernel_plan": plan5}}

    def c3_block():
        # config 3's whole corpus as --c3-shards row shards on this card when it fits (10 M x 768: 107 GB with the bf16 copies);
        # shard 0 IS the one-rank shard of the c3_shard leg (rows are a function of the global row range)
        parts = max(1, args.c3_shards)
        base, rem = divmod(args.c3_rows, parts)
        starts = [0]
        for r in range(parts):
            starts.append(starts[-1] + base + (1 if r < rem else 0))
        need = args.c3_rows * args.c3_dim * (8 + 4 + 2) * 1.06 + 4e9
        free_b, _tot = torch.cuda.mem_get_info(dev)
        full = (not args.no_c3_full) and free_b > need
        shards = [build_c3_shard(V, torch, dev, dev_index, args.c3_dim, starts[r], starts[r + 1]) for r in range(parts if full else 1)]
        torch.cuda.synchronize()
        other["c3_shard"] = run_c3(V, torch, None, args, dev, dev_index, 0, 1, False, k, args.block_cap_s, prebuilt=shards[0])
        out["nccl_ranks_seen"] = other["c3_shard"].get("nccl_ranks_seen")
        publish()
        if full:
            other["c3_full_one_card"] = run_c3_full_one_card(V, torch, args, dev, dev_index, k, shards, starts)
        else:
            other["c3_full_one_card"] = {"skipped": ("--no-c3-full" if args.no_c3_full else
                                                     f"needs ~{need / 1e9:.0f} GB of device memory, {free_b / 1e9:.0f} GB free")}
        del shards
        torch.cuda.empty_cache()

    # ---- the reference's many-readers usage (src/client.rs:398, src/server.rs:269): 16 host threads, single searches, default handle ----
    def concurrent_block():
        T, per = 16, 30
        Qt = unit_queries(1616, T * per, dim)
        lone = [idx.search_arrays(Qt[t * per], k, metric) for t in range(T)]
        # the first shared pass of a handle builds the batch filter's bf16 copy of the rows (once; ~30 ms at 10 M x 384):
        # outside the timed loop, like every other warm-up of this file
        idx.search_batch(Qt[:16], k, metric)
        def closed_loop():
            b0, q0 = idx.coalesce_stats()
            w0, us0 = idx.coalesce_gather()
            res = [None] * (T * per)
            lat = [0.0] * (T * per)
            bar = threading.Barrier(T + 1)

            def worker(t):
                bar.wait()
                for i in range(t * per, (t + 1) * per):
                    ta = time.perf_counter()
                    res[i] = idx.search_arrays(Qt[i], k, metric)
                    lat[i] = time.perf_counter() - ta
            th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            for x in th:
                x.start()
            bar.wait()
            tc = time.perf_counter()
            for x in th:
                x.join()
            el = time.perf_counter() - tc
            b1, q1 = idx.coalesce_stats()
            w1, us1 = idx.coalesce_gather()
            same = sum(int(res[t * per][0].tolist() == lone[t][0].tolist() and res[t * per][1].tolist() == lone[t][1].tolist()) for t in range(T))
            la = np.sort(np.asarray(lat)) * 1e3
            return {"value": round(T * per / el, 1), "unit": "queries/s",
                    "latency_ms": {"mean": round(float(la.mean()), 3), "p50": round(float(la[len(la) // 2]), 3), "p99": round(float(la[int(len(la) * 0.99)]), 3)},
                    "identical_to_lone_search": f"{same}/{T}",
                    "passes": int(b1 - b0), "queries_per_pass": round((q1 - q0) / max(b1 - b0, 1), 2),
                    "leader_waits": int(w1 - w0), "leader_wait_ms_total": round((us1 - us0) / 1e3, 2)}

        closed_loop()                      # settle: thread start-up, the coalescer's pass history
        r_on = closed_loop()
        idx.coalesce_gather(False)
        r_off = closed_loop()
        idx.coalesce_gather(True)

        def native_threads():
            """The same loop from NATIVE threads (tools/native_loadgen.c: pthreads on vl_index_search_cap): no GIL between a
            caller's return and its next call -- what a Rust or C server's workers do."""
            import ctypes as C
            import subprocess
            import tempfile
            from vectorlite_amd import _lib
            so = os.path.join(tempfile.mkdtemp(), "native_loadgen.so")
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(ROOT, "tools", "native_loadgen.c")],
                           check=True, capture_output=True, timeout=60)
            G = C.CDLL(so)
            fn = C.cast(_lib.load().vl_index_search_cap, C.c_void_p)
            Qn = np.ascontiguousarray(Qt)
            res = {}
            for label, adaptive in (("settle", True), ("value", True), ("without_adaptive_gather", False)):
                idx.coalesce_gather(adaptive)
                lat = np.zeros(T * per)
                fid = np.zeros(T * k, dtype=np.uint64)
                fsc = np.zeros(T * k)
                el = C.c_double(0.0)
                b0, q0 = idx.coalesce_stats()
                rc = G.vl_loadgen(fn, idx._h, Qn.ctypes.data_as(C.c_void_p), C.c_uint64(dim), C.c_uint64(k), C.c_int(int(metric)),
                                  C.c_int(T), C.c_int(per), lat.ctypes.data_as(C.c_void_p), fid.ctypes.data_as(C.c_void_p),
                                  fsc.ctypes.data_as(C.c_void_p), C.byref(el))
                b1, q1 = idx.coalesce_stats()
                if rc != 0:
                    raise RuntimeError(f"vl_index_search_cap returned {rc} under the native load generator")
                same = sum(int(fid[t * k:(t + 1) * k].tolist() == lone[t][0].tolist() and fsc[t * k:(t + 1) * k].tolist() == lone[t][1].tolist())
                           for t in range(T))
                la = np.sort(lat) * 1e3
                if label != "settle":
                    res[label] = {"value": round(T * per / el.value, 1), "unit": "queries/s",
                                  "latency_ms": {"mean": round(float(la.mean()), 3), "p50": round(float(la[len(la) // 2]), 3),
                                                 "p99": round(float(la[int(len(la) * 0.99)]), 3)},
                                  "queries_per_pass": round((q1 - q0) / max(b1 - b0, 1), 2), "identical_to_lone_search": f"{same}/{T}"}
            idx.coalesce_gather(True)
            return dict(res["value"], without_adaptive_gather=res["without_adaptive_gather"],
                        note="pthreads in a closed loop on vl_index_search_cap (tools/native_loadgen.c), same handle, same queries")
        try:
            r_native = native_threads()
        except Exception as e:  # no compiler on the box, ...: the Python-thread figures above stand on their own
            idx.coalesce_gather(True)
            r_native = {"skipped": repr(e)[:200]}
        out["concurrent_16_threads"] = {
            "threads": T, "queries": T * per, **r_on,
            "native_threads": r_native,
            "handle": "as created: coalescing on by default (max 256 per pass, window 0, adaptive gather); lone callers are the timed region above",
            "without_adaptive_gather": r_off,
            "note": "concurrent callers share slab passes (bf16 MFMA filter + exact f64 finalize): every answer is the lone search's; "
                    "the gather lets a leader wait (<= a quarter of the recent pass time) for as many callers as the last passes held, "
                    "instead of leading a pass of one while the other 15 are on their way back"}

    def c4_block():
        other["c4_hnsw"] = run_c4(V, torch, args, dev, dev_index, k, args.block_cap_s * 2.5, log)  # three builds at ef_construction 400 + the sweep

    # ---- one reference-faithful CPU query at FULL size (SURVEY 8(d) "run d384 fully"): is the x N scaling above true? ----
    def cpu_full_block():
        need = n * dim * 8 * 1.25 + 6e9
        avail = None
        try:
            for ln in open("/proc/meminfo"):
                if ln.startswith("MemAvailable:"):
                    avail = int(ln.split()[1]) * 1024
            for pth in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
                if os.path.exists(pth):
                    v = open(pth).read().strip()
                    if v != "max":
                        used = 0
                        for up in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                            if os.path.exists(up):
                                used = int(open(up).read().strip())
                        avail = min(avail, int(v) - used) if avail is not None else int(v) - used
        except Exception:
            pass
        cb = out.setdefault("cpu_baseline", {})
        if avail is None or avail < max(need, 48e9 if n >= 10_000_000 else 0):
            cb["full_size_check"] = {"ran": False, "why": f"needs ~{need / 1e9:.0f} GB of host memory for {n} AoS f64 rows, "
                                                          f"{'unknown' if avail is None else round(avail / 1e9, 1)} GB available"}
            return
        from oracle import oracle as O
        O.build()
        ref = O.FlatOracle(dim)
        d3, ci3 = 0, 0
        tb = time.perf_counter()
        while d3 < n:
            c = min(args.chunk, n - d3)
            x = gen_unit_rows(torch, dev, c, dim, 1234 + ci3)  # the corpus of the timed region, chunk by chunk
            ref.extend(ids_for(d3, c), x.cpu().numpy())
            d3 += c
            ci3 += 1
            del x
        build_s = time.perf_counter() - tb
        nqf = 2
        same = 0
        tq = time.perf_counter()
        ref_res = [ref.search(Qc[i], k, metric) for i in range(nqf)]
        cpu_s = time.perf_counter() - tq
        for i in range(nqf):
            gi, gs = idx.search_arrays(Qc[i], k, metric)
            same += int(gi.tolist() == ref_res[i][0].tolist() and gs.tolist() == ref_res[i][1].tolist())
        del ref
        full_qps = nqf / cpu_s
        cb["full_size_check"] = {"ran": True, "rows": n, "queries": nqf, "seconds_per_query": round(cpu_s / nqf, 3),
                                 "value": round(full_qps, 5), "unit": "queries/s",
                                 "scaled_sample_value": cb.get("value"),
                                 "scaled_over_full": round(cb["value"] / full_qps, 4) if cb.get("value") else None,
                                 "gpu_answer_bit_identical": f"{same}/{nqf} (ids and f64 scores, N={n})",
                                 "oracle_build_s": round(build_s, 1)}

    if extras:
        block("value_sustained", sustained_block)
        block("fast_vs_exact_full_size", exact_block)
        block("concurrent_16_threads", concurrent_block)
        if not args.no_other_configs:
            block("c2", c2_block)
            block("c5", c5_block)
            block("c3_shard", c3_block)
            block("c4_hnsw", c4_block)
        block("d2d_copy_ceiling", d2d_block)
        block("bf16_first_filter", bf16_block)
        if want_cpu and not args.no_full_size_cpu_check:
            block("cpu_full_size_check", cpu_full_block)

    publish()
    if not args.result_file:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    launched = "WORLD_SIZE" in os.environ  # torch.distributed.run (the driver's N > 1 form) set it
    if launched or args.worker or args.inline:
        if args.inline and not launched and args.gpus != 1:
            raise SystemExit("--inline runs exactly one rank: use --gpus 1, or drop --inline")
        return run_rank(args)
    return supervise(args, argv)


if __name__ == "__main__":
    sys.exit(main())
