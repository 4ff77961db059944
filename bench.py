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
(weak scaling).  The row-sharded batched mode with an RCCL all-gather (config 3) is
`tools/bench_sharded.py` over the C ABI's vl_shard_* entry points, not this headline line.
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
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
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

    def sharded_check():
        """Every rank, after the timed region (N > 1 only; informational, never part of `value`): the row-sharded
        batched search of config 3's kind on small shards -- each rank scans its own rows, ONE ncclAllGather inside
        libvectorlite_amd.so (vl_shard_search_batch), device merge -- and the ranks compare digests of the answer."""
        from vectorlite_amd.sharded import Comm, ShardedFlatIndex
        rows_per, nqs = 200_000, 256
        shard = V.FlatIndex(dim, device=dev_index)
        g = torch.Generator(device=dev)
        g.manual_seed(777 + rank)
        x = torch.randn((rows_per, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        shard.add_rows(ids_for(rank * rows_per, rows_per), x, validate=False)
        del x
        Qs = unit_queries(2468, nqs, dim)  # the same batch on every rank
        comm = None
        if rehearse:  # ranks share one card and RCCL refuses that: the records travel by gloo, the merge is the same kernel
            sh = ShardedFlatIndex(shard, transport="torch")
        else:
            comm = Comm.from_torch_distributed(device=dev_index)
            sh = ShardedFlatIndex(shard, comm=comm)
        assert (sh.offset, sh.total) == (rank * rows_per, world * rows_per)
        sh.search_batch(Qs, k, metric)
        barrier()
        ts = time.perf_counter()
        for _ in range(3):
            si, ss, sn, sp = sh.search_batch(Qs, k, metric, with_positions=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ts) / 3
        dig = torch.tensor([int(np.bitwise_xor.reduce(np.ascontiguousarray(si).reshape(-1).view(np.int64))),
                            int(np.bitwise_xor.reduce(np.ascontiguousarray(ss).reshape(-1).view(np.int64)))],
                           dtype=torch.int64, device="cpu" if rehearse else dev)
        lo, hi = dig.clone(), dig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool((lo == hi).all().item())
        own_ok = 0
        for qi in range(4):  # this rank's rows in the global answer are what its own single search() returns for them
            li, ls = shard.search_arrays(Qs[qi], k, metric)
            mine = [(int(i), float(sc)) for i, sc, pp in zip(si[qi], ss[qi], sp[qi]) if rank * rows_per <= int(pp) < (rank + 1) * rows_per]
            own_ok += int(mine == list(zip(li.tolist(), ls.tolist()))[: len(mine)])
        tm = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        if comm is not None:
            comm.close()
        return {"transport": "gloo records + device merge (rehearsal on one card)" if rehearse else "RCCL: one ncclAllGather per batch inside the library (vl_shard_search_batch)",
                "shards": world, "rows_per_shard": rows_per, "queries": nqs, "ms_per_batch": round(float(tm.item()) * 1e3, 3),
                "identical_on_every_rank": same, "own_rows_match_single_search": f"{own_ok}/4"}

    run_sharded = use_dist and world > 1 and not args.no_checks
    if rank != 0:
        dist.barrier()  # rank 0 has published the line
        if run_sharded:
            wd = threading.Timer(120.0, lambda: os._exit(0))  # never outlive a stuck collective
            wd.daemon = True
            wd.start()
            try:
                sharded_check()
            except BaseException as e:  # noqa: BLE001
                log(f"[bench] rank {rank}: row-sharded check failed: {type(e).__name__}: {e}")
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
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)["k_scan"]
        if tj["workload"] == {"rows": n, "dim": dim, "metric": args.metric}:
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

    if run_sharded:
        def on_stuck():  # a collective that never returns must not take the measured line with it
            errors.append({"block": "row_sharded_check", "error": "no answer within 120 s: abandoned"})
            publish()
            if not args.result_file:
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)
        wd = threading.Timer(120.0, on_stuck)
        wd.daemon = True
        wd.start()
        block("row_sharded_check", lambda: out["config"].__setitem__("row_sharded_check", sharded_check()))
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

    if extras:
        block("fast_vs_exact_full_size", exact_block)
        block("d2d_copy_ceiling", d2d_block)
        block("bf16_first_filter", bf16_block)

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
