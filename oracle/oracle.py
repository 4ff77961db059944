"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes wrapper over ``libvl_oracle.so`` (oracle/vl_oracle.c, a plain-C
restatement of the reference's distance-scan hot path) plus a pure-Python
second restatement of the same formulas for small cases (Python floats are
IEEE f64 and ``x * y`` then ``+`` are separately rounded, i.e. the same
operation sequence as the reference's scalar loops).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  ``vectorlite_amd`` never does.

Parity pinning: reference = Rust, not buildable here (no cargo/rustc); pinned
by the reference's own known-answer tests (tests/golden/reference_kats.json,
tests/test_oracle_kats.py).  HNSW graph walk (crate hnsw 0.11.0): UNPINNED.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

COSINE, EUCLIDEAN, MANHATTAN, DOT = 0, 1, 2, 3
METRICS = {"cosine": COSINE, "euclidean": EUCLIDEAN, "manhattan": MANHATTAN, "dotproduct": DOT}

OK, DIM_MISMATCH, DUP_ID, NOT_FOUND, METRIC_MISMATCH, NAN_PANIC = 0, 1, 2, 3, 4, 5

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvl_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("vl_oracle.c", "vl_hnsw_cpu.c", "vl_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libvl_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp, u64p = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
        for name in ("vlo_cosine", "vlo_euclidean", "vlo_manhattan", "vlo_dot"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [dp, dp, C.c_size_t]
        L.vlo_calculate.restype = C.c_double
        L.vlo_calculate.argtypes = [C.c_int, dp, dp, C.c_size_t]
        L.vlo_hnsw_distance.restype = C.c_uint64
        L.vlo_hnsw_distance.argtypes = [C.c_int, dp, dp, C.c_size_t]
        L.vlo_convert_distance_to_similarity.restype = C.c_double
        L.vlo_convert_distance_to_similarity.argtypes = [C.c_double, C.c_int]
        L.vlo_hnsw_score.restype = C.c_double
        L.vlo_hnsw_score.argtypes = [C.c_uint64, C.c_int]
        L.vlo_embed_f32.restype = None
        L.vlo_embed_f32.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_int, dp]
        L.vlo_flat_new.restype = C.c_void_p
        L.vlo_flat_new.argtypes = [C.c_size_t, u64p, dp, C.c_size_t]
        L.vlo_flat_extend.restype = C.c_int
        L.vlo_flat_extend.argtypes = [C.c_void_p, u64p, dp, C.c_size_t]
        L.vlo_flat_free.restype = None
        L.vlo_flat_free.argtypes = [C.c_void_p]
        L.vlo_flat_add.restype = C.c_int
        L.vlo_flat_add.argtypes = [C.c_void_p, C.c_uint64, dp, C.c_size_t]
        L.vlo_flat_delete.restype = C.c_int
        L.vlo_flat_delete.argtypes = [C.c_void_p, C.c_uint64]
        L.vlo_flat_len.restype = C.c_size_t
        L.vlo_flat_len.argtypes = [C.c_void_p]
        L.vlo_flat_dim.restype = C.c_size_t
        L.vlo_flat_dim.argtypes = [C.c_void_p]
        L.vlo_flat_get.restype = C.c_int
        L.vlo_flat_get.argtypes = [C.c_void_p, C.c_uint64, dp]
        L.vlo_flat_search.restype = C.c_int
        L.vlo_flat_search.argtypes = [C.c_void_p, dp, C.c_size_t, C.c_size_t, C.c_int, u64p, dp,
                                      C.POINTER(C.c_size_t)]
        L.vlo_hnsw_postprocess.restype = C.c_size_t
        L.vlo_hnsw_postprocess.argtypes = [u64p, u64p, dp, C.c_size_t, C.c_size_t, C.c_int]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def calculate(metric: int, a, b) -> float:
    a, b = _f64(a), _f64(b)
    assert a.shape == b.shape, "Vectors must have the same length"  # src/lib.rs:382
    return lib().vlo_calculate(metric, _dp(a), _dp(b), a.size)


def hnsw_distance(metric: int, a, b) -> int:
    a, b = _f64(a), _f64(b)
    return int(lib().vlo_hnsw_distance(metric, _dp(a), _dp(b), min(a.size, b.size)))


def convert_distance_to_similarity(distance: float, metric: int) -> float:
    return lib().vlo_convert_distance_to_similarity(float(distance), metric)


def hnsw_score(d: int, metric: int) -> float:
    return lib().vlo_hnsw_score(int(d), metric)


def hnsw_postprocess(ids, dists, k: int, metric: int) -> Tuple[np.ndarray, np.ndarray]:
    ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint64)).copy()
    dists = np.ascontiguousarray(np.asarray(dists, dtype=np.uint64))
    scores = np.empty(max(ids.size, 1), dtype=np.float64)
    m = lib().vlo_hnsw_postprocess(_u64p(ids), _u64p(dists), _dp(scores), ids.size, k, metric)
    return ids[:m].copy(), scores[:m].copy()


def embed_f32(emb, normalize: bool = True) -> np.ndarray:
    """EmbeddingGenerator's post-processing (src/embeddings.rs:169-181) of [n, dim] (or [dim]) f32 model output."""
    e = np.ascontiguousarray(np.asarray(emb, dtype=np.float32))
    rows = e.reshape(-1, e.shape[-1]) if e.ndim > 1 else e.reshape(1, -1)
    out = np.empty(rows.shape, dtype=np.float64)
    L = lib()
    for i in range(rows.shape[0]):
        L.vlo_embed_f32(rows[i].ctypes.data_as(C.POINTER(C.c_float)), rows.shape[1], 1 if normalize else 0,
                        out[i].ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(e.shape)


class OracleError(Exception):
    def __init__(self, code: int, detail=None):
        super().__init__(f"oracle status {code} detail={detail}")
        self.code = code
        self.detail = detail


class FlatOracle:
    """FlatIndex restatement (src/index/flat.rs:60-135)."""

    def __init__(self, dim: int, ids=None, values=None):
        ids = np.ascontiguousarray(np.asarray(ids if ids is not None else [], dtype=np.uint64))
        values = _f64(values if values is not None else np.zeros((0, dim)))
        n = ids.size
        assert values.size == n * dim
        self._h = lib().vlo_flat_new(dim, _u64p(ids), _dp(values), n)
        if not self._h:
            raise MemoryError("vlo_flat_new")
        self.dim = dim

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            _lib.vlo_flat_free(h)
            self._h = None

    def extend(self, ids, values) -> None:
        """More rows without validation (FlatIndex::new's data, appended chunk by chunk)."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint64))
        values = _f64(values)
        assert values.size == ids.size * self.dim
        if lib().vlo_flat_extend(self._h, _u64p(ids), _dp(values), ids.size) != OK:
            raise MemoryError("vlo_flat_extend")

    def add(self, id: int, values) -> None:
        v = _f64(values)
        rc = lib().vlo_flat_add(self._h, id, _dp(v), v.size)
        if rc != OK:
            raise OracleError(rc)

    def delete(self, id: int) -> None:
        lib().vlo_flat_delete(self._h, id)

    def __len__(self) -> int:
        return lib().vlo_flat_len(self._h)

    def get_vector(self, id: int):
        out = np.empty(self.dim, dtype=np.float64)
        rc = lib().vlo_flat_get(self._h, id, _dp(out))
        return out if rc == OK else None

    def search(self, query, k: int, metric: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _f64(query)
        m = min(k, len(self))
        ids = np.empty(max(m, 1), dtype=np.uint64)
        scores = np.empty(max(m, 1), dtype=np.float64)
        n_out = C.c_size_t(0)
        rc = lib().vlo_flat_search(self._h, _dp(q), q.size, k, metric, _u64p(ids), _dp(scores),
                                   C.byref(n_out))
        if rc == DIM_MISMATCH:
            raise OracleError(rc, {"expected": n_out.value, "actual": q.size})
        if rc != OK:
            raise OracleError(rc)
        return ids[: n_out.value].copy(), scores[: n_out.value].copy()


# ---------------------------------------------------------------------------
# Pure-Python second restatement (small cases only): independent of the C
# file, used by tests to cross-check it.
# ---------------------------------------------------------------------------

def py_calculate(metric: int, a: Sequence[float], b: Sequence[float]) -> float:
    a = [float(x) for x in a]
    b = [float(x) for x in b]
    assert len(a) == len(b)
    if metric == COSINE:  # src/lib.rs:425-444
        dot = na = nb = 0.0
        for x, y in zip(a, b):
            dot += x * y
            na += x * x
            nb += y * y
        norm_a, norm_b = math.sqrt(na), math.sqrt(nb)
        if norm_a == 0.0 or norm_b == 0.0:
            return 0.0
        return dot / (norm_a * norm_b)
    if metric == EUCLIDEAN:  # src/lib.rs:476-489
        s = -0.0
        for x, y in zip(a, b):
            d = x - y
            s += d * d
        return 1.0 / (1.0 + math.sqrt(s))
    if metric == MANHATTAN:  # src/lib.rs:521-532
        s = -0.0
        for x, y in zip(a, b):
            s += abs(x - y)
        return 1.0 / (1.0 + s)
    s = -0.0  # src/lib.rs:565-572
    for x, y in zip(a, b):
        s += x * y
    return s


def py_flat_search(rows: List[Tuple[int, Sequence[float]]], query, k: int, metric: int):
    """src/index/flat.rs:106-117 on a list of (id, values)."""
    scored = [(py_calculate(metric, v, query), i, rid) for i, (rid, v) in enumerate(rows)]
    if len(scored) >= 2 and any(s != s for s, _, _ in scored):
        raise OracleError(NAN_PANIC)
    # stable descending: Python's sort is stable; key = -score keeps storage order on ties
    scored.sort(key=lambda t: -t[0])
    top = scored[:k]
    return [t[2] for t in top], [t[0] for t in top]


class HnswCpuWalker:
    """The "CPU HNSW": single-threaded walks of a graph exported by vl_index_hnsw_graph_export, with the reference's
    u64 distance callbacks (oracle/vl_hnsw_cpu.c).  `graph` is what vectorlite_amd.HNSWIndex.graph(with_rows=True)
    returns.  Checker only."""

    def __init__(self, graph, metric: int):
        L = lib()
        vp = C.c_void_p
        L.vlo_hnsw_walk.restype = C.c_size_t
        L.vlo_hnsw_walk.argtypes = [C.c_int, vp, C.c_size_t, C.c_size_t, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32,
                                    C.c_uint32, C.c_int, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
        self.L, self.g, self.metric = L, graph, int(metric)
        self.rows = np.ascontiguousarray(graph["rows"], dtype=np.float64)
        self.keep = {k: np.ascontiguousarray(graph[k]) for k in ("level", "upper_off", "cnt0", "nbr0", "cntU", "nbrU")}
        self.stamp = np.zeros(max(graph["n"], 1), np.uint32)
        self.epoch = C.c_uint32(0)
        self.evals = C.c_uint64(0)

    def search(self, q, ef: int, k: int):
        """(nodes [<= k], u64 distances) in ascending distance."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        out_n = np.zeros(max(k, 1), np.uint32)
        out_d = np.zeros(max(k, 1), np.uint64)
        g, kp = self.g, self.keep
        p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        got = self.L.vlo_hnsw_walk(self.metric, p(self.rows), self.rows.shape[1], g["n"], p(kp["level"]), p(kp["upper_off"]),
                                   p(kp["cnt0"]), p(kp["nbr0"]), g["m0"], p(kp["cntU"]), p(kp["nbrU"]), g["m"], g["entry"],
                                   g["max_level"], p(q), int(ef), int(k), p(self.stamp), C.byref(self.epoch), p(out_n),
                                   p(out_d), C.byref(self.evals))
        return out_n[:got].copy(), out_d[:got].copy()
