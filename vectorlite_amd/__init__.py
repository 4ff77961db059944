"""vectorlite_amd -- MI355X-native distance-scan engine behind VectorLite's index interface.

Python host-side mirror of the reference's operator interface for the ONE path this repo
implements (reference = mmailhos/vectorlite v0.1.5, paths relative to /root/reference):

    trait VectorIndex            src/lib.rs:224-245     -> FlatIndex.{add, delete, search, len, ...}
    struct Vector / SearchResult src/lib.rs:164-203     -> Vector / SearchResult
    enum SimilarityMetric        src/lib.rs:363-378     -> SimilarityMetric
    VectorLiteError variants     src/errors.rs:18,42    -> DimensionMismatch / MetricMismatch

Every numeric operation happens in libvectorlite_amd.so (HIP kernels, C ABI in
include/vectorlite_amd.h).  This module only marshals buffers and re-attaches text/metadata to
the k winners, which is what a Rust `impl VectorIndex for GpuFlatIndex` would do.
No CPU fallback exists: without the built library, or without a GPU, calls raise.
"""
from __future__ import annotations

import ctypes as C
import enum
import threading
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

__all__ = [
    "SimilarityMetric", "Vector", "SearchResult", "FlatIndex", "MultiFlatIndex", "HNSWIndex", "VectorLiteError", "DimensionMismatch",
    "MetricMismatch", "NaNScore", "DeviceError", "IndexOpError", "hnsw_score", "runtime_info",
    "PATH_FAST", "PATH_EXACT_SELECT", "PATH_EXACT_SORT",
]

VL_OK, VL_ERR_DIM_MISMATCH, VL_ERR_DUP_ID, VL_ERR_NOT_FOUND, VL_ERR_METRIC_MISMATCH = 0, 1, 2, 3, 4
VL_ERR_NAN_SCORE, VL_ERR_DEVICE, VL_ERR_OOM, VL_ERR_INVALID_ARG = 5, 6, 7, 8
PATH_NONE, PATH_FAST, PATH_EXACT_SELECT, PATH_EXACT_SORT = 0, 1, 2, 3


class SimilarityMetric(enum.IntEnum):
    """enum SimilarityMetric (src/lib.rs:363-378); Cosine is the default."""
    Cosine = 0
    Euclidean = 1
    Manhattan = 2
    DotProduct = 3

    @classmethod
    def default(cls) -> "SimilarityMetric":
        return cls.Cosine


@dataclass
class Vector:
    """struct Vector (src/lib.rs:164-174)."""
    id: int
    values: Sequence[float]
    text: str = ""
    metadata: Optional[Any] = None


@dataclass
class SearchResult:
    """struct SearchResult (src/lib.rs:194-203)."""
    id: int
    score: float
    text: str = ""
    metadata: Optional[Any] = None


class VectorLiteError(Exception):
    """enum VectorLiteError (src/errors.rs:11-67), the variants a search can return."""


class DimensionMismatch(VectorLiteError):
    def __init__(self, expected: int, actual: int):
        super().__init__(f"Dimension mismatch: expected {expected}, got {actual}")
        self.expected = expected
        self.actual = actual


class MetricMismatch(VectorLiteError):
    def __init__(self, requested, index):
        super().__init__(f"Metric mismatch: requested {requested!r}, index {index!r}")
        self.requested = requested
        self.index = index


class NaNScore(VectorLiteError):
    """Stands in for the reference's panic in `partial_cmp().unwrap()` (src/index/flat.rs:116)."""


class DeviceError(VectorLiteError):
    """HIP runtime failure or no GPU: there is no CPU fallback."""


class IndexOpError(ValueError):
    """add/delete return Err(String) in the reference (src/index/flat.rs:82-96); the message is the
    reference's text ("Vector dimension mismatch", "Vector ID {id} already exists")."""


def _last_error() -> str:
    msg = _lib.load().vl_last_error()
    return msg.decode() if msg else ""


def _raise(rc: int):
    if rc == VL_OK:
        return
    msg = _last_error()
    if rc == VL_ERR_DIM_MISMATCH:
        e, a = C.c_uint64(0), C.c_uint64(0)
        _lib.load().vl_last_dim_mismatch(C.byref(e), C.byref(a))
        raise DimensionMismatch(e.value, a.value)
    if rc == VL_ERR_NAN_SCORE:
        raise NaNScore(msg)
    if rc in (VL_ERR_DEVICE, VL_ERR_OOM):
        raise DeviceError(f"status {rc}: {msg}")
    if rc == VL_ERR_METRIC_MISMATCH:
        raise MetricMismatch(None, None)
    raise VectorLiteError(f"status {rc}: {msg}")


def runtime_info() -> Tuple[int, int]:
    """(number of visible HIP devices, ABI version)."""
    n, v = C.c_int(0), C.c_int(0)
    _lib.load().vl_runtime_info(C.byref(n), C.byref(v))
    return n.value, v.value


def hnsw_score(d_u64: int, metric: int) -> float:
    """convert_distance_to_similarity(d as f64 / 1000.0, metric) (src/index/hnsw.rs:51-75, :478-479)."""
    return _lib.load().vl_hnsw_score(int(d_u64), int(metric))


def last_path() -> int:
    return _lib.load().vl_last_path()


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _pf64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pu64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _add_embeddings(L, h, dim: int, ids, embeddings, normalize: bool, validate: bool) -> None:
    """vl_index_add_embeddings_f32: [n, dim] f32 numpy array or torch tensor (a CUDA/HIP tensor on the index's
    device is read in place); widened and L2-normalised on the device as src/embeddings.rs:171-179 does."""
    ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint64))
    on_device = False
    try:
        import torch
        is_tensor = isinstance(embeddings, torch.Tensor)
    except Exception:  # pragma: no cover
        is_tensor = False
    if is_tensor and embeddings.is_cuda:
        import torch
        if embeddings.dtype != torch.float32 or not embeddings.is_contiguous():
            raise ValueError("device embeddings must be a contiguous float32 tensor")
        torch.cuda.current_stream(embeddings.device).synchronize()  # producer kernels are done
        ptr, count, on_device = C.c_void_p(embeddings.data_ptr()), embeddings.numel(), True
    else:
        emb = np.ascontiguousarray(np.asarray(embeddings.numpy() if is_tensor else embeddings, dtype=np.float32))
        ptr, count = C.c_void_p(emb.ctypes.data), emb.size
    if count != ids.size * dim:
        raise ValueError("embeddings must be [n, dim]")
    rc = L.vl_index_add_embeddings_f32(h, _pu64(ids), ptr, ids.size, 1 if normalize else 0, 1 if validate else 0,
                                       1 if on_device else 0)
    if rc == VL_ERR_DUP_ID:
        raise IndexOpError(_last_error())
    _raise(rc)


class FlatIndex:
    """GPU-resident counterpart of `FlatIndex` (src/index/flat.rs:60-135).

    `FlatIndex(dim, data)` is `FlatIndex::new(dim, data)`: nothing is validated.
    """

    def __init__(self, dim: int, data: Sequence[Vector] = (), device: int = 0, _handle=None):
        self._L = _lib.load()
        self._meta: Dict[int, Tuple[str, Any]] = {}
        self._h = C.c_void_p()
        self._tls = threading.local()
        self.device = device
        if _handle is not None:
            self._h = _handle
            return
        data = list(data)
        if data:
            ids = np.ascontiguousarray(np.array([v.id for v in data], dtype=np.uint64))
            vals = _f64([list(v.values) for v in data]).reshape(len(data), -1)
            if vals.shape[1] != dim:
                raise ValueError("FlatIndex(dim, data): rows must have `dim` values in this binding")
            _raise(self._L.vl_flat_from_rows(dim, _pu64(ids), _pf64(vals), len(data), device, C.byref(self._h)))
            for v in data:
                self._meta.setdefault(int(v.id), (v.text, v.metadata))
        else:
            _raise(self._L.vl_flat_create(dim, device, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._L.vl_index_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- trait VectorIndex ------------------------------------------------------------------
    def add(self, vector: Vector) -> None:
        vals = _f64(vector.values).ravel()
        rc = self._L.vl_index_add(self._h, int(vector.id), _pf64(vals), vals.size)
        if rc in (VL_ERR_DIM_MISMATCH, VL_ERR_DUP_ID):
            raise IndexOpError(_last_error())
        _raise(rc)
        self._meta[int(vector.id)] = (vector.text, vector.metadata)

    def delete(self, id: int) -> None:
        _raise(self._L.vl_index_delete(self._h, int(id)))
        self._meta.pop(int(id), None)

    def search(self, query, k: int, similarity_metric: int = SimilarityMetric.Cosine) -> List[SearchResult]:
        ids, scores = self.search_arrays(query, k, similarity_metric)
        out = []
        for i, s in zip(ids.tolist(), scores.tolist()):
            text, md = self._meta.get(i, ("", None))
            out.append(SearchResult(id=i, score=s, text=text, metadata=md))
        return out

    def len(self) -> int:
        return int(self._L.vl_index_len(self._h))

    __len__ = len

    def is_empty(self) -> bool:
        return bool(self._L.vl_index_is_empty(self._h))

    def dimension(self) -> int:
        return int(self._L.vl_index_dimension(self._h))

    def get_vector(self, id: int) -> Optional[Vector]:
        out = np.empty(max(self.dimension(), 1), dtype=np.float64)
        rc = self._L.vl_index_get_vector(self._h, int(id), _pf64(out))
        if rc == VL_ERR_NOT_FOUND:
            return None
        _raise(rc)
        text, md = self._meta.get(int(id), ("", None))
        return Vector(id=int(id), values=out[: self.dimension()].tolist(), text=text, metadata=md)

    def max_id(self) -> Optional[int]:
        out = C.c_uint64(0)
        rc = self._L.vl_index_max_id(self._h, C.byref(out))
        if rc == VL_ERR_NOT_FOUND:
            return None
        _raise(rc)
        return out.value

    # ---- array-level entry points -----------------------------------------------------------
    def _out_buffers(self, k: int):
        """Per-thread output buffers of the single-query call (ids, scores, count) with their raw addresses:
        made once per thread, regrown only for a larger k.  vl_index_search_cap bounds what the library writes by
        the buffers' own capacity, whatever the index length is by the time the search runs."""
        tl = self._tls
        buf = getattr(tl, "buf", None)
        if buf is None or buf[0] < k:
            cap = max(64, min(int(k), max(self.len(), 1)))
            ids = np.empty(cap, dtype=np.uint64)
            scores = np.empty(cap, dtype=np.float64)
            n = C.c_uint64(0)
            buf = (cap, ids, scores, n, ids.ctypes.data, scores.ctypes.data, C.addressof(n))
            tl.buf = buf
        return buf

    def search_arrays(self, query, k: int, metric: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        q = query
        if not (type(q) is np.ndarray and q.dtype == np.float64 and q.ndim == 1 and q.flags.c_contiguous):
            q = _f64(query).ravel()
        k = min(max(int(k), 0), 1 << 62)
        cap, ids, scores, n, p_ids, p_scores, p_n = self._out_buffers(k)
        rc = self._L.vl_index_search_cap(self._h, q.ctypes.data, q.size, k, metric, cap, p_ids, p_scores, p_n)
        if rc:
            _raise(rc)
        m = n.value
        return ids[:m].copy(), scores[:m].copy()

    def search_positions(self, query, k: int, metric: int = 0):
        """(positions, ids, scores): positions are storage positions, for row-shard merging."""
        q = _f64(query).ravel()
        m = max(min(int(k), self.len()), 1)
        pos = np.empty(m, dtype=np.uint64)
        ids = np.empty(m, dtype=np.uint64)
        scores = np.empty(m, dtype=np.float64)
        n = C.c_uint64(0)
        _raise(self._L.vl_index_search_positions(self._h, _pf64(q), q.size, min(int(k), m), int(metric), _pu64(pos),
                                                 _pu64(ids), _pf64(scores), C.byref(n)))
        return pos[: n.value].copy(), ids[: n.value].copy(), scores[: n.value].copy()

    def search_batch(self, queries, k: int, metric: int = 0):
        """New capability (no reference counterpart): nq independent searches.
        Returns (ids [nq, k], scores [nq, k], n [nq]); row i is exactly search(queries[i])."""
        Q = _f64(queries)
        if Q.ndim != 2:
            raise ValueError("queries must be [nq, dim]")
        nq, qlen = Q.shape
        kk = max(min(int(k), self.len()), 1)   # min(k, len) results at most; k is clamped to the buffers
        kc = min(int(k), kk)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        _raise(self._L.vl_index_search_batch(self._h, _pf64(Q), nq, qlen, kc, int(metric), _pu64(ids),
                                             _pf64(scores), _pu64(n)))
        return ids[:, :kc], scores[:, :kc], n[:nq]

    def search_batch_device(self, queries, k: int, metric: int = 0, with_positions: bool = False):
        """search_batch for queries that are already on this index's GPU: `queries` is a contiguous float64 torch tensor
        [nq, dim] on that device (vl_index_search_batch_dev: no host staging, no PCIe copy of the queries on the MFMA
        batch path).  Returns (ids, scores, n), or (positions, ids, scores, n) with `with_positions`."""
        import torch
        if not isinstance(queries, torch.Tensor) or queries.dtype != torch.float64 or not queries.is_contiguous() \
                or queries.dim() != 2 or not queries.is_cuda:
            raise ValueError("queries must be a contiguous float64 [nq, dim] tensor on the index's GPU")
        nq, qlen = queries.shape
        kk = max(min(int(k), self.len()), 1)
        kc = min(int(k), kk)
        pos = np.zeros((nq, kk), dtype=np.uint64) if with_positions else None
        ids = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        torch.cuda.current_stream(queries.device).synchronize()  # whatever produced the queries has finished
        _raise(self._L.vl_index_search_batch_dev(self._h, C.c_void_p(queries.data_ptr()), nq, qlen, kc, int(metric),
                                                 _pu64(pos) if with_positions else None, _pu64(ids), _pf64(scores), _pu64(n)))
        if with_positions:
            return pos[:, :kc], ids[:, :kc], scores[:, :kc], n[:nq]
        return ids[:, :kc], scores[:, :kc], n[:nq]

    def search_batch_embeddings(self, embeddings, k: int, metric: int = 0, normalize: bool = True):
        """The caller's embed -> search step for a batch (src/client.rs:393-401): `embeddings` is [nq, dim] float32 as the
        model emits it -- a numpy array, or a contiguous torch tensor on this index's GPU.  Widening to f64 and the L2
        normalisation of src/embeddings.rs:169-181 run on the device, bit for bit, then vl_index_search_batch_dev.
        Returns (ids, scores, n) like search_batch."""
        on_device = False
        try:
            import torch
            is_tensor = isinstance(embeddings, torch.Tensor)
        except Exception:  # pragma: no cover
            is_tensor = False
        if is_tensor:
            import torch
            if embeddings.dtype != torch.float32 or not embeddings.is_contiguous() or embeddings.dim() != 2:
                raise ValueError("embeddings must be a contiguous float32 [nq, dim] tensor")
            if embeddings.is_cuda:
                on_device = True
                torch.cuda.current_stream(embeddings.device).synchronize()
                ptr = C.c_void_p(embeddings.data_ptr())
                nq, dim = embeddings.shape
            else:
                embeddings = embeddings.numpy()
        if not on_device:
            E = np.ascontiguousarray(np.asarray(embeddings, dtype=np.float32))
            if E.ndim != 2:
                raise ValueError("embeddings must be [nq, dim]")
            nq, dim = E.shape
            ptr = C.c_void_p(E.ctypes.data)
        kk = max(min(int(k), self.len()), 1)
        kc = min(int(k), kk)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        _raise(self._L.vl_index_search_batch_embeddings_f32(self._h, ptr, nq, dim, 1 if normalize else 0, 1 if on_device else 0,
                                                            kc, int(metric), _pu64(ids), _pf64(scores), _pu64(n)))
        return ids[:, :kc], scores[:, :kc], n[:nq]

    def search_batch_positions(self, queries, k: int, metric: int = 0):
        """(positions, ids, scores, n), each [nq, k] ([nq] for n): the batched search_positions."""
        Q = _f64(queries)
        if Q.ndim != 2:
            raise ValueError("queries must be [nq, dim]")
        nq, qlen = Q.shape
        kk = max(min(int(k), self.len()), 1)
        kc = min(int(k), kk)
        pos = np.zeros((nq, kk), dtype=np.uint64)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        _raise(self._L.vl_index_search_batch_positions(self._h, _pf64(Q), nq, qlen, kc, int(metric), _pu64(pos),
                                                       _pu64(ids), _pf64(scores), _pu64(n)))
        return pos[:, :kc], ids[:, :kc], scores[:, :kc], n[:nq]

    def add_rows(self, ids, values, validate: bool = True) -> None:
        """n x add() in one device pass.  `values`: [n, dim] f64 numpy array, or a torch CUDA/HIP
        tensor (f64, contiguous, on this index's device) which is ingested device-to-device."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint64))
        n = ids.size
        on_device = False
        try:
            import torch
            is_tensor = isinstance(values, torch.Tensor)
        except Exception:  # pragma: no cover
            is_tensor = False
        if is_tensor:
            import torch
            if values.dtype != torch.float64 or not values.is_contiguous():
                raise ValueError("device rows must be a contiguous float64 tensor")
            if values.numel() != n * self.dimension():
                raise ValueError("values must be [n, dim]")
            if values.is_cuda:
                torch.cuda.current_stream(values.device).synchronize()  # producer kernels are done
                on_device = True
                ptr = C.c_void_p(values.data_ptr())
            else:
                values = values.numpy()
        if not on_device:
            vals = _f64(values)
            if vals.size != n * self.dimension():
                raise ValueError("values must be [n, dim]")
            ptr = C.c_void_p(vals.ctypes.data)
        rc = self._L.vl_index_add_bulk(self._h, _pu64(ids), ptr, n, 1 if validate else 0, 1 if on_device else 0)
        if rc == VL_ERR_DUP_ID:
            raise IndexOpError(_last_error())
        _raise(rc)

    def add_embeddings(self, ids, embeddings, normalize: bool = True, validate: bool = True) -> None:
        """The ingest step in front of add (src/embeddings.rs:169-181 then src/index/flat.rs:82-91 per row): f32
        model output -> normalised f64 rows on the device -> n x add()."""
        _add_embeddings(self._L, self._h, self.dimension(), ids, embeddings, normalize, validate)

    def reserve(self, n_rows: int) -> None:
        _raise(self._L.vl_index_reserve(self._h, int(n_rows)))

    def clone(self) -> "FlatIndex":
        h = C.c_void_p()
        _raise(self._L.vl_index_clone(self._h, C.byref(h)))
        c = FlatIndex(self.dimension(), device=self.device, _handle=h)
        c._meta = dict(self._meta)
        return c

    def export(self) -> Tuple[np.ndarray, np.ndarray]:
        n, d = self.len(), self.dimension()
        ids = np.empty(max(n, 1), dtype=np.uint64)
        vals = np.empty((max(n, 1), max(d, 1)), dtype=np.float64)
        _raise(self._L.vl_index_export(self._h, _pu64(ids), _pf64(vals)))
        return ids[:n].copy(), vals[:n, :d].copy()

    def hnsw_distances(self, query, positions, metric: int) -> np.ndarray:
        """Metric::distance(query, row) -> u64 for the rows at `positions` (src/index/hnsw.rs:113-174)."""
        q = _f64(query).ravel()
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.uint64))
        out = np.empty(max(pos.size, 1), dtype=np.uint64)
        _raise(self._L.vl_index_hnsw_distances(self._h, _pf64(q), q.size, int(metric), _pu64(pos), pos.size,
                                               _pu64(out)))
        return out[: pos.size].copy()

    # ---- diagnostics ------------------------------------------------------------------------
    def force_path(self, path: int) -> None:
        _raise(self._L.vl_index_force_path(self._h, int(path)))

    def set_single_filter(self, mode: str) -> None:
        """"f32" (default) or "bf16": which copy of the slab single queries scan first."""
        _raise(self._L.vl_index_set_single_filter(self._h, {"f32": 0, "bf16": 1}[mode]))

    def set_coalescing(self, max_batch: int, window_us: int = 0) -> None:
        """Answer concurrent search() calls (other threads) with shared slab passes; 0 turns it off."""
        _raise(self._L.vl_index_set_coalescing(self._h, int(max_batch), int(window_us)))

    def coalesce_stats(self) -> Tuple[int, int]:
        b, q = C.c_uint64(0), C.c_uint64(0)
        _raise(self._L.vl_index_coalesce_stats(self._h, C.byref(b), C.byref(q)))
        return int(b.value), int(q.value)

    def coalesce_gather(self, adaptive: Optional[bool] = None) -> Tuple[int, int]:
        """Switch the coalescer's adaptive gather (None leaves it) and read (passes whose leader waited, microseconds waited)."""
        w, us = C.c_uint64(0), C.c_uint64(0)
        _raise(self._L.vl_index_coalesce_gather(self._h, -1 if adaptive is None else int(bool(adaptive)), C.byref(w), C.byref(us)))
        return int(w.value), int(us.value)

    def last_scan(self) -> Dict[str, int]:
        """Which k_scan instantiation / grid answered the last single search (vl_index_last_scan)."""
        v, g, q = C.c_int(0), C.c_int(0), C.c_int(0)
        _raise(self._L.vl_index_last_scan(self._h, C.byref(v), C.byref(g), C.byref(q)))
        return {"variant": int(v.value), "grid": int(g.value), "query_in_kernarg": int(q.value)}

    def last_filter(self) -> Dict[str, int]:
        """The batch filter's last launch sequence on this handle (vl_index_last_filter)."""
        v = (C.c_int * 6)()
        _raise(self._L.vl_index_last_filter(self._h, v))
        return {"ksteps": v[0], "metric": v[1], "chunks": v[2], "grid_x": v[3], "stages": v[4], "sample_blocks": v[5]}

    def profile_enable(self, on: bool) -> None:
        _raise(self._L.vl_index_profile_enable(self._h, 1 if on else 0))

    def profile_read(self) -> Tuple[int, float, int]:
        n, ms, b = C.c_uint64(0), C.c_double(0.0), C.c_uint64(0)
        _raise(self._L.vl_index_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(b)))
        return n.value, ms.value, b.value


class MultiFlatIndex(FlatIndex):
    """ONE flat index over several GPUs in one process (vl_flat_create_multi): the same `VectorIndex` surface as
    `FlatIndex`, every method inherited -- only the handle differs.  mode "replicas": every GPU holds every row,
    concurrent searches are dealt to the least busy replica, batches are cut across them.  mode "row_shards": every
    GPU holds part of the rows, every search runs on all of them and the exact per-shard top-k are merged on the
    device, bit-identical to one index holding every row.  A device listed twice = two parts on that card."""

    REPLICAS, ROW_SHARDS = 0, 1

    def __init__(self, dim: int, devices: Sequence[int], mode="replicas"):
        m = {"replicas": 0, "row_shards": 1}.get(mode, mode)
        if m not in (0, 1):
            raise ValueError("mode must be 'replicas' or 'row_shards'")
        L = _lib.load()
        devs = [int(d) for d in devices]
        arr = (C.c_int * max(len(devs), 1))(*devs)
        h = C.c_void_p()
        _raise(L.vl_flat_create_multi(int(dim), arr, len(devs), int(m), C.byref(h)))
        super().__init__(dim, device=devs[0] if devs else 0, _handle=h)
        self.devices = devs
        self.mode = int(m)

    def parts(self) -> Dict[str, Any]:
        """Rows held and searches answered by each part (vl_index_parts)."""
        n, mode = C.c_int(0), C.c_int(0)
        _raise(self._L.vl_index_parts(self._h, C.byref(n), C.byref(mode), None, None, 0))
        rows = np.zeros(max(n.value, 1), dtype=np.uint64)
        srch = np.zeros(max(n.value, 1), dtype=np.uint64)
        _raise(self._L.vl_index_parts(self._h, C.byref(n), C.byref(mode), _pu64(rows), _pu64(srch), int(n.value)))
        return {"n_parts": int(n.value), "mode": int(mode.value), "rows": rows[: n.value].tolist(), "searches": srch[: n.value].tolist()}

    def clone(self) -> "MultiFlatIndex":
        h = C.c_void_p()
        _raise(self._L.vl_index_clone(self._h, C.byref(h)))
        c = MultiFlatIndex.__new__(MultiFlatIndex)
        FlatIndex.__init__(c, self.dimension(), device=self.device, _handle=h)
        c.devices, c.mode = list(self.devices), self.mode
        c._meta = dict(self._meta)
        return c


class HNSWIndex:
    """GPU counterpart of `HNSWIndex` (src/index/hnsw.rs:197-496), default profile M = 16, M0 = 32.

    The distance callbacks (u64 = trunc(dist * 1000)), the score conversion, ef = min(k, len), the
    tombstone deletes and every error are the reference's; the graph walk is this library's own
    (crate hnsw 0.11.0 is not in the reference tree), so results are approximate: judged by recall."""

    def __init__(self, dim: int, metric: int = SimilarityMetric.Cosine, device: int = 0, m: int = 16, m0: int = 32,
                 ef_construction: int = 400, seed: int = 0, _handle=None):
        self._L = _lib.load()
        self._meta: Dict[int, Tuple[str, Any]] = {}
        self._h = C.c_void_p()
        self.device = device
        if _handle is not None:  # adopt a handle made by the library (vl_vlc_build_index)
            self._h = _handle
            return
        if dim == 0:
            raise ValueError("HNSW index dimension cannot be 0")  # panics in the reference (:217-219)
        _raise(self._L.vl_hnsw_create_ex(dim, int(metric), m, m0, ef_construction, seed, device, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._L.vl_index_destroy(h)
            except Exception:
                pass
            self._h = None

    def set_coalescing(self, max_batch: int, window_us: int = 0) -> None:
        """Concurrent search() calls (other threads) share graph-walk launches; 0 turns it off."""
        _raise(self._L.vl_index_set_coalescing(self._h, int(max_batch), int(window_us)))

    def coalesce_stats(self) -> Tuple[int, int]:
        b, q = C.c_uint64(0), C.c_uint64(0)
        _raise(self._L.vl_index_coalesce_stats(self._h, C.byref(b), C.byref(q)))
        return int(b.value), int(q.value)

    def coalesce_gather(self, adaptive: Optional[bool] = None) -> Tuple[int, int]:
        """Switch the coalescer's adaptive gather (None leaves it) and read (passes whose leader waited, microseconds waited)."""
        w, us = C.c_uint64(0), C.c_uint64(0)
        _raise(self._L.vl_index_coalesce_gather(self._h, -1 if adaptive is None else int(bool(adaptive)), C.byref(w), C.byref(us)))
        return int(w.value), int(us.value)

    def graph(self, with_rows: bool = False) -> Dict[str, Any]:
        """The graph as it stands (vl_index_hnsw_graph_info / _export): entry, max_level, m, m0 and the arrays
        level[n], upper_off[n], cnt0[n], nbr0[n, m0], cntU[slots], nbrU[slots, m], node_ids[n], live[n] (+ rows[n, dim])."""
        n, slots = C.c_uint64(0), C.c_uint64(0)
        entry, m, m0 = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        lvl = C.c_int(0)
        _raise(self._L.vl_index_hnsw_graph_info(self._h, C.byref(n), C.byref(entry), C.byref(lvl), C.byref(m), C.byref(m0),
                                                C.byref(slots)))
        nn, ns, d = int(n.value), int(slots.value), self.dimension()
        g = {"entry": int(entry.value), "max_level": int(lvl.value), "m": int(m.value), "m0": int(m0.value), "n": nn,
             "level": np.zeros(max(nn, 1), np.uint8), "upper_off": np.zeros(max(nn, 1), np.uint32),
             "cnt0": np.zeros(max(nn, 1), np.uint32), "nbr0": np.zeros((max(nn, 1), int(m0.value)), np.uint32),
             "cntU": np.zeros(max(ns, 1), np.uint32), "nbrU": np.zeros((max(ns, 1), int(m.value)), np.uint32),
             "node_ids": np.zeros(max(nn, 1), np.uint64), "live": np.zeros(max(nn, 1), np.uint8)}
        rows = np.zeros((max(nn, 1), d), np.float64) if with_rows else None
        vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p()  # noqa: E731
        _raise(self._L.vl_index_hnsw_graph_export(self._h, vp(g["level"]), vp(g["upper_off"]), vp(g["cnt0"]), vp(g["nbr0"]),
                                                  vp(g["cntU"]), vp(g["nbrU"]), vp(g["node_ids"]), vp(g["live"]), vp(rows)))
        if with_rows:
            g["rows"] = rows[:nn]
        for key in ("level", "upper_off", "cnt0", "nbr0", "node_ids", "live"):
            g[key] = g[key][:nn]
        g["cntU"], g["nbrU"] = g["cntU"][:ns], g["nbrU"][:ns]
        return g

    def set_min_beam(self, min_beam: int) -> None:
        """Opt-in beam floor of searches that name no ef; default 0 = the reference's strict ef = min(k, len)."""
        _raise(self._L.vl_index_hnsw_set_min_beam(self._h, int(min_beam)))

    def walk_stats(self) -> Tuple[int, int]:
        """(queries walked, distance evaluations made for them) since creation."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _raise(self._L.vl_index_hnsw_walk_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def clone(self) -> "HNSWIndex":
        """Deep copy (rows + device graph, tombstones included): #[derive(Clone)] on HNSWIndex."""
        h = C.c_void_p()
        _raise(self._L.vl_index_clone(self._h, C.byref(h)))
        c = HNSWIndex(self.dimension(), device=self.device, _handle=h)
        c._meta = dict(self._meta)
        return c

    def export(self) -> Tuple[np.ndarray, np.ndarray]:
        """(ids, values) of the live rows in insertion order: the serialised `vector_values`."""
        n, d = self.len(), self.dimension()
        ids = np.empty(max(n, 1), dtype=np.uint64)
        vals = np.empty((max(n, 1), max(d, 1)), dtype=np.float64)
        _raise(self._L.vl_index_export(self._h, _pu64(ids), _pf64(vals)))
        return ids[:n].copy(), vals[:n, :d].copy()

    def metric(self) -> SimilarityMetric:
        m = C.c_int(0)
        _raise(self._L.vl_index_metric(self._h, C.byref(m)))
        return SimilarityMetric(m.value)

    def add(self, vector: Vector) -> None:
        vals = _f64(vector.values).ravel()
        rc = self._L.vl_index_add(self._h, int(vector.id), _pf64(vals), vals.size)
        if rc in (VL_ERR_DIM_MISMATCH, VL_ERR_DUP_ID):
            raise IndexOpError(_last_error())
        _raise(rc)
        self._meta[int(vector.id)] = (vector.text, vector.metadata)

    def add_rows(self, ids, values) -> None:
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint64))
        on_device = False
        try:
            import torch
            is_tensor = isinstance(values, torch.Tensor)
        except Exception:  # pragma: no cover
            is_tensor = False
        if is_tensor and values.is_cuda:
            import torch
            if values.dtype != torch.float64 or not values.is_contiguous():
                raise ValueError("device rows must be a contiguous float64 tensor")
            torch.cuda.current_stream(values.device).synchronize()
            ptr, on_device = C.c_void_p(values.data_ptr()), True
            count = values.numel()
        else:
            vals = _f64(values.numpy() if is_tensor else values)
            ptr, count = C.c_void_p(vals.ctypes.data), vals.size
        if count != ids.size * self.dimension():
            raise ValueError("values must be [n, dim]")
        rc = self._L.vl_index_add_bulk(self._h, _pu64(ids), ptr, ids.size, 1, 1 if on_device else 0)
        if rc == VL_ERR_DUP_ID:
            raise IndexOpError(_last_error())
        _raise(rc)

    def add_embeddings(self, ids, embeddings, normalize: bool = True) -> None:
        """f32 model output -> normalised f64 rows on the device (src/embeddings.rs:169-181) -> n x add()."""
        _add_embeddings(self._L, self._h, self.dimension(), ids, embeddings, normalize, True)

    def delete(self, id: int) -> None:
        rc = self._L.vl_index_delete(self._h, int(id))
        if rc == VL_ERR_NOT_FOUND:
            raise IndexOpError(_last_error())  # "Vector ID {id} does not exist" (:401-403)
        _raise(rc)
        self._meta.pop(int(id), None)

    def _raise_search(self, rc: int, requested: int):
        if rc == VL_ERR_METRIC_MISMATCH:
            raise MetricMismatch(SimilarityMetric(int(requested)), self.metric())
        _raise(rc)

    def search_batch(self, queries, k: int, metric: int, ef: int = 0):
        """(ids [nq, k], scores [nq, k], n [nq]); ef = 0 is the reference's ef = min(k, len)."""
        Q = _f64(queries)
        if Q.ndim == 1:
            Q = Q[None, :]
        nq, qlen = Q.shape
        # buffers hold min(k, len) entries per query and k is clamped to them (a prefix of the same ranking): a huge k
        # costs nothing, and an add() from another thread after len() was read cannot make the library overrun them
        kk = max(min(int(k), self.len()), 1)
        kc = min(int(k), kk)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        rc = self._L.vl_index_search_ef(self._h, _pf64(Q), nq, qlen, kc, int(ef), int(metric), _pu64(ids),
                                        _pf64(scores), _pu64(n))
        self._raise_search(rc, metric)
        return ids[:, :kc], scores[:, :kc], n[:nq]

    def search_arrays(self, query, k: int, metric: int, ef: int = 0):
        ids, scores, n = self.search_batch(_f64(query).ravel()[None, :], k, metric, ef)
        m = int(n[0])
        return ids[0, :m].copy(), scores[0, :m].copy()

    def search(self, query, k: int, similarity_metric: int) -> List[SearchResult]:
        q = _f64(query).ravel()
        kk = max(min(int(k), self.len()), 1)
        ids = np.zeros(kk, dtype=np.uint64)
        scores = np.zeros(kk, dtype=np.float64)
        n = C.c_uint64(0)
        rc = self._L.vl_index_search(self._h, _pf64(q), q.size, min(int(k), kk), int(similarity_metric), _pu64(ids),
                                     _pf64(scores), C.byref(n))
        self._raise_search(rc, similarity_metric)
        out = []
        for i, s in zip(ids[: n.value].tolist(), scores[: n.value].tolist()):
            text, md = self._meta.get(i, ("", None))
            out.append(SearchResult(id=i, score=s, text=text, metadata=md))
        return out

    def len(self) -> int:
        return int(self._L.vl_index_len(self._h))

    __len__ = len

    def is_empty(self) -> bool:
        return bool(self._L.vl_index_is_empty(self._h))

    def dimension(self) -> int:
        return int(self._L.vl_index_dimension(self._h))

    def get_vector(self, id: int) -> Optional[Vector]:
        out = np.empty(self.dimension(), dtype=np.float64)
        rc = self._L.vl_index_get_vector(self._h, int(id), _pf64(out))
        if rc == VL_ERR_NOT_FOUND:
            return None
        _raise(rc)
        text, md = self._meta.get(int(id), ("", None))
        return Vector(id=int(id), values=out.tolist(), text=text, metadata=md)

    def max_id(self) -> Optional[int]:
        out = C.c_uint64(0)
        rc = self._L.vl_index_max_id(self._h, C.byref(out))
        if rc == VL_ERR_NOT_FOUND:
            return None
        _raise(rc)
        return out.value
