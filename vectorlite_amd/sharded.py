"""Row-sharded flat index across the GPUs of one node: host harness over the C ABI's vl_comm_* / vl_shard_*
entry points (include/vectorlite_amd.h; design in csrc/shard.hpp).

No reference counterpart: the reference is single-process.  north_star shards the
*batched* flat search by rows: rank r holds the contiguous row range [offset_r, offset_r + n_r) of the corpus,
every rank scans its own shard with the same HIP path as the single-GPU index, ONE all-gather exchanges the
per-shard exact top-k, and a device kernel merges by (score desc, GLOBAL position asc) -- identical to one index
holding all rows (src/index/flat.rs:116), bit for bit, because every shard returns exact f64 scores.

Everything numeric is in libvectorlite_amd.so.  This module only (1) hands rank 0's ncclUniqueId to the other
ranks (torch.distributed's store / broadcast -- the job a Rust server's own RPC would do) and (2) offers two
transports for the one exchange:

  transport="rccl"   vl_shard_search_batch: local search -> ncclAllGather inside the library -> device merge.
                     The production form (one process per GPU, xGMI).
  transport="torch"  vl_shard_search_local -> torch.distributed.all_gather_into_tensor (gloo in the tests, where
                     several ranks share one card and RCCL refuses that) -> vl_shard_merge (the same device
                     merge kernel).

Exchange size: 8 * (4 + nq + 3 * nq * ks) bytes per rank (config 3: 1024 queries, k = 10 -> 254 KB per rank):
latency-bound, so it is a single collective, not a ring of small ones.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib

__all__ = ["ShardedFlatIndex", "OneProcessShards", "Comm", "shard_ranges", "pack_words", "unpack_record"]

VL_COMM_ID_BYTES = 128
SHARD_HDR_WORDS = 4


def shard_ranges(n_rows: int, world: int):
    """Contiguous, near-equal row ranges: rank r gets [starts[r], starts[r+1])."""
    base, rem = divmod(n_rows, world)
    starts = [0]
    for r in range(world):
        starts.append(starts[-1] + base + (1 if r < rem else 0))
    return starts


def _is_device_tensor(x) -> bool:
    try:
        import torch
    except Exception:  # pragma: no cover
        return False
    return isinstance(x, torch.Tensor) and x.is_cuda


def pack_words(nq: int, ks: int) -> int:
    """vl_shard_packed_words: u64 words of one rank's exchange record."""
    return SHARD_HDR_WORDS + nq + 3 * nq * ks


def unpack_record(rec: np.ndarray, nq: int, ks: int):
    """One exchange record -> (status, shard_len, dim, counts [nq], scores [nq, ks] f64, gpos [nq, ks], ids [nq, ks])."""
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    cnt = rec[SHARD_HDR_WORDS: SHARD_HDR_WORDS + nq]
    body = rec[SHARD_HDR_WORDS + nq:].reshape(3, nq, ks)
    return int(rec[0]), int(rec[1]), int(rec[2]), cnt, body[0].view(np.float64), body[1], body[2]


def _pu64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _pf64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Comm:
    """vl_comm: one rank's end of the RCCL communicator (ncclCommInitRank from a caller-supplied ncclUniqueId)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int):
        from . import _raise
        self._L = _lib.load()
        self._h = C.c_void_p()
        if len(unique_id) != VL_COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes vl_comm_unique_id returned on rank 0")
        buf = (C.c_uint8 * VL_COMM_ID_BYTES).from_buffer_copy(unique_id)
        _raise(self._L.vl_comm_create(buf, int(world), int(rank), int(device), C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        from . import _raise
        buf = (C.c_uint8 * VL_COMM_ID_BYTES)()
        _raise(_lib.load().vl_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device: int, group=None) -> "Comm":
        """Every rank of an initialised torch.distributed group calls this: rank 0's id travels by broadcast."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank, device)

    @property
    def world(self) -> int:
        return int(self._L.vl_comm_world(self._h))

    @property
    def rank(self) -> int:
        return int(self._L.vl_comm_rank(self._h))

    def profile_enable(self, on: bool = True) -> None:
        self._L.vl_comm_profile_enable(self._h, 1 if on else 0)

    def profile_read(self) -> dict:
        """Totals since the last read: calls, local search ms (host clock), and on the exchange stream (HIP events)
        the record's H2D copy, the ncclAllGather, the merge kernel + D2H."""
        calls, a, b, c, d = C.c_uint64(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        self._L.vl_comm_profile_read(self._h, C.byref(calls), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return {"calls": int(calls.value), "local_ms": a.value, "h2d_ms": b.value, "allgather_ms": c.value, "merge_ms": d.value}

    def record_paths(self) -> dict:
        """Batches whose exchange record the finalize kernel wrote on the device / that built it on the host."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._L.vl_comm_record_paths(self._h, C.byref(a), C.byref(b))
        return {"on_device": int(a.value), "via_host": int(b.value)}

    def close(self):
        if getattr(self, "_h", None):
            self._L.vl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedFlatIndex:
    """One rank's view of a row-sharded flat index.

    `local`: this rank's shard, a `vectorlite_amd.FlatIndex` on this rank's GPU.
    transport "rccl": `comm` is a `Comm`; offsets come from vl_shard_sync.
    transport "torch": torch.distributed (any backend) moves the records; the merge still runs on the device.
    Call `sync()` after building the shards and after any add/delete."""

    def __init__(self, local, comm: Comm = None, transport: str = None, group=None, gather_device=None):
        self._L = _lib.load()
        self.local = local
        self.comm = comm
        self.transport = transport or ("rccl" if comm is not None else "torch")
        self.group = group
        self.gather_device = gather_device  # torch transport: where the gathered tensor lives (None = CPU, gloo)
        self.offset = 0
        self.total = 0
        self.max_len = 0
        if self.transport == "rccl":
            if comm is None:
                raise ValueError('transport "rccl" needs a Comm')
            self.world, self.rank = comm.world, comm.rank
        elif self.transport == "torch":
            import torch.distributed as dist
            self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
            self.world = self._dist.get_world_size(group) if self._dist else 1
            self.rank = self._dist.get_rank(group) if self._dist else 0
        else:
            raise ValueError("transport must be 'rccl' or 'torch'")
        self.sync()

    # ---- the table every rank agrees on ---------------------------------------------------------------
    def _torch_all_gather(self, words: np.ndarray) -> np.ndarray:
        """u64 [m] -> [world, m] through torch.distributed (one collective)."""
        if self.world == 1:
            return words[None, :].copy()
        import torch
        t = torch.from_numpy(words.view(np.int64))
        if self.gather_device is not None:
            t = t.to(self.gather_device)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=t.device)
        self._dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().view(np.uint64).reshape(self.world, -1)

    def sync(self) -> Tuple[int, int]:
        """Collective.  Returns (this shard's global offset, total rows)."""
        from . import _raise
        if self.transport == "rccl":
            off, tot = C.c_uint64(0), C.c_uint64(0)
            _raise(self._L.vl_shard_sync(self.local._h, self.comm._h, C.byref(off), C.byref(tot)))
            self.offset, self.total = int(off.value), int(tot.value)
            self.max_len = None  # kept inside the library
        else:
            mine = np.array([len(self.local), self.local.dimension()], dtype=np.uint64)
            allv = self._torch_all_gather(mine)
            if len(set(allv[:, 1].tolist())) != 1:
                raise ValueError(f"shards disagree on the dimension: {allv[:, 1].tolist()}")
            lens = allv[:, 0].astype(np.int64)
            self.offset, self.total, self.max_len = int(lens[: self.rank].sum()), int(lens.sum()), int(lens.max())
        return self.offset, self.total

    def global_len(self) -> int:
        return self.total

    # ---- search ---------------------------------------------------------------------------------------
    def search_batch(self, queries, k: int, metric: int = 0, with_positions: bool = False):
        """nq independent searches over the whole (sharded) corpus.
        Returns (ids [nq, k], scores [nq, k], n [nq]) -- identical on every rank (+ gpos [nq, k] on request)."""
        from . import _raise
        on_device = _is_device_tensor(queries)  # a contiguous float64 [nq, dim] torch tensor on this shard's GPU: no host staging
        if on_device:
            import torch
            if queries.dtype != torch.float64 or not queries.is_contiguous() or queries.dim() != 2:
                raise ValueError("device queries must be a contiguous float64 [nq, dim] tensor")
            torch.cuda.current_stream(queries.device).synchronize()  # whatever produced them has finished
            Q, qptr = queries, C.c_void_p(queries.data_ptr())
        else:
            Q = np.ascontiguousarray(np.asarray(queries, dtype=np.float64))
            if Q.ndim == 1:
                Q = Q[None, :]
        nq, qlen = Q.shape
        k = int(k)
        kk = max(min(k, max(self.total, 1)), 1)  # output row stride: min(k, total rows) results at most
        ids = np.zeros((nq, kk), dtype=np.uint64)
        gpos = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        k_call = min(k, kk)  # results for a smaller k are a prefix: the buffers can never be overrun
        if self.transport == "rccl":
            if on_device:
                _raise(self._L.vl_shard_search_batch_dev(self.local._h, self.comm._h, qptr, nq, qlen, k_call, int(metric),
                                                         _pu64(gpos), _pu64(ids), _pf64(scores), _pu64(n)))
            else:
                _raise(self._L.vl_shard_search_batch(self.local._h, self.comm._h, _pf64(Q), nq, qlen, k_call, int(metric),
                                                     _pu64(gpos), _pu64(ids), _pf64(scores), _pu64(n)))
        elif k_call > 0 and self.total > 0 and nq > 0:
            ks = min(k_call, self.max_len)
            if on_device:
                rec = np.zeros(pack_words(nq, ks), dtype=np.uint64)
                _raise(self._L.vl_shard_search_local_dev(self.local._h, self.offset, 1 if self.total else 0, qptr, nq, qlen,
                                                         ks, int(metric), _pu64(rec)))
            else:
                rec = self._local_record(Q, ks, int(metric))
            gathered = np.ascontiguousarray(self._torch_all_gather(rec))
            self._merge(gathered, nq, ks, k_call, gpos, ids, scores, n)
        out = (ids[:, :k_call], scores[:, :k_call], n[:nq])
        return out + (gpos[:, :k_call],) if with_positions else out

    # the two halves around the "torch" transport's exchange; both are one C-ABI call
    def _local_record(self, Q: np.ndarray, ks: int, metric: int) -> np.ndarray:
        """vl_shard_search_local: this shard's exchange record for the batch (status in word 0)."""
        from . import _raise
        nq, qlen = Q.shape
        rec = np.zeros(pack_words(nq, ks), dtype=np.uint64)
        _raise(self._L.vl_shard_search_local(self.local._h, self.offset, 1 if self.total else 0, _pf64(Q), nq, qlen, ks,
                                             metric, _pu64(rec)))
        return rec

    def _merge(self, gathered: np.ndarray, nq: int, ks: int, k: int, gpos, ids, scores, n) -> None:
        """vl_shard_merge: the device merge kernel over the `world` gathered records."""
        from . import _raise
        _raise(self._L.vl_shard_merge(int(self.local.device), _pu64(gathered), self.world, nq, ks, k, _pu64(gpos),
                                      _pu64(ids), _pf64(scores), _pu64(n)))

    def search(self, query, k: int, metric: int = 0):
        ids, scores, n = self.search_batch(np.asarray(query, dtype=np.float64)[None, :], k, metric)
        m = int(n[0])
        return ids[0, :m].copy(), scores[0, :m].copy()


class OneProcessShards:
    """The row shards of one corpus as separate flat handles in ONE process (one card, or several): a batch is answered by
    vl_shard_search_local(_dev) on every shard and ONE vl_shard_merge over the `world` records -- exactly the calls an
    N-rank run makes (ShardedFlatIndex, transport "torch"), minus the wire.  BASELINE config 3 at its own size runs through
    this on a single MI355X (8 shards of 1.25 M x 768 fit one card: 107 GB with the bf16 copies); the answer is bit-identical
    to one index holding every row in shard order (src/index/flat.rs:98-119 on the union)."""

    def __init__(self, shards, merge_device: int = None):
        if not shards:
            raise ValueError("at least one shard")
        self._L = _lib.load()
        self.shards = list(shards)
        self.world = len(self.shards)
        dims = {s.dimension() for s in self.shards}
        if len(dims) != 1:
            raise ValueError(f"shards disagree on the dimension: {sorted(dims)}")
        self.merge_device = int(self.shards[0].device if merge_device is None else merge_device)
        self.sync()

    def sync(self) -> int:
        """Re-read the shard lengths (after add / delete on any shard).  Returns the total row count."""
        lens = [len(s) for s in self.shards]
        self.offsets = [int(x) for x in np.concatenate([[0], np.cumsum(lens)[:-1]])]
        self.total = int(sum(lens))
        self.max_len = int(max(lens))
        return self.total

    def search_batch(self, queries, k: int, metric: int = 0, with_positions: bool = False, timings: dict = None):
        """(ids [nq, k'], scores [nq, k'], n [nq]) with k' = min(k, total rows) (+ global positions on request).
        timings (optional dict): receives 'local_ms' (per shard, host clock around the shard's call) and 'merge_ms'."""
        import time
        from . import _raise
        on_device = _is_device_tensor(queries)
        if on_device:
            import torch
            if queries.dtype != torch.float64 or not queries.is_contiguous() or queries.dim() != 2:
                raise ValueError("device queries must be a contiguous float64 [nq, dim] tensor")
            torch.cuda.current_stream(queries.device).synchronize()
            nq, qlen = int(queries.shape[0]), int(queries.shape[1])
            qptr = C.c_void_p(queries.data_ptr())
        else:
            Q = np.ascontiguousarray(np.asarray(queries, dtype=np.float64))
            if Q.ndim == 1:
                Q = Q[None, :]
            nq, qlen = Q.shape
        k = int(k)
        kk = max(min(k, max(self.total, 1)), 1)
        k_call = min(k, kk)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        gpos = np.zeros((nq, kk), dtype=np.uint64)
        scores = np.zeros((nq, kk), dtype=np.float64)
        n = np.zeros(max(nq, 1), dtype=np.uint64)
        if k_call > 0 and self.total > 0 and nq > 0:
            ks = min(k_call, self.max_len)
            gathered = np.zeros((self.world, pack_words(nq, ks)), dtype=np.uint64)
            local_ms = []
            for r, s in enumerate(self.shards):
                t0 = time.perf_counter()
                if on_device:
                    _raise(self._L.vl_shard_search_local_dev(s._h, self.offsets[r], 1, qptr, nq, qlen, ks, int(metric),
                                                             _pu64(gathered[r])))
                else:
                    _raise(self._L.vl_shard_search_local(s._h, self.offsets[r], 1, _pf64(Q), nq, qlen, ks, int(metric),
                                                         _pu64(gathered[r])))
                local_ms.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            _raise(self._L.vl_shard_merge(self.merge_device, _pu64(gathered), self.world, nq, ks, k_call, _pu64(gpos), _pu64(ids),
                                          _pf64(scores), _pu64(n)))
            if timings is not None:
                timings["local_ms"] = local_ms
                timings["merge_ms"] = (time.perf_counter() - t0) * 1e3
        out = (ids[:, :k_call], scores[:, :k_call], n[:nq])
        return out + (gpos[:, :k_call],) if with_positions else out
