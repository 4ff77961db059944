"""Row-sharded flat index across the GPUs of one node (one process per GPU, torch.distributed).

No reference counterpart: the reference is single-process.  The north star shards the
*batched* flat search by rows: rank r holds the contiguous row range [offset_r, offset_r + n_r) of the
corpus, every rank scans its own shard with the same HIP path as the single-GPU index, and ONE
all-gather (RCCL over xGMI with backend "nccl"; gloo in the CPU tests) exchanges the per-shard exact
top-k.  Because each shard returns the reference's exact f64 scores, merging is just the reference's
ordering on the union: score descending, ties by GLOBAL storage position (shard offset + local
position) ascending -- identical to a single index holding all rows (src/index/flat.rs:116).

Exchange size: nq * (k + 1) * 24 bytes per rank (config 3: 1024 queries, k = 10 -> 270 KB per rank):
latency-bound, so it is a single collective, not a ring of small ones.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

__all__ = ["ShardedFlatIndex", "merge_shard_results", "shard_ranges"]


def shard_ranges(n_rows: int, world: int):
    """Contiguous, near-equal row ranges: rank r gets [starts[r], starts[r+1])."""
    base, rem = divmod(n_rows, world)
    starts = [0]
    for r in range(world):
        starts.append(starts[-1] + base + (1 if r < rem else 0))
    return starts


def merge_shard_results(scores: np.ndarray, gpos: np.ndarray, ids: np.ndarray, counts: np.ndarray, k: int):
    """Merge per-shard top-k lists of ONE query.

    scores/gpos/ids: [world, k]; counts: [world].  Returns (ids, scores, gpos) of the global top-k in
    the reference's order: score descending, global position ascending on ties."""
    sel_s, sel_p, sel_i = [], [], []
    for r in range(scores.shape[0]):
        c = int(counts[r])
        sel_s.append(scores[r, :c])
        sel_p.append(gpos[r, :c])
        sel_i.append(ids[r, :c])
    s = np.concatenate(sel_s) if sel_s else np.zeros(0)
    p = np.concatenate(sel_p) if sel_p else np.zeros(0, dtype=np.int64)
    i = np.concatenate(sel_i) if sel_i else np.zeros(0, dtype=np.uint64)
    order = np.lexsort((p, -s))  # primary: -score ascending (= score descending); secondary: position
    order = order[:k]
    return i[order], s[order], p[order]


class ShardedFlatIndex:
    """One rank's view of a row-sharded flat index.

    `local` is this rank's shard: a `vectorlite_amd.FlatIndex` on this rank's GPU (anything exposing
    `search_positions(query, k, metric)` and `len()` works, which is how the CPU tests drive the
    collective path).  `offset` is the global position of the shard's first row.
    """

    def __init__(self, local, offset: int, group=None, device=None):
        self.local = local
        self.offset = int(offset)
        self.group = group
        self.device = device
        try:
            import torch.distributed as dist
            self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        except Exception:  # pragma: no cover
            self._dist = None
        self.world = self._dist.get_world_size(group) if self._dist else 1
        self.rank = self._dist.get_rank(group) if self._dist else 0

    # ---- collective ---------------------------------------------------------------------------
    def _all_gather(self, packed: np.ndarray) -> np.ndarray:
        """packed: int64 [m]; returns [world, m]."""
        if self.world == 1:
            return packed[None, :]
        import torch
        t = torch.from_numpy(packed)
        if self.device is not None:
            t = t.to(self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=t.device)
        self._dist.all_gather_into_tensor(out, t, group=self.group)  # one collective (RCCL on GPU ranks)
        return out.cpu().numpy().reshape(self.world, -1)

    def search_batch(self, queries, k: int, metric: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """nq independent searches over the whole (sharded) corpus.
        Returns (ids [nq, k], scores [nq, k], n [nq]); identical on every rank."""
        Q = np.ascontiguousarray(np.asarray(queries, dtype=np.float64))
        if Q.ndim == 1:
            Q = Q[None, :]
        nq = Q.shape[0]
        kk = max(int(k), 1)
        # per query: kk rows of (score bits, global pos, id bits) + 1 row whose first word is the count
        packed = np.zeros((nq, kk + 1, 3), dtype=np.int64)
        if len(self.local) != 0 and k != 0:
            if hasattr(self.local, "search_batch_positions"):  # GPU shard: queries share slab passes
                bpos, bids, bsc, bn = self.local.search_batch_positions(Q, k, metric)
                packed[:, : int(k), 0] = np.ascontiguousarray(bsc, dtype=np.float64).view(np.int64)
                packed[:, : int(k), 1] = bpos.astype(np.int64) + self.offset
                packed[:, : int(k), 2] = np.ascontiguousarray(bids, dtype=np.uint64).view(np.int64)
                packed[:, kk, 0] = bn.astype(np.int64)
            else:
                for qi in range(nq):
                    pos, ids, scores = self.local.search_positions(Q[qi], k, metric)
                    c = len(pos)
                    packed[qi, :c, 0] = np.asarray(scores, dtype=np.float64).view(np.int64)
                    packed[qi, :c, 1] = np.asarray(pos, dtype=np.int64) + self.offset
                    packed[qi, :c, 2] = np.asarray(ids, dtype=np.uint64).view(np.int64)
                    packed[qi, kk, 0] = c
        gathered = self._all_gather(packed.reshape(-1)).reshape(self.world, nq, kk + 1, 3)
        # merge for all queries at once: [nq, world * kk] candidates, entries beyond a shard's count pushed
        # behind everything, then the reference's order -- score descending, global position ascending
        counts = gathered[:, :, kk, 0]                                             # [world, nq]
        valid = np.arange(kk)[None, None, :] < counts[:, :, None]                  # [world, nq, kk]
        sc = np.ascontiguousarray(gathered[:, :, :kk, 0]).view(np.float64)
        sc = np.where(valid, sc, -np.inf).transpose(1, 0, 2).reshape(nq, -1)
        gp = np.where(valid, gathered[:, :, :kk, 1], np.iinfo(np.int64).max).transpose(1, 0, 2).reshape(nq, -1)
        gi = gathered[:, :, :kk, 2].transpose(1, 0, 2).reshape(nq, -1)
        order = np.lexsort((gp, -sc), axis=-1)[:, :kk]
        total = np.minimum(counts.sum(axis=0), int(k)).astype(np.uint64)           # [nq]
        out_scores = np.take_along_axis(sc, order, axis=1)
        out_ids = np.ascontiguousarray(np.take_along_axis(gi, order, axis=1)).view(np.uint64)
        live = np.arange(kk)[None, :] < total[:, None]
        out_scores = np.where(live, out_scores, 0.0)
        out_ids = np.where(live, out_ids, np.uint64(0))
        return out_ids[:, : int(k)], out_scores[:, : int(k)], total

    def search(self, query, k: int, metric: int = 0):
        ids, scores, n = self.search_batch(np.asarray(query, dtype=np.float64)[None, :], k, metric)
        m = int(n[0])
        return ids[0, :m].copy(), scores[0, :m].copy()

    def global_len(self) -> int:
        n = np.array([len(self.local)], dtype=np.int64)
        return int(self._all_gather(n).sum())
