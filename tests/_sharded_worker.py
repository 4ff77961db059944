"""Worker for tests/test_sharded_gloo.py and tests/test_gpu_sharded.py: one rank of a row-sharded search.

mode "cpu":  gloo, no GPU.  What is under test is the host harness of vectorlite_amd/sharded.py -- sync(),
             offsets, the exchange record's layout, the one all-gather, k clamping.  The two C-ABI halves
             (vl_shard_search_local / vl_shard_merge need a GPU) are replaced BY THE TEST with checker-side
             stand-ins: the oracle fills the record, a numpy lexsort merges.  Product code has no such path.
mode "gpu":  gloo rendezvous, every rank owns a real GPU shard on the one visible card (RCCL refuses two ranks
             on one device, so the records travel by gloo); vl_shard_search_local and the device merge kernel
             (vl_shard_merge) are the product's.
mode "rccl": world 1 only on a one-GPU box: vl_comm_create + vl_shard_sync + vl_shard_search_batch (ncclAllGather).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_cpu_checked_class():
    from oracle import oracle as O
    from vectorlite_amd.sharded import ShardedFlatIndex, pack_words, unpack_record, SHARD_HDR_WORDS

    class OracleShard:
        """Stands where the GPU shard stands: len(), dimension(), and the rows for the oracle."""

        def __init__(self, dim, ids, rows):
            self.o = O.FlatOracle(dim, ids, rows)
            self.dim = dim
            self.pos_of = {int(i): p for p, i in enumerate(ids)}
            self.device = 0
            self._h = None

        def __len__(self):
            return len(self.o)

        def dimension(self):
            return self.dim

    class CpuCheckedSharded(ShardedFlatIndex):
        def _local_record(self, Q, ks, metric):
            nq = Q.shape[0]
            rec = np.zeros(pack_words(nq, ks), dtype=np.uint64)
            rec[1], rec[2] = len(self.local), self.local.dimension()
            cnt = rec[SHARD_HDR_WORDS: SHARD_HDR_WORDS + nq]
            body = rec[SHARD_HDR_WORDS + nq:].reshape(3, nq, ks)
            if len(self.local):
                for qi in range(nq):
                    ids, scores = self.local.o.search(Q[qi], ks, metric)
                    c = len(ids)
                    cnt[qi] = c
                    body[0, qi, :c] = np.asarray(scores, dtype=np.float64).view(np.uint64)
                    body[1, qi, :c] = np.array([self.local.pos_of[int(i)] + self.offset for i in ids], dtype=np.uint64)
                    body[2, qi, :c] = np.asarray(ids, dtype=np.uint64)
            return rec

        def _merge(self, gathered, nq, ks, k, gpos, ids, scores, n):
            recs = [unpack_record(gathered[r], nq, ks) for r in range(self.world)]
            assert all(r[0] == 0 for r in recs)
            for qi in range(nq):
                s = np.concatenate([r[4][qi, : int(r[3][qi])] for r in recs])
                p = np.concatenate([r[5][qi, : int(r[3][qi])] for r in recs])
                i = np.concatenate([r[6][qi, : int(r[3][qi])] for r in recs])
                order = np.lexsort((p, -s))[:k]  # score descending, then global position
                m = len(order)
                scores[qi, :m], gpos[qi, :m], ids[qi, :m], n[qi] = s[order], p[order], i[order], m

    return OracleShard, CpuCheckedSharded


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from vectorlite_amd.sharded import Comm, ShardedFlatIndex, shard_ranges
    data = np.load(os.path.join(out_dir, "data.npz"))
    rows, ids, Q = data["rows"], data["ids"], data["Q"]
    ks_list = [int(x) for x in data["ks"]]
    starts = [int(x) for x in data["starts"]] if "starts" in data else shard_ranges(rows.shape[0], world)
    lo, hi = starts[rank], starts[rank + 1]
    if mode == "cpu":
        OracleShard, Cls = make_cpu_checked_class()
        idx = Cls(OracleShard(rows.shape[1], ids[lo:hi], rows[lo:hi]), transport="torch")
    else:
        import vectorlite_amd as V
        shard = V.FlatIndex(rows.shape[1], device=0)
        if hi > lo:
            shard.add_rows(ids[lo:hi], rows[lo:hi], validate=False)
        if mode == "rccl":
            assert world == 1
            idx = ShardedFlatIndex(shard, comm=Comm.from_torch_distributed(device=0))
        else:
            idx = ShardedFlatIndex(shard, transport="torch")
    assert idx.global_len() == rows.shape[0] and idx.offset == lo
    res = {}
    for m in range(4):
        for k in ks_list:
            i, s, n, p = idx.search_batch(Q, k, m, with_positions=True)
            res[f"ids_{m}_{k}"], res[f"scores_{m}_{k}"], res[f"n_{m}_{k}"], res[f"gpos_{m}_{k}"] = i, s, n, p
    np.savez(os.path.join(out_dir, f"out_rank{rank}.npz"), **res)
    if mode != "cpu":  # errors travel inside the exchange: every rank raises the same thing, nobody hangs
        import vectorlite_amd as V
        try:
            idx.search_batch(Q[:, :-1], 5, 0)
            raise SystemExit("a query of the wrong length was accepted")
        except V.DimensionMismatch as e:
            assert (e.expected, e.actual) == (rows.shape[1], rows.shape[1] - 1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
