"""Worker for tests/test_sharded_gloo.py: one rank of a row-sharded search over gloo (CPU).
The local shard is driven by the ORACLE here (tests may use it as the checker's stand-in for the
GPU shard); what is under test is the collective + merge path of vectorlite_amd/sharded.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleShard:
    """search_positions() on a FlatOracle (ids are unique in these tests, so id -> position is a dict)."""

    def __init__(self, dim, ids, rows):
        from oracle import oracle as O
        self.o = O.FlatOracle(dim, ids, rows)
        self.pos_of = {int(i): p for p, i in enumerate(ids)}

    def __len__(self):
        return len(self.o)

    def search_positions(self, q, k, metric):
        ids, scores = self.o.search(q, k, metric)
        pos = np.array([self.pos_of[int(i)] for i in ids], dtype=np.uint64)
        return pos, ids, scores


def main():
    out_dir = sys.argv[1]
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from vectorlite_amd.sharded import ShardedFlatIndex, shard_ranges
    data = np.load(os.path.join(out_dir, "data.npz"))
    rows, ids, Q = data["rows"], data["ids"], data["Q"]
    starts = shard_ranges(rows.shape[0], world)
    lo, hi = starts[rank], starts[rank + 1]
    shard = OracleShard(rows.shape[1], ids[lo:hi], rows[lo:hi])
    idx = ShardedFlatIndex(shard, offset=lo)
    assert idx.global_len() == rows.shape[0]
    res = {}
    for m in range(4):
        for k in (1, 10, 50):
            i, s, n = idx.search_batch(Q, k, m)
            res[f"ids_{m}_{k}"], res[f"scores_{m}_{k}"], res[f"n_{m}_{k}"] = i, s, n
    np.savez(os.path.join(out_dir, f"out_rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
