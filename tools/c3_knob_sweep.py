#!/usr/bin/env python3
"""Config 3's shard (1024 queries x 1.25 M x 768, Euclidean): the batch filter's launch-plan knobs, ONE process, combinations
interleaved over rounds (guide rule 24).  The knobs are read per launch sequence (csrc/mfma_scan.hip): sample size
(VL_MFMA_SAMPLE_MIN), pass-1 stages (VL_MFMA_STAGES) and where the first stage ends (VL_MFMA_STAGE1, sixteenths).
Filter time = HIP events around the filter's launch sequence; answers of every combination are compared with the default's."""
import itertools, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rows, dim, nq, metric = int(os.environ.get("ROWS", 1_250_000)), int(os.environ.get("DIM", 768)), int(os.environ.get("NQ", 1024)), int(os.environ.get("METRIC", 1))
    import torch
    import vectorlite_amd as V
    dev = torch.device("cuda", 0)
    idx = V.FlatIndex(dim)
    idx.reserve(rows)
    done = ci = 0
    while done < rows:
        c = min(250_000, rows - done)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + ci)
        x = torch.randn((c, dim), dtype=torch.float64, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
        idx.add_rows(np.arange(done, done + c, dtype=np.uint64), x, validate=False)
        done += c
        ci += 1
        del x
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((nq, dim))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    dQ = torch.from_numpy(Q).to(dev)
    knobs = ("VL_MFMA_SAMPLE_MIN", "VL_MFMA_STAGES", "VL_MFMA_STAGE1")
    combos = [(None, None, None)] + [c for c in itertools.product(("16384", "65536", None), ("2", "3"), ("1", "2", "3")) if c != (None, "2", "2")]

    def apply(c):
        for k, v in zip(knobs, c):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    apply(combos[0])
    base = idx.search_batch_device(dQ, 10, metric)
    t = {c: [] for c in combos}
    same = {c: True for c in combos}
    for rnd in range(3):
        for c in combos:
            apply(c)
            a = idx.search_batch_device(dQ, 10, metric)
            same[c] = same[c] and bool(np.array_equal(a[0], base[0]) and np.array_equal(a[1], base[1]))
            idx.profile_read()
            idx.profile_enable(True)
            for _ in range(4):
                idx.search_batch_device(dQ, 10, metric)
            idx.profile_enable(False)
            n_pass, ms, _ = idx.profile_read()
            t[c].append(ms / 4)
    apply(combos[0])
    out = []
    for c in combos:
        out.append({"sample_min": c[0] or "default(32768)", "stages": c[1] or "default(2)", "stage1_16ths": c[2] or "default(2)",
                    "filter_ms_median": round(float(np.median(t[c])), 4), "filter_ms_min": round(float(np.min(t[c])), 4), "answers_identical": same[c]})
    out.sort(key=lambda e: e["filter_ms_median"])
    for e in out:
        print(json.dumps(e), flush=True)


if __name__ == "__main__":
    main()
