import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
from oracle import oracle as O
rng = np.random.default_rng(2)
n, dim = 5000, 96
emb = rng.standard_normal((n, dim)).astype(np.float32)
emb[3] = 0.0; emb[4] = 1e30; emb[5] = 1e-30; emb[6] = emb[7]
idx = V.FlatIndex(dim); idx.add_embeddings(np.arange(n, dtype=np.uint64), emb)
rows = O.embed_f32(emb)
print("oracle embed available:", rows is not None)
if rows is not None:
    ref = O.FlatOracle(dim, np.arange(n, dtype=np.uint64), rows)
    e_ids, e_vals = idx.export()
    assert np.array_equal(e_vals, rows), "stored rows differ from the oracle's embedding post-processing"
    Qe = rng.standard_normal((6, dim)).astype(np.float32); Qe[0] = 0.0; Qe[1] = emb[7]; Qe[2] = 1e30; Qe[3] = 1e-38
    bi, bs, bn = idx.search_batch_embeddings(Qe, 10, 0)
    for j in range(6):
        q = O.embed_f32(Qe[j:j+1])[0]
        wi, ws = ref.search(q, 10, 0)
        assert bi[j, : bn[j]].tolist() == wi.tolist() and bs[j, : bn[j]].tolist() == ws.tolist(), j
    print("embedding probe ok")
