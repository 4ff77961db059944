import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ctypes as C
import vectorlite_amd as V
from oracle import oracle as O
dim, n = 1, 65
rng = np.random.default_rng(1234 + dim)
def unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)); x /= np.linalg.norm(x, axis=1, keepdims=True); return x
for nn in (1, 2, 63, 64, 65):
    rows = unit_rows(rng, nn, dim)
    qs = [unit_rows(rng, 1, dim)[0] for _ in range(3)]
ids = np.arange(n, dtype=np.uint64)
gpu = V.FlatIndex(dim); gpu.add_rows(ids, rows, validate=False)
ref = O.FlatOracle(dim, ids, rows)
q = qs[1]
print("q", q, "rows+", int((rows[:,0]*q[0] > 0).sum()))
L = gpu._L
for path in (0, 2, 3):
    gpu.force_path(path)
    for k in (10, 32, 64, 65):
        pos = np.zeros(80, dtype=np.uint64); idb = np.zeros(80, dtype=np.uint64); sc = np.zeros(80); nout = C.c_uint64(0)
        rc = L.vl_index_search_positions(gpu._h, q.ctypes.data_as(C.POINTER(C.c_double)), 1, k, 0,
              pos.ctypes.data_as(C.POINTER(C.c_uint64)), idb.ctypes.data_as(C.POINTER(C.c_uint64)), sc.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nout))
        ri, rs = ref.search(q, k, 0)
        print("path", path, "k", k, "rc", rc, "lastpath", V.last_path(), V._last_error() if rc else "")
        if rc == 0:
            print("  gpu", pos[:nout.value].tolist()); print("  ref", ri.tolist())
