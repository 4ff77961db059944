#!/usr/bin/env python3
"""Phase timing inside k_merge_finalize for single searches (diagnostic build with -DVL_DBG_STAMPS):
  tools/build_variant_k.sh stamps -DVL_DBG_STAMPS && VL_LIB_PATH=$PWD/vectorlite_amd/libvl_stamps.so python tools/finalize_stamps.py
Stamps are s_memrealtime (100 MHz): 0 kernel start, 1 lists folded, 2 tree merge done, 3 candidates in LDS, 4 row loads issued,
5 query staged, 6 first tile summed, 7 all tiles summed, 8 rescoring done, 9 ranked + block stored, 10 stamped."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vectorlite_amd as V
from vectorlite_amd import _lib
n, dim = int(os.environ.get("ROWS", 10_000_000)), 384
idx = V.FlatIndex(dim); idx.reserve(n)
for ci, c0 in enumerate(range(0, n, 500_000)):
    c = min(500_000, n - c0)
    g = torch.Generator(device="cuda:0"); g.manual_seed(1234 + ci)
    x = torch.randn((c, dim), dtype=torch.float64, device="cuda:0", generator=g); x /= torch.linalg.vector_norm(x, dim=1, keepdim=True)
    idx.add_rows(np.arange(c0, c0 + c, dtype=np.uint64), x, validate=False); del x
rng = np.random.default_rng(5); Q = rng.standard_normal((64, dim)); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
L = _lib.load()
for i in range(20): idx.search_arrays(Q[i], 10, 0)
acc = np.zeros(11)
clk = []
reps = 50
buf = (C.c_ulonglong * 32)()
for i in range(reps):
    idx.search_arrays(Q[i % 64], 10, 0)
    torch.cuda.synchronize()
    L.vl_dbg_read_stamps(buf)
    s = np.array(buf[:11], dtype=np.float64)
    acc += (s - s[0]) * 0.01
    c = np.array(buf[16:27], dtype=np.float64)
    clk.append((c[10] - c[0]) / max(s[10] - s[0], 1) * 100.0)  # shader cycles per 10 ns tick -> MHz
print("us since kernel start:", np.round(acc / reps, 2).tolist())
print("shader clock during the kernel (MHz): median", round(float(np.median(clk))), "min", round(min(clk)), "max", round(max(clk)))
