"""Time loading a .vlc file: native streaming reader (vl_vlc_*) vs json.load + bulk add.
usage: python tools/bench_vlc_load.py [rows] [dim]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vectorlite_amd as V
from vectorlite_amd import persistence as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
rng = np.random.default_rng(1)
rows = rng.standard_normal((n, dim))
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
path = os.path.join(tempfile.mkdtemp(prefix="vlc_"), "c.vlc")
t = time.perf_counter()
with open(path, "w") as f:
    f.write('{"header": {"version": "1.0.0", "format": "vectorlite-collection", "created_at": "x"},\n'
            '"metadata": {"name": "bench", "created_at": "x", "vector_count": %d, "dimension": %d, "index_type": "Flat"},\n'
            '"index": {"Flat": {"dim": %d, "data": [\n' % (n, dim, dim))
    for i in range(n):
        f.write('{"id": %d, "values": %s, "text": "row %d", "metadata": null}%s\n'
                % (i, json.dumps(rows[i].tolist()), i, "," if i + 1 < n else ""))
    f.write("]}}}\n")
size = os.path.getsize(path)
print(f"wrote {path}: {size / 1e6:.0f} MB in {time.perf_counter() - t:.1f}s", flush=True)
V.FlatIndex(4).add_rows(np.arange(2, dtype=np.uint64), np.ones((2, 4)))  # runtime warm-up

t = time.perf_counter()
doc = P.VlcDocument(path)
t_open = time.perf_counter() - t
name, idx = doc.name, doc.build_index()
t_native = time.perf_counter() - t
ids, vals = idx.export()
assert np.array_equal(vals, rows) and ids.tolist() == list(range(n))
print(f"native reader: open (map + structural pass + header checks) {t_open:.2f}s, total {t_native:.2f}s "
      f"= {size / 1e6 / t_native:.0f} MB/s, {n / t_native:.0f} rows/s; rows bit-identical", flush=True)

t = time.perf_counter()
with open(path) as f:
    data = P.parse_collection(f.read())
idx2 = P.index_from_payload(data["index"])
t_py = time.perf_counter() - t
print(f"json.loads + add_rows: {t_py:.2f}s = {size / 1e6 / t_py:.0f} MB/s  -> native is {t_py / t_native:.1f}x faster", flush=True)
q = rows[123]
assert [r.id for r in idx.search(q, 5, 0)] == [r.id for r in idx2.search(q, 5, 0)]
os.remove(path)
