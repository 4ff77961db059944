import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, vectorlite_amd as V
from vectorlite_amd import persistence as P
rng = np.random.default_rng(3)
rows = rng.standard_normal((500, 20)); ids = np.arange(500, dtype=np.uint64) * np.uint64(3)
for mode in ("replicas", "row_shards"):
    m = V.MultiFlatIndex(20, [0, 0, 0], mode)
    for i, r in zip(ids[:50], rows[:50]):
        m.add(V.Vector(int(i), r, f"text {int(i)}", {"k": int(i)}))
    m.add_rows(ids[50:], rows[50:])
    m.delete(int(ids[7]))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "c.vlc")
        P.save_collection_to_file("multi", m, path)
        name, back = P.load_collection_from_file(path)
        assert name == "multi" and len(back) == len(m) == 499
        q = rows[33] * 1.01
        a, b = m.search(q, 5, 0), back.search(q, 5, 0)
        assert [(x.id, x.score, x.text, x.metadata) for x in a] == [(x.id, x.score, x.text, x.metadata) for x in b], (a[:2], b[:2])
        assert a[0].text == "text 99" and a[0].metadata == {"k": 99}
    print(mode, "persistence round trip ok")
