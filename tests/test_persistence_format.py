""".vlc format handling.  Header validation runs on CPU; the round trips need a GPU."""
import json

import numpy as np
import pytest


def _doc(version="1.0.0", fmt="vectorlite-collection"):
    return {"header": {"version": version, "format": fmt, "created_at": "2025-01-01T00:00:00Z"},
            "metadata": {"name": "c", "created_at": "2025-01-01T00:00:00Z", "vector_count": 2, "dimension": 3,
                         "index_type": "Flat"},
            "index": {"Flat": {"dim": 3, "data": [
                {"id": 0, "values": [1.0, 2.0, 3.0], "text": "First vector", "metadata": None},
                {"id": 1, "values": [4.0, 5.0, 6.0], "text": "Second vector", "metadata": {"k": 1}}]}}}


def test_header_validation_matches_reference():
    """src/persistence.rs:160-173 and its tests :293-351 (version / format rejection)."""
    from vectorlite_amd import persistence as P
    assert P.parse_collection(json.dumps(_doc()))["metadata"]["name"] == "c"
    with pytest.raises(P.VersionMismatch) as e:
        P.parse_collection(json.dumps(_doc(version="2.0.0")))
    assert (e.value.expected, e.value.actual) == ("1.0.0", "2.0.0")
    with pytest.raises(P.InvalidFormat, match="Expected format 'vectorlite-collection', got 'invalid-format'"):
        P.parse_collection(json.dumps(_doc(fmt="invalid-format")))
    with pytest.raises(P.PersistenceError):
        P.parse_collection("{not json")
    with pytest.raises(P.FileNotFound):
        P.load_collection_from_file("/nonexistent/x.vlc")


@pytest.mark.gpu
def test_flat_vlc_round_trip(tmp_path):
    import vectorlite_amd as V
    from vectorlite_amd import persistence as P
    p = tmp_path / "c.vlc"
    p.write_text(json.dumps(_doc(), indent=2))
    name, idx = P.load_collection_from_file(str(p))
    assert name == "c" and isinstance(idx, V.FlatIndex) and len(idx) == 2 and idx.dimension() == 3
    # src/persistence.rs:247-249: q=[1.1,2.1,3.1] -> id 0 first
    res = idx.search([1.1, 2.1, 3.1], 1, V.SimilarityMetric.Cosine)
    assert res[0].id == 0 and res[0].text == "First vector"
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((50, 7))
    big = V.FlatIndex(7)
    big.add_rows(np.arange(50, dtype=np.uint64) * 11, rows)
    out = tmp_path / "sub" / "big.vlc"
    P.save_collection_to_file("big", big, str(out))
    assert not (tmp_path / "sub" / "big.tmp").exists()
    doc = json.loads(out.read_text())
    assert doc["metadata"] == {**doc["metadata"], "vector_count": 50, "dimension": 7, "index_type": "Flat"}
    name2, back = P.load_collection_from_file(str(out))
    ids, vals = back.export()
    assert name2 == "big" and ids.tolist() == (np.arange(50) * 11).tolist() and np.array_equal(vals, rows)


@pytest.mark.gpu
def test_hnsw_vlc_round_trip(tmp_path):
    import vectorlite_amd as V
    from vectorlite_amd import persistence as P
    idx = V.HNSWIndex(3, V.SimilarityMetric.Euclidean)
    for i, r in ((1, [1, 0, 0]), (2, [0, 1, 0]), (3, [0, 0, 1]), (4, [1, 1, 1])):
        idx.add(V.Vector(i, r, f"t{i}"))
    out = tmp_path / "h.vlc"
    P.save_collection_to_file("h", idx, str(out))
    name, back = P.load_collection_from_file(str(out))
    assert isinstance(back, V.HNSWIndex) and len(back) == 4 and back.metric() == V.SimilarityMetric.Euclidean
    res = back.search([1.1, 0.1, 0.1], 2, V.SimilarityMetric.Euclidean)  # src/index/hnsw.rs:736-742
    assert res[0].id == 1 and res[0].text == "t1"
