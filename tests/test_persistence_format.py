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


# ---- native streaming reader (csrc/vlc_loader.cpp): host-only part, no GPU needed --------------------
def _write(tmp_path, doc, name="c.vlc", **kw):
    p = tmp_path / name
    p.write_text(doc if isinstance(doc, str) else json.dumps(doc, **kw))
    return str(p)


def test_native_reader_structure_and_values(tmp_path):
    from vectorlite_amd import persistence as P
    rng = np.random.default_rng(3)
    n, dim = 700, 9
    rows = rng.standard_normal((n, dim)) * np.exp(rng.uniform(-40, 40, size=(n, 1)))
    rows[5] = [0.0, -0.0, 1.0, -1.0, 1e-320, 5e-324, 1.7976931348623157e308, 123456789012345678.0, 3.0]
    data = [{"id": int(i * 7 + 1), "values": rows[i].tolist(), "text": f"t{i} \"q\" \\ é中\U0001F600\n",
             "metadata": None if i % 3 == 0 else {"k": [i, "]", {"x": "}\""}], "s": "a\"]b"}} for i in range(n)]
    doc = _doc()
    doc["metadata"].update(name="näme \"x\"", vector_count=n, dimension=dim)
    doc["index"] = {"Flat": {"dim": dim, "data": data}}
    for kw in ({"indent": 2}, {"separators": (",", ":")}, {"ensure_ascii": False, "indent": 1}):
        d = P.VlcDocument(_write(tmp_path, doc, **kw))
        assert (d.index_type, d.metric, d.dim, d.rows, d.vector_count, d.dimension) == ("Flat", None, dim, n, n, dim)
        assert d.name == "näme \"x\""
        got = d.values()
        assert got.shape == (n, dim) and np.array_equal(got.view(np.uint64), rows.view(np.uint64))  # bit for bit, -0.0 too
        assert np.array_equal(d.values(10, 5).view(np.uint64), rows[10:15].view(np.uint64))
        side = d.side_data(first_wins=True)
        assert list(side.keys()) == [r["id"] for r in data]
        assert all(side[r["id"]] == (r["text"], r["metadata"]) for r in data)
        d.close()


def test_native_reader_field_order_unknown_fields_and_integers(tmp_path):
    from vectorlite_amd import persistence as P
    text = '''{"index": {"Flat": {"extra": [1, {"a": "]"}], "data": [
        {"metadata": {"a": 1}, "text": "x", "values": [1, -2, 3e0, 4.5E+1], "id": 18446744073709551615, "zzz": null},
        {"id": 0, "values": [ 0.1 ,0.2,
            0.3, 1e-7 ], "text": ""}], "dim": 4}},
      "metadata": {"index_type": "Flat", "dimension": 4, "vector_count": 2, "created_at": "x", "name": "n"},
      "header": {"created_at": "2025-01-01T00:00:00Z", "format": "vectorlite-collection", "version": "1.0.0"}}'''
    d = P.VlcDocument(_write(tmp_path, text))
    assert d.rows == 2 and d.dim == 4
    assert d.values().tolist() == [[1.0, -2.0, 3.0, 45.0], [0.1, 0.2, 0.3, 1e-7]]
    ids = d.side_table()[0].tolist()
    assert ids == [18446744073709551615, 0]
    assert d.side_data(True) == {18446744073709551615: ("x", {"a": 1}), 0: ("", None)}


def test_native_reader_hnsw_payload(tmp_path):
    from vectorlite_amd import persistence as P
    import vectorlite_amd as V
    doc = _doc()
    doc["metadata"]["index_type"] = "HNSW"
    doc["index"] = {"HNSW": {"dim": 3, "metric": "Euclidean", "id_to_index": {"5": 0, "9": 1}, "index_to_id": {"0": 5, "1": 9},
                             "metadata": {"9": {"text": "nine", "metadata": {"a": 1}}, "5": {"text": "five", "metadata": None}},
                             "vector_values": {"5": [1.0, 0.0, 0.0], "9": [0.0, 1.0, 0.5]}}}
    d = P.VlcDocument(_write(tmp_path, doc, indent=2))
    assert (d.index_type, d.metric, d.dim, d.rows) == ("HNSW", V.SimilarityMetric.Euclidean, 3, 2)
    assert d.values().tolist() == [[1.0, 0.0, 0.0], [0.0, 1.0, 0.5]]
    assert d.side_data(False) == {5: ("five", None), 9: ("nine", {"a": 1})}
    doc["index"]["HNSW"]["dim"] = 0
    with pytest.raises(P.PersistenceError, match="Invalid dimension: cannot be 0"):  # src/index/hnsw.rs:288-290
        P.VlcDocument(_write(tmp_path, doc))
    doc["index"]["HNSW"].update(dim=3, metric="Chebyshev")
    with pytest.raises(P.PersistenceError, match="unknown variant `Chebyshev`"):
        P.VlcDocument(_write(tmp_path, doc))


def test_native_reader_errors_match_reference_categories(tmp_path):
    """src/persistence.rs:149-176 (+ its tests :293-351): serde errors first, then version, then format."""
    from vectorlite_amd import persistence as P
    with pytest.raises(P.FileNotFound):
        P.VlcDocument(str(tmp_path / "missing.vlc"))
    with pytest.raises(P.VersionMismatch) as e:
        P.VlcDocument(_write(tmp_path, _doc(version="2.0.0")))
    assert (e.value.expected, e.value.actual) == ("1.0.0", "2.0.0")
    with pytest.raises(P.InvalidFormat, match="Expected format 'vectorlite-collection', got 'invalid-format'"):
        P.VlcDocument(_write(tmp_path, _doc(fmt="invalid-format")))
    both = _doc(version="9", fmt="nope")  # version is checked before format (:160-173)
    with pytest.raises(P.VersionMismatch):
        P.VlcDocument(_write(tmp_path, both))
    bad_and_wrong_version = json.dumps(_doc(version="2.0.0"))[:-3]  # malformed document: serde speaks first
    for text in ("", "{not json", "[]", bad_and_wrong_version, json.dumps(_doc()) + " x",
                 json.dumps({k: v for k, v in _doc().items() if k != "header"})):
        with pytest.raises(P.PersistenceError, match="Serialization error"):
            P.VlcDocument(_write(tmp_path, text))

    def flat(rows, dim=3):
        d = _doc()
        d["index"] = {"Flat": {"dim": dim, "data": rows}}
        return d
    for rows, msg in (([{"id": 1, "values": [1, 2, 3]}], "missing field `text`"),
                      ([{"id": -1, "values": [1, 2, 3], "text": ""}], "expected u64"),
                      ([{"id": 1.5, "values": [1, 2, 3], "text": ""}], "floating point"),
                      ([{"values": [1, 2, 3], "text": ""}], "missing field `id`"),
                      ([{"id": 1, "values": 7, "text": ""}], "expected a sequence"),
                      ([{"id": 1, "values": [1, 2, 3], "text": 5}], "expected a string")):
        with pytest.raises(P.PersistenceError, match=msg):
            P.VlcDocument(_write(tmp_path, flat(rows)))
    # rows are converted lazily: bad numbers / ragged rows surface when the values are read
    for vals, msg in (("[1, 2]", "Vector dimension mismatch: expected 3, got 2"), ("[1, 2, 3, 4]", "expected 3, got 4"),
                      ("[1, 2, x]", "expected f64"), ("[1, 2, 01]", "invalid number"), ("[1, 2, 1e999]", "out of range"),
                      ("[1, 2, 3,]", "expected f64"), ("[1, 2, +3]", "expected f64"), ("[1, 2, NaN]", "expected f64"),
                      ("[1, 2, 1.]", "invalid number"), ("[1 2 3]", "expected `,` or `]`")):
        text = json.dumps(flat([{"id": 1, "values": "@@", "text": ""}])).replace('"@@"', vals)
        d = P.VlcDocument(_write(tmp_path, text))
        with pytest.raises(P.PersistenceError, match=msg):
            d.values()


def test_native_reader_large_file_parallel_conversion(tmp_path):
    """40k x 64 rows (~50 MB of JSON): the threaded conversion returns exactly what Python's json does."""
    from vectorlite_amd import persistence as P
    rng = np.random.default_rng(9)
    n, dim = 40000, 64
    rows = rng.standard_normal((n, dim))
    p = tmp_path / "big.vlc"
    with open(p, "w") as f:
        f.write('{"header": {"version": "1.0.0", "format": "vectorlite-collection", "created_at": "x"},\n'
                '"metadata": {"name": "big", "created_at": "x", "vector_count": %d, "dimension": %d, "index_type": "Flat"},\n'
                '"index": {"Flat": {"dim": %d, "data": [\n' % (n, dim, dim))
        for i in range(n):
            f.write('{"id": %d, "values": %s, "text": "row %d", "metadata": null}%s\n'
                    % (i, json.dumps(rows[i].tolist()), i, "," if i + 1 < n else ""))
        f.write("]}}}\n")
    d = P.VlcDocument(str(p))
    assert d.rows == n
    assert np.array_equal(d.values(), rows)
    assert d.side_data(True)[n - 1] == (f"row {n - 1}", None)


@pytest.mark.gpu
def test_native_loader_builds_the_same_index_as_bulk_add(tmp_path):
    """vl_vlc_build_index (chunked, conversion overlapped with ingest) == add_rows of the same rows:
    duplicate ids kept (FlatIndex{dim, data} is not validated, src/index/flat.rs:59), search identical."""
    import vectorlite_amd as V
    from vectorlite_amd import persistence as P
    rng = np.random.default_rng(4)
    n, dim = 3000, 48
    rows = rng.standard_normal((n, dim))
    ids = np.arange(n, dtype=np.uint64) * 3
    ids[10] = ids[2]  # duplicate id, kept
    doc = _doc()
    doc["index"] = {"Flat": {"dim": dim, "data": [
        {"id": int(ids[i]), "values": rows[i].tolist(), "text": f"t{i}", "metadata": {"i": i}} for i in range(n)]}}
    p = tmp_path / "n.vlc"
    p.write_text(json.dumps(doc))
    name, idx = P.load_collection_from_file(str(p))
    ref = V.FlatIndex(dim)
    ref.add_rows(ids, rows, validate=False)
    gi, gv = idx.export()
    assert gi.tolist() == ids.tolist() and np.array_equal(gv, rows)
    for m in range(4):
        a, b = idx.search_arrays(rows[7] + 0.01, 20, m), ref.search_arrays(rows[7] + 0.01, 20, m)
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
    assert idx.get_vector(int(ids[2])).text == "t2"  # first row of a duplicated id
    res = idx.search(rows[5], 1, V.SimilarityMetric.Cosine)
    assert res[0].id == int(ids[5]) and res[0].metadata == {"i": 5}
    ragged = _doc()
    ragged["index"]["Flat"]["data"][1]["values"] = [1.0, 2.0]
    p.write_text(json.dumps(ragged))
    with pytest.raises(P.PersistenceError, match="Vector dimension mismatch: expected 3, got 2"):
        P.load_collection_from_file(str(p))


def test_native_reader_survives_mutated_files(tmp_path):
    """1500 seeded mutations (byte flips, cuts, duplicated spans, stray quotes / brackets / escapes, truncations)
    of valid Flat and HNSW documents: the reader either accepts the file or reports a PersistenceError --
    it never crashes, and whatever it accepts can be read back completely."""
    import random
    from vectorlite_amd import persistence as P
    doc = _doc()
    doc["index"]["Flat"]["data"][0]["text"] = "a \" b \\ é"
    doc["index"]["Flat"]["data"][1]["metadata"] = {"k": [1, {"z": "]}"}]}
    hn = json.loads(json.dumps(_doc()))
    hn["index"] = {"HNSW": {"dim": 2, "metric": "Cosine", "id_to_index": {}, "index_to_id": {},
                            "metadata": {"5": {"text": "t", "metadata": None}},
                            "vector_values": {"5": [1.0, 2.0], "6": [0.5, -1]}}}
    bases = [json.dumps(doc, indent=2).encode(), json.dumps(doc, separators=(",", ":")).encode(), json.dumps(hn, indent=1).encode()]
    rnd = random.Random(7)
    path = str(tmp_path / "f.vlc")
    accepted = 0
    for _ in range(1500):
        b = bytearray(rnd.choice(bases))
        for _ in range(rnd.randint(1, 4)):
            op = rnd.randint(0, 4)
            if op == 0:
                b[rnd.randrange(len(b))] = rnd.randrange(256)
            elif op == 1 and len(b) > 1:
                del b[rnd.randrange(len(b)):rnd.randrange(len(b)) or None]
            elif op == 2:
                i = rnd.randrange(len(b) + 1)
                b[i:i] = rnd.choice([b'"', b"\\", b"[", b"]", b"{", b"}", b",", b":", b"\\u12", b"1e999", b"-", b"\x00", b'"\\'])
            elif op == 3 and len(b) > 10:
                i = rnd.randrange(len(b) - 5)
                j = i + rnd.randint(1, 5)
                b[i:j] = b[i:j] * rnd.randint(2, 4)
            else:
                b = b[: rnd.randrange(len(b) + 1)]
            if not b:
                b = bytearray(b" ")
        with open(path, "wb") as f:
            f.write(bytes(b))
        try:
            d = P.VlcDocument(path)
        except (P.PersistenceError, UnicodeDecodeError):
            continue
        try:
            d.values()
            d.side_data(True)
        except (P.PersistenceError, ValueError, UnicodeDecodeError):
            pass
        d.close()
        accepted += 1
    assert 0 < accepted < 1500
