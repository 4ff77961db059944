"""The reference's client-layer tests (src/client.rs:526-850) replayed on the GPU-backed mirror
(vectorlite_amd/client.py): same mock embedder (every text -> vec![1.0; dim], src/client.rs:504-523),
same expectations."""
import json

import pytest

pytestmark = pytest.mark.gpu


class MockEmbeddingFunction:
    def __init__(self, dimension):
        self._d = dimension

    def generate_embedding(self, text):
        return [1.0] * self._d

    def dimension(self):
        return self._d


@pytest.fixture()
def C():
    import vectorlite_amd as V
    assert V.runtime_info()[0] > 0
    from vectorlite_amd import client
    return client


def test_create_get_delete_collections(C):  # :526-597
    cl = C.VectorLiteClient(MockEmbeddingFunction(3))
    assert cl.list_collections() == []
    cl.create_collection("test_collection", C.IndexType.Flat)
    assert cl.has_collection("test_collection") and cl.list_collections() == ["test_collection"]
    with pytest.raises(C.CollectionAlreadyExists):
        cl.create_collection("test_collection", C.IndexType.Flat)
    assert cl.get_collection("test_collection") is not None and cl.get_collection("nope") is None
    cl.delete_collection("test_collection")
    assert not cl.has_collection("test_collection")
    with pytest.raises(C.CollectionNotFound):
        cl.delete_collection("test_collection")
    with pytest.raises(C.CollectionNotFound):  # :625-633
        cl.add_text_to_collection("non_existent", "Hello world")


def test_collection_operations(C):  # :600-683
    import vectorlite_amd as V
    cl = C.VectorLiteClient(MockEmbeddingFunction(3))
    cl.create_collection("test_collection", C.IndexType.Flat)
    info = cl.get_collection_info("test_collection")
    assert info.is_empty and info.count == 0 and info.name == "test_collection"
    assert cl.add_text_to_collection("test_collection", "Hello world") == 0  # first id is 0
    assert cl.add_text_to_collection("test_collection", "Another text", {"a": 1}) == 1
    assert cl.get_collection_info("test_collection").count == 2
    # every stored vector is identical: all scores tie, the stable order returns id 0 (:665-667)
    res = cl.search_text_in_collection("test_collection", "Hello", 1, V.SimilarityMetric.Cosine)
    assert len(res) == 1 and res[0].id == 0 and res[0].text == "Hello world"
    res = cl.search_text_in_collection("test_collection", "Hello", 5)  # metric defaults to Cosine
    assert [r.id for r in res] == [0, 1] and res[1].metadata == {"a": 1}
    assert cl.get_vector_from_collection("test_collection", 0).id == 0
    cl.delete_from_collection("test_collection", 0)
    assert cl.get_collection_info("test_collection").count == 1
    assert cl.get_vector_from_collection("test_collection", 0) is None
    assert cl.add_text_to_collection("test_collection", "third") == 2  # ids are never reused


def test_hnsw_collection_and_metric_rules(C):  # :686-722
    import vectorlite_amd as V
    cl = C.VectorLiteClient(MockEmbeddingFunction(3))
    with pytest.raises(C.MetricRequired):
        cl.create_collection("hnsw_collection", C.IndexType.HNSW)
    cl.create_collection("hnsw_collection", C.IndexType.HNSW, V.SimilarityMetric.Euclidean)
    assert cl.add_text_to_collection("hnsw_collection", "First document") == 0
    assert cl.add_text_to_collection("hnsw_collection", "Second document") == 1
    assert cl.get_collection_info("hnsw_collection").count == 2
    assert len(cl.search_text_in_collection("hnsw_collection", "First", 1, V.SimilarityMetric.Euclidean)) == 1
    assert len(cl.search_text_in_collection("hnsw_collection", "First", 1)) == 1  # defaults to the index metric
    with pytest.raises(V.MetricMismatch):
        cl.search_text_in_collection("hnsw_collection", "First", 1, V.SimilarityMetric.Cosine)
    with pytest.raises(C.VectorNotFound):
        cl.delete_from_collection("hnsw_collection", 99)


def test_collection_save_and_load(C, tmp_path):  # :725-850
    import vectorlite_amd as V
    from vectorlite_amd import persistence as P
    cl = C.VectorLiteClient(MockEmbeddingFunction(3))
    cl.create_collection("test_collection", C.IndexType.Flat)
    cl.add_text_to_collection("test_collection", "Hello world")
    cl.add_text_to_collection("test_collection", "Another text")
    path = tmp_path / "nested" / "dir" / "test_collection.vlc"  # parent directories are created (:810-827)
    cl.get_collection("test_collection").save_to_file(str(path))
    assert path.exists()
    loaded = C.Collection.load_from_file(str(path))
    info = loaded.get_info()
    assert (info.name, info.count, info.dimension) == ("test_collection", 2, 3)
    assert loaded.next_id() == 2  # max_id + 1 (:297-308)
    assert loaded.add_text("new", MockEmbeddingFunction(3)) == 2
    res = loaded.search_text("Hello", 1, V.SimilarityMetric.Cosine, MockEmbeddingFunction(3))
    assert res[0].id == 0 and res[0].text == "Hello world"
    with pytest.raises(P.FileNotFound):
        C.Collection.load_from_file(str(tmp_path / "missing.vlc"))
    bad = tmp_path / "bad.vlc"
    bad.write_text("invalid json content")
    with pytest.raises(P.PersistenceError):
        C.Collection.load_from_file(str(bad))
    # HNSW round trip (:762-807)
    cl.create_collection("h", C.IndexType.HNSW, V.SimilarityMetric.Euclidean)
    cl.add_text_to_collection("h", "a")
    cl.add_text_to_collection("h", "b")
    hp = tmp_path / "h.vlc"
    cl.get_collection("h").save_to_file(str(hp))
    assert json.loads(hp.read_text())["metadata"]["index_type"] == "HNSW"
    lh = C.Collection.load_from_file(str(hp))
    assert lh.get_info().count == 2 and lh.next_id() == 2
    assert len(lh.search_text("a", 1, V.SimilarityMetric.Euclidean, MockEmbeddingFunction(3))) == 1


def test_dimension_mismatch_is_retyped(C):  # :334-345
    import vectorlite_amd as V

    class Bad(MockEmbeddingFunction):
        def generate_embedding(self, text):
            return [1.0, 2.0]

    cl = C.VectorLiteClient(MockEmbeddingFunction(3))
    cl.create_collection("c", C.IndexType.Flat)
    with pytest.raises(V.DimensionMismatch) as e:
        cl.get_collection("c").add_text("x", Bad(3))
    assert (e.value.expected, e.value.actual) == (3, 2)
