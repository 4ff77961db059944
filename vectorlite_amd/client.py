"""Caller-side mirror of the reference's client layer (src/client.rs) -- SURVEY section 8(f) row f1.

`VectorLiteClient` / `Collection` keep the reference's behaviour: collections by name, `IndexType`
{Flat, HNSW} (HNSW requires a metric, src/client.rs:69-72), ids allocated from an atomic counter that
starts at 0 or at max_id + 1 after a load (:295-315), embedding OUTSIDE the index lock (:353, :395),
string errors of `add`/`delete` re-typed by substring (:334-345, :384-390), metric defaulting to the
index's own metric or Cosine (:143-155).  The index underneath is the GPU one.  Embedding models are
out of scope: `embedding_function` is any object with `generate_embedding(text) -> list[float]` and
`dimension()` (trait EmbeddingFunction, src/embeddings.rs:135-141).
"""
from __future__ import annotations

import enum
import itertools
import threading
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

from . import (DimensionMismatch, FlatIndex, HNSWIndex, IndexOpError, SearchResult, SimilarityMetric, Vector,
               VectorLiteError)


class IndexType(enum.Enum):
    Flat = "flat"
    HNSW = "hnsw"


class CollectionAlreadyExists(VectorLiteError):
    pass


class CollectionNotFound(VectorLiteError):
    pass


class MetricRequired(VectorLiteError):
    pass


class DuplicateVectorId(VectorLiteError):
    pass


class VectorNotFound(VectorLiteError):
    pass


@dataclass
class CollectionInfo:
    name: str
    count: int
    is_empty: bool
    dimension: int


class Collection:
    def __init__(self, name: str, index):
        self._name = name
        self.index = index
        mx = index.max_id()
        self._next = itertools.count(0 if mx is None else mx + 1)  # src/client.rs:297-308
        self._next_peek = 0 if mx is None else mx + 1
        self._lock = threading.Lock()  # guards the counter only; the index has its own RW discipline

    def name(self) -> str:
        return self._name

    def next_id(self) -> int:
        return self._next_peek

    def _alloc_id(self) -> int:
        with self._lock:
            i = next(self._next)
            self._next_peek = i + 1
            return i

    def add_text_with_metadata(self, text: str, metadata: Optional[Any], embedding_function) -> int:
        vid = self._alloc_id()
        embedding = embedding_function.generate_embedding(text)  # outside any index lock
        try:
            self.index.add(Vector(id=vid, values=embedding, text=text, metadata=metadata))
        except IndexOpError as e:
            msg = str(e)
            if "dimension" in msg:
                raise DimensionMismatch(self.index.dimension(), len(embedding)) from e
            if "already exists" in msg:
                raise DuplicateVectorId(f"Vector ID {vid} already exists") from e
            raise VectorLiteError(msg) from e
        return vid

    def add_text(self, text: str, embedding_function) -> int:
        return self.add_text_with_metadata(text, None, embedding_function)

    def delete(self, id: int) -> None:
        try:
            self.index.delete(id)
        except IndexOpError as e:
            if "does not exist" in str(e):
                raise VectorNotFound(f"Vector ID {id} not found") from e
            raise VectorLiteError(str(e)) from e

    def search_text(self, query_text: str, k: int, similarity_metric: SimilarityMetric, embedding_function
                    ) -> List[SearchResult]:
        return self.index.search(embedding_function.generate_embedding(query_text), k, similarity_metric)

    def get_vector(self, id: int) -> Optional[Vector]:
        return self.index.get_vector(id)

    def get_info(self) -> CollectionInfo:
        return CollectionInfo(self._name, len(self.index), self.index.is_empty(), self.index.dimension())

    def save_to_file(self, path: str) -> None:
        from . import persistence
        persistence.save_collection_to_file(self._name, self.index, path)

    @classmethod
    def load_from_file(cls, path: str, device: int = 0) -> "Collection":
        from . import persistence
        name, index = persistence.load_collection_from_file(path, device=device)
        return cls(name, index)


class VectorLiteClient:
    def __init__(self, embedding_function, device: int = 0):
        self.embedding_function = embedding_function
        self.device = device
        self.collections: Dict[str, Collection] = {}

    def create_collection(self, name: str, index_type: IndexType, metric: Optional[SimilarityMetric] = None) -> None:
        if name in self.collections:
            raise CollectionAlreadyExists(name)
        dim = self.embedding_function.dimension()
        if index_type == IndexType.Flat:
            index = FlatIndex(dim, device=self.device)
        else:
            if metric is None:
                raise MetricRequired("HNSW requires a similarity metric")
            index = HNSWIndex(dim, metric, device=self.device)
        self.collections[name] = Collection(name, index)

    def add_collection(self, collection: Collection) -> None:
        if collection.name() in self.collections:
            raise CollectionAlreadyExists(collection.name())
        self.collections[collection.name()] = collection

    def get_collection(self, name: str) -> Optional[Collection]:
        return self.collections.get(name)

    def list_collections(self) -> List[str]:
        return list(self.collections.keys())

    def has_collection(self, name: str) -> bool:
        return name in self.collections

    def delete_collection(self, name: str) -> None:
        if self.collections.pop(name, None) is None:
            raise CollectionNotFound(name)

    def _get(self, name: str) -> Collection:
        c = self.collections.get(name)
        if c is None:
            raise CollectionNotFound(name)
        return c

    def add_text_to_collection(self, collection_name: str, text: str, metadata: Optional[Any] = None) -> int:
        return self._get(collection_name).add_text_with_metadata(text, metadata, self.embedding_function)

    def search_text_in_collection(self, collection_name: str, query_text: str, k: int,
                                  similarity_metric: Optional[SimilarityMetric] = None) -> List[SearchResult]:
        c = self._get(collection_name)
        if similarity_metric is None:  # src/client.rs:143-155
            similarity_metric = c.index.metric() if isinstance(c.index, HNSWIndex) else SimilarityMetric.Cosine
        return c.search_text(query_text, k, similarity_metric, self.embedding_function)

    def delete_from_collection(self, collection_name: str, id: int) -> None:
        self._get(collection_name).delete(id)

    def get_vector_from_collection(self, collection_name: str, id: int) -> Optional[Vector]:
        return self._get(collection_name).get_vector(id)

    def get_collection_info(self, collection_name: str) -> CollectionInfo:
        return self._get(collection_name).get_info()
