""".vlc collection files <-> GPU indexes (the step before the hot path).

Format (reference src/persistence.rs:60-96): pretty JSON
    {"header":   {"version": "1.0.0", "format": "vectorlite-collection", "created_at": ...},
     "metadata": {"name", "created_at", "vector_count", "dimension", "index_type": "Flat" | "HNSW"},
     "index":    {"Flat": {"dim", "data": [{"id", "values", "text", "metadata"}, ...]}}
               | {"HNSW": {"dim", "metric", "id_to_index", "index_to_id", "metadata": {id: {"text", "metadata"}},
                           "vector_values": {id: [f64, ...]}}}}
Loading validates version and format like load_collection_from_file (src/persistence.rs:149-176);
a Flat payload is taken as is (serde fills FlatIndex{dim, data} with no validation, src/index/flat.rs:59),
an HNSW payload is rebuilt by re-inserting every vector (src/index/hnsw.rs:272-360).  Saving writes a
temporary file and renames it (src/persistence.rs:129-146).  Only JSON handling happens here: rows go
to the device through the same C ABI as every other ingest.
"""
from __future__ import annotations

import datetime
import json
import os
from typing import Tuple

import numpy as np

from . import FlatIndex, HNSWIndex, SimilarityMetric, VectorLiteError

VERSION = "1.0.0"
FORMAT = "vectorlite-collection"
_METRIC_NAMES = {"Cosine": SimilarityMetric.Cosine, "Euclidean": SimilarityMetric.Euclidean,
                 "Manhattan": SimilarityMetric.Manhattan, "DotProduct": SimilarityMetric.DotProduct}


class PersistenceError(VectorLiteError):
    pass


class VersionMismatch(PersistenceError):
    def __init__(self, expected, actual):
        super().__init__(f"Version mismatch: expected {expected}, got {actual}")
        self.expected, self.actual = expected, actual


class InvalidFormat(PersistenceError):
    pass


class FileNotFound(PersistenceError):
    pass


def parse_collection(text: str) -> dict:
    """JSON text -> validated CollectionData dict (no device work)."""
    try:
        data = json.loads(text)
    except json.JSONDecodeError as e:
        raise PersistenceError(f"Serialization error: {e}") from e
    for key in ("header", "metadata", "index"):
        if key not in data:
            raise PersistenceError(f"Serialization error: missing field `{key}`")
    if data["header"].get("version") != VERSION:
        raise VersionMismatch(VERSION, data["header"].get("version"))
    if data["header"].get("format") != FORMAT:
        raise InvalidFormat(f"Expected format '{FORMAT}', got '{data['header'].get('format')}'")
    return data


def index_from_payload(payload: dict, device: int = 0):
    """The `index` member (externally tagged VectorIndexWrapper) -> a GPU index."""
    if "Flat" in payload:
        p = payload["Flat"]
        dim = int(p["dim"])
        idx = FlatIndex(dim, device=device)
        rows = p.get("data", [])
        if rows:
            ids = np.array([int(r["id"]) for r in rows], dtype=np.uint64)
            vals = np.array([r["values"] for r in rows], dtype=np.float64).reshape(len(rows), -1)
            if vals.shape[1] != dim:
                raise PersistenceError("Serialization error: row length differs from dim")
            idx.add_rows(ids, vals, validate=False)
            for r in rows:
                idx._meta.setdefault(int(r["id"]), (r.get("text", ""), r.get("metadata")))
        return idx
    if "HNSW" in payload:
        p = payload["HNSW"]
        dim = int(p["dim"])
        if dim == 0:
            raise PersistenceError("Invalid dimension: cannot be 0")  # src/index/hnsw.rs:288-290
        metric = _METRIC_NAMES[p["metric"]]
        idx = HNSWIndex(dim, metric, device=device)
        vv = p.get("vector_values", {})
        if vv:
            ids = np.array([int(k) for k in vv.keys()], dtype=np.uint64)
            vals = np.array(list(vv.values()), dtype=np.float64).reshape(len(vv), -1)
            if vals.shape[1] != dim:
                raise PersistenceError(f"Vector dimension mismatch: expected {dim}, got {vals.shape[1]}")
            idx.add_rows(ids, vals)
        for k, m in p.get("metadata", {}).items():
            idx._meta[int(k)] = (m.get("text", ""), m.get("metadata"))
        return idx
    raise PersistenceError("Serialization error: unknown index variant")


def load_collection_from_file(path: str, device: int = 0) -> Tuple[str, object]:
    """(collection name, GPU index) from a .vlc file."""
    try:
        with open(path, "r") as f:
            text = f.read()
    except FileNotFoundError as e:
        raise FileNotFound(str(path)) from e
    data = parse_collection(text)
    return data["metadata"]["name"], index_from_payload(data["index"], device=device)


def _now() -> str:
    return datetime.datetime.now(datetime.timezone.utc).isoformat().replace("+00:00", "Z")


def payload_from_index(index) -> Tuple[str, dict]:
    if isinstance(index, FlatIndex):
        ids, vals = index.export()
        rows = []
        for i, v in zip(ids.tolist(), vals.tolist()):
            text, md = index._meta.get(i, ("", None))
            rows.append({"id": i, "values": v, "text": text, "metadata": md})
        return "Flat", {"Flat": {"dim": index.dimension(), "data": rows}}
    if isinstance(index, HNSWIndex):
        ids = sorted(index._meta.keys()) if index._meta else []
        live = [i for i in ids if index.get_vector(i) is not None]
        vv = {str(i): index.get_vector(i).values for i in live}
        md = {str(i): {"text": index._meta[i][0], "metadata": index._meta[i][1]} for i in live}
        n2i = {str(i): n for n, i in enumerate(live)}
        return "HNSW", {"HNSW": {"dim": index.dimension(), "metric": index.metric().name,
                                 "id_to_index": n2i, "index_to_id": {str(n): i for n, i in enumerate(live)},
                                 "metadata": md, "vector_values": vv}}
    raise PersistenceError("unsupported index type")


def save_collection_to_file(name: str, index, path: str) -> None:
    kind, payload = payload_from_index(index)
    data = {"header": {"version": VERSION, "format": FORMAT, "created_at": _now()},
            "metadata": {"name": name, "created_at": _now(), "vector_count": len(index),
                         "dimension": index.dimension(), "index_type": kind},
            "index": payload}
    parent = os.path.dirname(os.path.abspath(path))
    os.makedirs(parent, exist_ok=True)
    tmp = os.path.splitext(path)[0] + ".tmp"
    with open(tmp, "w") as f:
        json.dump(data, f, indent=2)
    os.replace(tmp, path)
