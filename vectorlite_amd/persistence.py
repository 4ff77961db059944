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
temporary file and renames it (src/persistence.rs:129-146).

Loading goes through the library's streaming reader (csrc/vlc_loader.cpp, C ABI vl_vlc_*): the file is
mapped, one structural pass locates the rows, host threads convert the numbers into staging blocks that
are ingested on the device -- no Vec<Vector> (or Python list of lists) is ever built.  This module only
decodes the k-independent side data (text / metadata tokens) from the byte ranges the reader reports.
`parse_collection` / `index_from_payload` remain for callers that already hold a parsed document.
"""
from __future__ import annotations

import ctypes as C
import datetime
import json
import mmap
import os
from typing import Tuple

import numpy as np

from . import FlatIndex, HNSWIndex, SimilarityMetric, VectorLiteError, _last_error, _lib, _raise

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


VL_ERR_IO, VL_ERR_FILE_NOT_FOUND, VL_ERR_SERIALIZATION, VL_ERR_VERSION_MISMATCH, VL_ERR_INVALID_FORMAT = 9, 10, 11, 12, 13


def _raise_vlc(rc: int, path: str):
    if rc == 0:
        return
    msg = _last_error()
    if rc == VL_ERR_FILE_NOT_FOUND:
        raise FileNotFound(str(path))
    if rc == VL_ERR_VERSION_MISMATCH:
        raise VersionMismatch(VERSION, msg.split("got ", 1)[-1])
    if rc == VL_ERR_INVALID_FORMAT:
        raise InvalidFormat(msg.split("Invalid file format: ", 1)[-1])
    if rc in (VL_ERR_SERIALIZATION, VL_ERR_IO):
        raise PersistenceError(msg)
    _raise(rc)


class VlcDocument:
    """An opened .vlc file: header validated, rows located, nothing converted yet (host work only)."""

    def __init__(self, path: str):
        self._L = _lib.load()
        self._d = C.c_void_p()
        self.path = str(path)
        _raise_vlc(self._L.vl_vlc_open(self.path.encode(), C.byref(self._d)), path)
        it, me = C.c_int(0), C.c_int(0)
        dim, rows, vc, dm = (C.c_uint64(0) for _ in range(4))
        self._L.vl_vlc_info(self._d, C.byref(it), C.byref(me), C.byref(dim), C.byref(rows), C.byref(vc), C.byref(dm))
        self.index_type = "Flat" if it.value == 0 else "HNSW"
        self.metric = None if me.value < 0 else SimilarityMetric(me.value)
        self.dim, self.rows = int(dim.value), int(rows.value)
        self.vector_count, self.dimension = int(vc.value), int(dm.value)
        self.name = self._L.vl_vlc_name(self._d).decode("utf-8")

    def close(self):
        if getattr(self, "_d", None):
            self._L.vl_vlc_close(self._d)
            self._d = None

    __del__ = close

    def values(self, first: int = 0, n: int = None) -> np.ndarray:
        """rows [first, first + n) as an [n, dim] f64 array (converted on the library's host threads)."""
        n = self.rows - first if n is None else n
        out = np.empty((max(n, 1), max(self.dim, 1)), dtype=np.float64)
        _raise_vlc(self._L.vl_vlc_read_values(self._d, first, n, out.ctypes.data_as(C.POINTER(C.c_double))), self.path)
        return out[:n, : self.dim]

    def side_table(self):
        """ids plus the (offset, length) byte ranges of each row's text / metadata token in the file."""
        arrs = [np.zeros(max(self.rows, 1), dtype=np.uint64) for _ in range(5)]
        self._L.vl_vlc_side_table(self._d, *[a.ctypes.data_as(C.POINTER(C.c_uint64)) for a in arrs])
        return [a[: self.rows] for a in arrs]

    def side_data(self, first_wins: bool) -> dict:
        """id -> (text, metadata), decoded from the file's own tokens."""
        ids, toff, tlen, moff, mlen = self.side_table()
        out = {}
        if self.rows == 0:
            return out
        with open(self.path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            for i in range(self.rows):
                key = int(ids[i])
                if first_wins and key in out:
                    continue
                text = json.loads(mm[int(toff[i]): int(toff[i] + tlen[i])]) if tlen[i] else ""
                md = json.loads(mm[int(moff[i]): int(moff[i] + mlen[i])]) if mlen[i] else None
                out[key] = (text, md)
        return out

    def build_index(self, device: int = 0):
        h = C.c_void_p()
        _raise_vlc(self._L.vl_vlc_build_index(self._d, int(device), C.byref(h)), self.path)
        if self.index_type == "Flat":
            idx = FlatIndex(self.dim, device=device, _handle=h)
            idx._meta = self.side_data(first_wins=True)  # get_vector finds the first row of an id
        else:
            idx = HNSWIndex(self.dim, self.metric, device=device, _handle=h)
            idx._meta = self.side_data(first_wins=False)
        return idx


def load_collection_from_file(path: str, device: int = 0) -> Tuple[str, object]:
    """(collection name, GPU index) from a .vlc file (src/persistence.rs:149-176)."""
    doc = VlcDocument(path)
    try:
        return doc.name, doc.build_index(device)
    finally:
        doc.close()


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
        ids, vals = index.export()  # live rows, insertion order
        live = ids.tolist()
        vv = {str(i): v for i, v in zip(live, vals.tolist())}
        md = {str(i): {"text": index._meta.get(i, ("", None))[0], "metadata": index._meta.get(i, ("", None))[1]} for i in live}
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
