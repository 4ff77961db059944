"""ctypes binding of libvectorlite_amd.so (include/vectorlite_amd.h).

The library is the product: there is no Python or CPU fallback.  If the shared object has not
been built this module raises, and on a machine without a HIP device every compute entry
point returns VL_ERR_DEVICE (surfaced as :class:`vectorlite_amd.DeviceError`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VL_LIB_PATH: diagnostic builds of the same library (tools/, kernel anatomy runs); never a different implementation
SO_PATH = os.environ.get("VL_LIB_PATH") or os.path.join(_HERE, "libvectorlite_amd.so")

# every symbol include/vectorlite_amd.h declares
SYMBOLS = [
    "vl_flat_create", "vl_flat_from_rows", "vl_flat_create_multi", "vl_index_parts", "vl_hnsw_create", "vl_hnsw_create_ex", "vl_index_type", "vl_index_metric", "vl_index_search_ef", "vl_index_clone", "vl_index_destroy", "vl_index_reserve",
    "vl_index_add", "vl_index_add_bulk", "vl_index_add_embeddings_f32", "vl_index_delete", "vl_index_search", "vl_index_search_batch", "vl_index_search_cap", "vl_index_search_batch_cap",
    "vl_index_len", "vl_index_is_empty", "vl_index_dimension", "vl_index_get_vector", "vl_index_max_id",
    "vl_index_export", "vl_index_search_positions", "vl_index_search_batch_positions", "vl_index_search_batch_dev", "vl_index_search_batch_embeddings_f32", "vl_index_hnsw_distances", "vl_hnsw_score",
    "vl_last_error", "vl_last_dim_mismatch", "vl_last_path", "vl_index_force_path", "vl_index_set_single_filter",
    "vl_index_set_coalescing", "vl_index_coalesce_stats", "vl_index_coalesce_gather", "vl_index_hnsw_walk_stats",
    "vl_vlc_open", "vl_vlc_close", "vl_vlc_name", "vl_vlc_info", "vl_vlc_side_table", "vl_vlc_read_values", "vl_vlc_build_index",
    "vl_index_profile_enable", "vl_index_profile_read", "vl_index_last_scan", "vl_index_last_filter", "vl_runtime_info",
    "vl_comm_unique_id", "vl_comm_create", "vl_comm_destroy", "vl_comm_world", "vl_comm_rank", "vl_comm_profile_enable", "vl_comm_profile_read", "vl_comm_record_paths",
    "vl_index_hnsw_set_min_beam", "vl_index_hnsw_graph_info", "vl_index_hnsw_graph_export",
    "vl_shard_sync", "vl_shard_search_batch", "vl_shard_packed_words", "vl_shard_search_local", "vl_shard_merge", "vl_shard_search_batch_dev", "vl_shard_search_local_dev",
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build it with `python -m vectorlite_amd.build` "
            "(hipcc --offload-arch=gfx950).  vectorlite_amd has no CPU fallback.")
    # PyTorch's ROCm wheel bundles its own libamdhip64.so.7; loading it first makes this library
    # resolve to that same runtime instance, so torch device pointers and ours share one context.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = C.CDLL(SO_PATH)
    u64, i32, f64 = C.c_uint64, C.c_int, C.c_double
    p_u64, p_f64, vp = C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_void_p
    pp = C.POINTER(C.c_void_p)
    p_i32 = C.POINTER(C.c_int)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("vl_flat_create", i32, [u64, i32, pp])
    sig("vl_flat_from_rows", i32, [u64, p_u64, p_f64, u64, i32, pp])
    sig("vl_flat_create_multi", i32, [u64, p_i32, i32, i32, pp])
    sig("vl_index_parts", i32, [vp, p_i32, p_i32, p_u64, p_u64, i32])
    sig("vl_hnsw_create", i32, [u64, i32, i32, pp])
    sig("vl_hnsw_create_ex", i32, [u64, i32, C.c_uint32, C.c_uint32, C.c_uint32, u64, i32, pp])
    sig("vl_index_type", i32, [vp])
    sig("vl_index_metric", i32, [vp, C.POINTER(C.c_int)])
    sig("vl_index_search_ef", i32, [vp, p_f64, u64, u64, u64, C.c_uint32, i32, p_u64, p_f64, p_u64])
    sig("vl_index_clone", i32, [vp, pp])
    sig("vl_index_destroy", None, [vp])
    sig("vl_index_reserve", i32, [vp, u64])
    sig("vl_index_add", i32, [vp, u64, p_f64, u64])
    sig("vl_index_add_bulk", i32, [vp, p_u64, vp, u64, i32, i32])
    sig("vl_index_add_embeddings_f32", i32, [vp, p_u64, vp, u64, i32, i32, i32])
    sig("vl_index_delete", i32, [vp, u64])
    sig("vl_index_search", i32, [vp, p_f64, u64, u64, i32, p_u64, p_f64, p_u64])
    sig("vl_index_search_batch", i32, [vp, p_f64, u64, u64, u64, i32, p_u64, p_f64, p_u64])
    # the hot single-query entry takes raw addresses (no ctypes pointer objects made per call)
    sig("vl_index_search_cap", i32, [vp, vp, u64, u64, i32, u64, vp, vp, vp])
    sig("vl_index_search_batch_cap", i32, [vp, p_f64, u64, u64, u64, i32, u64, p_u64, p_f64, p_u64])
    sig("vl_index_len", u64, [vp])
    sig("vl_index_is_empty", i32, [vp])
    sig("vl_index_dimension", u64, [vp])
    sig("vl_index_get_vector", i32, [vp, u64, p_f64])
    sig("vl_index_max_id", i32, [vp, p_u64])
    sig("vl_index_export", i32, [vp, p_u64, p_f64])
    sig("vl_index_search_positions", i32, [vp, p_f64, u64, u64, i32, p_u64, p_u64, p_f64, p_u64])
    sig("vl_index_search_batch_positions", i32, [vp, p_f64, u64, u64, u64, i32, p_u64, p_u64, p_f64, p_u64])
    sig("vl_index_search_batch_dev", i32, [vp, vp, u64, u64, u64, i32, p_u64, p_u64, p_f64, p_u64])
    sig("vl_index_search_batch_embeddings_f32", i32, [vp, vp, u64, u64, i32, i32, u64, i32, p_u64, p_f64, p_u64])
    sig("vl_index_hnsw_distances", i32, [vp, p_f64, u64, i32, p_u64, u64, p_u64])
    sig("vl_hnsw_score", f64, [u64, i32])
    sig("vl_last_error", C.c_char_p, [])
    sig("vl_last_dim_mismatch", None, [p_u64, p_u64])
    sig("vl_last_path", i32, [])
    sig("vl_index_force_path", i32, [vp, i32])
    sig("vl_index_set_single_filter", i32, [vp, i32])
    sig("vl_vlc_open", i32, [C.c_char_p, C.POINTER(vp)])
    sig("vl_vlc_close", None, [vp])
    sig("vl_vlc_name", C.c_char_p, [vp])
    sig("vl_vlc_info", i32, [vp, p_i32, p_i32, p_u64, p_u64, p_u64, p_u64])
    sig("vl_vlc_side_table", i32, [vp, p_u64, p_u64, p_u64, p_u64, p_u64])
    sig("vl_vlc_read_values", i32, [vp, u64, u64, p_f64])
    sig("vl_vlc_build_index", i32, [vp, i32, C.POINTER(vp)])
    sig("vl_index_set_coalescing", i32, [vp, i32, i32])
    sig("vl_index_coalesce_stats", i32, [vp, p_u64, p_u64])
    sig("vl_index_coalesce_gather", i32, [vp, i32, p_u64, p_u64])
    sig("vl_index_hnsw_walk_stats", i32, [vp, p_u64, p_u64])
    sig("vl_index_hnsw_set_min_beam", i32, [vp, C.c_uint32])
    p_u32 = C.POINTER(C.c_uint32)
    sig("vl_index_hnsw_graph_info", i32, [vp, p_u64, p_u32, p_i32, p_u32, p_u32, p_u64])
    sig("vl_index_hnsw_graph_export", i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp])
    sig("vl_index_profile_enable", i32, [vp, i32])
    sig("vl_index_profile_read", i32, [vp, p_u64, p_f64, p_u64])
    sig("vl_index_last_scan", i32, [vp, p_i32, p_i32, p_i32])
    sig("vl_index_last_filter", i32, [vp, p_i32])
    sig("vl_runtime_info", i32, [C.POINTER(C.c_int), C.POINTER(C.c_int)])
    p_u8 = C.POINTER(C.c_uint8)
    sig("vl_comm_unique_id", i32, [p_u8])
    sig("vl_comm_create", i32, [p_u8, i32, i32, i32, pp])
    sig("vl_comm_destroy", None, [vp])
    sig("vl_comm_world", i32, [vp])
    sig("vl_comm_rank", i32, [vp])
    sig("vl_comm_profile_enable", i32, [vp, i32])
    sig("vl_comm_profile_read", i32, [vp, p_u64, p_f64, p_f64, p_f64, p_f64])
    sig("vl_comm_record_paths", i32, [vp, p_u64, p_u64])
    sig("vl_shard_sync", i32, [vp, vp, p_u64, p_u64])
    sig("vl_shard_search_batch", i32, [vp, vp, p_f64, u64, u64, u64, i32, p_u64, p_u64, p_f64, p_u64])
    sig("vl_shard_packed_words", u64, [u64, u64])
    sig("vl_shard_search_local", i32, [vp, u64, i32, p_f64, u64, u64, u64, i32, p_u64])
    sig("vl_shard_search_batch_dev", i32, [vp, vp, vp, u64, u64, u64, i32, p_u64, p_u64, p_f64, p_u64])
    sig("vl_shard_search_local_dev", i32, [vp, u64, i32, vp, u64, u64, u64, i32, p_u64])
    sig("vl_shard_merge", i32, [i32, p_u64, C.c_uint32, u64, u64, u64, p_u64, p_u64, p_f64, p_u64])
    _lib = L
    return L
