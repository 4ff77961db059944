//! GPU-resident indexes for vectorlite: `impl VectorIndex` over libvectorlite_amd.so
//! (C ABI: include/vectorlite_amd.h of the vectorlite_amd repository).
//!
//! NOT compiled in the repository that ships it (its image has no Rust toolchain); every `extern`
//! signature below is checked against the header by tests/test_rust_binding_signatures.py.
//!
//! Vectors live in HBM (f64 master rows + an f32 scan slab); `text` / `metadata` stay on the host in a
//! side table keyed by id and are re-attached to the k winners only (the CPU FlatIndex clones them for
//! all N rows before sorting).
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

use serde::{Deserialize, Deserializer, Serialize, Serializer};

use crate::errors::{VectorLiteError, VectorLiteResult};
use crate::{SearchResult, SimilarityMetric, Vector, VectorIndex};

#[repr(C)]
pub struct vl_index {
    _private: [u8; 0],
}
#[repr(C)]
pub struct vl_vlc_doc {
    _private: [u8; 0],
}
#[repr(C)]
pub struct vl_comm {
    _private: [u8; 0],
}

extern "C" {
    fn vl_flat_create(dim: u64, device: c_int, out: *mut *mut vl_index) -> c_int;
    fn vl_flat_from_rows(dim: u64, ids: *const u64, values: *const f64, n: u64, device: c_int, out: *mut *mut vl_index) -> c_int;
    fn vl_hnsw_create(dim: u64, metric: c_int, device: c_int, out: *mut *mut vl_index) -> c_int;
    fn vl_index_clone(h: *const vl_index, out: *mut *mut vl_index) -> c_int;
    fn vl_index_destroy(h: *mut vl_index);
    fn vl_index_add(h: *mut vl_index, id: u64, values: *const f64, len: u64) -> c_int;
    fn vl_index_add_bulk(h: *mut vl_index, ids: *const u64, values: *const f64, n: u64, validate: c_int, values_on_device: c_int) -> c_int;
    fn vl_index_add_embeddings_f32(h: *mut vl_index, ids: *const u64, embeddings: *const f32, n: u64, normalize: c_int, validate: c_int, embeddings_on_device: c_int) -> c_int;
    fn vl_index_delete(h: *mut vl_index, id: u64) -> c_int;
    fn vl_index_search(h: *const vl_index, query: *const f64, q_len: u64, k: u64, metric: c_int, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_index_search_cap(h: *const vl_index, query: *const f64, q_len: u64, k: u64, metric: c_int, out_capacity: u64, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_index_search_batch_cap(h: *const vl_index, queries: *const f64, nq: u64, q_len: u64, k: u64, metric: c_int, out_stride: u64, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_flat_create_multi(dim: u64, device_ids: *const c_int, n_dev: c_int, mode: c_int, out: *mut *mut vl_index) -> c_int;
    fn vl_index_len(h: *const vl_index) -> u64;
    fn vl_index_get_vector(h: *const vl_index, id: u64, out: *mut f64) -> c_int;
    fn vl_index_export(h: *const vl_index, out_ids: *mut u64, out_values: *mut f64) -> c_int;
    #[allow(dead_code)]
    fn vl_index_set_coalescing(h: *mut vl_index, max_batch: c_int, window_us: c_int) -> c_int;
    fn vl_index_search_batch(h: *const vl_index, queries: *const f64, nq: u64, q_len: u64, k: u64, metric: c_int, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_index_search_ef(h: *const vl_index, queries: *const f64, nq: u64, q_len: u64, k: u64, ef: u32, metric: c_int, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_vlc_open(path: *const c_char, out: *mut *mut vl_vlc_doc) -> c_int;
    fn vl_vlc_close(doc: *mut vl_vlc_doc);
    fn vl_vlc_build_index(doc: *const vl_vlc_doc, device: c_int, out: *mut *mut vl_index) -> c_int;
    fn vl_last_error() -> *const c_char;
    fn vl_last_dim_mismatch(expected: *mut u64, actual: *mut u64);
    // row-sharded batched search: one process per GPU, one RCCL all-gather per batch
    fn vl_comm_unique_id(out_id: *mut u8) -> c_int;
    fn vl_comm_create(id: *const u8, world: c_int, rank: c_int, device: c_int, out: *mut *mut vl_comm) -> c_int;
    fn vl_comm_destroy(comm: *mut vl_comm);
    fn vl_shard_sync(shard: *const vl_index, comm: *mut vl_comm, out_offset: *mut u64, out_total: *mut u64) -> c_int;
    fn vl_shard_search_batch(shard: *const vl_index, comm: *mut vl_comm, queries: *const f64, nq: u64, q_len: u64, k: u64, metric: c_int, out_gpos: *mut u64, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_shard_search_batch_dev(shard: *const vl_index, comm: *mut vl_comm, d_queries: *const f64, nq: u64, q_len: u64, k: u64, metric: c_int, out_gpos: *mut u64, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_index_search_batch_dev(h: *const vl_index, d_queries: *const f64, nq: u64, q_len: u64, k: u64, metric: c_int, out_pos: *mut u64, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
    fn vl_index_search_batch_embeddings_f32(h: *const vl_index, embeddings: *const f32, nq: u64, dim: u64, normalize: c_int, embeddings_on_device: c_int, k: u64, metric: c_int, out_ids: *mut u64, out_scores: *mut f64, out_n: *mut u64) -> c_int;
}

pub const VL_COMM_ID_BYTES: usize = 128;

const VL_OK: c_int = 0;
const VL_ERR_DIM_MISMATCH: c_int = 1;
const VL_ERR_DUP_ID: c_int = 2;
const VL_ERR_NOT_FOUND: c_int = 3;
const VL_ERR_METRIC_MISMATCH: c_int = 4;
const VL_ERR_NAN_SCORE: c_int = 5;

fn last_error() -> String {
    unsafe { CStr::from_ptr(vl_last_error()).to_string_lossy().into_owned() }
}

/// Declaration order of `enum SimilarityMetric` = the ABI's metric codes.
fn metric_code(m: SimilarityMetric) -> c_int {
    match m {
        SimilarityMetric::Cosine => 0,
        SimilarityMetric::Euclidean => 1,
        SimilarityMetric::Manhattan => 2,
        SimilarityMetric::DotProduct => 3,
    }
}

type Side = HashMap<u64, (String, Option<serde_json::Value>)>;

/// Shared body of the two index types: the handle dispatches like `VectorIndexWrapper`.
#[derive(Debug)]
struct Handle {
    raw: *mut vl_index,
    dim: usize,
    side: Side,
}
// The library takes its own locks: searches are re-entrant (callers hold RwLock::read together),
// add/delete take the handle exclusively.
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}

impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { vl_index_destroy(self.raw) }
    }
}

impl Clone for Handle {
    fn clone(&self) -> Self {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { vl_index_clone(self.raw, &mut raw) };
        assert_eq!(rc, VL_OK, "vl_index_clone: {}", last_error());
        Handle { raw, dim: self.dim, side: self.side.clone() }
    }
}

impl Handle {
    fn add(&mut self, v: Vector) -> Result<(), String> {
        match unsafe { vl_index_add(self.raw, v.id, v.values.as_ptr(), v.values.len() as u64) } {
            VL_OK => {
                self.side.insert(v.id, (v.text, v.metadata));
                Ok(())
            }
            VL_ERR_DIM_MISMATCH => Err("Vector dimension mismatch".to_string()),
            VL_ERR_DUP_ID => Err(format!("Vector ID {} already exists", v.id)),
            _ => Err(last_error()),
        }
    }

    fn search(&self, query: &[f64], k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<SearchResult>> {
        // vl_index_search_cap writes min(k, len at search time, cap) entries: the bound is this buffer's own size,
        // whatever the index length is by the time the search runs (in Rust `&mut self` on add already excludes a
        // concurrent grow; a C or Python caller has no such lock).
        let cap = k.min(self.len()).max(1);
        let (mut ids, mut scores, mut n) = (vec![0u64; cap], vec![0f64; cap], 0u64);
        let rc = unsafe {
            vl_index_search_cap(self.raw, query.as_ptr(), query.len() as u64, k as u64, metric_code(metric), cap as u64, ids.as_mut_ptr(), scores.as_mut_ptr(), &mut n)
        };
        match rc {
            VL_OK => Ok((0..n as usize)
                .map(|i| {
                    let (text, metadata) = self.side.get(&ids[i]).cloned().unwrap_or_default();
                    SearchResult { id: ids[i], score: scores[i], text, metadata }
                })
                .collect()),
            VL_ERR_DIM_MISMATCH => {
                let (mut e, mut a) = (0u64, 0u64);
                unsafe { vl_last_dim_mismatch(&mut e, &mut a) };
                Err(VectorLiteError::DimensionMismatch { expected: e as usize, actual: a as usize })
            }
            VL_ERR_METRIC_MISMATCH => Err(VectorLiteError::InternalError(last_error())), // caller maps to MetricMismatch
            VL_ERR_NAN_SCORE => panic!("NaN similarity score"), // what partial_cmp().unwrap() does on the CPU index
            _ => Err(VectorLiteError::InternalError(last_error())),
        }
    }

    fn len(&self) -> usize {
        unsafe { vl_index_len(self.raw) as usize }
    }

    fn get_vector(&self, id: u64) -> Option<Vector> {
        let mut values = vec![0f64; self.dim];
        if unsafe { vl_index_get_vector(self.raw, id, values.as_mut_ptr()) } != VL_OK {
            return None;
        }
        let (text, metadata) = self.side.get(&id).cloned().unwrap_or_default();
        Some(Vector { id, values, text, metadata })
    }

    /// (ids, row-major values) of every live row, insertion order: the serde payload.
    fn export(&self) -> (Vec<u64>, Vec<f64>) {
        let n = self.len();
        let (mut ids, mut values) = (vec![0u64; n], vec![0f64; n * self.dim]);
        if n > 0 {
            let rc = unsafe { vl_index_export(self.raw, ids.as_mut_ptr(), values.as_mut_ptr()) };
            assert_eq!(rc, VL_OK, "vl_index_export: {}", last_error());
        }
        (ids, values)
    }
}

// ------------------------------------------------------------------------------------------------
// Flat
// ------------------------------------------------------------------------------------------------
#[derive(Debug, Clone)]
pub struct GpuFlatIndex(Handle);

impl GpuFlatIndex {
    /// Mirrors `FlatIndex::new(dim, data)`: no validation of `data`.
    pub fn new(dim: usize, data: Vec<Vector>) -> Self {
        let ids: Vec<u64> = data.iter().map(|v| v.id).collect();
        let mut values = Vec::with_capacity(data.len() * dim);
        for v in &data {
            values.extend_from_slice(&v.values);
        }
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { vl_flat_from_rows(dim as u64, ids.as_ptr(), values.as_ptr(), ids.len() as u64, 0, &mut raw) };
        assert_eq!(rc, VL_OK, "vl_flat_from_rows: {}", last_error());
        let mut side = Side::new();
        for v in data {
            side.entry(v.id).or_insert((v.text, v.metadata)); // first row of an id wins, like get_vector's find()
        }
        // tokio workers search concurrently under RwLock::read and share slab passes: the library does that by default
        // (up to 256 queries per pass, window 0, and a leader waits briefly for peers still on their way back from the
        // previous pass -- vl_index_coalesce_gather); vl_index_set_coalescing(raw, 0, 0) would turn it off
        GpuFlatIndex(Handle { raw, dim, side })
    }
}

/// How `GpuFlatIndex::new_multi` spreads one index over the GPUs of the node (include/vectorlite_amd.h).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum MultiGpuMode {
    /// every GPU holds every row; concurrent `search` calls (tokio workers under `RwLock::read`, src/client.rs:398)
    /// are dealt to the least busy replica
    Replicas = 0,
    /// every GPU holds part of the rows; each search runs on all of them, exact per-shard top-k merged on the device
    RowShards = 1,
}

impl GpuFlatIndex {
    /// `FlatIndex::new(dim, data)` over several GPUs in THIS process (`vl_flat_create_multi`): the value behaves as any
    /// other `GpuFlatIndex` behind `VectorIndexWrapper` -- same trait, same results -- but the server's one
    /// `Arc<RwLock<Collection>>` (src/client.rs:243-247) now keeps all `devices` busy.  No rank processes, no id handshake.
    pub fn new_multi(dim: usize, data: Vec<Vector>, devices: &[i32], mode: MultiGpuMode) -> Self {
        let mut raw = std::ptr::null_mut();
        let devs: Vec<c_int> = devices.iter().map(|d| *d as c_int).collect();
        let rc = unsafe { vl_flat_create_multi(dim as u64, devs.as_ptr(), devs.len() as c_int, mode as c_int, &mut raw) };
        assert_eq!(rc, VL_OK, "vl_flat_create_multi: {}", last_error());
        let ids: Vec<u64> = data.iter().map(|v| v.id).collect();
        let mut values = Vec::with_capacity(data.len() * dim);
        for v in &data {
            values.extend_from_slice(&v.values);
        }
        if !ids.is_empty() {
            // FlatIndex::new validates nothing (src/index/flat.rs:68-73): validate = 0
            let rc = unsafe { vl_index_add_bulk(raw, ids.as_ptr(), values.as_ptr(), ids.len() as u64, 0, 0) };
            assert_eq!(rc, VL_OK, "vl_index_add_bulk: {}", last_error());
        }
        let mut side = Side::new();
        for v in data {
            side.entry(v.id).or_insert((v.text, v.metadata));
        }
        // coalescing is on by default, one queue per replica
        GpuFlatIndex(Handle { raw, dim, side })
    }
}

impl GpuFlatIndex {
    /// The ingest step of `Collection::add_text` for a batch (`src/client.rs:313-345`): the embedding model's raw
    /// f32 outputs are widened and L2-normalised on the device exactly as `EmbeddingGenerator::generate_embedding`
    /// does on the host (`src/embeddings.rs:169-181`), then added row by row.  `texts[i]` / `metadata[i]` stay here.
    pub fn add_embeddings(&mut self, ids: &[u64], embeddings_f32: &[f32], texts: Vec<String>, metadata: Vec<Option<serde_json::Value>>) -> Result<(), String> {
        assert_eq!(embeddings_f32.len(), ids.len() * self.0.dim);
        let before = self.0.len();
        let rc = unsafe { vl_index_add_embeddings_f32(self.0.raw, ids.as_ptr(), embeddings_f32.as_ptr(), ids.len() as u64, 1, 1, 0) };
        let taken = self.0.len() - before; // rows in front of a duplicate id are kept, like sequential add() calls
        for ((id, t), m) in ids.iter().zip(texts).zip(metadata).take(taken) {
            self.0.side.insert(*id, (t, m));
        }
        if rc == VL_OK { Ok(()) } else { Err(last_error()) }
    }

    /// nq independent searches sharing slab passes (no reference counterpart); row i is exactly `search(queries[i])`.
    pub fn search_batch(&self, queries: &[f64], nq: usize, k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<Vec<(u64, f64)>>> {
        let dim = self.0.dim;
        assert_eq!(queries.len(), nq * dim);
        let stride = k.min(self.len()).max(1); // rows of the output: the library writes min(k, len, stride) entries each
        let (mut ids, mut scores, mut n) = (vec![0u64; nq * stride], vec![0f64; nq * stride], vec![0u64; nq]);
        let rc = unsafe {
            vl_index_search_batch_cap(self.0.raw, queries.as_ptr(), nq as u64, dim as u64, k as u64, metric_code(metric), stride as u64, ids.as_mut_ptr(), scores.as_mut_ptr(), n.as_mut_ptr())
        };
        if rc != VL_OK {
            return Err(VectorLiteError::InternalError(last_error()));
        }
        Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * stride + i], scores[q * stride + i])).collect()).collect())
    }

    /// The embed -> search step of `Collection::search_text` (src/client.rs:393-401) for a batch: `embeddings` is `[nq, dim]`
    /// f32 as the model emits it; widening and L2 normalisation (src/embeddings.rs:169-181) run on the GPU, bit for bit.
    pub fn search_batch_embeddings(&self, embeddings: &[f32], nq: usize, k: usize, metric: SimilarityMetric, normalize: bool) -> VectorLiteResult<Vec<Vec<(u64, f64)>>> {
        let dim = self.0.dim;
        assert_eq!(embeddings.len(), nq * dim);
        let stride = k.min(self.len()).max(1);
        let (mut ids, mut scores, mut n) = (vec![0u64; nq * stride], vec![0f64; nq * stride], vec![0u64; nq]);
        let rc = unsafe {
            vl_index_search_batch_embeddings_f32(self.0.raw, embeddings.as_ptr(), nq as u64, dim as u64, normalize as c_int, 0, k.min(stride) as u64, metric_code(metric), ids.as_mut_ptr(), scores.as_mut_ptr(), n.as_mut_ptr())
        };
        if rc != VL_OK {
            return Err(VectorLiteError::InternalError(last_error()));
        }
        Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * stride + i], scores[q * stride + i])).collect()).collect())
    }

    /// `search_batch` for queries that already sit in this GPU's memory (embeddings computed there): `d_queries` is a
    /// device pointer to `[nq, dim]` f64; no host staging, no PCIe copy of the queries on the MFMA batch path.
    ///
    /// # Safety
    /// `d_queries` must be a valid device allocation of `nq * dim` f64 on the index's GPU, not written until the call returns.
    pub unsafe fn search_batch_device(&self, d_queries: *const f64, nq: usize, k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<Vec<(u64, f64)>>> {
        let stride = k.min(self.len()).max(1); // min(k, len) results at most; k is clamped to the buffers
        let (mut ids, mut scores, mut n) = (vec![0u64; nq * stride], vec![0f64; nq * stride], vec![0u64; nq]);
        let rc = vl_index_search_batch_dev(self.0.raw, d_queries, nq as u64, self.0.dim as u64, k.min(stride) as u64, metric_code(metric), std::ptr::null_mut(), ids.as_mut_ptr(), scores.as_mut_ptr(), n.as_mut_ptr());
        if rc != VL_OK {
            return Err(VectorLiteError::InternalError(last_error()));
        }
        Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * stride + i], scores[q * stride + i])).collect()).collect())
    }
}

impl VectorIndex for GpuFlatIndex {
    fn add(&mut self, vector: Vector) -> Result<(), String> {
        self.0.add(vector)
    }
    fn delete(&mut self, id: u64) -> Result<(), String> {
        unsafe { vl_index_delete(self.0.raw, id) }; // a missing id is Ok for the flat index
        self.0.side.remove(&id);
        Ok(())
    }
    fn search(&self, query: &[f64], k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<SearchResult>> {
        self.0.search(query, k, metric)
    }
    fn len(&self) -> usize {
        self.0.len()
    }
    fn is_empty(&self) -> bool {
        self.0.len() == 0
    }
    fn get_vector(&self, id: u64) -> Option<Vector> {
        self.0.get_vector(id)
    }
    fn dimension(&self) -> usize {
        self.0.dim
    }
}

/// Same on-disk payload as the CPU `FlatIndex` (`{dim, data: [Vector]}`): .vlc files stay interchangeable.
#[derive(Serialize, Deserialize)]
struct FlatPayload {
    dim: usize,
    data: Vec<Vector>,
}

impl Serialize for GpuFlatIndex {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let (ids, values) = self.0.export();
        let dim = self.0.dim;
        let data = ids
            .iter()
            .enumerate()
            .map(|(i, &id)| {
                let (text, metadata) = self.0.side.get(&id).cloned().unwrap_or_default();
                Vector { id, values: values[i * dim..(i + 1) * dim].to_vec(), text, metadata }
            })
            .collect();
        FlatPayload { dim, data }.serialize(s)
    }
}

impl<'de> Deserialize<'de> for GpuFlatIndex {
    fn deserialize<D: Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
        let p = FlatPayload::deserialize(d)?;
        Ok(GpuFlatIndex::new(p.dim, p.data))
    }
}

// ------------------------------------------------------------------------------------------------
// Row-sharded flat index: one process per GPU, this process holds the rows [offset, offset + len) of the corpus.
// Every rank calls the same methods with the same arguments (the library's all-gather is collective).
// ------------------------------------------------------------------------------------------------
pub struct ShardedGpuFlatIndex {
    shard: GpuFlatIndex,
    comm: *mut vl_comm,
    pub offset: u64,
    pub total: u64,
}
unsafe impl Send for ShardedGpuFlatIndex {}
unsafe impl Sync for ShardedGpuFlatIndex {}

impl ShardedGpuFlatIndex {
    /// Rank 0 creates the id and ships the 128 bytes to the other ranks over the server's own channel.
    pub fn unique_id() -> [u8; VL_COMM_ID_BYTES] {
        let mut id = [0u8; VL_COMM_ID_BYTES];
        let rc = unsafe { vl_comm_unique_id(id.as_mut_ptr()) };
        assert_eq!(rc, VL_OK, "vl_comm_unique_id: {}", last_error());
        id
    }

    /// Collective (ncclCommInitRank, then the length exchange).
    pub fn new(shard: GpuFlatIndex, id: &[u8; VL_COMM_ID_BYTES], world: usize, rank: usize, device: i32) -> Result<Self, String> {
        let mut comm = std::ptr::null_mut();
        if unsafe { vl_comm_create(id.as_ptr(), world as c_int, rank as c_int, device as c_int, &mut comm) } != VL_OK {
            return Err(last_error());
        }
        let mut s = ShardedGpuFlatIndex { shard, comm, offset: 0, total: 0 };
        s.sync()?;
        Ok(s)
    }

    /// Collective; call again after add/delete on any shard.
    pub fn sync(&mut self) -> Result<(), String> {
        match unsafe { vl_shard_sync(self.shard.0.raw, self.comm, &mut self.offset, &mut self.total) } {
            VL_OK => Ok(()),
            _ => Err(last_error()),
        }
    }

    pub fn shard_mut(&mut self) -> &mut GpuFlatIndex {
        &mut self.shard
    }

    /// `search_batch` for a batch that already sits in this rank's GPU memory (`d_queries`: device pointer to `[nq, dim]`
    /// f64 -- e.g. what an `ncclBroadcast` of the batch left there): no host staging, no PCIe copy of the queries.
    ///
    /// # Safety
    /// `d_queries` must be a valid device allocation of `nq * dim` f64 on the shard's GPU, not written until the call returns.
    pub unsafe fn search_batch_device(&self, d_queries: *const f64, nq: usize, k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<Vec<(u64, f64)>>> {
        let dim = self.shard.0.dim;
        let stride = k.min(self.total as usize).max(1);
        let (mut ids, mut scores, mut n) = (vec![0u64; nq * stride], vec![0f64; nq * stride], vec![0u64; nq]);
        let rc = vl_shard_search_batch_dev(self.shard.0.raw, self.comm, d_queries, nq as u64, dim as u64, k.min(stride) as u64, metric_code(metric), std::ptr::null_mut(), ids.as_mut_ptr(), scores.as_mut_ptr(), n.as_mut_ptr());
        match rc {
            VL_OK => Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * stride + i], scores[q * stride + i])).collect()).collect()),
            VL_ERR_DIM_MISMATCH => {
                let (mut e, mut a) = (0u64, 0u64);
                vl_last_dim_mismatch(&mut e, &mut a);
                Err(VectorLiteError::DimensionMismatch { expected: e as usize, actual: a as usize })
            }
            _ => Err(VectorLiteError::InternalError(last_error())),
        }
    }

    /// Collective: nq searches over the whole corpus; identical on every rank; row q is exactly what
    /// `FlatIndex::search(queries[q])` returns on one index holding every shard's rows in rank order.
    pub fn search_batch(&self, queries: &[f64], nq: usize, k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<Vec<(u64, f64)>>> {
        let dim = self.shard.0.dim;
        assert_eq!(queries.len(), nq * dim);
        let stride = k.min(self.total as usize).max(1); // at most min(k, total) results; k is clamped to the buffers
        let (mut ids, mut scores, mut n) = (vec![0u64; nq * stride], vec![0f64; nq * stride], vec![0u64; nq]);
        let rc = unsafe {
            vl_shard_search_batch(self.shard.0.raw, self.comm, queries.as_ptr(), nq as u64, dim as u64, k.min(stride) as u64, metric_code(metric), std::ptr::null_mut(), ids.as_mut_ptr(), scores.as_mut_ptr(), n.as_mut_ptr())
        };
        match rc {
            VL_OK => Ok((0..nq).map(|q| (0..n[q] as usize).map(|i| (ids[q * stride + i], scores[q * stride + i])).collect()).collect()),
            VL_ERR_DIM_MISMATCH => {
                let (mut e, mut a) = (0u64, 0u64);
                unsafe { vl_last_dim_mismatch(&mut e, &mut a) };
                Err(VectorLiteError::DimensionMismatch { expected: e as usize, actual: a as usize })
            }
            _ => Err(VectorLiteError::InternalError(last_error())),
        }
    }
}

impl Drop for ShardedGpuFlatIndex {
    fn drop(&mut self) {
        unsafe { vl_comm_destroy(self.comm) };
    }
}

// ------------------------------------------------------------------------------------------------
// HNSW (graph built and walked on the GPU; tombstone deletes, ef = min(k, len), metric fixed at creation)
// ------------------------------------------------------------------------------------------------
#[derive(Debug, Clone)]
pub struct GpuHnswIndex {
    h: Handle,
    metric: SimilarityMetric,
}

impl GpuHnswIndex {
    pub fn new(dim: usize, metric: SimilarityMetric) -> Self {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { vl_hnsw_create(dim as u64, metric_code(metric), 0, &mut raw) };
        assert_eq!(rc, VL_OK, "vl_hnsw_create: {}", last_error());
        GpuHnswIndex { h: Handle { raw, dim, side: Side::new() }, metric }
    }
    pub fn metric(&self) -> SimilarityMetric {
        self.metric
    }
    /// Bulk insert (one batched graph build on the device) for the custom Deserialize.
    pub fn add_all(&mut self, ids: &[u64], values: &[f64]) -> Result<(), String> {
        match unsafe { vl_index_add_bulk(self.h.raw, ids.as_ptr(), values.as_ptr(), ids.len() as u64, 1, 0) } {
            VL_OK => Ok(()),
            _ => Err(last_error()),
        }
    }
}

/// Same on-disk payload as the CPU `HNSWIndex` (`src/index/hnsw.rs:197-214`); its custom Deserialize only
/// reads `dim`, `metric`, `metadata` and `vector_values` (`:277-283`), so the two id maps are written as the
/// identity over the live rows.
#[derive(Serialize, Deserialize)]
struct HnswMeta {
    text: String,
    metadata: Option<serde_json::Value>,
}
#[derive(Serialize)]
struct HnswPayloadOut {
    dim: usize,
    metric: SimilarityMetric,
    id_to_index: HashMap<u64, usize>,
    index_to_id: HashMap<usize, u64>,
    metadata: HashMap<u64, HnswMeta>,
    vector_values: HashMap<u64, Vec<f64>>,
}
#[derive(Deserialize)]
struct HnswPayloadIn {
    dim: usize,
    metric: SimilarityMetric,
    metadata: HashMap<u64, HnswMeta>,
    vector_values: HashMap<u64, Vec<f64>>,
}

impl Serialize for GpuHnswIndex {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let (ids, values) = self.h.export();
        let dim = self.h.dim;
        let mut out = HnswPayloadOut {
            dim,
            metric: self.metric,
            id_to_index: HashMap::new(),
            index_to_id: HashMap::new(),
            metadata: HashMap::new(),
            vector_values: HashMap::new(),
        };
        for (i, &id) in ids.iter().enumerate() {
            let (text, metadata) = self.h.side.get(&id).cloned().unwrap_or_default();
            out.id_to_index.insert(id, i);
            out.index_to_id.insert(i, id);
            out.metadata.insert(id, HnswMeta { text, metadata });
            out.vector_values.insert(id, values[i * dim..(i + 1) * dim].to_vec());
        }
        out.serialize(s)
    }
}

impl<'de> Deserialize<'de> for GpuHnswIndex {
    fn deserialize<D: Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
        let p = HnswPayloadIn::deserialize(d)?;
        if p.dim == 0 {
            return Err(serde::de::Error::custom("Invalid dimension: cannot be 0"));
        }
        let mut idx = GpuHnswIndex::new(p.dim, p.metric);
        let mut ids = Vec::with_capacity(p.vector_values.len());
        let mut values = Vec::with_capacity(p.vector_values.len() * p.dim);
        for (id, v) in &p.vector_values {
            if v.len() != p.dim {
                return Err(serde::de::Error::custom(format!("Vector dimension mismatch: expected {}, got {}", p.dim, v.len())));
            }
            ids.push(*id);
            values.extend_from_slice(v);
        }
        idx.add_all(&ids, &values).map_err(serde::de::Error::custom)?;
        for (id, m) in p.metadata {
            idx.h.side.insert(id, (m.text, m.metadata));
        }
        Ok(idx)
    }
}

impl VectorIndex for GpuHnswIndex {
    fn add(&mut self, vector: Vector) -> Result<(), String> {
        self.h.add(vector)
    }
    fn delete(&mut self, id: u64) -> Result<(), String> {
        match unsafe { vl_index_delete(self.h.raw, id) } {
            VL_OK => {
                self.h.side.remove(&id);
                Ok(())
            }
            VL_ERR_NOT_FOUND => Err(format!("Vector ID {} does not exist", id)),
            _ => Err(last_error()),
        }
    }
    fn search(&self, query: &[f64], k: usize, metric: SimilarityMetric) -> VectorLiteResult<Vec<SearchResult>> {
        if metric != self.metric {
            return Err(VectorLiteError::MetricMismatch { requested: metric, index: self.metric });
        }
        self.h.search(query, k, metric)
    }
    fn len(&self) -> usize {
        self.h.len()
    }
    fn is_empty(&self) -> bool {
        self.h.len() == 0
    }
    fn get_vector(&self, id: u64) -> Option<Vector> {
        self.h.get_vector(id)
    }
    fn dimension(&self) -> usize {
        self.h.dim
    }
}
