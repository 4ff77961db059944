/*
 * vectorlite_amd.h -- C ABI of the MI355X-native distance-scan engine.
 *
 * Drop-in boundary for ONE path of mmailhos/vectorlite v0.1.5: the distance
 * scan + top-k behind `trait VectorIndex` (src/lib.rs:224-245), i.e. what
 * `FlatIndex` (src/index/flat.rs:60-135) and the distance callbacks / score
 * post-processing of `HNSWIndex` (src/index/hnsw.rs:113-174, :415-496) do.
 * A Rust `impl VectorIndex for GpuFlatIndex` binds these symbols through
 * `extern "C"`; text/metadata stay on the Rust side keyed
 * by id -- this library returns (id, score) pairs only.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every in/out buffer and the
 *    library keeps no pointer past the call;
 *  - every function returns a vl_status; no C++ exception crosses the boundary;
 *  - vectors, queries and scores are f64 like the reference
 *    (src/lib.rs:168,198,232); ids are arbitrary u64;
 *  - search calls are re-entrant and may run concurrently from many threads on
 *    one handle (the reference searches under RwLock::read, src/client.rs:398);
 *    add/delete take the handle exclusively (RwLock::write, src/client.rs:333,383);
 *  - there is NO CPU fallback: without a usable HIP device every compute entry
 *    point returns VL_ERR_DEVICE.
 */
#ifndef VECTORLITE_AMD_H
#define VECTORLITE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum SimilarityMetric, declaration order (src/lib.rs:363-378) */
typedef enum vl_metric {
    VL_COSINE = 0,     /* cosine_similarity     src/lib.rs:425-444 */
    VL_EUCLIDEAN = 1,  /* euclidean_similarity  src/lib.rs:476-489 */
    VL_MANHATTAN = 2,  /* manhattan_similarity  src/lib.rs:521-532 */
    VL_DOTPRODUCT = 3  /* dot_product           src/lib.rs:565-572 */
} vl_metric;

typedef enum vl_status {
    VL_OK = 0,
    VL_ERR_DIM_MISMATCH = 1,    /* DimensionMismatch{expected,actual} src/errors.rs:18; "Vector dimension mismatch" src/index/flat.rs:84 */
    VL_ERR_DUP_ID = 2,          /* "Vector ID {} already exists"      src/index/flat.rs:87 */
    VL_ERR_NOT_FOUND = 3,       /* "Vector ID {} does not exist"      src/index/hnsw.rs:401-403; get_vector -> None */
    VL_ERR_METRIC_MISMATCH = 4, /* MetricMismatch{requested,index}    src/errors.rs:42, src/index/hnsw.rs:425-430 */
    VL_ERR_NAN_SCORE = 5,       /* stands in for the partial_cmp().unwrap() panic, src/index/flat.rs:116 */
    VL_ERR_DEVICE = 6,          /* HIP runtime / kernel failure, or no device */
    VL_ERR_OOM = 7,             /* host or device allocation failed */
    VL_ERR_INVALID_ARG = 8,     /* null pointer, unknown metric, ... */
    /* .vlc loader (PersistenceError, src/persistence.rs:34-54) */
    VL_ERR_IO = 9,               /* "IO error: {0}" */
    VL_ERR_FILE_NOT_FOUND = 10,  /* "File not found: {0}" */
    VL_ERR_SERIALIZATION = 11,   /* "Serialization error: {0}" (malformed JSON, missing field, bad row) */
    VL_ERR_VERSION_MISMATCH = 12,/* "Version mismatch: expected 1.0.0, got {actual}" */
    VL_ERR_INVALID_FORMAT = 13   /* "Invalid file format: Expected format 'vectorlite-collection', got '{}'" */
} vl_status;

/* Which search pipeline produced the last result on this thread (diagnostics). */
typedef enum vl_path {
    VL_PATH_NONE = 0,
    VL_PATH_FAST = 1,       /* f32 slab scan -> top-64 candidates -> exact f64 rescoring, bound check passed */
    VL_PATH_EXACT_SELECT = 2, /* full f64 scan in reference arithmetic + exact top-k select (k <= 64) */
    VL_PATH_EXACT_SORT = 3    /* full f64 scan + device-wide sort (any k) */
} vl_path;

typedef struct vl_index vl_index; /* opaque; Send + Sync */

/* ---- lifecycle ---------------------------------------------------------- */

/* FlatIndex::new(dim, Vec::new())  (src/index/flat.rs:68-73) on HIP device `device`. */
int vl_flat_create(uint64_t dim, int device, vl_index **out);

/* FlatIndex::new(dim, data) with n rows: like the reference it validates
 * nothing (duplicate ids are kept).  `values` is [n, dim] row-major f64 on the host. */
int vl_flat_from_rows(uint64_t dim, const uint64_t *ids, const double *values, uint64_t n, int device,
                      vl_index **out);

/* ONE flat index over several GPUs of the node, in ONE process (the reference is one process: collections live in
 * Arc<RwLock<..>>, src/client.rs:243-247; searches run under read() on any tokio worker, src/client.rs:398,
 * src/server.rs:269,379-392).  The handle answers every vl_index_* entry point like a vl_flat_create handle
 * (VectorIndexWrapper::Flat); no per-GPU process, no id handshake.  device_ids[n_dev]: HIP ordinals, one part each
 * (an ordinal listed twice makes two parts on that card).
 *   VL_MULTI_REPLICAS    every GPU holds every row; add / delete go to all of them; a vl_index_search is answered by
 *                        the replica with the fewest searches in flight, on the calling thread (concurrent callers
 *                        spread over the GPUs; vl_index_set_coalescing gives each replica its own queue); a
 *                        vl_index_search_batch is cut into one run of queries per replica.
 *   VL_MULTI_ROW_SHARDS  every GPU holds part of the rows (add appends to the shortest shard; bulk adds are cut into
 *                        contiguous runs that level the shards).  Every search runs on all shards side by side, the
 *                        per-shard exact top-k are merged on the first GPU by (score desc, insertion order asc): the
 *                        answer is bit-identical to one index holding every row (src/index/flat.rs:116).
 * Device-side inputs (values_on_device, embeddings_on_device, d_queries) are expected on device_ids[0].
 * vl_index_search_positions returns global insertion numbers for a row-sharded handle. */
#define VL_MULTI_REPLICAS 0
#define VL_MULTI_ROW_SHARDS 1
int vl_flat_create_multi(uint64_t dim, const int *device_ids, int n_dev, int mode, vl_index **out);

/* Parts of a handle (1 for a single-GPU handle, mode -1): rows held by each part and searches each has answered
 * (queries, for batches) -- how evenly the replica dealer / the shard filler spread the work.  rows / searches hold
 * `capacity` entries and are filled when capacity >= *n_parts; any pointer may be NULL. */
int vl_index_parts(const vl_index *h, int *n_parts, int *mode, uint64_t *rows, uint64_t *searches, int capacity);

/* HNSWIndex::new(dim, metric) (src/index/hnsw.rs:216-259), default cargo profile M = 16, M0 = 32
 * (src/index/hnsw.rs:95-109).  The handle then behaves like VectorIndexWrapper::HNSW
 * (src/lib.rs:271-327) under every vl_index_* trait entry point below: add validates dimension and
 * duplicate ids (:363-371), delete of an absent id is VL_ERR_NOT_FOUND (:401-403) and tombstones the
 * node (:405-411), search checks the dimension even when empty (:416-421), rejects another metric with
 * VL_ERR_METRIC_MISMATCH (:425-430), walks the graph with ef = min(k, len) (:437,454), converts
 * u64 distances to scores (:51-75, :478-479), stable-sorts and truncates (:493-494).
 * The graph walk itself is this library's own (crate hnsw 0.11.0 is not part of the reference tree): it
 * navigates by f32 distances and gives every node of the final beam the reference's exact f64 callback value,
 * from which the returned scores are computed; results are approximate and judged by recall.
 * vl_hnsw_create builds with ef_construction = 400 -- what crate hnsw 0.11.0's Params::default() is recalled to be (the
 * reference passes no parameters, src/index/hnsw.rs:226-244; the crate is not in the reference tree, so unverified);
 * vl_hnsw_create_ex takes M, M0 (<= 64), ef_construction (1..512) and the level seed. */
int vl_hnsw_create(uint64_t dim, int metric, int device, vl_index **out);
int vl_hnsw_create_ex(uint64_t dim, int metric, uint32_t m, uint32_t m0, uint32_t ef_construction, uint64_t seed,
                      int device, vl_index **out);

/* #[derive(Clone)] (src/index/flat.rs:59, src/index/hnsw.rs:197; persistence clones the wrapper,
 * src/persistence.rs:118).  Flat: rows copied device to device.  HNSW: rows and the device graph are
 * copied as they are (tombstoned nodes included), so the clone answers exactly like the original. */
int vl_index_clone(const vl_index *h, vl_index **out);

void vl_index_destroy(vl_index *h);

/* Pre-size device storage for `n_rows` (optional; storage grows geometrically otherwise). */
int vl_index_reserve(vl_index *h, uint64_t n_rows);

/* ---- trait VectorIndex (src/lib.rs:224-245) ------------------------------- */

/* add(Vector): VL_ERR_DIM_MISMATCH if len != dim, VL_ERR_DUP_ID if id exists (src/index/flat.rs:82-91). */
int vl_index_add(vl_index *h, uint64_t id, const double *values, uint64_t len);

/* n x add() in order, one device pass.  validate != 0: stop at the first row whose id
 * already exists (rows before it are kept, like n sequential adds) and return VL_ERR_DUP_ID;
 * validate == 0: FlatIndex::new semantics (no checks).  values_on_device != 0: `values`
 * is a device pointer on the index's device ([n, dim] f64). */
int vl_index_add_bulk(vl_index *h, const uint64_t *ids, const double *values, uint64_t n, int validate,
                      int values_on_device);

/* NEW -- the ingest step in front of add (src/embeddings.rs:169-181): `embeddings` is [n, dim] f32 as the model
 * emits it (host pointer, or a device pointer on the index's device when embeddings_on_device != 0).  Each value
 * is widened to f64 (`x as f64`, :172) and, with normalize != 0, the row is L2-normalised on the device with the
 * host's arithmetic (:175-179: norm = sqrt of the in-order sum of squares; x / norm when norm > 0, else the row
 * unchanged), bit for bit.  The rows are then appended like vl_index_add_bulk (same `validate` meaning and
 * duplicate-id behaviour; HNSW handles always validate).  PCIe carries 4 bytes per value instead of 8. */
int vl_index_add_embeddings_f32(vl_index *h, const uint64_t *ids, const float *embeddings, uint64_t n, int normalize,
                                int validate, int embeddings_on_device);

/* delete(id): removes every row with that id, order-preserving; VL_OK even when
 * the id is absent (src/index/flat.rs:93-96). */
int vl_index_delete(vl_index *h, uint64_t id);

/* search(query, k, metric) (src/index/flat.rs:98-119): writes min(k, len) results, best first,
 * ties in insertion order.  out_ids/out_scores must hold min(k, len) entries; a caller that sized its
 * buffers from an earlier vl_index_len() passes k = min(k, capacity) -- results for a smaller k are a prefix
 * of those for a larger k, so the call can never write past the buffer even if the index grew since.
 * Empty index: any q_len is accepted and *out_n = 0.  Otherwise q_len != dim ->
 * VL_ERR_DIM_MISMATCH (vl_last_dim_mismatch gives expected/actual). */
int vl_index_search(const vl_index *h, const double *query, uint64_t q_len, uint64_t k, int metric,
                    uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* vl_index_search with an explicit output capacity: writes min(k, len, out_capacity) results -- the first
 * out_capacity entries of what vl_index_search would write (`truncate(k)`, src/index/flat.rs:117, applied once
 * more).  On an HNSW handle the walk still runs with the caller's k (its beam is ef = min(k, len),
 * src/index/hnsw.rs:437): only the copy-out is capped, so the entries ARE that prefix.  A caller that sized its buffers from an earlier vl_index_len() cannot be overrun by a concurrent add():
 * the bound is the caller's own number, not the index's length at search time.  The Rust / Python / C bindings of
 * this repository all call this form. */
int vl_index_search_cap(const vl_index *h, const double *query, uint64_t q_len, uint64_t k, int metric,
                        uint64_t out_capacity, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* NEW capability (the reference has no batch entry, src/lib.rs:224-245): nq independent
 * searches, each with exactly vl_index_search's result.  queries is [nq, q_len];
 * out_ids/out_scores are [nq, k] (row stride k), out_n is [nq]. */
int vl_index_search_batch(const vl_index *h, const double *queries, uint64_t nq, uint64_t q_len, uint64_t k,
                          int metric, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* The batch form with an explicit row capacity: out_ids / out_scores are [nq, out_stride]; row i receives
 * min(k, len, out_stride) entries, out_n[i] says how many (HNSW: the first out_stride entries of the walk with the
 * caller's k, as above). */
int vl_index_search_batch_cap(const vl_index *h, const double *queries, uint64_t nq, uint64_t q_len, uint64_t k,
                              int metric, uint64_t out_stride, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

uint64_t vl_index_len(const vl_index *h);      /* len()       src/index/flat.rs:121-123 */
int vl_index_is_empty(const vl_index *h);      /* is_empty()  src/index/flat.rs:125-127 */
uint64_t vl_index_dimension(const vl_index *h);/* dimension() src/index/flat.rs:133-135 */

/* VectorIndexWrapper::index_type() / ::metric() (src/lib.rs:329-346): 0 = Flat, 1 = HNSW;
 * vl_index_metric returns VL_ERR_NOT_FOUND for None (flat). */
int vl_index_type(const vl_index *h);
int vl_index_metric(const vl_index *h, int *out_metric);

/* HNSW only, own extension (the reference has no ef knob): nq walks with beam width
 * max(ef, min(k, len)); outputs as vl_index_search_batch.  ef = 0 is what vl_index_search does: the reference's
 * ef = min(k, len) (src/index/hnsw.rs:437,454), raised to the handle's beam floor only if the caller set one (below).
 * The walk kernel holds beams of up to 512 entries; ef > 512 is VL_ERR_INVALID_ARG (never silently narrowed), and
 * min(k, len) > 512 is answered by the exact scan of the row store. */
int vl_index_search_ef(const vl_index *h, const double *queries, uint64_t nq, uint64_t q_len, uint64_t k,
                       uint32_t ef, int metric, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* HNSW handle: OPT-IN beam floor of searches that name no ef.  DEFAULT 0 = the reference's rule: the walk keeps
 * ef = min(k, len) entries (src/index/hnsw.rs:437) -- 10 for k = 10.  With min_beam > 0 (at most 512) the walk keeps at
 * least that many entries and returns the best min(k, len) of them: the same number of results, recall@10 0.96
 * instead of 0.78 at N = 1 M on embedding-like data with min_beam = 32.  That is a DEVIATION from the reference's walk
 * width, which is why it is off unless asked for (env VL_HNSW_MIN_BEAM sets it for handles created afterwards). */
int vl_index_hnsw_set_min_beam(vl_index *h, uint32_t min_beam);

/* HNSW handle: the graph as it stands, every node (tombstoned ones included; node = insertion position).  level[n]
 * = top layer of each node; layer 0: cnt0[n] neighbours of node i at nbr0[i * m0 ..]; layer L >= 1 of node i: slot
 * upper_off[i] + L - 1 with cntU[slot] neighbours at nbrU[slot * m ..]; node_ids[n] the caller's ids, live[n] 0 for
 * tombstones, rows[n, dim] the f64 rows.  Any output pointer may be NULL.  The reference cannot export its graph
 * (crate hnsw 0.11.0 keeps it private; HNSWIndex::hnsw is #[serde(skip)], src/index/hnsw.rs:199-200); this is what a
 * CPU-side walk of the SAME graph needs (the repository's checker, oracle/vl_hnsw_cpu.c, is one). */
int vl_index_hnsw_graph_info(const vl_index *h, uint64_t *n_nodes, uint32_t *entry, int *max_level, uint32_t *m,
                             uint32_t *m0, uint64_t *upper_slots);
int vl_index_hnsw_graph_export(const vl_index *h, uint8_t *level, uint32_t *upper_off, uint32_t *cnt0, uint32_t *nbr0,
                               uint32_t *cntU, uint32_t *nbrU, uint64_t *node_ids, uint8_t *live, double *rows);

/* get_vector(id) (src/index/flat.rs:129-131): first row with that id -> out[dim];
 * VL_ERR_NOT_FOUND stands for None. */
int vl_index_get_vector(const vl_index *h, uint64_t id, double *out_values);

/* max_id() (src/index/flat.rs:76-78): VL_ERR_NOT_FOUND stands for None (empty index). */
int vl_index_max_id(const vl_index *h, uint64_t *out_id);

/* Rows in storage order, for Serialize (src/index/flat.rs:59): ids[len], values[len, dim].
 * HNSW handle: the live rows in insertion order (the `vector_values` member, src/index/hnsw.rs:197-214). */
int vl_index_export(const vl_index *h, uint64_t *out_ids, double *out_values);

/* ---- .vlc collection files (src/persistence.rs:149-176; the step before the path) --------- */

/* load_collection_from_file without materialising CollectionData: the file is mapped, one structural
 * pass locates every row's `values` / `text` / `metadata` tokens, header.version and header.format
 * are validated after the document has been accepted (the reference's order), and the numbers are
 * converted by host threads straight into staging blocks that are ingested on the device.
 * vl_vlc_open does host work only (usable without a GPU); vl_vlc_build_index creates the GPU index
 * ("Flat" payload: FlatIndex{dim, data} taken as is, duplicate ids kept, src/index/flat.rs:59;
 * "HNSW" payload: every vector re-inserted, src/index/hnsw.rs:272-360, in file order). */
typedef struct vl_vlc_doc vl_vlc_doc;
int vl_vlc_open(const char *path, vl_vlc_doc **out);
void vl_vlc_close(vl_vlc_doc *doc);
const char *vl_vlc_name(const vl_vlc_doc *doc); /* metadata.name, UTF-8; valid until vl_vlc_close */
/* index_type 0 = Flat / 1 = HNSW; metric -1 for Flat; dim = the payload's dim; rows = vectors found;
 * vector_count / dimension = the metadata block's fields (informational, like the reference). */
int vl_vlc_info(const vl_vlc_doc *doc, int *index_type, int *metric, uint64_t *dim, uint64_t *rows,
                uint64_t *vector_count, uint64_t *dimension);
/* Per row, file order: id and the byte ranges (offset, length) in the file of its `text` string token
 * (quotes included) and `metadata` value token; length 0 = absent.  Arrays of `rows` entries. */
int vl_vlc_side_table(const vl_vlc_doc *doc, uint64_t *ids, uint64_t *text_off, uint64_t *text_len,
                      uint64_t *meta_off, uint64_t *meta_len);
/* rows [first, first + n) converted to out[n, dim] f64 on the host (correctly rounded). */
int vl_vlc_read_values(const vl_vlc_doc *doc, uint64_t first, uint64_t n, double *out_values);
int vl_vlc_build_index(const vl_vlc_doc *doc, int device, vl_index **out);

/* ---- multi-GPU row shards (no reference counterpart) ---------------------- */

/* Like vl_index_search but returns storage POSITIONS (0-based, this shard) instead of ids,
 * so that a caller holding contiguous row shards can merge per-shard top-k by
 * (score desc, shard offset + position asc) -- the reference's stable order. */
int vl_index_search_positions(const vl_index *h, const double *query, uint64_t q_len, uint64_t k, int metric,
                              uint64_t *out_pos, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* Batched form of vl_index_search_positions: out_pos/out_ids/out_scores are [nq, k], out_n is [nq]. */
int vl_index_search_batch_positions(const vl_index *h, const double *queries, uint64_t nq, uint64_t q_len,
                                    uint64_t k, int metric, uint64_t *out_pos, uint64_t *out_ids,
                                    double *out_scores, uint64_t *out_n);

/* vl_index_search_batch_positions with the queries ALREADY IN DEVICE MEMORY of the index's GPU (e.g. embeddings computed
 * there, or a batch another rank broadcast with RCCL): d_queries is a device pointer to [nq, q_len] f64, the outputs stay
 * host buffers ([nq, k] / [nq]; out_pos and out_ids may each be NULL).  Batches the bf16 MFMA filter serves (cosine, dot,
 * Euclidean; >= 2 queries; an index of >= 8192 rows with dim <= 768) are staged by a kernel -- no host staging, no PCIe copy
 * of the queries; anything else (Manhattan, one query, a small index, an HNSW handle, queries the filter cannot certify) is
 * copied to the host and answered exactly as vl_index_search_batch would.  Same results, same errors as the host form.
 * The caller keeps d_queries valid and unmodified until the call returns (work already queued on other streams that
 * writes it must have completed). */
int vl_index_search_batch_dev(const vl_index *h, const double *d_queries, uint64_t nq, uint64_t q_len, uint64_t k,
                              int metric, uint64_t *out_pos, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* The caller's whole step (src/client.rs:393-401: embed -> index.search) for a batch: `embeddings` is [nq, dim] f32 as the
 * model emits it (host pointer, or a device pointer on the index's GPU when embeddings_on_device != 0).  Each query is
 * widened to f64 and, with normalize != 0, L2-normalised on the device with the arithmetic of src/embeddings.rs:169-181
 * (bit for bit, like vl_index_add_embeddings_f32 does for rows), then searched through vl_index_search_batch_dev: with
 * device embeddings the batch never visits the host, with host embeddings PCIe carries 4 bytes per value instead of 8.
 * Outputs as vl_index_search_batch; row i is exactly vl_index_search of the processed query i. */
int vl_index_search_batch_embeddings_f32(const vl_index *h, const float *embeddings, uint64_t nq, uint64_t dim,
                                         int normalize, int embeddings_on_device, uint64_t k, int metric,
                                         uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* ---- row-sharded batched search over RCCL (north_star's config 3; no reference counterpart) ----------
 *
 * One process per GPU.  Rank r holds the contiguous row range [offset_r, offset_r + len_r) of the corpus as an
 * ordinary flat handle (vl_flat_create + vl_index_add_bulk ...).  A batch is answered by every rank on its own
 * shard with the single-GPU pipeline (exact f64 scores), ONE ncclAllGather over xGMI exchanges the per-shard
 * top-k records, and a device kernel merges them by (score desc, GLOBAL position asc) -- the reference's stable
 * sort (src/index/flat.rs:116) applied to the whole corpus, so the answer is bit-identical to one index holding
 * every row.  All ranks must make the same calls with the same queries / k / metric; errors of any one shard
 * travel inside the exchange, so every rank returns the same status and nobody is left waiting.  That covers the
 * local search (status in word 0 of the record) and the growth of the exchange buffers (a pre-flight status
 * all-gather through buffers made at vl_comm_create, run by every rank exactly when the buffers have to grow); a
 * device failure that leaves a rank no way to join a collective ends in ncclCommAbort on that rank -- its peers'
 * collective fails instead of waiting -- and the communicator then refuses every later call (VL_ERR_DEVICE).
 *
 * vl_comm_unique_id: rank 0 makes the ncclUniqueId and hands the 128 bytes to the other ranks by whatever
 * channel the host has (the Rust server would use its own RPC; the Python harness broadcasts it with
 * torch.distributed).  vl_comm_create is collective (ncclCommInitRank). */
#define VL_COMM_ID_BYTES 128
typedef struct vl_comm vl_comm;
int vl_comm_unique_id(uint8_t *out_id);
int vl_comm_create(const uint8_t *id, int world, int rank, int device, vl_comm **out);
void vl_comm_destroy(vl_comm *comm);
int vl_comm_world(const vl_comm *comm);
int vl_comm_rank(const vl_comm *comm);

/* Where a batch's time goes on this rank, summed over the vl_shard_search_batch* calls since the last read: the
 * local search (host clock) and, from HIP events on the exchange stream, the H2D copy of this rank's record, the
 * ncclAllGather over xGMI, and the merge kernel + D2H of the answer.  Reading clears the totals. */
int vl_comm_profile_enable(vl_comm *comm, int enable);
int vl_comm_profile_read(vl_comm *comm, uint64_t *calls, double *local_ms, double *h2d_ms, double *allgather_ms,
                         double *merge_ms);

/* How this rank's exchange records were made, since the communicator was created: on_device = batches whose record the
 * finalize kernel wrote straight into the all-gather's send buffer (only the 32-byte header then crosses PCIe in front of
 * the collective; round 4), via_host = batches that built it in pinned host memory and copied it over (Manhattan, tiny
 * shards, k > 60, failures travelling in word 0).  Either pointer may be NULL. */
int vl_comm_record_paths(const vl_comm *comm, uint64_t *on_device, uint64_t *via_host);

/* Collective: the ranks exchange (len, dimension) of their shards; each learns the global position of its
 * first row (*out_offset, rank order) and the total row count.  Call after building the shards and again
 * after any add/delete on any of them; a shard that changed since is reported as VL_ERR_INVALID_ARG by the
 * next search on every rank. */
int vl_shard_sync(const vl_index *shard, vl_comm *comm, uint64_t *out_offset, uint64_t *out_total);

/* Collective: nq searches over the WHOLE sharded corpus; outputs as vl_index_search_batch ([nq, k], row
 * stride k; out_n[q] = min(k, total rows)), identical on every rank.  out_gpos (optional) receives global
 * storage positions.  Semantics per query are FlatIndex::search's (src/index/flat.rs:98-119) on the union:
 * empty corpus accepts any q_len; otherwise q_len != dim -> VL_ERR_DIM_MISMATCH; NaN -> VL_ERR_NAN_SCORE. */
int vl_shard_search_batch(const vl_index *shard, vl_comm *comm, const double *queries, uint64_t nq, uint64_t q_len,
                          uint64_t k, int metric, uint64_t *out_gpos, uint64_t *out_ids, double *out_scores,
                          uint64_t *out_n);

/* vl_shard_search_batch with the queries already in this rank's GPU memory (e.g. one rank embedded the batch and
 * ncclBroadcast it): the local search is vl_index_search_batch_dev.  Every rank of a call uses the same form. */
int vl_shard_search_batch_dev(const vl_index *shard, vl_comm *comm, const double *d_queries, uint64_t nq, uint64_t q_len,
                              uint64_t k, int metric, uint64_t *out_gpos, uint64_t *out_ids, double *out_scores,
                              uint64_t *out_n);

/* The two halves of vl_shard_search_batch for a host that moves the records with its own transport (MPI, gloo,
 * the server's RPC): vl_shard_search_local fills this shard's exchange record (vl_shard_packed_words(nq, ks)
 * u64 words; ks = min(k, longest shard); the local status travels in word 0 and the call itself returns VL_OK
 * unless an argument is null), the host gathers the `world` records in rank order, and vl_shard_merge runs the
 * same device merge kernel on them. */
uint64_t vl_shard_packed_words(uint64_t nq, uint64_t ks);
int vl_shard_search_local(const vl_index *shard, uint64_t row_offset, int corpus_has_rows, const double *queries,
                          uint64_t nq, uint64_t q_len, uint64_t ks, int metric, uint64_t *out_packed);
/* ... the same with device-resident queries (vl_index_search_batch_dev underneath) */
int vl_shard_search_local_dev(const vl_index *shard, uint64_t row_offset, int corpus_has_rows, const double *d_queries,
                              uint64_t nq, uint64_t q_len, uint64_t ks, int metric, uint64_t *out_packed);
int vl_shard_merge(int device, const uint64_t *gathered, uint32_t world, uint64_t nq, uint64_t ks, uint64_t k,
                   uint64_t *out_gpos, uint64_t *out_ids, double *out_scores, uint64_t *out_n);

/* ---- HNSW distance callbacks (src/index/hnsw.rs:113-174) ------------------ */

/* For each of the m stored rows at `positions`, Metric::distance(query, row) -> u64
 * (trunc(dist * 1000), Rust `as u64` saturation), computed on the device in the
 * reference's f64 operation order. */
int vl_index_hnsw_distances(const vl_index *h, const double *query, uint64_t q_len, int metric,
                            const uint64_t *positions, uint64_t m, uint64_t *out_dist);

/* convert_distance_to_similarity(d_u64 as f64 / 1000.0, metric) (src/index/hnsw.rs:51-75, :478-479). */
double vl_hnsw_score(uint64_t d_u64, int metric);

/* ---- diagnostics ---------------------------------------------------------- */

const char *vl_last_error(void);                                 /* thread-local message */
void vl_last_dim_mismatch(uint64_t *expected, uint64_t *actual); /* of the last VL_ERR_DIM_MISMATCH on this thread */
int vl_last_path(void);                                          /* vl_path of the last search on this thread */

/* Force every search onto an exact pipeline (testing): 0 = automatic, 2 / 3 = vl_path value. */
int vl_index_force_path(vl_index *h, int path);

/* Single-query candidate filter: 0 = stream the f32 slab (default, north_star's layout);
 * 1 = stream a bf16 copy of the slab first (half the HBM bytes) and fall back to the f32 scan when
 * the exactness bound cannot certify the answer.  Results are identical either way. */
int vl_index_set_single_filter(vl_index *h, int mode);

/* Coalescing of concurrent vl_index_search calls, flat or HNSW handle (the reference serves searches
 * concurrently under RwLock::read, src/client.rs:398; src/server.rs:269 -- one full scan each).
 * max_batch >= 2: callers that arrive while a scan is in flight are answered together by the next
 * slab pass / graph-walk launch (at most max_batch per pass; vl_index_search_batch's kernels); each caller still gets
 * exactly the result and status a lone vl_index_search returns.  window_us > 0 additionally lets a
 * lone caller wait that long for company.  max_batch 0/1 = off.
 * DEFAULT (round 4): ON with max_batch 256 and window 0 for every handle -- the reference's many-readers usage
 * (16 threads on one 10 M x 384 index: 456 QPS / 35 ms per query with separate scans, 9.8 k QPS / 1.6 ms coalesced,
 * identical answers); a lone caller finds no pass in flight, leads a pass of one and runs exactly the single-search
 * path (no wait, no batch kernels).  The first pass that answers two or more queries on an index of >= 8192 rows
 * builds the batch filter's bf16 copy of the rows (+2 bytes per value).  VL_COALESCE=0 in the
 * environment makes handles created afterwards start with it off; vl_index_set_coalescing(h, 0, 0) turns it off.
 * vl_index_coalesce_stats: passes run and queries answered by them since creation.
 * ADAPTIVE GATHER (window 0 only; on by default, VL_COALESCE_ADAPTIVE=0 or vl_index_coalesce_gather(h, 0, ..) turns it
 * off): N callers in a closed loop otherwise fall into passes of 1 and N - 1 in turn (the first caller back leads at
 * once, alone).  Every pass leaves an estimate of the callers in the loop (requests answered + compatible requests queued
 * behind it); a leader that finds fewer queued than the larger of the last two estimates waits for them, at most a quarter
 * of the recent pass time and never more than 400 us -- and only while requests arrive faster during a gather than going
 * at once would answer them (many more threads than cores: the peers sit in the run queue, the leader goes); a lone
 * caller (estimates 1, 1) never waits.  Native threads on one 10 M x 384 index: 16 -> 11.5 k QPS at 1.39 ms, 64 -> 39 k,
 * 256 -> 67 k (tools/concurrent_native.py).
 * vl_index_coalesce_gather: adaptive = 1 / 0 sets the switch, -1 leaves it; waits / waited_us (may be NULL) report the
 * passes whose leader waited and the microseconds spent waiting since creation. */
int vl_index_set_coalescing(vl_index *h, int max_batch, int window_us);
int vl_index_coalesce_stats(const vl_index *h, uint64_t *batches, uint64_t *queries);
int vl_index_coalesce_gather(vl_index *h, int adaptive, uint64_t *waits, uint64_t *waited_us);

/* HNSW handle: queries walked and Metric::distance evaluations (src/index/hnsw.rs:113-174) made for them since
 * creation: navigation evaluations on the f32 rows (dim x 4 bytes each) plus one exact f64 evaluation per entry of
 * the final beam (dim x 8 bytes each). */
int vl_index_hnsw_walk_stats(const vl_index *h, uint64_t *queries, uint64_t *distance_evals);

/* Kernel timing with HIP events on the stream the scan kernel runs on.
 * enable != 0 starts accumulating; vl_index_profile_read returns and clears the totals. */
int vl_index_profile_enable(vl_index *h, int enable);
int vl_index_profile_read(vl_index *h, uint64_t *n_scan_launches, double *scan_ms_total,
                          uint64_t *scan_bytes_total);

/* Which instantiation of the f32 scan kernel answered the last single search on this handle: variant = G * 10000 +
 * VPL * 100 + U of k_scan<metric, G, VPL, U> (negative: k_scan_generic's G), the number of workgroups it was launched
 * on, and whether the query travelled in the kernel arguments.  bench.py ties its PMC traffic figure to these. */
int vl_index_last_scan(const vl_index *h, int *variant, int *grid, int *query_in_kernarg);

/* The last launch sequence of the batch filter (bf16 MFMA, k_mfma_rows) on this handle, out6 = {K steps of 16 (row stride
 * / 16), metric, query chunks, workgroups per chunk of the last pass-1 stage, pass-1 stages, 32-row blocks sampled}; all
 * zero before the first such batch.  bench.py ties the PMC traffic figures of configs 3 and 5 to these. */
int vl_index_last_filter(const vl_index *h, int *out6);

/* Library/version probe; also reports how many HIP devices are visible (0 -> no GPU). */
int vl_runtime_info(int *n_devices, int *abi_version);

#ifdef __cplusplus
}
#endif
#endif /* VECTORLITE_AMD_H */
