/*
 * vl_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's distance-scan hot path
 * (mmailhos/vectorlite v0.1.5).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (vectorlite_amd/) never links, imports or calls it.
 *
 * Parity pinning: the reference is Rust and cannot be built in this image
 * (no cargo/rustc), so this restatement is pinned by the reference's own
 * known-answer tests (tests/golden/reference_kats.json),
 * see tests/test_oracle_kats.py.  The HNSW graph walk lives in the third-party
 * crate hnsw 0.11.0 (Cargo.lock:1111-1123), whose source is absent: for that
 * part parity is UNPINNED (only the distance callbacks and the score
 * conversion, which are in the reference tree, are restated here).
 *
 * Each function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef VL_ORACLE_H
#define VL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:363-378  enum SimilarityMetric (declaration order) */
enum { VLO_COSINE = 0, VLO_EUCLIDEAN = 1, VLO_MANHATTAN = 2, VLO_DOT = 3 };

/* status codes of the flat / hnsw search restatement */
enum {
    VLO_OK = 0,
    VLO_DIM_MISMATCH = 1,    /* VectorLiteError::DimensionMismatch  src/index/flat.rs:100-103 */
    VLO_DUP_ID = 2,          /* "Vector ID {} already exists"        src/index/flat.rs:87      */
    VLO_NOT_FOUND = 3,       /* "Vector ID {} does not exist"        src/index/hnsw.rs:401-403 */
    VLO_METRIC_MISMATCH = 4, /* VectorLiteError::MetricMismatch      src/index/hnsw.rs:425-430 */
    VLO_NAN_PANIC = 5        /* partial_cmp().unwrap() panic         src/index/flat.rs:116     */
};

/* ---- distance math: src/lib.rs:380-572 --------------------------------- */
double vlo_cosine(const double *a, const double *b, size_t n);     /* src/lib.rs:425-444 */
double vlo_euclidean(const double *a, const double *b, size_t n);  /* src/lib.rs:476-489 */
double vlo_manhattan(const double *a, const double *b, size_t n);  /* src/lib.rs:521-532 */
double vlo_dot(const double *a, const double *b, size_t n);        /* src/lib.rs:565-572 */
double vlo_calculate(int metric, const double *a, const double *b, size_t n); /* src/lib.rs:380-391 */

/* ---- ingest step: src/embeddings.rs:169-181 -------------------------------- */
/* f32 model output -> stored row: `x as f64` (:172); norm = sqrt(sum of x*x) (:175);
 * x / norm if norm > 0, else the widened values unchanged (:176-180).  normalize == 0: widen only. */
void vlo_embed_f32(const float *emb, size_t n, int normalize, double *out);

/* ---- HNSW boundary: src/index/hnsw.rs ----------------------------------- */
/* impl Metric<Vec<f64>> for {Euclidean,Cosine,Manhattan,DotProduct}: :113-174 */
uint64_t vlo_hnsw_distance(int metric, const double *a, const double *b, size_t n);
/* convert_distance_to_similarity: :51-75 */
double vlo_convert_distance_to_similarity(double distance, int metric);
/* the composition applied at :478-479 (d_u64 as f64 / 1000.0, then convert) */
double vlo_hnsw_score(uint64_t d, int metric);

/* ---- FlatIndex: src/index/flat.rs --------------------------------------- */
typedef struct vlo_flat vlo_flat;

/* FlatIndex::new (:68-73): no validation of dims or duplicate ids. */
vlo_flat *vlo_flat_new(size_t dim, const uint64_t *ids, const double *values, size_t n);
void vlo_flat_free(vlo_flat *f);
int vlo_flat_add(vlo_flat *f, uint64_t id, const double *values, size_t len); /* :82-91 */
int vlo_flat_delete(vlo_flat *f, uint64_t id);                                /* :93-96 */
size_t vlo_flat_len(const vlo_flat *f);                                       /* :121-123 */
size_t vlo_flat_dim(const vlo_flat *f);                                       /* :133-135 */
int vlo_flat_get(const vlo_flat *f, uint64_t id, double *out);                /* :129-131 */
/* FlatIndex::search (:98-119).  out_* must hold min(k, len) entries.
 * On VLO_DIM_MISMATCH, *out_n receives the expected dim. */
int vlo_flat_search(const vlo_flat *f, const double *q, size_t q_len, size_t k, int metric,
                    uint64_t *out_ids, double *out_scores, size_t *out_n);

/* Post-processing of HNSWIndex::search (:468-495) applied to a list of
 * (node id, d_u64) neighbours that a graph walk returned: score conversion,
 * stable descending sort, truncate(k).  ids/dists are rewritten in place. */
size_t vlo_hnsw_postprocess(uint64_t *ids, const uint64_t *dists, double *scores, size_t n,
                            size_t k, int metric);

#ifdef __cplusplus
}
#endif
/* vl_hnsw_cpu.c: the "CPU HNSW" -- a single-threaded walk of an exported graph with the reference's u64 distances
 * (graph walk of crate hnsw 0.11.0: parity unpinned; see the file's header). */
size_t vlo_hnsw_walk(int metric, const double *rows, size_t dim, size_t n_nodes, const uint8_t *level,
                     const uint32_t *upper_off, const uint32_t *cnt0, const uint32_t *nbr0, uint32_t m0,
                     const uint32_t *cntU, const uint32_t *nbrU, uint32_t m, uint32_t entry, int max_level,
                     const double *q, uint32_t ef, uint32_t k, uint32_t *stamp, uint32_t *epoch, uint32_t *out_nodes,
                     uint64_t *out_dist, uint64_t *evals);

#endif
