// kernels.hpp -- launch interface of the gfx950 kernels (kernels.hip).
//
// Every kernel here serves the one hot path this repo implements: the distance
// scan + top-k of the reference's FlatIndex::search (src/index/flat.rs:98-119,
// metric math src/lib.rs:425-572) and the HNSW distance callbacks
// (src/index/hnsw.rs:113-174).  Paths are relative to /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vl {

enum Metric : int { COSINE = 0, EUCLIDEAN = 1, MANHATTAN = 2, DOT = 3 };

constexpr int KP = 64;            // candidate-list length (one entry per lane of a wave)
constexpr int KFAST_MAX = 60;     // largest k tried on the fast paths: the 64-entry candidate list must keep a margin
                                  // behind the k-th entry for the bound check to pass (checked per query; on 10 M random
                                  // unit rows one rank at the top is worth ~2e-4, the cosine bound ~5e-5)
constexpr uint32_t POS_SENTINEL = 0xFFFFFFFFu;

struct Cand32 {  // f32 candidate: scan key (larger = better) + storage position
    float key;
    uint32_t pos;
};
struct Cand64 {  // exact candidate: reference f64 score + storage position
    double key;
    uint32_t pos;
    uint32_t pad;
};

// Device -> host result block of one search.
struct SearchResultBlock {
    uint32_t n_out;
    uint32_t flags;  // RESULT_* bits
    uint32_t pos[KP];
    double score[KP];
    // completion stamp of a single search: written LAST (system-scope release) by the finalize kernel when the
    // caller passed a non-zero `seq`; the host waits on it instead of on the stream (flat_index.cpp, wait_result)
    uint32_t seq;
    uint32_t pad;
};
constexpr uint32_t RESULT_NEEDS_EXACT = 1u;  // bound check failed / tie at the cut: redo on the exact path
constexpr uint32_t RESULT_HAS_NAN = 2u;      // some score is NaN

// Device planes of a row shard's exchange record (shard.hpp) that the finalize kernel fills directly, so that the record of a
// batch never visits the host on its way into the all-gather: for query q (q0 + block index) the certified answer's
// (score bits, shard offset + position, id) at [q * ks + rank], and cnt[q] = entries offered -- 0 when the bound check did
// not certify the answer (the host redoes that query and patches its slice of the record).  cnt == nullptr: no record.
struct ShardRecordSink {
    unsigned long long* cnt = nullptr;         // [nq]
    unsigned long long* score_bits = nullptr;  // [nq, ks]
    unsigned long long* gpos = nullptr;        // [nq, ks]
    unsigned long long* ids = nullptr;         // [nq, ks]
    const unsigned long long* pos_to_id = nullptr;  // [n_rows] this shard's position -> id table on the device
    unsigned long long row_offset = 0;         // global position of the shard's first row
    uint32_t ks = 0;                           // record row stride
    uint32_t q0 = 0;                           // first query of this launch within the record
};

// Per-index device statistics maintained by the ingest kernel.
struct IngestStats {
    unsigned long long max_norm_bits;  // bits of the largest row L2 norm (f64, >= 0)
    unsigned int n_out_of_domain;      // rows outside the f32 fast-path domain
    unsigned int pad;
};

// Row flags produced by ingest.
constexpr uint8_t ROW_OUT_OF_DOMAIN = 1;

struct ScanPlan {
    int grid;      // workgroups launched (= partial lists written)
    int variant;   // which instantiation ran (diagnostics)
};

// f32 embeddings [n, dim] -> f64 rows [n, dim], widened and (normalize) L2-normalised with the host arithmetic of
// src/embeddings.rs:171-179 (sequential sum of squares, sqrt, one division per value; a zero row stays as it is).
hipError_t launch_embed_f32(hipStream_t s, const float* emb, uint64_t n, uint32_t dim, bool normalize, double* out);

// f64 master rows [n, dim] -> f32 slab rows [n, ld] (zero padded), inv_norm[n], flags[n], stats.
hipError_t launch_ingest(hipStream_t s, const double* master, float* slab, float* inv_norm, uint8_t* flags,
                         IngestStats* stats, uint64_t n, uint32_t dim, uint32_t ld);

// Upper bound on the workgroups launch_scan uses (partials must hold SCAN_MAX_GRID*KP entries).
constexpr int SCAN_MAX_GRID = 4096;
constexpr int SELECT_MAX_GRID = 1024;
constexpr int SCAN_BATCH_QB = 8;           // queries served by one slab pass of k_scan_batch
constexpr int SCAN_BATCH_MAX_GRID = 2048;
constexpr int SCAN_BATCH_MAX_QUERIES = 512;  // queries one launch_scan_batch call answers: groups of QB as blockIdx.y
// partial-list buffers: the scan lists (single: SCAN_MAX_GRID; batch: QB x SCAN_BATCH_MAX_GRID),
// followed by two ping-pong merge regions of QB x 64 lists
constexpr size_t PARTIALS32_LISTS = (size_t)SCAN_BATCH_QB * SCAN_BATCH_MAX_GRID;
static_assert(PARTIALS32_LISTS >= (size_t)SCAN_MAX_GRID, "single-query lists must fit");
constexpr size_t PARTIALS32_ENTRIES = (PARTIALS32_LISTS + 2 * (size_t)SCAN_BATCH_QB * 64) * KP;
constexpr size_t PARTIALS64_ENTRIES = (size_t)(SELECT_MAX_GRID + 128) * KP;

// K1: f32 slab scan -> per-workgroup top-KP partial lists.
// Two ways to hand over the query.  q32_host != nullptr (and scan_takes_qarg(ld)): the f32 query, zero padded to
// `ld` floats, rounded from the f64 query to nearest even, is copied into the kernel arguments -- nothing has to
// be on the device before the launch.  Otherwise q64 is the f64 query in device memory and each lane rounds its
// slice itself.  Both forms produce the same keys.
constexpr int SCAN_QARG_FLOATS = 768;
bool scan_takes_qarg(uint32_t ld);
hipError_t launch_scan(hipStream_t s, int metric, const float* slab, const float* inv_norm, const double* q64,
                       uint64_t n, uint32_t dim, uint32_t ld, Cand32* partials, ScanPlan* plan,
                       const float* q32_host = nullptr);

// `out` may be pinned host memory (the result block is written once, by one wave).
// K2: merge partial lists -> top-KP, rescore them in reference f64 arithmetic from the master
// rows, rank by (score desc, pos asc), run the exactness bound check, write the result block.
// One workgroup per query: `partials` holds nq x n_lists lists (query-major), q64 is [nq, dim],
// q_norms[nq] the f64 query norms, out[nq] the result blocks.
// 60 < k <= KMULTI_MAX: the scan's workgroup lists cut into n_parts partitions of 64 candidates each, rescored and
// ranked together; out = n_parts result blocks (ranks 64 i .. 64 i + 63 in block i; n_out and flags in block 0).
constexpr int KMULTI_MAX = 220;
hipError_t launch_merge_finalize_multi(hipStream_t s, int metric, Cand32* partials, int n_lists_total, int n_parts,
                                       const double* master, const double* q64, const double* q_norm, uint32_t dim,
                                       uint64_t n_rows, uint32_t k, double max_row_norm, SearchResultBlock* out);
// q64 / q_norms may be device memory or device-visible pinned host memory (a single search reads its 3 KB query
// straight from the pinned staging block: one workgroup, one PCIe round trip hidden behind the list merge).
// seq != 0 (nq == 1 only): out->seq = seq is stored last, system-scope release, after the result block.
hipError_t launch_merge_finalize(hipStream_t s, int metric, Cand32* partials, int n_lists, int nq,
                                 const double* master, const double* q64, const double* q_norms, uint32_t dim,
                                 uint64_t n_rows, uint32_t k, double max_row_norm, SearchResultBlock* out,
                                 double in_extra = 0.0, uint32_t seq = 0, const ShardRecordSink* sink = nullptr);

// K2 for a batch whose candidates already are ONE sorted top-64 list per query (lists[nq][KP], the MFMA filter's output):
// the rescoring of each query's 64 rows split over four 256-thread workgroups + one wave per query that ranks, checks the
// bound and emits -- the same arithmetic and result blocks as launch_merge_finalize(n_lists = 1), at batch throughput.
// score_scratch: nq x KP doubles of device memory.
hipError_t launch_batch_finalize(hipStream_t s, int metric, const Cand32* lists, int nq, const double* master, const double* q64,
                                 const double* q_norms, uint32_t dim, uint64_t n_rows, uint32_t k, double max_row_norm,
                                 SearchResultBlock* out, double in_extra, double* score_scratch,
                                 const ShardRecordSink* sink = nullptr);

// K3: one launch for nq <= SCAN_BATCH_MAX_QUERIES queries (q64 is [nq, dim]) in groups of SCAN_BATCH_QB -- every group is one
// pass over the slab (blockIdx.y = group); lists are written query-major, plan->grid of them per query (<= 64 when there is
// more than one group, so that the finalize kernel takes them without a merge level).
bool scan_batch_supported(uint32_t ld);
hipError_t launch_scan_batch(hipStream_t s, int metric, const float* slab, const float* inv_norm, const double* q64,
                             uint32_t nq, uint64_t n, uint32_t dim, uint32_t ld, Cand32* partials, ScanPlan* plan);

// Exact path: reference-order f64 score of every row.
hipError_t launch_exact_scan(hipStream_t s, int metric, const double* master, const double* q64, uint64_t n,
                             uint32_t dim, double* scores, uint32_t* nan_flag);
// Exact path, k <= KP: top-k of scores[] by (score desc, pos asc).
int select_grid_for(uint64_t n);
// Ranks 64 r .. 64 r + k - 1 of the exact order: `after` = the (full, k = 64) block of round r - 1, nullptr for r = 0.
hipError_t launch_exact_select(hipStream_t s, const double* scores, uint64_t n, uint32_t k, Cand64* partials,
                               const uint32_t* nan_flag, SearchResultBlock* out,
                               const SearchResultBlock* after = nullptr);
constexpr int SELECT_MAX_ROUNDS = 16;  // k <= 1024 is answered by rounds of 64; larger k by the full sort
// Exact path, any k: device-wide sort of (score desc, pos asc); writes the first k.
// okeys/opos are scratch of next_pow2(n) entries.
uint64_t sort_capacity_for(uint64_t n);
hipError_t launch_exact_sort(hipStream_t s, const double* scores, uint64_t n, uint64_t k, uint64_t* okeys,
                             uint32_t* opos, uint32_t* out_pos, double* out_scores);

// HNSW distance callbacks: u64 distance of query vs the rows at positions[0..m).
hipError_t launch_hnsw_distances(hipStream_t s, int metric, const double* master, const double* q64,
                                 uint32_t dim, const uint32_t* positions, uint32_t m, uint64_t* out);

}  // namespace vl
