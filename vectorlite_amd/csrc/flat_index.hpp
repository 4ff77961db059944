// flat_index.hpp -- C++ host side of the GPU flat index: the mirror of the reference's
// `impl VectorIndex for FlatIndex` (src/index/flat.rs:60-135) above the HIP kernels.
// The reference is compiled code (Rust, not buildable in this image), so the host layer is
// C++; include/vectorlite_amd.h exposes it as the C ABI a Rust `impl VectorIndex` would bind.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "rwlock.hpp"
#include "coalescer.hpp"
#include "kernels.hpp"
#include "mfma_scan.hpp"

namespace vl {

// vl_status values (include/vectorlite_amd.h)
enum Status : int {
    OK = 0,
    ERR_DIM_MISMATCH = 1,
    ERR_DUP_ID = 2,
    ERR_NOT_FOUND = 3,
    ERR_METRIC_MISMATCH = 4,
    ERR_NAN_SCORE = 5,
    ERR_DEVICE = 6,
    ERR_OOM = 7,
    ERR_INVALID_ARG = 8,
};

constexpr int K3_PIPE_QUERIES = SCAN_BATCH_MAX_QUERIES;  // queries staged, launched (one scan + one finalize) and synchronised together on the f32 batch path
constexpr int COALESCE_DEFAULT_BATCH = 256;  // concurrent callers one pass answers at most, by default (window 0)

enum Path : int { PATH_NONE = 0, PATH_FAST = 1, PATH_EXACT_SELECT = 2, PATH_EXACT_SORT = 3 };

// thread-local diagnostics (vl_last_error & friends)
void set_last_error(const std::string& msg);
const char* last_error();
void set_dim_mismatch(uint64_t expected, uint64_t actual);
void get_dim_mismatch(uint64_t* expected, uint64_t* actual);
void set_last_path(int p);
int last_path();

// Per-search scratch: one HIP stream plus device/pinned buffers.  Searches are re-entrant
// (the reference searches under RwLock::read, src/client.rs:398): each call borrows one.
struct Workspace {
    int device = 0;
    hipStream_t stream = nullptr;
    double* d_q64 = nullptr;
    double* h_q64 = nullptr;  // pinned
    size_t q_cap = 0;
    Cand32* d_partials = nullptr;
    Cand64* d_partials64 = nullptr;
    SearchResultBlock* d_result = nullptr;
    SearchResultBlock* h_result = nullptr;  // pinned
    uint32_t* d_nan = nullptr;
    uint32_t* h_nan = nullptr;  // pinned
    // exact path (lazy)
    double* d_scores = nullptr;
    size_t scores_cap = 0;
    uint64_t* d_okeys = nullptr;
    uint32_t* d_opos = nullptr;
    size_t sort_cap = 0;
    uint32_t* d_out_pos = nullptr;
    double* d_out_scores = nullptr;
    size_t out_cap = 0;
    // hnsw distances (lazy)
    uint32_t* d_positions = nullptr;
    uint64_t* d_dists = nullptr;
    size_t hn_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<float> q32;   // the f32 query handed to k_scan in its kernel arguments
    uint32_t seq = 0;         // stamp of the last single search issued from this workspace (h_result->seq)
    // f32 batch path (K3), several passes in flight: staging for K3_PIPE_QUERIES queries and their result blocks (lazy)
    double* k3_d_q64 = nullptr;                // [K3_PIPE_QUERIES, dim] queries, then their norms
    double* k3_h_q64 = nullptr;                // pinned
    SearchResultBlock* k3_h_result = nullptr;  // pinned [K3_PIPE_QUERIES]
    // large-batch MFMA path (lazy)
    MfmaScratch mf;
    double* mf_d_q64 = nullptr;             // [MFMA_MAX_BATCH, dim] queries, then their norms
    double* mf_h_q64 = nullptr;             // pinned
    Cand32* mf_lists = nullptr;             // [MFMA_MAX_BATCH, KP]
    double* mf_scores = nullptr;            // [MFMA_MAX_BATCH, KP] reference scores of the candidates (batch finalize scratch)
    SearchResultBlock* mf_h_result = nullptr;  // pinned [2][MFMA_MAX_BATCH]: two launch sequences in flight
    unsigned char* mf_h_dom = nullptr;      // pinned [2][MFMA_MAX_BATCH]: in-domain flags of device-resident queries
    hipEvent_t mf_ev_done[2] = {nullptr, nullptr};  // behind each sequence's finalize
    hipEvent_t mf_ev_h2d = nullptr;         // behind the copies out of the pinned query staging area
    bool mf_h2d_pending = false;

    ~Workspace();
};

// Search scratch (streams, partial-list buffers, the batch filter's buffers: ~10 MB, 80+ MB once a large batch ran) is
// shared by every handle of one (device, dimension): the reference keeps a HashMap of collections (src/client.rs:243-247),
// and thousands of small handles each holding their own scratch held 14.6 MB apiece (tools/many_handles_probe.py).
// Reference-counted: the pool of a (device, dimension) goes when its last handle does.
struct WorkspacePool {
    std::mutex mu;
    std::vector<Workspace*> free_;
    std::vector<std::unique_ptr<Workspace>> all;
    size_t users = 0;
};

class GpuFlatIndex {
public:
    struct CoalesceReq {  // one caller waiting in search_coalesced()
        const double* query;
        uint64_t k;
        int metric;
        uint64_t* out_pos;
        uint64_t* out_ids;
        double* out_scores;
        uint64_t* out_n;
        int rc = 6;  // ERR_DEVICE until the pass answers it (a pass that dies must not read as success)
        int path = 0;
        std::string err = "coalesced pass ended without answering this request";
        bool done = false;
    };

    // FlatIndex::new(dim, Vec::new())
    static int create(uint64_t dim, int device, GpuFlatIndex** out);
    ~GpuFlatIndex();

    // trait VectorIndex (src/lib.rs:224-245)
    int add(uint64_t id, const double* values, uint64_t len);
    // values_on_device: `values` is device memory; src_device >= 0 names the GPU it lives on when that is not this
    // index's own (a multi-GPU handle replicating rows: hipMemcpyPeerAsync)
    int add_bulk(const uint64_t* ids, const double* values, uint64_t n, bool validate, bool values_on_device,
                 int src_device = -1);
    int remove(uint64_t id);  // `delete`
    // the same, reporting which storage positions went (descending) -- a sharded handle keeps a per-row table beside this index
    int remove_report(uint64_t id, std::vector<uint64_t>* removed_positions);
    bool contains(uint64_t id) const;                       // O(1) after the first call (the duplicate-id table)
    int find_first(uint64_t id, uint64_t* out_pos) const;   // first row with that id (get_vector's rule); ERR_NOT_FOUND
    int get_row_at(uint64_t pos, double* out) const;
    int search(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
               uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    // NEW (no reference counterpart): nq independent searches sharing slab passes; outputs are
    // [nq, k] with row stride k.
    int search_batch(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                     uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    // NEW: search_batch with the queries already in device memory of this index's GPU (embeddings computed there): the
    // MFMA batch path stages them with a kernel -- no host staging, no PCIe copy of the queries; whatever that path does
    // not serve (Manhattan, one query, small indexes, queries it cannot certify) is copied to the host and answered there.
    int search_batch_device(const double* d_queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                            uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    // A row shard's answer written straight into the DEVICE planes of its exchange record (shard.hpp; `record` = the
    // record's first word in this GPU's memory, stride ks = k): the finalize kernel fills count / score / global position /
    // id of every query it certifies, the few it cannot are redone through the host paths and their slices patched.  Queries
    // on the host (queries_on_device = false) or in this GPU's memory.  *handled = false (and nothing written): this batch
    // does not take the MFMA filter (Manhattan, one query, a small or empty shard, out-of-domain rows, k > 60 ...) -- the
    // caller then builds the record on the host as before.  Word 0..3 of the record (status, len, dim) are the caller's.
    int search_batch_to_record(const double* queries, bool queries_on_device, uint64_t nq, uint64_t q_len, uint64_t ks,
                               int metric, uint64_t row_offset, unsigned long long* d_record, bool* handled) const;
    uint64_t len() const;
    bool is_empty() const { return len() == 0; }
    uint64_t dimension() const { return dim_; }
    int get_vector(uint64_t id, double* out) const;
    int max_id(uint64_t* out) const;

    int clone(GpuFlatIndex** out) const;
    int reserve(uint64_t n_rows);
    // forget the rows at positions >= n_rows (host bookkeeping only, cannot fail): how a multi-GPU handle takes back the
    // part of a bulk add that another part could not complete
    void truncate(uint64_t n_rows);
    int export_rows(uint64_t* out_ids, double* out_values) const;
    int hnsw_distances(const double* query, uint64_t q_len, int metric, const uint64_t* positions, uint64_t m,
                       uint64_t* out) const;

    // storage access for the HNSW graph layered on top of this row store (hnsw_index.cpp)
    const double* device_master() const { return d_master_; }
    const float* device_slab() const { return d_slab_; }
    const float* device_inv_norm() const { return d_inv_norm_; }
    uint32_t slab_ld() const { return ld_; }
    uint64_t capacity() const { return cap_; }

    void force_path(int p) { force_path_.store(p); }
    // 0: single queries scan the f32 slab (default); 1: try the bf16 slab first (half the bytes)
    void set_single_filter(int mode)
    {
        single_filter_.store(mode);
        bf16_tries_.store(0);
        bf16_fails_.store(0);
    }
    // Group concurrent single-query search() calls into shared slab passes (coalescer.hpp, search_coalesced()).
    // max_batch <= 1 turns it off.  window_us: how long a lone caller waits for company.  create() turns it on with
    // (COALESCE_DEFAULT_BATCH, 0) unless VL_COALESCE=0.
    void set_coalescing(int max_batch, int window_us) { co_.configure(max_batch, window_us, (int)MFMA_MAX_BATCH); }
    void coalesce_stats(uint64_t* batches, uint64_t* queries) const { co_.stats(batches, queries); }
    // adaptive: 1 / 0 switch the leader's adaptive gather (coalescer.hpp) on / off, -1 leaves it; waits, waited_us: passes whose
    // leader waited for its peers and the time spent waiting, since creation
    void coalesce_gather(int adaptive, uint64_t* waits, uint64_t* waited_us) const
    {
        if (adaptive >= 0) co_.set_adaptive(adaptive != 0);
        co_.gather_stats(waits, waited_us);
    }
    void profile_enable(bool on);
    void profile_read(uint64_t* n, double* ms, uint64_t* bytes);
    // which k_scan instantiation (G * 10000 + VPL * 100 + U; negative: the generic kernel's G), on how many workgroups,
    // the last single f32 scan used, and whether its query travelled in the kernel arguments
    void last_scan(int* variant, int* grid, int* qarg) const
    {
        if (variant) *variant = last_scan_variant_.load();
        if (grid) *grid = last_scan_grid_.load();
        if (qarg) *qarg = last_scan_qarg_.load();
    }
    // the batch filter's last launch sequence on this handle: {K steps of 16, metric, query chunks, workgroups per chunk of
    // the last pass-1 stage, pass-1 stages, 32-row blocks sampled}; all zero before the first MFMA batch
    void last_filter(int out[6]) const
    {
        for (int i = 0; i < 6; ++i) out[i] = last_filter_[i].load(std::memory_order_relaxed);
    }
    int device() const { return device_; }

private:
    GpuFlatIndex(uint64_t dim, int device);
    int ensure_capacity(uint64_t rows);  // caller holds the unique lock
    int ingest_range(uint64_t first, uint64_t n);
    int remove_position(uint64_t pos);
    void rebuild_id_counts() const;
    Workspace* acquire_ws() const;
    void release_ws(Workspace* ws) const;
    int prepare_ws(Workspace* ws) const;
    int search_direct(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos, uint64_t* out_ids,
                      double* out_scores, uint64_t* out_n) const;
    int search_coalesced(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                         uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    void run_coalesced(std::vector<CoalesceReq*>& batch) const;
    int search_locked(Workspace* ws, const double* query, uint64_t k_eff, int metric, uint64_t* out_pos,
                      uint64_t* out_ids, double* out_scores, uint64_t* out_n, bool skip_fast) const;
    int run_exact(Workspace* ws, int metric, uint64_t n, uint64_t k_eff, std::vector<uint32_t>* pos,
                  std::vector<double>* scores) const;
    int wait_result(Workspace* ws, uint32_t seq) const;
    int ensure_bf16_slab(bool frag_major) const;  // lazily builds the bf16 slab (row-major, or MFMA fragment order) a filter streams
    int ensure_mfma_scratch(Workspace* ws) const;
    int search_batch_locked(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                            uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;  // mu_ held (shared)
    int search_batch_mfma(Workspace* ws, const double* queries, const double* d_queries, uint64_t nq, uint64_t k,
                          uint64_t k_eff, int metric, uint64_t* out_pos, uint64_t* out_ids, double* out_scores,
                          uint64_t* out_n, std::vector<uint8_t>* done,
                          const ShardRecordSink* sink = nullptr) const;  // queries on the host, or d_queries on the device
    int ensure_device_ids() const;  // lazily uploads the position -> id table (what a device-written exchange record needs)

    const uint64_t dim_;
    const uint32_t ld_;  // slab row stride in floats: dim rounded up to 4 (16-byte vector loads)
    const int device_;

    mutable RwLock mu_;  // search: shared; add/delete: unique
    // device storage
    double* d_master_ = nullptr;  // [cap, dim] f64: exact rows
    float* d_slab_ = nullptr;     // [cap, ld]  f32: what the scan streams
    float* d_inv_norm_ = nullptr; // [cap]      f32: 1/|row|, 0 for zero rows
    uint8_t* d_flags_ = nullptr;  // [cap]
    mutable void* d_slab16_ = nullptr;     // [cap, ldb] bf16: candidate filter of the MFMA batch path (lazy)
    mutable float* d_norm16_ = nullptr;    // [cap] f32 |row| (the bf16 slab rows are unit-normalised)
    mutable float* d_sqnorm_ = nullptr;    // [cap] f32 |row|^2 for the GEMM-form Euclidean key (with d_slab16_)
    mutable uint64_t slab16_rows_ = 0;     // rows converted so far (== len() once built)
    mutable void* d_slab16f_ = nullptr;    // the same rows in MFMA fragment order: what k_mfma_rows streams (lazy; dims <= 384)
    mutable uint64_t slab16f_rows_ = 0;
    mutable std::mutex bf16_mu_;
    mutable unsigned long long* d_ids_ = nullptr;  // [d_ids_cap_] position -> id on the device (lazy: row-sharded batches only)
    mutable uint64_t d_ids_cap_ = 0, d_ids_rows_ = 0;  // rows uploaded so far (a delete rewinds it like the bf16 copies)
    IngestStats* d_stats_ = nullptr;
    uint64_t cap_ = 0;  // rows every array holds (the minimum of the four below)
    uint64_t cap_master_ = 0, cap_slab_ = 0, cap_inv_ = 0, cap_flags_ = 0;
    hipStream_t mut_stream_ = nullptr;
    void* d_bounce_ = nullptr;  // delete compaction buffer
    size_t bounce_bytes_ = 0;

    // host bookkeeping
    std::vector<uint64_t> ids_;        // position -> id (insertion order)
    std::vector<uint8_t> row_flags_;   // position -> ROW_* flags
    uint64_t n_out_of_domain_ = 0;
    double max_row_norm_ = 0.0;        // upper bound over in-domain rows ever stored
    mutable std::unordered_map<uint64_t, uint32_t> id_counts_;  // id -> multiplicity (lazy)
    mutable bool id_counts_valid_ = true;

    // workspace pool
    static WorkspacePool* attach_pool(int device, uint64_t dim);
    static void detach_pool(int device, uint64_t dim);
    WorkspacePool* ws_pool_ = nullptr;

    mutable Coalescer<CoalesceReq> co_;

    std::atomic<int> force_path_{0};
    std::atomic<int> single_filter_{0};
    mutable std::atomic<uint64_t> bf16_tries_{0}, bf16_fails_{0};
    std::atomic<bool> profile_{false};
    // single searches in flight on this handle: up to SPIN_MAX_SEARCHERS of them poll their result stamp,
    // more than that sleep in hipStreamSynchronize (wait_result)
    static constexpr int SPIN_MAX_SEARCHERS = 2;
    static constexpr int SPIN_MAX_MS = 200;
    mutable std::atomic<int> active_searches_{0};
    mutable std::atomic<int> last_scan_variant_{0}, last_scan_grid_{0}, last_scan_qarg_{0};
    mutable std::atomic<int> last_filter_[6] = {};
    mutable std::mutex prof_mu_;
    mutable uint64_t prof_n_ = 0;
    mutable double prof_ms_ = 0.0;
    mutable uint64_t prof_bytes_ = 0;
};

// NEW: f32 embeddings [n, dim] (host or device) -> widened, optionally L2-normalised f64 rows on the
// device with the arithmetic of src/embeddings.rs:171-179, appended through `append(ids, device rows, count)`.
int add_embeddings_f32(int device, uint64_t dim, const uint64_t* ids, const float* emb, uint64_t n, bool normalize,
                       bool emb_on_device, const std::function<int(const uint64_t*, const double*, uint64_t)>& append);

// NEW: the query side of the same step (src/client.rs:393-401: embed -> index.search): nq f32 embeddings [nq, dim] (host or
// device) are widened and optionally L2-normalised on the device exactly like add_embeddings_f32 does for rows, then handed
// to `search(device f64 queries, count)` -- for the flat index that is search_batch_device: the batch never visits the host.
int search_embeddings_f32(int device, uint64_t dim, const float* emb, uint64_t nq, bool normalize, bool emb_on_device,
                          const std::function<int(const double*, uint64_t)>& search);

}  // namespace vl
