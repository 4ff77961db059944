// hnsw_index.hpp -- host side of the GPU HNSW index: mirror of the reference's
// `impl VectorIndex for HNSWIndex` (src/index/hnsw.rs:363-496) above the device graph walk
// (hnsw.hip).  Rows live in a GpuFlatIndex row store (node index = storage position; deletes are
// tombstones exactly like the reference, src/index/hnsw.rs:400-414).
#pragma once

#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "rwlock.hpp"
#include "flat_index.hpp"
#include "hnsw.hpp"

namespace vl {

// convert_distance_to_similarity(d as f64 / 1000.0, metric) (src/index/hnsw.rs:51-75, :478-479)
double hnsw_score(uint64_t d_u64, int metric);

struct HnswParams {
    uint32_t m = 16;                // MAXIMUM_NUMBER_CONNECTIONS   (src/index/hnsw.rs:95-101, default profile)
    uint32_t m0 = 32;               // MAXIMUM_NUMBER_CONNECTIONS_0 (src/index/hnsw.rs:103-109)
    uint32_t ef_construction = 400; // HNSWIndex::new calls Hnsw::new(metric) = the crate's Params::default() (src/index/hnsw.rs:226-244);
                                    // SURVEY 9.5 recalls that default as 400 (crate hnsw 0.11.0 is not in the tree: unverified).
                                    // Rounds 1-3 built with 128 because the beam stopped there; measured at N = 1 M x 384
                                    // (profiles/r04_hnsw_efc_sweep_1m_d384.jsonl): 128 -> 400 lifts the strict-beam recall@10 on
                                    // clustered rows 0.48 -> 0.57 (0.66 -> 0.76 at ef 128), nothing on latent-16 rows, build 3.6 -> 9.1 s
    uint64_t seed = 0;
};

// What ONE walk launch needs besides the graph: a stream, the visited sets of its walks (per walk slot a bitmap over the
// nodes + a log of the nodes it set, hnsw.hip: Visited) and the query / result staging buffers.  Launches borrow one from
// a pool inside the index, so searches from different threads run their walk kernels at the same time (the reference
// searches under RwLock::read, src/client.rs:398) -- nothing is shared between two launches but the read-only graph.
struct WalkScratch {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t* d_bits = nullptr;   // [n_slots, words], all zero between walks
    uint32_t* d_log = nullptr;    // [n_slots, log_cap]
    uint32_t n_slots = 0, words = 0, log_cap = 0;
    double* d_q = nullptr;
    unsigned long long* d_out = nullptr;  // [nq*k ids][nq*k score bits][nq counts], one D2H copy per batch
    unsigned long long* h_out = nullptr;  // pinned mirror
    uint64_t q_cap = 0, out_cap = 0;
    ~WalkScratch();
};

class HnswIndex {
public:
    struct CoalesceReq {  // one caller waiting in search() while coalescing is on
        const double* query;
        uint64_t k;
        uint32_t ef;
        uint64_t* out_ids;
        double* out_scores;
        uint64_t* out_n;
        uint64_t out_limit;  // entries the caller's buffers hold
        int rc = 6;  // ERR_DEVICE until the walk answers it
        std::string err = "coalesced walk ended without answering this request";
        bool done = false;
    };
    // concurrent single-query search() calls share walk launches (coalescer.hpp); 0 / 1 = off; on by default (create())
    void set_coalescing(int max_batch, int window_us) { co_.configure(max_batch, window_us, 4096); }
    void coalesce_stats(uint64_t* batches, uint64_t* queries) const { co_.stats(batches, queries); }
    void coalesce_gather(int adaptive, uint64_t* waits, uint64_t* waited_us) const
    {
        if (adaptive >= 0) co_.set_adaptive(adaptive != 0);
        co_.gather_stats(waits, waited_us);
    }

    static int create(uint64_t dim, int metric, const HnswParams& p, int device, HnswIndex** out);
    ~HnswIndex();

    int add(uint64_t id, const double* values, uint64_t len);
    int add_bulk(const uint64_t* ids, const double* values, uint64_t n, bool values_on_device);
    int remove(uint64_t id);
    // ef == 0: the reference's rule ef = min(k, len) (src/index/hnsw.rs:437,454); ef > 0: own extension
    // out_stride (0 = k): the caller's buffers hold out_stride entries per query -- rows are out_stride apart and at most
    // out_stride results are written per row, while the WALK still runs with the caller's k (ef = min(k, len)): the
    // first out_stride entries of what the uncapped call would return (vl_index_search_cap / _batch_cap)
    int search(const double* query, uint64_t q_len, uint64_t k, int metric, uint32_t ef, uint64_t* out_ids,
               double* out_scores, uint64_t* out_n, uint64_t out_stride = 0) const;
    int search_batch(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint32_t ef,
                     uint64_t* out_ids, double* out_scores, uint64_t* out_n, uint64_t out_stride = 0) const;
    uint64_t len() const;
    uint64_t dimension() const { return dim_; }
    int device() const { return device_; }
    int metric() const { return metric_; }
    int get_vector(uint64_t id, double* out) const;
    int max_id(uint64_t* out) const;
    uint64_t graph_nodes() const { return n_nodes_; }
    // #[derive(Clone)] on HNSWIndex (persistence clones the wrapper, src/persistence.rs:118): a deep copy of
    // the rows and of the device graph, tombstones included.
    // queries walked and Metric::distance evaluations made by them since creation (SURVEY 8(d) C4:
    // navigation evaluations read f32 rows, the final beam's exact evaluations f64 rows)
    void walk_stats(uint64_t* queries, uint64_t* distance_evals) const;
    // beam floor of searches that name no ef (0 = the reference's strict ef = min(k, len)); at most HNSW_MAX_EF
    void set_min_beam(uint32_t b) { min_beam_.store(b > (uint32_t)HNSW_MAX_EF ? (uint32_t)HNSW_MAX_EF : b); }
    uint32_t min_beam() const { return min_beam_.load(); }
    int clone(HnswIndex** out) const;
    // live rows in node (insertion) order: the `vector_values` member of the serialised form
    int export_rows(uint64_t* out_ids, double* out_values) const;
    // the graph as it stands (every node, tombstoned ones included): for inspection and CPU-side walks
    void graph_info(uint64_t* n_nodes, uint32_t* entry, int* max_level, uint32_t* m, uint32_t* m0, uint64_t* upper_slots) const;
    int graph_export(uint8_t* level, uint32_t* upper_off, uint32_t* cnt0, uint32_t* nbr0, uint32_t* cntU, uint32_t* nbrU,
                     uint64_t* node_ids, uint8_t* live, double* rows) const;

private:
    HnswIndex(uint64_t dim, int metric, const HnswParams& p, int device);
    int ensure_graph(uint64_t nodes, uint64_t upper_slots);
    HnswGraphView view() const;
    HnswGraphView view(const WalkScratch* ws) const;
    WalkScratch* acquire_scratch(uint64_t walks) const;  // nullptr + last_error on failure
    void release_scratch(WalkScratch* ws) const;
    void drop_scratch_pool();                             // the graph's capacity changed (caller holds the unique lock)
    int ensure_io(WalkScratch* ws, uint64_t nq, uint64_t k) const;
    int search_exact_fallback(const double* query, uint64_t k, uint64_t* out_ids, double* out_scores, uint64_t* out_n,
                              uint64_t out_limit) const;

    const uint64_t dim_;
    const int metric_;
    const HnswParams params_;
    const int device_;
    std::unique_ptr<GpuFlatIndex> store_;
    mutable RwLock mu_;
    hipStream_t stream_ = nullptr;  // mutators (graph growth, level upload, tombstones)
    mutable std::mutex pool_mu_;
    mutable std::condition_variable pool_cv_;
    mutable std::vector<std::unique_ptr<WalkScratch>> pool_all_;
    mutable std::vector<WalkScratch*> pool_free_;
    mutable size_t pool_pending_ = 0;    // scratches being allocated (slot and bytes already counted)
    mutable uint64_t pool_bytes_ = 0;    // visited-set bytes of pool_all_ + pending

    // device graph
    uint32_t* d_nbr0_ = nullptr;
    unsigned long long* d_dist0_ = nullptr;
    uint32_t* d_cnt0_ = nullptr;
    uint8_t* d_level_ = nullptr;
    uint32_t* d_upper_off_ = nullptr;
    uint32_t* d_nbrU_ = nullptr;
    unsigned long long* d_distU_ = nullptr;
    uint32_t* d_cntU_ = nullptr;
    uint32_t* d_lock_ = nullptr;
    uint32_t* d_indeg0_ = nullptr;
    uint64_t g_cap_ = 0, u_cap_ = 0;

    // host bookkeeping
    std::vector<uint8_t> level_;
    std::vector<uint32_t> upper_off_;
    uint64_t n_upper_ = 0;
    uint32_t entry_ = HNSW_NONE;
    int max_level_ = -1;
    uint64_t n_nodes_ = 0;
    std::unordered_map<uint64_t, uint32_t> id_to_node_;  // live ids only (id_to_index, src/index/hnsw.rs:205)
    std::vector<uint64_t> node_id_;
    std::vector<uint8_t> live_;
    uint64_t live_count_ = 0;

    std::atomic<uint32_t> min_beam_{0};  // 0 = the reference's strict ef = min(k, len); a wider floor is opt-in
    mutable std::atomic<uint64_t> stat_queries_{0}, stat_evals_{0};  // stat_evals_: the exact-fallback's share only
    mutable Coalescer<CoalesceReq> co_;

    unsigned long long* d_node_id_ = nullptr;      // [g_cap_] node -> caller's id (device copy of node_id_)
    uint8_t* d_live_ = nullptr;                    // [g_cap_] 0 = tombstoned
    unsigned long long* d_stat_evals_ = nullptr;   // distance evaluations of all query walks
};

}  // namespace vl
