// multi_index.hpp -- ONE flat-index handle over several GPUs in ONE process.
//
// The reference is a single process: collections live in Arc<RwLock<..>> (src/client.rs:243-247), every search
// runs under read() on whatever tokio worker took the request (src/client.rs:398, src/server.rs:269,379-392).
// A drop-in GpuFlatIndex inside that process can use one GPU; this class gives the same `impl VectorIndex`
// surface (flat_index.hpp) to all of them, with no process-per-GPU launcher and no id handshake:
//
//   REPLICAS    every GPU holds the whole corpus (15-46 GB of 288).  add / delete go to every replica; a
//               search() is answered by the replica with the fewest searches in flight, on the caller's own
//               thread (the reference's concurrency model: many readers); a search_batch() is cut into one
//               contiguous run of queries per replica, answered side by side.  No data-path exchange.
//   ROW_SHARDS  every GPU holds a subset of the rows (north_star's batched configuration).  Every shard answers
//               the whole batch on its own rows with the single-GPU pipeline (exact f64 scores); the per-shard
//               top-k records (shard.hpp's layout) land in ONE pinned host block, because the position -> id
//               table of a shard lives on the host, go to the merge GPU in one copy, and shard.hip's
//               k_shard_merge ranks them by (score desc, GLOBAL insertion order asc) -- the reference's stable
//               sort (src/index/flat.rs:116) on the whole corpus.  Rows carry a global insertion number, so
//               shards need not be contiguous ranges: add() appends to the shortest shard.
//
// Same device listed twice = two replicas / two shards on one card (how the one-GPU test box exercises this).
// The multi-PROCESS form of ROW_SHARDS (one rank per GPU, ncclAllGather over xGMI) stays in shard_comm.cpp.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

#include "rwlock.hpp"
#include "flat_index.hpp"
#include "shard.hpp"

namespace vl {

class MultiFlatIndex {
public:
    enum Mode : int { REPLICAS = 0, ROW_SHARDS = 1 };

    static int create(uint64_t dim, const int* devices, int n_dev, int mode, MultiFlatIndex** out);
    ~MultiFlatIndex();

    // trait VectorIndex (src/lib.rs:224-245), same argument meaning as GpuFlatIndex
    int add(uint64_t id, const double* values, uint64_t len);
    int add_bulk(const uint64_t* ids, const double* values, uint64_t n, bool validate, bool values_on_device,
                 int src_device = -1);
    int remove(uint64_t id);
    // out_pos: global insertion numbers (ROW_SHARDS) / storage positions (REPLICAS: every replica stores alike)
    int search(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos, uint64_t* out_ids,
               double* out_scores, uint64_t* out_n) const;
    int search_batch(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                     uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    // queries in the memory of part 0's GPU: staged through the host (every part needs them)
    int search_batch_device(const double* d_queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                            uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    uint64_t len() const;
    bool is_empty() const { return len() == 0; }
    uint64_t dimension() const { return dim_; }
    int get_vector(uint64_t id, double* out) const;
    int max_id(uint64_t* out) const;
    int clone(MultiFlatIndex** out) const;
    int reserve(uint64_t n_rows);
    int export_rows(uint64_t* out_ids, double* out_values) const;

    void force_path(int p);
    void set_single_filter(int mode);
    void set_coalescing(int max_batch, int window_us);
    void coalesce_stats(uint64_t* batches, uint64_t* queries) const;
    void coalesce_gather(int adaptive, uint64_t* waits, uint64_t* waited_us) const;
    void profile_enable(bool on);
    void profile_read(uint64_t* n, double* ms, uint64_t* bytes);
    void last_scan(int* variant, int* grid, int* qarg) const { parts_[0]->last_scan(variant, grid, qarg); }
    void last_filter(int out[6]) const { parts_[0]->last_filter(out); }
    int device() const { return parts_[0]->device(); }  // where device-side inputs are expected

    int mode() const { return mode_; }
    int n_parts() const { return (int)parts_.size(); }
    // rows held by each part and searches each has answered (load balance of the replica dealer / the shard filler)
    void part_stats(uint64_t* rows, uint64_t* searches) const;

private:
    MultiFlatIndex(uint64_t dim, int mode) : dim_(dim), mode_(mode) {}
    // one worker thread per part: part-wise work of one call runs side by side (different GPUs)
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<void()> task;
        bool has_task = false, done = true, stop = false;
    };
    void start_workers();
    void run_parts(const std::function<void(int)>& fn) const;  // fn(i) for every part, concurrently; returns when all are done
    int shard_search(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                     uint64_t* out_ids, double* out_scores, uint64_t* out_n) const;
    int pick_replica() const;

    const uint64_t dim_;
    const int mode_;
    std::vector<std::unique_ptr<GpuFlatIndex>> parts_;
    mutable std::vector<std::unique_ptr<Worker>> workers_;
    mutable std::mutex run_mu_;  // one fan-out at a time uses the workers; a caller that finds them busy runs its parts itself

    mutable RwLock mu_;  // search: shared; add / delete: unique (RwLock, src/client.rs:333,383,398)
    // ROW_SHARDS: global insertion number of every row of every shard (ascending within a shard)
    std::vector<std::vector<uint64_t>> seq_;
    uint64_t next_seq_ = 0;

    mutable std::vector<std::unique_ptr<std::atomic<int>>> inflight_;       // REPLICAS: searches running on each replica
    mutable std::vector<std::unique_ptr<std::atomic<uint64_t>>> answered_;  // searches each part has answered
    mutable std::atomic<uint32_t> rr_{0};
    // a delete that failed on some parts only (the device died in the middle of a compaction) cannot be taken back: the
    // parts no longer agree, so the handle refuses every later call instead of serving divergent answers
    std::atomic<bool> broken_{false};
    int refuse_if_broken() const;

    // ROW_SHARDS exchange: a pinned host block of all parts' records + a merger on part 0's GPU per search IN FLIGHT.
    // The reference serves many readers at once (RwLock::read, src/client.rs:398): concurrent searches of a sharded handle
    // borrow separate slots (up to EXCHANGE_SLOTS; then they wait for one) instead of queueing behind one set of buffers.
    struct ExchangeSlot {
        std::unique_ptr<ShardMerger> merger;
        unsigned long long* h_records = nullptr;  // pinned [parts][words]
        uint64_t h_records_cap = 0;
        bool busy = false;
    };
    static constexpr size_t EXCHANGE_SLOTS = 4;
    ExchangeSlot* acquire_slot() const;
    void release_slot(ExchangeSlot* s) const;
    mutable std::mutex slots_mu_;
    mutable std::condition_variable slots_cv_;
    mutable std::vector<std::unique_ptr<ExchangeSlot>> slots_;
};

}  // namespace vl
