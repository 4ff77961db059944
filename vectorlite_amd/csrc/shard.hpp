// shard.hpp -- row-sharded batched flat search across the GPUs of one node (no reference counterpart:
// the reference is one process).  One process per GPU; rank r holds the contiguous
// row range [offset_r, offset_r + len_r) of the corpus as an ordinary GpuFlatIndex; every rank answers the
// whole query batch on its own shard with the single-GPU pipeline (exact f64 scores), ONE ncclAllGather
// (RCCL over xGMI) exchanges the per-shard top-k, and a device kernel merges them in the reference's order:
// score descending, ties by GLOBAL storage position ascending (src/index/flat.rs:116 on the whole corpus).
//
// Exchange record of one rank ("packed shard result"), u64 words, nq queries, stride ks entries per query:
//     [0] status (vl_status of the local search)   [1] shard length   [2] dimension   [3] reserved
//     [4 .. 4+nq)            count[q]              (results this shard has for query q, <= ks)
//     then nq*ks score bits (f64), nq*ks global positions, nq*ks ids   (three planes, row stride ks)
// A rank ALWAYS takes part in the collective, whatever its local status: errors travel in word 0, so one
// failing shard fails the call on every rank instead of leaving the others waiting in the all-gather.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <vector>

namespace vl {

class GpuFlatIndex;

constexpr uint32_t SHARD_HDR_WORDS = 4;
constexpr int SHARD_MAX_WORLD = 64;
constexpr uint32_t SHARD_ID_BYTES = 128;  // sizeof(ncclUniqueId)

inline uint64_t shard_packed_words(uint64_t nq, uint64_t ks) { return SHARD_HDR_WORDS + nq + 3 * nq * ks; }

struct ShardMergeOut {  // written by the merge kernel's block 0
    unsigned long long status;  // first non-OK status among the ranks (0 = all fine)
    unsigned long long rank;    // the rank that reported it
};

// gathered: [world][shard_packed_words(nq, ks)] on the device.  Outputs have row stride k_out.
hipError_t launch_shard_merge(hipStream_t stream, const unsigned long long* gathered, uint32_t world, uint32_t nq,
                              uint32_t ks, uint32_t k_out, unsigned long long* out_gpos, unsigned long long* out_ids,
                              double* out_scores, unsigned long long* out_n, ShardMergeOut* out_status);

// Local half: nq searches on `shard` with stride ks, packed into `packed` (host, shard_packed_words words).
// Never fails as a call: the status goes into word 0.  expected_len: the length the other ranks were told
// (UINT64_MAX = do not check).  any_rows: the sharded index as a whole holds rows (an empty SHARD of a
// non-empty index still rejects a query of the wrong length, like one index holding all rows would).
void shard_search_local(const GpuFlatIndex* shard, uint64_t row_offset, uint64_t expected_len, bool any_rows,
                        const double* queries, uint64_t nq, uint64_t q_len, uint64_t ks, int metric,
                        unsigned long long* packed, bool queries_on_device = false,  // queries: host, or this GPU's memory
                        const uint64_t* pos_to_global = nullptr);  // shards that are not one contiguous range: a table instead of the offset

// Device buffers + stream for gather/merge on one GPU.
class ShardMerger {
public:
    explicit ShardMerger(int device) : device_(device) {}
    ~ShardMerger();
    int ensure(uint64_t world, uint64_t nq, uint64_t ks, uint64_t k_out);
    int ensure_stream();
    bool needs_growth(uint64_t world, uint64_t nq, uint64_t ks, uint64_t k_out) const;  // would ensure() allocate?
    // gathered records already on the host (another transport did the exchange): H2D, merge, D2H
    int merge_host(const unsigned long long* gathered, uint32_t world, uint64_t nq, uint64_t ks, uint64_t k,
                   uint64_t* out_gpos, uint64_t* out_ids, double* out_scores, uint64_t* out_n);
    // after the collective filled d_recv(): merge + copy out (k = the caller's row stride)
    int merge_device(uint32_t world, uint64_t nq, uint64_t ks, uint64_t k, uint64_t* out_gpos, uint64_t* out_ids,
                     double* out_scores, uint64_t* out_n, hipEvent_t done = nullptr);  // `done`: recorded behind the D2H copy
    hipStream_t stream() const { return stream_; }
    unsigned long long* d_send() const { return d_send_; }
    unsigned long long* d_recv() const { return d_recv_; }
    unsigned long long* h_send() const { return h_send_; }
    unsigned long long* h_hdr() const { return h_hdr_; }  // pinned: the 4 header words of a record the device wrote the rest of
    int device() const { return device_; }

private:
    int device_;
    hipStream_t stream_ = nullptr;
    unsigned long long* d_send_ = nullptr;
    unsigned long long* d_recv_ = nullptr;
    unsigned long long* h_send_ = nullptr;  // pinned
    unsigned long long* h_hdr_ = nullptr;   // pinned, SHARD_HDR_WORDS words
    uint64_t send_cap_ = 0, recv_cap_ = 0;
    unsigned long long* d_out_ = nullptr;   // gpos | ids | scores | n | status
    unsigned long long* h_out_ = nullptr;   // pinned
    uint64_t out_cap_ = 0;
};

// One rank's end of the RCCL communicator + what the ranks agreed on at the last sync().
class ShardComm {
public:
    static int unique_id(uint8_t out[SHARD_ID_BYTES]);
    static int create(const uint8_t id[SHARD_ID_BYTES], int world, int rank, int device, ShardComm** out);
    ~ShardComm();
    int world() const { return world_; }
    int rank() const { return rank_; }
    // collective: all-gather of (len, dim); caches every rank's length, returns this rank's offset and the total
    int sync(const GpuFlatIndex* shard, uint64_t* out_offset, uint64_t* out_total);
    // collective: local search + ONE all-gather + device merge; outputs [nq, k] (row stride k), identical on every rank
    int search_batch(const GpuFlatIndex* shard, const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric,
                     uint64_t* out_gpos, uint64_t* out_ids, double* out_scores, uint64_t* out_n,
                     bool queries_on_device = false);
    // where a batch's time goes, summed over calls: the local search (host clock), and on the exchange stream (HIP
    // events) the H2D copy of this rank's record, the ncclAllGather, the merge kernel + D2H of the answer
    void profile_enable(bool on);
    void profile_read(uint64_t* calls, double* local_ms, double* h2d_ms, double* allgather_ms, double* merge_ms);
    // of the batches answered so far: how many had their exchange record written on the device by the finalize kernel (only
    // the 32-byte header then crosses PCIe in front of the all-gather) / how many went through the pinned host record
    void record_paths(uint64_t* on_device, uint64_t* via_host) const;

private:
    ShardComm(int world, int rank, int device) : world_(world), rank_(rank), merger_(device) {}
    int fail_and_abort(int rc);
    int exchange_status(unsigned long long mine, unsigned long long* first_bad, int* bad_rank);
    int ensure_exchange(uint64_t nq, uint64_t ks, uint64_t k_out);
    bool dead_ = false;                         // aborted after a local failure in front of a collective
    unsigned long long* d_status_ = nullptr;    // [8] send + [2 * world] receive, made at create()
    unsigned long long* h_status_ = nullptr;    // pinned, same shape
    hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
    bool profile_ = false;
    uint64_t rec_device_ = 0, rec_host_ = 0;
    uint64_t prof_calls_ = 0;
    double prof_local_ms_ = 0.0, prof_h2d_ms_ = 0.0, prof_allgather_ms_ = 0.0, prof_merge_ms_ = 0.0;
    int world_, rank_;
    void* comm_ = nullptr;  // ncclComm_t
    ShardMerger merger_;
    std::mutex mu_;         // one collective at a time per communicator
    bool synced_ = false;
    uint64_t dim_ = 0;
    std::vector<uint64_t> lens_;  // every rank's shard length at the last sync()
    uint64_t offset_ = 0, total_ = 0, max_len_ = 0;
};

}  // namespace vl
