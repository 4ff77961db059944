// shard_comm.cpp -- host side of the row-sharded batched search: RCCL communicator, the one all-gather, and the
// buffers around the device merge (shard.hpp has the design and the exchange record's layout).
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "flat_index.hpp"
#include "shard.hpp"

namespace vl {

static_assert(sizeof(ncclUniqueId) == SHARD_ID_BYTES, "vl_comm id size must be sizeof(ncclUniqueId)");
static_assert(sizeof(unsigned long long) == sizeof(uint64_t) && sizeof(double) == sizeof(uint64_t), "u64 planes");

#define SH_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));        \
            (void)hipGetLastError(); /* a failed call (hipMalloc out of memory ...) leaves the thread's sticky error behind: the next launch's hipGetLastError() must not report it */ \
            return (e_ == hipErrorOutOfMemory) ? (int)ERR_OOM : (int)ERR_DEVICE;      \
        }                                                                             \
    } while (0)

#define SH_NCCL(expr)                                                                 \
    do {                                                                              \
        ncclResult_t r_ = (expr);                                                     \
        if (r_ != ncclSuccess) {                                                      \
            set_last_error(std::string(#expr) + ": " + ncclGetErrorString(r_));       \
            return (int)ERR_DEVICE;                                                   \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// local half
// ---------------------------------------------------------------------------------------------------------
void shard_search_local(const GpuFlatIndex* shard, uint64_t row_offset, uint64_t expected_len, bool any_rows,
                        const double* queries, uint64_t nq, uint64_t q_len, uint64_t ks, int metric,
                        unsigned long long* packed, bool queries_on_device, const uint64_t* pos_to_global)
{
    const uint64_t words = shard_packed_words(nq, ks), plane = nq * ks;
    std::memset(packed, 0, words * sizeof(unsigned long long));
    int status = OK;
    const uint64_t len = shard ? shard->len() : 0, dim = shard ? shard->dimension() : 0;
    packed[1] = len;
    packed[2] = dim;
    if (!shard || (!queries && nq && q_len)) {
        status = ERR_INVALID_ARG;
        set_last_error("shard search: null shard or queries");
    } else if (expected_len != UINT64_MAX && len != expected_len) {
        status = ERR_INVALID_ARG;
        set_last_error("this shard holds " + std::to_string(len) + " rows but the ranks agreed on " +
                       std::to_string(expected_len) + ": call vl_shard_sync after add/delete");
    } else if (any_rows && q_len != dim) {  // src/index/flat.rs:99-104 on the index as a whole
        status = ERR_DIM_MISMATCH;
        set_dim_mismatch(dim, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim) + ", got " + std::to_string(q_len));
    } else if (len != 0 && nq != 0 && ks != 0) {
        unsigned long long* cnt = packed + SHARD_HDR_WORDS;
        unsigned long long* sc = cnt + nq;
        unsigned long long* gp = sc + plane;
        unsigned long long* id = gp + plane;
        try {
            status = queries_on_device
                         ? shard->search_batch_device(queries, nq, q_len, ks, metric, reinterpret_cast<uint64_t*>(gp),
                                                      reinterpret_cast<uint64_t*>(id), reinterpret_cast<double*>(sc),
                                                      reinterpret_cast<uint64_t*>(cnt))
                         : shard->search_batch(queries, nq, q_len, ks, metric, reinterpret_cast<uint64_t*>(gp),
                                               reinterpret_cast<uint64_t*>(id), reinterpret_cast<double*>(sc),
                                               reinterpret_cast<uint64_t*>(cnt));
        } catch (const std::bad_alloc&) {
            status = ERR_OOM;
            set_last_error("host allocation failed in the shard search");
        } catch (...) {
            status = ERR_DEVICE;
            set_last_error("internal error in the shard search");
        }
        if (status == OK) {
            for (uint64_t q = 0; q < nq; ++q)
                for (uint64_t j = 0; j < cnt[q]; ++j) {  // local -> global position
                    unsigned long long& g = gp[q * ks + j];
                    g = pos_to_global ? pos_to_global[g] : g + row_offset;
                }
        } else {
            std::memset(cnt, 0, (nq + 3 * plane) * sizeof(unsigned long long));  // a failed shard offers nothing
        }
    }
    packed[0] = (unsigned long long)status;
}

// ---------------------------------------------------------------------------------------------------------
// ShardMerger
// ---------------------------------------------------------------------------------------------------------
ShardMerger::~ShardMerger()
{
    (void)hipSetDevice(device_);
    if (d_send_) (void)hipFree(d_send_);
    if (d_recv_) (void)hipFree(d_recv_);
    if (h_send_) (void)hipHostFree(h_send_);
    if (h_hdr_) (void)hipHostFree(h_hdr_);
    if (d_out_) (void)hipFree(d_out_);
    if (h_out_) (void)hipHostFree(h_out_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

static uint64_t out_words(uint64_t nq, uint64_t k_out) { return 3 * nq * k_out + nq + 2; }

int ShardMerger::ensure_stream()
{
    SH_HIP(hipSetDevice(device_));
    if (!stream_) SH_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    return OK;
}

bool ShardMerger::needs_growth(uint64_t world, uint64_t nq, uint64_t ks, uint64_t k_out) const
{
    const uint64_t words = shard_packed_words(nq, ks);
    return !stream_ || words > send_cap_ || words * world > recv_cap_ || out_words(nq, k_out) > out_cap_;
}

int ShardMerger::ensure(uint64_t world, uint64_t nq, uint64_t ks, uint64_t k_out)
{
    SH_HIP(hipSetDevice(device_));
    if (!stream_) SH_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    if (!h_hdr_) SH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_hdr_), SHARD_HDR_WORDS * 8, hipHostMallocDefault));
    const uint64_t words = shard_packed_words(nq, ks);
    if (words > send_cap_) {
        if (d_send_) (void)hipFree(d_send_);
        if (h_send_) (void)hipHostFree(h_send_);
        d_send_ = nullptr;
        h_send_ = nullptr;
        send_cap_ = 0;
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_send_), words * 8));
        SH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_send_), words * 8, hipHostMallocDefault));
        send_cap_ = words;
    }
    if (words * world > recv_cap_) {
        if (d_recv_) (void)hipFree(d_recv_);
        d_recv_ = nullptr;
        recv_cap_ = 0;
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_recv_), words * world * 8));
        recv_cap_ = words * world;
    }
    const uint64_t ow = out_words(nq, k_out);
    if (ow > out_cap_) {
        if (d_out_) (void)hipFree(d_out_);
        if (h_out_) (void)hipHostFree(h_out_);
        d_out_ = nullptr;
        h_out_ = nullptr;
        out_cap_ = 0;
        SH_HIP(hipMalloc(reinterpret_cast<void**>(&d_out_), ow * 8));
        SH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_out_), ow * 8, hipHostMallocDefault));
        out_cap_ = ow;
    }
    return OK;
}

int ShardMerger::merge_device(uint32_t world, uint64_t nq, uint64_t ks, uint64_t k, uint64_t* out_gpos,
                              uint64_t* out_ids, double* out_scores, uint64_t* out_n, hipEvent_t done)
{
    const uint64_t k_out = std::min<uint64_t>(k, (uint64_t)world * ks);
    const uint64_t plane = nq * k_out;
    unsigned long long* d_gpos = d_out_;
    unsigned long long* d_ids = d_out_ + plane;
    double* d_scores = reinterpret_cast<double*>(d_out_ + 2 * plane);
    unsigned long long* d_n = d_out_ + 3 * plane;
    ShardMergeOut* d_st = reinterpret_cast<ShardMergeOut*>(d_out_ + 3 * plane + nq);
    SH_HIP(launch_shard_merge(stream_, d_recv_, world, (uint32_t)nq, (uint32_t)ks, (uint32_t)k_out, d_gpos, d_ids, d_scores,
                              d_n, d_st));
    SH_HIP(hipMemcpyAsync(h_out_, d_out_, out_words(nq, k_out) * 8, hipMemcpyDeviceToHost, stream_));
    if (done) SH_HIP(hipEventRecord(done, stream_));
    SH_HIP(hipStreamSynchronize(stream_));
    const unsigned long long st = h_out_[3 * plane + nq], rk = h_out_[3 * plane + nq + 1];
    if (st != 0) {
        set_last_error("shard rank " + std::to_string(rk) + " reported status " + std::to_string(st));
        return (int)st;
    }
    for (uint64_t q = 0; q < nq; ++q) {
        const uint64_t m = std::min<uint64_t>(h_out_[3 * plane + q], k_out);
        if (out_gpos) std::memcpy(out_gpos + q * k, h_out_ + q * k_out, m * 8);
        if (out_ids) std::memcpy(out_ids + q * k, h_out_ + plane + q * k_out, m * 8);
        std::memcpy(out_scores + q * k, h_out_ + 2 * plane + q * k_out, m * 8);
        out_n[q] = m;
    }
    return OK;
}

int ShardMerger::merge_host(const unsigned long long* gathered, uint32_t world, uint64_t nq, uint64_t ks, uint64_t k,
                            uint64_t* out_gpos, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    const uint64_t k_out = std::min<uint64_t>(k, (uint64_t)world * ks);
    const int rc = ensure(world, nq, ks, k_out);
    if (rc != OK) return rc;
    SH_HIP(hipMemcpyAsync(d_recv_, gathered, shard_packed_words(nq, ks) * world * 8, hipMemcpyHostToDevice, stream_));
    return merge_device(world, nq, ks, k, out_gpos, out_ids, out_scores, out_n);
}

// ---------------------------------------------------------------------------------------------------------
// ShardComm
// ---------------------------------------------------------------------------------------------------------
int ShardComm::unique_id(uint8_t out[SHARD_ID_BYTES])
{
    if (!out) return ERR_INVALID_ARG;
    ncclUniqueId id;
    SH_NCCL(ncclGetUniqueId(&id));
    std::memcpy(out, &id, SHARD_ID_BYTES);
    return OK;
}

int ShardComm::create(const uint8_t id_bytes[SHARD_ID_BYTES], int world, int rank, int device, ShardComm** out)
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    if (!id_bytes || world < 1 || world > SHARD_MAX_WORLD || rank < 0 || rank >= world) {
        set_last_error("vl_comm_create: need an id, 1 <= world <= " + std::to_string(SHARD_MAX_WORLD) + " and 0 <= rank < world");
        return ERR_INVALID_ARG;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        set_last_error("no usable HIP device (vectorlite_amd has no CPU fallback)");
        return ERR_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        set_last_error("invalid device ordinal");
        return ERR_INVALID_ARG;
    }
    SH_HIP(hipSetDevice(device));
    ShardComm* c = new (std::nothrow) ShardComm(world, rank, device);
    if (!c) return ERR_OOM;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, SHARD_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);  // collective: every rank calls it
    if (r != ncclSuccess) {
        set_last_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        delete c;
        return ERR_DEVICE;
    }
    c->comm_ = comm;
    // what every later collective may need without allocating: the exchange stream, the 8 + 2 * world word status
    // buffers (sync's table, the pre-flight of a growing exchange) and the timing events
    const size_t sw = 8 + 2 * (size_t)world;
    if (c->merger_.ensure_stream() != OK || hipMalloc(reinterpret_cast<void**>(&c->d_status_), sw * 8) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_status_), sw * 8, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("vl_comm_create: could not allocate the status exchange buffers");
        delete c;  // collective resources are released; the peers' next collective on this communicator fails
        return ERR_OOM;
    }
    for (auto& e : c->ev_)
        if (hipEventCreate(&e) != hipSuccess) {
            (void)hipGetLastError();
            delete c;
            return ERR_DEVICE;
        }
    *out = c;
    return OK;
}

ShardComm::~ShardComm()
{
    (void)hipSetDevice(merger_.device());
    if (comm_) (void)ncclCommDestroy(static_cast<ncclComm_t>(comm_));
    if (d_status_) (void)hipFree(d_status_);
    if (h_status_) (void)hipHostFree(h_status_);
    for (auto& e : ev_)
        if (e) (void)hipEventDestroy(e);
}

// A rank that cannot reach a collective its peers are about to post must not simply return: they would wait in
// the all-gather for ever.  Whatever fails locally BEFORE a collective is therefore either carried into it (word 0
// of the record; the pre-flight status exchange below for buffer growth) or, for a device failure that leaves no
// way to take part (hipSetDevice / a copy on the exchange stream), ends with ncclCommAbort: the peers' collective
// then fails instead of hanging, and this communicator refuses every later call.
int ShardComm::fail_and_abort(int rc)
{
    if (comm_) {
        (void)ncclCommAbort(static_cast<ncclComm_t>(comm_));
        comm_ = nullptr;
    }
    dead_ = true;
    synced_ = false;
    set_last_error(std::string(last_error()) + " (before a collective: communicator aborted so that the other ranks fail instead of waiting)");
    return rc;
}

#define SH_HIP_OR_ABORT(expr)                                                          \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));        \
            (void)hipGetLastError(); /* a failed call (hipMalloc out of memory ...) leaves the thread's sticky error behind: the next launch's hipGetLastError() must not report it */ \
            return fail_and_abort((e_ == hipErrorOutOfMemory) ? (int)ERR_OOM : (int)ERR_DEVICE); \
        }                                                                             \
    } while (0)

// collective: every rank's `mine` (0 = fine) -> the first non-zero status of any rank, through the 8-word buffers
// made at create(): nothing is allocated on the way
int ShardComm::exchange_status(unsigned long long mine, unsigned long long* first_bad, int* bad_rank)
{
    h_status_[0] = mine;
    SH_HIP_OR_ABORT(hipSetDevice(merger_.device()));
    SH_HIP_OR_ABORT(hipMemcpyAsync(d_status_, h_status_, 8, hipMemcpyHostToDevice, merger_.stream()));
    const ncclResult_t r = ncclAllGather(d_status_, d_status_ + 8, 1, ncclUint64, static_cast<ncclComm_t>(comm_), merger_.stream());
    if (r != ncclSuccess) {
        set_last_error(std::string("ncclAllGather (status): ") + ncclGetErrorString(r));
        return fail_and_abort(ERR_DEVICE);
    }
    SH_HIP_OR_ABORT(hipMemcpyAsync(h_status_ + 8, d_status_ + 8, (size_t)world_ * 8, hipMemcpyDeviceToHost, merger_.stream()));
    SH_HIP(hipStreamSynchronize(merger_.stream()));
    *first_bad = 0;
    *bad_rank = -1;
    for (int r2 = 0; r2 < world_; ++r2)
        if (h_status_[8 + r2] != 0) {
            *first_bad = h_status_[8 + r2];
            *bad_rank = r2;
            break;
        }
    return OK;
}

// Buffers for an exchange of (nq, ks): whether they have to grow depends only on the history of (nq, ks, world) --
// the same on every rank -- so all ranks take the pre-flight together or skip it together; a rank whose allocation
// fails says so THERE and every rank returns VL_ERR_OOM, nobody is left in the big all-gather.
int ShardComm::ensure_exchange(uint64_t nq, uint64_t ks, uint64_t k_out)
{
    if (!merger_.needs_growth((uint64_t)world_, nq, ks, k_out)) return OK;
    int rc = merger_.ensure((uint64_t)world_, nq, ks, k_out);
    if (const char* inj = getenv("VL_SHARD_INJECT_OOM"))  // tests: "<rank>" makes that rank's growth fail
        if (atoi(inj) == rank_) {
            set_last_error("injected allocation failure (VL_SHARD_INJECT_OOM)");
            rc = ERR_OOM;
        }
    const std::string local_msg = rc != OK ? std::string(last_error()) : std::string();
    unsigned long long bad = 0;
    int bad_rank = -1;
    const int xrc = exchange_status((unsigned long long)rc, &bad, &bad_rank);
    if (xrc != OK) return xrc;
    if (bad != 0) {
        set_last_error("shard rank " + std::to_string(bad_rank) + " could not size its exchange buffers (status " +
                       std::to_string(bad) + ")" + (local_msg.empty() ? "" : "; this rank: " + local_msg));
        return (int)bad;
    }
    return OK;
}

int ShardComm::sync(const GpuFlatIndex* shard, uint64_t* out_offset, uint64_t* out_total)
{
    if (!shard) return ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(mu_);
    if (dead_) {
        set_last_error("this communicator was aborted after a local failure");
        return ERR_DEVICE;
    }
    synced_ = false;
    // (len, dim) travel through the status buffers of create(): no allocation between here and the collective
    h_status_[0] = shard->len();
    h_status_[1] = shard->dimension();
    SH_HIP_OR_ABORT(hipSetDevice(merger_.device()));
    SH_HIP_OR_ABORT(hipMemcpyAsync(d_status_, h_status_, 2 * 8, hipMemcpyHostToDevice, merger_.stream()));
    {
        const ncclResult_t r = ncclAllGather(d_status_, d_status_ + 8, 2, ncclUint64, static_cast<ncclComm_t>(comm_), merger_.stream());
        if (r != ncclSuccess) {
            set_last_error(std::string("ncclAllGather (sync): ") + ncclGetErrorString(r));
            return fail_and_abort(ERR_DEVICE);
        }
    }
    std::vector<unsigned long long> all(2 * (size_t)world_);
    SH_HIP_OR_ABORT(hipMemcpyAsync(h_status_ + 8, d_status_ + 8, all.size() * 8, hipMemcpyDeviceToHost, merger_.stream()));
    SH_HIP(hipStreamSynchronize(merger_.stream()));
    std::memcpy(all.data(), h_status_ + 8, all.size() * 8);
    lens_.assign((size_t)world_, 0);
    uint64_t off = 0, total = 0, mx = 0;
    for (int r = 0; r < world_; ++r) {
        const uint64_t len = all[2 * r], dim = all[2 * r + 1];
        if (dim != shard->dimension()) {  // every rank sees the same table, so every rank fails alike
            set_last_error("shard rank " + std::to_string(r) + " has dimension " + std::to_string(dim) + ", this rank " +
                           std::to_string(shard->dimension()));
            return ERR_DIM_MISMATCH;
        }
        lens_[(size_t)r] = len;
        if (r < rank_) off += len;
        total += len;
        mx = std::max(mx, len);
    }
    if (total >= 0xFFFFFFFFFFFFull) return ERR_INVALID_ARG;
    dim_ = shard->dimension();
    offset_ = off;
    total_ = total;
    max_len_ = mx;
    synced_ = true;
    if (out_offset) *out_offset = off;
    if (out_total) *out_total = total;
    return OK;
}

int ShardComm::search_batch(const GpuFlatIndex* shard, const double* queries, uint64_t nq, uint64_t q_len, uint64_t k,
                            int metric, uint64_t* out_gpos, uint64_t* out_ids, double* out_scores, uint64_t* out_n,
                            bool queries_on_device)
{
    if (nq == 0) return OK;
    if (!out_n) return ERR_INVALID_ARG;
    for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
    if (metric < 0 || metric > 3) {
        set_last_error("unknown metric");
        return ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> g(mu_);
    if (dead_) {
        set_last_error("this communicator was aborted after a local failure");
        return ERR_DEVICE;
    }
    // Everything decided before the collective is a function of the arguments (the same on every rank by
    // contract) and of the table agreed at the last sync(): all ranks return early together or go on together.
    if (!synced_) {
        set_last_error("vl_shard_search_batch: call vl_shard_sync first (and again after any add/delete)");
        return ERR_INVALID_ARG;
    }
    if (k == 0 || total_ == 0) return OK;  // truncate(0); empty index accepts any query length (src/index/flat.rs:99)
    if (!shard || !queries || !out_scores) return ERR_INVALID_ARG;
    const uint64_t ks = std::min<uint64_t>(k, max_len_);  // the most any one shard can offer
    const uint64_t k_out = std::min<uint64_t>(k, (uint64_t)world_ * ks);
    const uint64_t words = shard_packed_words(nq, ks);
    if (nq > 0x7FFFFFFFull || ks > 0x7FFFFFFFull || words * (uint64_t)world_ > (1ull << 29)) {  // 4 GiB of records
        set_last_error("vl_shard_search_batch: nq x k too large for one exchange");
        return ERR_INVALID_ARG;
    }
    int rc = ensure_exchange(nq, ks, k_out);  // collective when the buffers have to grow (the same on every rank)
    if (rc != OK) return rc;
    const bool prof = profile_;
    const auto t_local = std::chrono::steady_clock::now();
    // from here to the all-gather nothing returns: a local failure travels in word 0 of this rank's record.
    // First choice: the finalize kernel writes this rank's record where the all-gather reads it (d_send) -- no pinned host
    // record, no H2D copy of it, only the 4 header words cross PCIe; batches that do not take the MFMA filter (and every
    // failure) build the record on the host as before.
    bool on_device = false;
    {
        // the 4 header words of a device-written record ("ok", this shard's length, the dimension) are known now: they
        // travel to d_send while the search runs, not after it (a record built on the host overwrites them below)
        unsigned long long* hdr = merger_.h_hdr();
        hdr[0] = OK;
        hdr[1] = lens_[(size_t)rank_];
        hdr[2] = dim_;
        hdr[3] = 0;
        SH_HIP_OR_ABORT(hipSetDevice(merger_.device()));
        SH_HIP_OR_ABORT(hipMemcpyAsync(merger_.d_send(), hdr, SHARD_HDR_WORDS * 8, hipMemcpyHostToDevice, merger_.stream()));
        int drc = OK;
        try {
            drc = shard->search_batch_to_record(queries, queries_on_device, nq, q_len, ks, metric, offset_, merger_.d_send(), &on_device);
        } catch (...) {
            drc = ERR_DEVICE;
            on_device = false;
        }
        if (drc != OK) on_device = false;  // the host path below repeats the search and reports the failure in word 0
        if (on_device && shard->len() != lens_[(size_t)rank_]) on_device = false;  // mutated since sync(): the host path says so
    }
    std::string local_msg;
    if (on_device) {
        ++rec_device_;
    } else {
        shard_search_local(shard, offset_, lens_[(size_t)rank_], total_ != 0, queries, nq, q_len, ks, metric, merger_.h_send(),
                           queries_on_device);
        if (merger_.h_send()[0] != 0) local_msg = std::string(last_error());
        ++rec_host_;
    }
    const double local_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_local).count();
    SH_HIP_OR_ABORT(hipSetDevice(merger_.device()));
    if (prof) SH_HIP_OR_ABORT(hipEventRecord(ev_[0], merger_.stream()));
    if (!on_device)
        SH_HIP_OR_ABORT(hipMemcpyAsync(merger_.d_send(), merger_.h_send(), words * 8, hipMemcpyHostToDevice, merger_.stream()));
    if (prof) SH_HIP_OR_ABORT(hipEventRecord(ev_[1], merger_.stream()));
    // THE exchange step of the path: one all-gather of per-shard top-k records (config 3: 1024 queries x k 10
    // -> 254 KB per rank), latency-bound on xGMI -- one collective, not a ring of small ones
    {
        const ncclResult_t r = ncclAllGather(merger_.d_send(), merger_.d_recv(), words, ncclUint64,
                                             static_cast<ncclComm_t>(comm_), merger_.stream());
        if (r != ncclSuccess) {
            set_last_error(std::string("ncclAllGather: ") + ncclGetErrorString(r));
            return fail_and_abort(ERR_DEVICE);
        }
    }
    if (prof) SH_HIP(hipEventRecord(ev_[2], merger_.stream()));
    rc = merger_.merge_device((uint32_t)world_, nq, ks, k, out_gpos, out_ids, out_scores, out_n, prof ? ev_[3] : nullptr);
    if (prof) {  // merge_device has synchronised the stream (or failed before its event: the read below then fails too)
        float h2d = 0.f, ag = 0.f, mg = 0.f;
        if (hipEventElapsedTime(&h2d, ev_[0], ev_[1]) == hipSuccess && hipEventElapsedTime(&ag, ev_[1], ev_[2]) == hipSuccess &&
            hipEventElapsedTime(&mg, ev_[2], ev_[3]) == hipSuccess) {
            prof_calls_ += 1;
            prof_local_ms_ += local_ms;
            prof_h2d_ms_ += h2d;
            prof_allgather_ms_ += ag;
            prof_merge_ms_ += mg;
        } else {
            (void)hipGetLastError();
        }
    }
    if (rc != OK && !local_msg.empty()) set_last_error(std::string(last_error()) + "; this rank: " + local_msg);
    return rc;
}

void ShardComm::record_paths(uint64_t* on_device, uint64_t* via_host) const
{
    if (on_device) *on_device = rec_device_;
    if (via_host) *via_host = rec_host_;
}

void ShardComm::profile_enable(bool on)
{
    std::lock_guard<std::mutex> g(mu_);
    profile_ = on;
}

void ShardComm::profile_read(uint64_t* calls, double* local_ms, double* h2d_ms, double* allgather_ms, double* merge_ms)
{
    std::lock_guard<std::mutex> g(mu_);
    if (calls) *calls = prof_calls_;
    if (local_ms) *local_ms = prof_local_ms_;
    if (h2d_ms) *h2d_ms = prof_h2d_ms_;
    if (allgather_ms) *allgather_ms = prof_allgather_ms_;
    if (merge_ms) *merge_ms = prof_merge_ms_;
    prof_calls_ = 0;
    prof_local_ms_ = prof_h2d_ms_ = prof_allgather_ms_ = prof_merge_ms_ = 0.0;
}

}  // namespace vl
