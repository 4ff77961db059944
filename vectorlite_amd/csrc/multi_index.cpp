// multi_index.cpp -- one flat-index handle over several GPUs in one process (multi_index.hpp has the design).
#include "multi_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_set>

namespace vl {

namespace {
// what a part's call leaves in the worker thread's thread-local diagnostics, carried back to the caller's thread
struct PartOutcome {
    int rc = OK;
    std::string err;
    int path = PATH_NONE;
    uint64_t dim_expected = 0, dim_actual = 0;
    void capture(int r)
    {
        rc = r;
        path = last_path();
        if (r != OK) {
            err = last_error();
            get_dim_mismatch(&dim_expected, &dim_actual);
        }
    }
};

// A part's call on a worker thread: whatever it throws (a host allocation inside the part) becomes that part's outcome --
// an exception swallowed by the worker would leave the default rc = OK behind and the call would report success.
template <typename F>
void guarded(PartOutcome& o, F&& call)
{
    try {
        o.capture(call());
    } catch (const std::bad_alloc&) {
        o.rc = ERR_OOM;
        o.err = "host allocation failed in one part of the index";
    } catch (...) {
        o.rc = ERR_DEVICE;
        o.err = "internal error in one part of the index";
    }
}

int publish_first_error(const std::vector<PartOutcome>& o)
{
    for (size_t i = 0; i < o.size(); ++i)
        if (o[i].rc != OK) {
            set_last_error(o[i].err + " (part " + std::to_string(i) + ")");
            if (o[i].rc == ERR_DIM_MISMATCH) set_dim_mismatch(o[i].dim_expected, o[i].dim_actual);
            return o[i].rc;
        }
    return OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------------
int MultiFlatIndex::create(uint64_t dim, const int* devices, int n_dev, int mode, MultiFlatIndex** out)
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_dev < 1 || n_dev > SHARD_MAX_WORLD || (mode != REPLICAS && mode != ROW_SHARDS)) {
        set_last_error("vl_flat_create_multi: need 1.." + std::to_string(SHARD_MAX_WORLD) + " device ordinals and mode 0 (replicas) or 1 (row shards)");
        return ERR_INVALID_ARG;
    }
    std::unique_ptr<MultiFlatIndex> m(new MultiFlatIndex(dim, mode));
    for (int i = 0; i < n_dev; ++i) {
        GpuFlatIndex* p = nullptr;
        const int rc = GpuFlatIndex::create(dim, devices[i], &p);
        if (rc != OK) return rc;
        m->parts_.emplace_back(p);
        m->inflight_.emplace_back(new std::atomic<int>(0));
        m->answered_.emplace_back(new std::atomic<uint64_t>(0));
    }
    m->seq_.resize(mode == ROW_SHARDS ? (size_t)n_dev : 0);
    m->start_workers();
    *out = m.release();
    return OK;
}

void MultiFlatIndex::start_workers()
{
    for (size_t i = 1; i < parts_.size(); ++i) {  // part 0 runs on the calling thread
        workers_.emplace_back(new Worker());
        Worker* w = workers_.back().get();
        w->th = std::thread([w]() {
            std::unique_lock<std::mutex> lk(w->mu);
            for (;;) {
                w->cv.wait(lk, [w]() { return w->has_task || w->stop; });
                if (w->stop) return;
                std::function<void()> t = std::move(w->task);
                w->has_task = false;
                lk.unlock();
                try {
                    t();
                } catch (...) {  // the task records its own outcome; nothing may escape a worker
                }
                lk.lock();
                w->done = true;
                w->cv.notify_all();
            }
        });
    }
}

MultiFlatIndex::~MultiFlatIndex()
{
    for (auto& w : workers_) {
        {
            std::lock_guard<std::mutex> g(w->mu);
            w->stop = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    for (auto& sl : slots_)
        if (sl && sl->h_records) {
            (void)hipSetDevice(parts_.empty() ? 0 : parts_[0]->device());
            (void)hipHostFree(sl->h_records);
        }
}

MultiFlatIndex::ExchangeSlot* MultiFlatIndex::acquire_slot() const
{
    std::unique_lock<std::mutex> lk(slots_mu_);
    for (;;) {
        for (auto& sl : slots_)
            if (!sl->busy) {
                sl->busy = true;
                return sl.get();
            }
        if (slots_.size() < EXCHANGE_SLOTS) {
            slots_.emplace_back(new ExchangeSlot());
            slots_.back()->busy = true;
            return slots_.back().get();
        }
        slots_cv_.wait(lk);
    }
}

void MultiFlatIndex::release_slot(ExchangeSlot* s) const
{
    {
        std::lock_guard<std::mutex> lk(slots_mu_);
        s->busy = false;
    }
    slots_cv_.notify_one();
}

void MultiFlatIndex::run_parts(const std::function<void(int)>& fn) const
{
    const int P = (int)parts_.size();
    if (P == 1) {
        fn(0);
        return;
    }
    // the workers serve one fan-out at a time; a caller that finds them taken (another search or a batch in flight) does not
    // queue behind it: it walks its parts on its own thread -- every part's entry points are thread-safe and select their
    // own device -- so concurrent callers keep all GPUs busy between them
    std::unique_lock<std::mutex> g(run_mu_, std::try_to_lock);
    if (!g.owns_lock()) {
        for (int i = 0; i < P; ++i) {
            try {
                fn(i);
            } catch (...) {
            }
        }
        return;
    }
    for (int i = 1; i < P; ++i) {
        Worker* w = workers_[(size_t)i - 1].get();
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->task = [&fn, i]() { fn(i); };
            w->has_task = true;
            w->done = false;
        }
        w->cv.notify_all();
    }
    try {
        fn(0);
    } catch (...) {
    }
    for (int i = 1; i < P; ++i) {
        Worker* w = workers_[(size_t)i - 1].get();
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w]() { return w->done; });
    }
}

// ---------------------------------------------------------------------------------------------
// add / delete
// ---------------------------------------------------------------------------------------------
int MultiFlatIndex::add(uint64_t id, const double* values, uint64_t len)
{
    if (len != dim_) {  // src/index/flat.rs:83-85
        set_dim_mismatch(dim_, len);
        set_last_error("Vector dimension mismatch");
        return ERR_DIM_MISMATCH;
    }
    if (!values && dim_) return ERR_INVALID_ARG;
    return add_bulk(&id, values, 1, /*validate=*/true, /*values_on_device=*/false);
}

int MultiFlatIndex::add_bulk(const uint64_t* ids, const double* values, uint64_t n, bool validate, bool values_on_device,
                             int src_device)
{
    if (n == 0) return OK;
    if (!ids || (!values && dim_)) return ERR_INVALID_ARG;
    std::unique_lock<RwLock> lk(mu_);
    if (const int brc = refuse_if_broken()) return brc;
    const int P = (int)parts_.size();
    if (values_on_device && src_device < 0) src_device = parts_[0]->device();
    std::vector<PartOutcome> oc((size_t)P);

    // Test hook: VL_MULTI_INJECT_ADD_FAIL=<part> makes that part's share of the next bulk adds fail (after the pre-flight).
    int inject = -1;
    if (const char* inj = getenv("VL_MULTI_INJECT_ADD_FAIL"))
        if (*inj) inject = atoi(inj);
    auto part_add = [&](int i, const uint64_t* pi, const double* pv, uint64_t cnt) -> int {
        if (i == inject) {
            set_last_error("injected add failure on part " + std::to_string(i) + " (VL_MULTI_INJECT_ADD_FAIL)");
            return (int)ERR_DEVICE;
        }
        return parts_[(size_t)i]->add_bulk(pi, pv, cnt, validate, values_on_device, src_device);
    };
    std::vector<uint64_t> len0((size_t)P);
    for (int p = 0; p < P; ++p) len0[(size_t)p] = parts_[(size_t)p]->len();
    // A fan-out is all-or-nothing (advisor, round 3): (1) every part makes room BEFORE any part takes a row -- running out
    // of memory, the one failure a healthy device produces, then leaves every part as it was; (2) if a part fails anyway,
    // every part forgets the rows this call gave it, so replicas never diverge and a sharded index never holds a
    // non-prefix subset of the batch.
    auto roll_back = [&]() {
        for (int p = 0; p < P; ++p) parts_[(size_t)p]->truncate(len0[(size_t)p]);
    };

    if (mode_ == REPLICAS) {  // the same n sequential adds on every replica: identical state, identical outcome
        run_parts([&](int i) { guarded(oc[(size_t)i], [&]() { return parts_[(size_t)i]->reserve(len0[(size_t)i] + n); }); });
        {
            const int rc = publish_first_error(oc);
            if (rc != OK) return rc;  // nothing was added anywhere
        }
        run_parts([&](int i) { guarded(oc[(size_t)i], [&]() { return part_add(i, ids, values, n); }); });
        bool same = true;
        const uint64_t len_a = parts_[0]->len();
        for (int p = 1; p < P; ++p) same = same && oc[(size_t)p].rc == oc[0].rc && parts_[(size_t)p]->len() == len_a;
        if (!same) {  // a replica failed where the others did not: back to the common state, report the failure
            roll_back();
            for (int p = 0; p < P; ++p)
                if (oc[(size_t)p].rc != OK && oc[(size_t)p].rc != ERR_DUP_ID) {
                    set_last_error(oc[(size_t)p].err + " (no replica kept a row of this call)");
                    return oc[(size_t)p].rc;
                }
            set_last_error("the replicas disagreed on a bulk add (no replica kept a row of this call)");
            return ERR_DEVICE;
        }
        return publish_first_error(oc);
    }

    // ROW_SHARDS.  validate: n sequential add() calls on the WHOLE index -- stop at the first id that exists in any
    // shard or earlier in this call (src/index/flat.rs:86-88), rows before it are kept
    uint64_t n_take = n;
    int rc_after = OK;
    std::string dup_msg;
    if (validate) {
        std::unordered_set<uint64_t> seen;
        seen.reserve((size_t)std::min<uint64_t>(n, 1u << 20) * 2);
        for (uint64_t i = 0; i < n && rc_after == OK; ++i) {
            bool dup = !seen.insert(ids[i]).second;
            for (int p = 0; p < P && !dup; ++p) dup = parts_[(size_t)p]->contains(ids[i]);
            if (dup) {
                n_take = i;
                rc_after = ERR_DUP_ID;
                dup_msg = "Vector ID " + std::to_string(ids[i]) + " already exists";
            }
        }
    }
    if (n_take == 0) {
        if (rc_after != OK) set_last_error(dup_msg);
        return rc_after;
    }
    // contiguous runs of the new rows, dealt so that the shards level out (the shortest shards fill up first)
    std::vector<uint64_t> lens((size_t)P), start((size_t)P, 0), count((size_t)P, 0);
    uint64_t total = n_take;
    for (int p = 0; p < P; ++p) total += (lens[(size_t)p] = parts_[(size_t)p]->len());
    const uint64_t ideal = (total + (uint64_t)P - 1) / (uint64_t)P;
    uint64_t given = 0;
    for (int p = 0; p < P && given < n_take; ++p) {
        const uint64_t room = ideal > lens[(size_t)p] ? ideal - lens[(size_t)p] : 0;
        start[(size_t)p] = given;
        count[(size_t)p] = std::min<uint64_t>(room, n_take - given);
        given += count[(size_t)p];
    }
    if (given < n_take) {  // (cannot happen: the rooms add up to at least n_take) -- never drop rows
        count[(size_t)P - 1] += n_take - given;
    }
    run_parts([&](int i) {  // pre-flight: room on every shard before any shard takes a row
        const size_t p = (size_t)i;
        if (count[p] == 0) return;
        guarded(oc[p], [&]() { return parts_[p]->reserve(lens[p] + count[p]); });
    });
    {
        const int rc = publish_first_error(oc);
        if (rc != OK) return rc;  // nothing was added anywhere
    }
    run_parts([&](int i) {
        const size_t p = (size_t)i;
        if (count[p] == 0) return;
        // ids were validated against the whole index above; the shard's own check keeps its id table current
        guarded(oc[p], [&]() { return part_add(i, ids + start[p], values ? values + start[p] * dim_ : nullptr, count[p]); });
    });
    {
        const int rc = publish_first_error(oc);
        if (rc != OK) {  // a shard failed: no shard keeps its run (the index would hold a non-prefix subset of the batch)
            const std::string msg = last_error();
            roll_back();
            set_last_error(msg + " (no row of this call was kept)");
            return rc;
        }
    }
    for (int p = 0; p < P; ++p) {
        if (count[(size_t)p] == 0) continue;
        std::vector<uint64_t>& sq = seq_[(size_t)p];
        for (uint64_t j = 0; j < count[(size_t)p]; ++j) sq.push_back(next_seq_ + start[(size_t)p] + j);
    }
    next_seq_ += n_take;
    if (rc_after != OK) set_last_error(dup_msg);
    return rc_after;
}

int MultiFlatIndex::refuse_if_broken() const
{
    if (!broken_.load(std::memory_order_relaxed)) return OK;
    set_last_error("this multi-GPU handle failed on part of its GPUs during a delete and no longer answers (its parts disagree); rebuild it");
    return ERR_DEVICE;
}

int MultiFlatIndex::remove(uint64_t id)
{
    std::unique_lock<RwLock> lk(mu_);
    if (const int brc = refuse_if_broken()) return brc;
    const int P = (int)parts_.size();
    std::vector<PartOutcome> oc((size_t)P);
    std::vector<std::vector<uint64_t>> gone((size_t)P);
    const char* inj = getenv("VL_MULTI_INJECT_DELETE_FAIL");  // test hook: that part's delete fails
    const int inject = inj && *inj ? atoi(inj) : -1;
    run_parts([&](int i) {
        guarded(oc[(size_t)i], [&]() {
            if (i == inject) {
                set_last_error("injected delete failure (VL_MULTI_INJECT_DELETE_FAIL)");
                return (int)ERR_DEVICE;
            }
            return parts_[(size_t)i]->remove_report(id, &gone[(size_t)i]);
        });
    });
    for (int p = 0; p < P; ++p)
        if (oc[(size_t)p].rc != OK) broken_.store(true);  // a compaction died half way on that part: nothing to roll back to
    if (mode_ == ROW_SHARDS)
        for (int p = 0; p < P; ++p)
            for (uint64_t pos : gone[(size_t)p])  // descending: each erase leaves the earlier positions in place
                if (pos < seq_[(size_t)p].size()) seq_[(size_t)p].erase(seq_[(size_t)p].begin() + (std::ptrdiff_t)pos);
    return publish_first_error(oc);  // an absent id is Ok(()) (src/index/flat.rs:93-96)
}

// ---------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------
int MultiFlatIndex::pick_replica() const
{
    const int P = (int)parts_.size();
    const int first = (int)(rr_.fetch_add(1, std::memory_order_relaxed) % (uint32_t)P);  // ties go round
    int best = first, load = inflight_[(size_t)first]->load(std::memory_order_relaxed);
    for (int d = 1; d < P; ++d) {
        const int i = (first + d) % P;
        const int l = inflight_[(size_t)i]->load(std::memory_order_relaxed);
        if (l < load) {
            best = i;
            load = l;
        }
    }
    return best;
}

int MultiFlatIndex::search(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos, uint64_t* out_ids,
                           double* out_scores, uint64_t* out_n) const
{
    if (!out_n) return ERR_INVALID_ARG;
    std::shared_lock<RwLock> lk(mu_);
    if (const int brc = refuse_if_broken()) return brc;
    if (mode_ == REPLICAS) {
        const int i = pick_replica();
        inflight_[(size_t)i]->fetch_add(1, std::memory_order_relaxed);
        const int rc = parts_[(size_t)i]->search(query, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
        inflight_[(size_t)i]->fetch_sub(1, std::memory_order_relaxed);
        answered_[(size_t)i]->fetch_add(1, std::memory_order_relaxed);
        return rc;
    }
    *out_n = 0;
    if (metric < 0 || metric > 3) {
        set_last_error("unknown metric");
        return ERR_INVALID_ARG;
    }
    return shard_search(query, 1, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
}

int MultiFlatIndex::search_batch(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                                 uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (nq == 0) return OK;
    if (!out_n) return ERR_INVALID_ARG;
    for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
    if (metric < 0 || metric > 3) {
        set_last_error("unknown metric");
        return ERR_INVALID_ARG;
    }
    std::shared_lock<RwLock> lk(mu_);
    if (const int brc = refuse_if_broken()) return brc;
    if (mode_ == ROW_SHARDS) return shard_search(queries, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n);

    // REPLICAS: one contiguous run of queries per replica (row stride k in every output, so the runs are plain offsets)
    const int P = (int)parts_.size();
    if (P == 1 || nq < 2) {
        const int i = pick_replica();
        answered_[(size_t)i]->fetch_add(nq, std::memory_order_relaxed);
        return parts_[(size_t)i]->search_batch(queries, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
    }
    // (an empty index accepts any q_len: the parts decide, each on its own -- they hold the same rows)
    std::vector<PartOutcome> oc((size_t)P);
    const uint64_t per = (nq + (uint64_t)P - 1) / (uint64_t)P;
    run_parts([&](int i) {
        const uint64_t q0 = std::min<uint64_t>(nq, per * (uint64_t)i), q1 = std::min<uint64_t>(nq, q0 + per);
        if (q0 == q1) return;
        // q_len may be wrong (then every part reports the mismatch before reading a query): offsets use the caller's q_len
        guarded(oc[(size_t)i], [&]() {
            return parts_[(size_t)i]->search_batch(queries ? queries + q0 * q_len : nullptr, q1 - q0, q_len, k, metric,
                                                   out_pos ? out_pos + q0 * k : nullptr, out_ids ? out_ids + q0 * k : nullptr,
                                                   out_scores ? out_scores + q0 * k : nullptr, out_n + q0);
        });
        answered_[(size_t)i]->fetch_add(q1 - q0, std::memory_order_relaxed);
    });
    set_last_path(oc[0].path);
    return publish_first_error(oc);
}

int MultiFlatIndex::search_batch_device(const double* d_queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric,
                                        uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (nq == 0) return OK;
    if (!out_n) return ERR_INVALID_ARG;
    if (parts_.size() == 1 && mode_ == REPLICAS)
        return parts_[0]->search_batch_device(d_queries, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
    // every part needs the batch: one D2H copy, then the host form (each part stages its own copy)
    std::vector<double> hq;
    static const double never_read = 0.0;
    const double* hp = d_queries ? &never_read : nullptr;
    if (d_queries && q_len && q_len == dim_) {
        hq.resize((size_t)nq * q_len);
        if (hipSetDevice(parts_[0]->device()) != hipSuccess ||
            hipMemcpy(hq.data(), d_queries, hq.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            set_last_error("copying the device queries to the host failed");
            return ERR_DEVICE;
        }
        hp = hq.data();
    }
    return search_batch(hp, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
}

// the caller holds mu_ (shared) and has zeroed out_n
int MultiFlatIndex::shard_search(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                                 uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    const int P = (int)parts_.size();
    uint64_t total = 0, max_len = 0;
    std::vector<uint64_t> lens((size_t)P);
    for (int p = 0; p < P; ++p) {
        lens[(size_t)p] = parts_[(size_t)p]->len();
        total += lens[(size_t)p];
        max_len = std::max(max_len, lens[(size_t)p]);
    }
    if (total != 0 && q_len != dim_) {  // src/index/flat.rs:99-104 (skipped while the index is empty)
        set_dim_mismatch(dim_, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(q_len));
        return ERR_DIM_MISMATCH;
    }
    if (total == 0 || k == 0) return OK;
    if (!queries || !out_scores) return ERR_INVALID_ARG;
    const uint64_t ks = std::min<uint64_t>(k, max_len);
    const uint64_t words = shard_packed_words(nq, ks);
    if (nq > 0x7FFFFFFFull || ks > 0x7FFFFFFFull || words * (uint64_t)P > (1ull << 29)) {
        set_last_error("nq x k too large for one exchange");
        return ERR_INVALID_ARG;
    }
    // the record block and the merger of THIS search (one set per search in flight)
    ExchangeSlot* slot = acquire_slot();
    struct SlotGuard {
        const MultiFlatIndex* m;
        ExchangeSlot* s;
        ~SlotGuard() { m->release_slot(s); }
    } slot_guard{this, slot};
    if (hipSetDevice(parts_[0]->device()) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("hipSetDevice failed");
        return ERR_DEVICE;
    }
    if (words * (uint64_t)P > slot->h_records_cap) {
        if (slot->h_records) (void)hipHostFree(slot->h_records);
        slot->h_records = nullptr;
        slot->h_records_cap = 0;
        // portable: the shards' GPUs write nothing here, but their worker threads fill it while another device is current
        if (hipHostMalloc(reinterpret_cast<void**>(&slot->h_records), words * (uint64_t)P * 8, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            set_last_error("host allocation of the shard records failed");
            return ERR_OOM;
        }
        slot->h_records_cap = words * (uint64_t)P;
    }
    if (!slot->merger) slot->merger.reset(new ShardMerger(parts_[0]->device()));
    unsigned long long* const recs = slot->h_records;
    ShardMerger* const merger = slot->merger.get();
    // every shard answers the whole batch on its own rows; positions become global insertion numbers on the way
    std::vector<std::string> errs((size_t)P);
    std::vector<int> paths((size_t)P, PATH_NONE);
    run_parts([&](int i) {
        const size_t p = (size_t)i;
        unsigned long long* rec = recs + p * words;
        shard_search_local(parts_[p].get(), 0, UINT64_MAX, true, queries, nq, q_len, ks, metric, rec, false, seq_[p].data());
        if (rec[0] != 0) errs[p] = last_error();
        paths[p] = last_path();
        answered_[p]->fetch_add(nq, std::memory_order_relaxed);
    });
    // A shard of ONE row returns its score even when it is NaN (a 1-element sort never compares); with two or more
    // rows in the index the reference's sort panics on it (src/index/flat.rs:116) wherever the row is stored
    if (total >= 2)
        for (int p = 0; p < P; ++p) {
            if (lens[(size_t)p] != 1) continue;
            const unsigned long long* rec = recs + (size_t)p * words;
            if (rec[0] != 0) continue;
            const double* sc = reinterpret_cast<const double*>(rec + SHARD_HDR_WORDS + nq);
            for (uint64_t q = 0; q < nq; ++q)
                if (rec[SHARD_HDR_WORDS + q] >= 1 && sc[q * ks] != sc[q * ks]) {
                    set_last_error("NaN similarity score: the reference panics in partial_cmp().unwrap()");
                    return ERR_NAN_SCORE;
                }
        }
    const int rc = merger->merge_host(recs, (uint32_t)P, nq, ks, k, out_pos, out_ids, out_scores, out_n);
    if (rc != OK) {
        for (int p = 0; p < P; ++p)
            if (!errs[(size_t)p].empty()) {
                set_last_error(errs[(size_t)p] + " (shard " + std::to_string(p) + ")");
                break;
            }
        return rc;
    }
    int path = PATH_FAST;
    for (int p = 0; p < P; ++p)
        if (lens[(size_t)p] != 0 && paths[(size_t)p] > path) path = paths[(size_t)p];
    set_last_path(path);
    return OK;
}

// ---------------------------------------------------------------------------------------------
// lookups / export / clone
// ---------------------------------------------------------------------------------------------
uint64_t MultiFlatIndex::len() const
{
    std::shared_lock<RwLock> lk(mu_);
    if (mode_ == REPLICAS) return parts_[0]->len();
    uint64_t n = 0;
    for (auto& p : parts_) n += p->len();
    return n;
}

int MultiFlatIndex::get_vector(uint64_t id, double* out) const
{
    std::shared_lock<RwLock> lk(mu_);
    if (mode_ == REPLICAS) return parts_[0]->get_vector(id, out);
    // the first row with that id in insertion order (src/index/flat.rs:129-131), whichever shard holds it
    int best = -1;
    uint64_t best_pos = 0, best_seq = UINT64_MAX;
    for (size_t p = 0; p < parts_.size(); ++p) {
        uint64_t pos = 0;
        if (parts_[p]->find_first(id, &pos) != OK || pos >= seq_[p].size()) continue;
        if (seq_[p][pos] < best_seq) {
            best_seq = seq_[p][pos];
            best_pos = pos;
            best = (int)p;
        }
    }
    if (best < 0) return ERR_NOT_FOUND;
    return parts_[(size_t)best]->get_row_at(best_pos, out);
}

int MultiFlatIndex::max_id(uint64_t* out) const
{
    if (!out) return ERR_INVALID_ARG;
    std::shared_lock<RwLock> lk(mu_);
    if (mode_ == REPLICAS) return parts_[0]->max_id(out);
    bool any = false;
    uint64_t mx = 0;
    for (auto& p : parts_) {
        uint64_t v = 0;
        if (p->max_id(&v) == OK) {
            mx = any ? std::max(mx, v) : v;
            any = true;
        }
    }
    if (!any) return ERR_NOT_FOUND;
    *out = mx;
    return OK;
}

int MultiFlatIndex::reserve(uint64_t n_rows)
{
    std::unique_lock<RwLock> lk(mu_);
    const uint64_t P = parts_.size();
    const uint64_t per = mode_ == REPLICAS ? n_rows : (n_rows + P - 1) / P;
    for (auto& p : parts_) {
        const int rc = p->reserve(per);
        if (rc != OK) return rc;
    }
    return OK;
}

int MultiFlatIndex::export_rows(uint64_t* out_ids, double* out_values) const
{
    std::shared_lock<RwLock> lk(mu_);
    if (mode_ == REPLICAS) return parts_[0]->export_rows(out_ids, out_values);
    // storage order of the whole index = ascending insertion number: rank every shard's rows in the union
    const size_t P = parts_.size();
    std::vector<size_t> cur(P, 0);
    std::vector<std::vector<uint64_t>> rank(P);
    uint64_t total = 0;
    for (size_t p = 0; p < P; ++p) {
        rank[p].resize(seq_[p].size());
        total += seq_[p].size();
    }
    for (uint64_t r = 0; r < total; ++r) {
        size_t bp = P;
        for (size_t p = 0; p < P; ++p)
            if (cur[p] < seq_[p].size() && (bp == P || seq_[p][cur[p]] < seq_[bp][cur[bp]])) bp = p;
        rank[bp][cur[bp]++] = r;
    }
    std::vector<uint64_t> ids;
    std::vector<double> vals;
    for (size_t p = 0; p < P; ++p) {
        const uint64_t n = seq_[p].size();
        if (n == 0) continue;
        ids.resize(n);
        vals.resize(n * dim_);
        const int rc = parts_[p]->export_rows(ids.data(), vals.data());
        if (rc != OK) return rc;
        for (uint64_t j = 0; j < n; ++j) {
            if (out_ids) out_ids[rank[p][j]] = ids[j];
            if (out_values && dim_) std::memcpy(out_values + rank[p][j] * dim_, vals.data() + j * dim_, dim_ * sizeof(double));
        }
    }
    return OK;
}

int MultiFlatIndex::clone(MultiFlatIndex** out) const
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    std::shared_lock<RwLock> lk(mu_);
    std::unique_ptr<MultiFlatIndex> m(new MultiFlatIndex(dim_, mode_));
    for (auto& p : parts_) {
        GpuFlatIndex* c = nullptr;
        const int rc = p->clone(&c);
        if (rc != OK) return rc;
        m->parts_.emplace_back(c);
        m->inflight_.emplace_back(new std::atomic<int>(0));
        m->answered_.emplace_back(new std::atomic<uint64_t>(0));
    }
    m->seq_ = seq_;
    m->next_seq_ = next_seq_;
    m->start_workers();
    *out = m.release();
    return OK;
}

// ---------------------------------------------------------------------------------------------
// knobs and statistics: applied to / summed over the parts
// ---------------------------------------------------------------------------------------------
void MultiFlatIndex::force_path(int p)
{
    for (auto& x : parts_) x->force_path(p);
}
void MultiFlatIndex::set_single_filter(int mode)
{
    for (auto& x : parts_) x->set_single_filter(mode);
}
void MultiFlatIndex::set_coalescing(int max_batch, int window_us)
{
    for (auto& x : parts_) x->set_coalescing(max_batch, window_us);  // one queue per replica / shard
}
void MultiFlatIndex::coalesce_stats(uint64_t* batches, uint64_t* queries) const
{
    uint64_t b = 0, q = 0;
    for (auto& x : parts_) {
        uint64_t bb = 0, qq = 0;
        x->coalesce_stats(&bb, &qq);
        b += bb;
        q += qq;
    }
    if (batches) *batches = b;
    if (queries) *queries = q;
}
void MultiFlatIndex::coalesce_gather(int adaptive, uint64_t* waits, uint64_t* waited_us) const
{
    uint64_t w = 0, us = 0;
    for (auto& x : parts_) {
        uint64_t ww = 0, uu = 0;
        x->coalesce_gather(adaptive, &ww, &uu);
        w += ww;
        us += uu;
    }
    if (waits) *waits = w;
    if (waited_us) *waited_us = us;
}
void MultiFlatIndex::profile_enable(bool on)
{
    for (auto& x : parts_) x->profile_enable(on);
}
void MultiFlatIndex::profile_read(uint64_t* n, double* ms, uint64_t* bytes)
{
    uint64_t tn = 0, tb = 0;
    double tm = 0.0;
    for (auto& x : parts_) {
        uint64_t a = 0, c = 0;
        double b = 0.0;
        x->profile_read(&a, &b, &c);
        tn += a;
        tm += b;
        tb += c;
    }
    if (n) *n = tn;
    if (ms) *ms = tm;
    if (bytes) *bytes = tb;
}
void MultiFlatIndex::part_stats(uint64_t* rows, uint64_t* searches) const
{
    for (size_t p = 0; p < parts_.size(); ++p) {
        if (rows) rows[p] = parts_[p]->len();
        if (searches) searches[p] = answered_[p]->load(std::memory_order_relaxed);
    }
}

}  // namespace vl
