// hnsw_index.cpp -- host logic of the GPU HNSW index (see hnsw_index.hpp).
//
// Mirrors `impl VectorIndex for HNSWIndex` (reference src/index/hnsw.rs:363-496): argument meaning,
// error texts, tombstone deletes, ef = min(k, len), score conversion, stable descending sort and
// truncate(k).  Every distance is evaluated on the GPU (hnsw.hip); the host only keeps the id maps,
// draws node levels and post-processes the <= ef (node, u64 distance) pairs a walk returns.
#include "hnsw_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace vl {

#define VL_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                   \
            (void)hipGetLastError(); /* a failed call (hipMalloc out of memory ...) leaves the thread's sticky error behind: the next launch's hipGetLastError() must not report it */ \
            return (e_ == hipErrorOutOfMemory) ? (int)ERR_OOM : (int)ERR_DEVICE;                 \
        }                                                                                        \
    } while (0)
#define VL_TRY(expr)               \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != OK) return rc_; \
    } while (0)

double hnsw_score(uint64_t d_u64, int metric)
{
    const double distance = (double)d_u64 / 1000.0;  // src/index/hnsw.rs:478
    switch (metric) {                                 // src/index/hnsw.rs:51-75
    case EUCLIDEAN:
    case MANHATTAN: return 1.0 / (1.0 + distance);
    case COSINE: return 1.0 - distance / 1000.0;
    default: {
        double v = (1000.0 - distance) / 1000.0;
        if (v < 0.0) v = 0.0;
        if (v > 1.0) v = 1.0;
        return v;
    }
    }
}

namespace {
constexpr uint32_t INSERT_BATCH_MAX = 4096;

uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// level = floor(-ln(u) / ln(M)), u uniform in (0, 1]: the standard HNSW level law
int draw_level(uint64_t seed, uint64_t node, uint32_t m)
{
    const uint64_t r = splitmix64(seed * 0x100000001B3ull + node);
    const double u = ((double)(r >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    int l = (int)std::floor(-std::log(u) / std::log((double)m));
    if (l < 0) l = 0;
    if (l > HNSW_MAX_LEVEL) l = HNSW_MAX_LEVEL;
    return l;
}

template <typename T>
int regrow(T** p, uint64_t old_count, uint64_t new_count, hipStream_t s, bool zero_new)
{
    T* np_ = nullptr;
    VL_HIP(hipMalloc(reinterpret_cast<void**>(&np_), new_count * sizeof(T)));
    if (zero_new) VL_HIP(hipMemsetAsync(np_, 0, new_count * sizeof(T), s));
    if (*p && old_count) VL_HIP(hipMemcpyAsync(np_, *p, old_count * sizeof(T), hipMemcpyDeviceToDevice, s));
    VL_HIP(hipStreamSynchronize(s));
    if (*p) (void)hipFree(*p);
    *p = np_;
    return OK;
}
}  // namespace

HnswIndex::HnswIndex(uint64_t dim, int metric, const HnswParams& p, int device)
    : dim_(dim), metric_(metric), params_(p), device_(device)
{
    // concurrent single searches share walk launches by default (window 0: a lone caller walks alone, at once);
    // here rather than in create() so that clones start the same way
    const char* ce = getenv("VL_COALESCE");
    if (!(ce && ce[0] == '0')) set_coalescing(COALESCE_DEFAULT_BATCH, 0);
}

int HnswIndex::create(uint64_t dim, int metric, const HnswParams& p, int device, HnswIndex** out)
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    if (metric < 0 || metric > 3) return ERR_INVALID_ARG;
    if (dim == 0) {  // HNSWIndex::new panics on dim 0 (src/index/hnsw.rs:217-219)
        set_last_error("HNSW index dimension cannot be 0");
        return ERR_INVALID_ARG;
    }
    if (p.m == 0 || p.m > 64 || p.m0 == 0 || p.m0 > 64 || p.ef_construction == 0 ||
        p.ef_construction > (uint32_t)HNSW_MAX_EF || dim > 3072) {  // the walk keeps the query in LDS: 48 B per dimension per workgroup
        set_last_error("HNSW parameters out of range (M, M0 <= 64; 1 <= ef_construction <= " + std::to_string(HNSW_MAX_EF) + "; dim <= 3072)");
        return ERR_INVALID_ARG;
    }
    std::unique_ptr<HnswIndex> h(new HnswIndex(dim, metric, p, device));
    GpuFlatIndex* st = nullptr;
    VL_TRY(GpuFlatIndex::create(dim, device, &st));
    h->store_.reset(st);
    VL_HIP(hipSetDevice(device));
    VL_HIP(hipStreamCreateWithFlags(&h->stream_, hipStreamNonBlocking));
    VL_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_stat_evals_), sizeof(unsigned long long)));
    VL_HIP(hipMemset(h->d_stat_evals_, 0, sizeof(unsigned long long)));
    if (const char* mb = getenv("VL_HNSW_MIN_BEAM"))
        if (*mb) h->set_min_beam((uint32_t)std::max(0, atoi(mb)));
    *out = h.release();
    return OK;
}

HnswIndex::~HnswIndex()
{
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    void* dev[] = {d_nbr0_, d_dist0_, d_cnt0_, d_level_, d_upper_off_, d_nbrU_, d_distU_, d_cntU_,
                   d_lock_, d_indeg0_, d_node_id_, d_live_, d_stat_evals_};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    pool_free_.clear();
    pool_all_.clear();
    if (stream_) (void)hipStreamDestroy(stream_);
}

uint64_t HnswIndex::len() const
{
    std::shared_lock<RwLock> lk(mu_);
    return live_count_;
}

// ---------------------------------------------------------------------------------------------
// walk scratch pool
// ---------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t WALK_LOG_CAP = 8192;          // nodes a walk may mark before clear() falls back to wiping its bitmap
constexpr uint32_t WALK_SLOTS_MAX = 4096;
constexpr uint64_t WALK_SCRATCH_BYTES = 2ull << 30;  // ceiling for one scratch's visited sets
constexpr size_t WALK_POOL_MAX = 16;
constexpr uint64_t WALK_POOL_BYTES = 8ull << 30;     // ceiling for ALL scratches of one index (idle ones are trimmed to stay under it)
}  // namespace

WalkScratch::~WalkScratch()
{
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (d_bits) (void)hipFree(d_bits);
    if (d_log) (void)hipFree(d_log);
    if (d_q) (void)hipFree(d_q);
    if (d_out) (void)hipFree(d_out);
    if (h_out) (void)hipHostFree(h_out);
    if (stream) (void)hipStreamDestroy(stream);
}

// A scratch for up to `walks` walks in flight (fewer slots only mean a query waits for a free wave inside the kernel).
// Caller holds mu_ (shared or unique), so g_cap_ is stable.
// The pool is bounded by BYTES of visited sets (WALK_POOL_BYTES), not only by count: when a new scratch would go over
// the budget the largest idle one is freed first, and when nothing is idle the caller waits for a release.  The device
// allocation itself runs without pool_mu_ (the slot and its bytes are reserved under the lock), so other searchers keep
// acquiring and releasing while a scratch is being made.
WalkScratch* HnswIndex::acquire_scratch(uint64_t walks) const
{
    const uint32_t words = (uint32_t)((g_cap_ + 31) / 32);
    const uint64_t per_slot = ((uint64_t)words + WALK_LOG_CAP) * sizeof(uint32_t);
    uint64_t max_slots = std::min<uint64_t>(WALK_SLOTS_MAX, WALK_SCRATCH_BYTES / std::max<uint64_t>(per_slot, 1));
    max_slots = std::max<uint64_t>(max_slots & ~3ull, 4);
    uint64_t want = 64;  // sizes come in a few classes so that scratches are reused: 64, 512, max
    if (walks > 64) want = 512;
    if (walks > 512) want = max_slots;
    want = std::min<uint64_t>(want, max_slots);
    const uint64_t need = want * per_slot;
    std::vector<std::unique_ptr<WalkScratch>> doomed;  // freed after the lock is dropped (the destructor syncs a stream)
    {
        std::unique_lock<std::mutex> lk(pool_mu_);
        for (;;) {
            WalkScratch* best = nullptr;
            size_t best_i = 0;
            for (size_t i = 0; i < pool_free_.size(); ++i) {
                WalkScratch* c = pool_free_[i];
                if (c->n_slots >= want && (!best || c->n_slots < best->n_slots)) {
                    best = c;
                    best_i = i;
                }
            }
            const size_t made = pool_all_.size() + pool_pending_;
            const bool room = made < WALK_POOL_MAX && (pool_bytes_ + need <= WALK_POOL_BYTES || made == 0);
            if (!best && !room && !pool_free_.empty()) {
                // over the count or the byte budget: a smaller idle scratch serves (the kernel queues walks on its
                // slots) unless freeing idle ones makes room for the size asked for
                uint64_t idle = 0;
                for (WalkScratch* c : pool_free_) idle += (uint64_t)c->n_slots * (c->words + c->log_cap) * sizeof(uint32_t);
                if (pool_all_.size() + pool_pending_ <= WALK_POOL_MAX && pool_bytes_ - idle + need <= WALK_POOL_BYTES) {
                    while (!pool_free_.empty() && (pool_bytes_ + need > WALK_POOL_BYTES || pool_all_.size() + pool_pending_ >= WALK_POOL_MAX)) {
                        size_t big = 0;
                        for (size_t i = 1; i < pool_free_.size(); ++i)
                            if (pool_free_[i]->n_slots > pool_free_[big]->n_slots) big = i;
                        WalkScratch* v = pool_free_[big];
                        pool_free_.erase(pool_free_.begin() + (long)big);
                        pool_bytes_ -= (uint64_t)v->n_slots * (v->words + v->log_cap) * sizeof(uint32_t);
                        for (size_t i = 0; i < pool_all_.size(); ++i)
                            if (pool_all_[i].get() == v) {
                                doomed.push_back(std::move(pool_all_[i]));
                                pool_all_.erase(pool_all_.begin() + (long)i);
                                break;
                            }
                    }
                    continue;  // re-evaluate: there is room now
                }
                best = pool_free_.back();
                best_i = pool_free_.size() - 1;
            }
            if (best) {
                pool_free_.erase(pool_free_.begin() + (long)best_i);
                lk.unlock();
                doomed.clear();
                return best;
            }
            if (room) break;
            pool_cv_.wait(lk);
        }
        pool_pending_ += 1;  // the slot and its bytes are reserved; the allocation runs unlocked
        pool_bytes_ += need;
    }
    doomed.clear();
    auto unreserve = [&]() {
        {
            std::lock_guard<std::mutex> lk(pool_mu_);
            pool_pending_ -= 1;
            pool_bytes_ -= need;
        }
        pool_cv_.notify_all();
    };
    std::unique_ptr<WalkScratch> ws(new (std::nothrow) WalkScratch());
    if (!ws) {
        unreserve();
        set_last_error("host allocation failed");
        return nullptr;
    }
    ws->device = device_;
    ws->words = words;
    ws->log_cap = WALK_LOG_CAP;
    ws->n_slots = (uint32_t)want;
    hipError_t e = hipSetDevice(device_);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ws->d_bits), (size_t)want * words * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ws->d_log), (size_t)want * WALK_LOG_CAP * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(ws->d_bits, 0, (size_t)want * words * sizeof(uint32_t), ws->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ws->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        unreserve();
        set_last_error(std::string("walk scratch allocation failed: ") + hipGetErrorString(e));
        return nullptr;  // ~WalkScratch frees what was made
    }
    WalkScratch* raw = ws.get();
    {
        std::lock_guard<std::mutex> lk(pool_mu_);
        pool_pending_ -= 1;
        pool_all_.push_back(std::move(ws));
    }
    return raw;
}

void HnswIndex::release_scratch(WalkScratch* ws) const
{
    if (!ws) return;
    {
        std::lock_guard<std::mutex> lk(pool_mu_);
        pool_free_.push_back(ws);
    }
    pool_cv_.notify_one();
}

void HnswIndex::drop_scratch_pool()
{
    std::lock_guard<std::mutex> lk(pool_mu_);
    pool_free_.clear();
    pool_all_.clear();  // bitmaps are sized by the graph's capacity: rebuilt on demand
    pool_bytes_ = 0;
}

HnswGraphView HnswIndex::view(const WalkScratch* ws) const
{
    HnswGraphView g;
    g.master = store_->device_master();
    g.slab = store_->device_slab();
    g.inv_norm = store_->device_inv_norm();
    g.ld = store_->slab_ld();
    g.dim = (uint32_t)dim_;
    g.m = params_.m;
    g.m0 = params_.m0;
    g.nbr0 = d_nbr0_;
    g.dist0 = d_dist0_;
    g.cnt0 = d_cnt0_;
    g.level = d_level_;
    g.upper_off = d_upper_off_;
    g.nbrU = d_nbrU_;
    g.distU = d_distU_;
    g.cntU = d_cntU_;
    g.lock = d_lock_;
    g.indeg0 = d_indeg0_;
    g.vis_bits = ws ? ws->d_bits : nullptr;
    g.vis_log = ws ? ws->d_log : nullptr;
    g.vis_words = ws ? ws->words : 0;
    g.vis_log_cap = ws ? ws->log_cap : 0;
    g.n_slots = ws ? ws->n_slots : 0;
    g.cap = g_cap_;
    g.node_id = d_node_id_;
    g.live = d_live_;
    return g;
}

int HnswIndex::ensure_graph(uint64_t nodes, uint64_t upper_slots)
{
    if (nodes > g_cap_) {
        if (nodes >= 0x7FFFFFF0ull) {
            set_last_error("HNSW node count exceeds 2^31");
            return ERR_INVALID_ARG;
        }
        const uint64_t nc = std::max<uint64_t>({nodes, g_cap_ * 2, 1024});
        VL_TRY(regrow(&d_nbr0_, g_cap_ * params_.m0, nc * params_.m0, stream_, false));
        VL_TRY(regrow(&d_dist0_, g_cap_ * params_.m0, nc * params_.m0, stream_, false));
        VL_TRY(regrow(&d_cnt0_, g_cap_, nc, stream_, true));
        VL_TRY(regrow(&d_level_, g_cap_, nc, stream_, true));
        VL_TRY(regrow(&d_upper_off_, g_cap_, nc, stream_, true));
        VL_TRY(regrow(&d_lock_, g_cap_, nc, stream_, true));
        VL_TRY(regrow(&d_indeg0_, g_cap_, nc, stream_, true));
        VL_TRY(regrow(&d_node_id_, g_cap_, nc, stream_, true));
        VL_TRY(regrow(&d_live_, g_cap_, nc, stream_, true));
        drop_scratch_pool();  // the walks' bitmaps are sized by the capacity
        g_cap_ = nc;
    }
    if (upper_slots > u_cap_) {
        const uint64_t nc = std::max<uint64_t>({upper_slots, u_cap_ * 2, 256});
        VL_TRY(regrow(&d_nbrU_, u_cap_ * params_.m, nc * params_.m, stream_, false));
        VL_TRY(regrow(&d_distU_, u_cap_ * params_.m, nc * params_.m, stream_, false));
        VL_TRY(regrow(&d_cntU_, u_cap_, nc, stream_, true));
        u_cap_ = nc;
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------
// add / delete (src/index/hnsw.rs:363-414)
// ---------------------------------------------------------------------------------------------
int HnswIndex::add(uint64_t id, const double* values, uint64_t len)
{
    if (len != dim_) {  // :364-366
        set_dim_mismatch(dim_, len);
        set_last_error("Vector dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(len));
        return ERR_DIM_MISMATCH;
    }
    return add_bulk(&id, values, 1, false);
}

int HnswIndex::add_bulk(const uint64_t* ids, const double* values, uint64_t n, bool values_on_device)
{
    if (n == 0) return OK;
    if (!ids || !values) return ERR_INVALID_ARG;
    std::unique_lock<RwLock> lk(mu_);
    VL_HIP(hipSetDevice(device_));

    // n sequential add() calls: stop at the first id that already exists (:368-370)
    uint64_t n_take = n;
    int rc_after = OK;
    {
        std::unordered_map<uint64_t, uint32_t> seen;
        for (uint64_t i = 0; i < n; ++i) {
            if (id_to_node_.count(ids[i]) || seen.count(ids[i])) {
                n_take = i;
                rc_after = ERR_DUP_ID;
                set_last_error("Vector ID " + std::to_string(ids[i]) + " already exists");
                break;
            }
            seen.emplace(ids[i], 0);
        }
    }
    if (n_take == 0) return rc_after;

    const uint64_t first = n_nodes_;
    if (store_->len() != first) {
        set_last_error("row store and graph out of step");
        return ERR_DEVICE;
    }
    // node levels and upper-layer slots
    std::vector<uint8_t> lv(n_take);
    std::vector<uint32_t> off(n_take);
    uint64_t upper = n_upper_;
    for (uint64_t i = 0; i < n_take; ++i) {
        const int l = draw_level(params_.seed, first + i, params_.m);
        lv[i] = (uint8_t)l;
        off[i] = (uint32_t)upper;
        upper += (uint64_t)l;
    }
    // Everything that can run out of memory comes BEFORE the row store is touched: if growing the graph arrays or
    // the visited stamps fails, nothing has been appended and the index still answers and accepts adds.  What is
    // left after the rows are in (copies and launches on an allocated graph) fails only with the device itself.
    VL_TRY(ensure_graph(first + n_take, upper));
    VL_TRY(store_->reserve(first + n_take));
    VL_TRY(store_->add_bulk(ids, values, n_take, /*validate=*/false, values_on_device));
    VL_HIP(hipMemcpyAsync(d_level_ + first, lv.data(), n_take, hipMemcpyHostToDevice, stream_));
    VL_HIP(hipMemcpyAsync(d_upper_off_ + first, off.data(), n_take * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
    VL_HIP(hipStreamSynchronize(stream_));  // lv/off are stack-owned
    level_.insert(level_.end(), lv.begin(), lv.end());
    upper_off_.insert(upper_off_.end(), off.begin(), off.end());
    n_upper_ = upper;

    // batched insertion: every batch walks the graph of all earlier nodes (phase A), then links
    // itself in (phase B).  Batches stay small against the graph they search (<= 1/8 of it).
    {
        WalkScratch* ws = acquire_scratch(std::min<uint64_t>(n_take, INSERT_BATCH_MAX));
        if (!ws) return ERR_OOM;
        struct Back {  // the scratch goes back whichever way this block is left
            const HnswIndex* h;
            WalkScratch* w;
            ~Back()
            {
                (void)hipStreamSynchronize(w->stream);  // idle before it goes back, also on an error path
                h->release_scratch(w);
            }
        } back{this, ws};
        hipStream_t bs = ws->stream;
        VL_HIP(hipStreamSynchronize(stream_));  // level / offset uploads above ran on the index's own stream
        uint64_t pos = first;
        const uint64_t end = first + n_take;
        const char* sf = getenv("VL_HNSW_SELECT");  // tuning: 0 closest-M, 1 heuristic, 3 heuristic + back-fill, 7 (default) + same-batch predecessors
        const uint32_t select_flags = sf && *sf ? (uint32_t)atoi(sf) : 7u;  // + 4: same-batch predecessors join the beam
        const char* bd = getenv("VL_HNSW_BATCH_DIV");
        const uint64_t batch_div = bd && *bd ? (uint64_t)std::max(1, atoi(bd)) : 8;
        if (entry_ == HNSW_NONE) {
            entry_ = (uint32_t)pos;
            max_level_ = level_[pos];
            ++pos;
        }
        while (pos < end) {
            uint64_t b = std::max<uint64_t>(1, pos / batch_div);
            b = std::min<uint64_t>({b, (uint64_t)INSERT_BATCH_MAX, end - pos});
            const HnswGraphView g = view(ws);
            VL_HIP(launch_hnsw_insert_search(bs, metric_, g, (uint32_t)pos, (uint32_t)b, params_.ef_construction,
                                             entry_, max_level_, select_flags));
            VL_HIP(launch_hnsw_insert_link(bs, g, (uint32_t)pos, (uint32_t)b, max_level_));
            for (uint64_t i = pos; i < pos + b; ++i) {
                if ((int)level_[i] > max_level_) {  // a taller node becomes the entry point
                    max_level_ = level_[i];
                    entry_ = (uint32_t)i;
                }
            }
            pos += b;
        }
        VL_HIP(hipStreamSynchronize(bs));
    }

    // device copies used by the query kernel's result stage
    VL_HIP(hipMemcpyAsync(d_node_id_ + first, ids, n_take * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
    VL_HIP(hipMemsetAsync(d_live_ + first, 1, n_take, stream_));
    VL_HIP(hipStreamSynchronize(stream_));
    node_id_.insert(node_id_.end(), ids, ids + n_take);
    live_.insert(live_.end(), n_take, 1);
    for (uint64_t i = 0; i < n_take; ++i) id_to_node_[ids[i]] = (uint32_t)(first + i);
    n_nodes_ += n_take;
    live_count_ += n_take;
    return rc_after;
}

int HnswIndex::remove(uint64_t id)
{
    std::unique_lock<RwLock> lk(mu_);
    auto it = id_to_node_.find(id);
    if (it == id_to_node_.end()) {  // :401-403
        set_last_error("Vector ID " + std::to_string(id) + " does not exist");
        return ERR_NOT_FOUND;
    }
    // tombstone: the node stays in the graph and is still walked (:407); host state follows the device's
    VL_HIP(hipSetDevice(device_));
    VL_HIP(hipMemsetAsync(d_live_ + it->second, 0, 1, stream_));
    VL_HIP(hipStreamSynchronize(stream_));
    live_[it->second] = 0;
    id_to_node_.erase(it);
    --live_count_;
    return OK;
}

int HnswIndex::get_vector(uint64_t id, double* out) const
{
    std::shared_lock<RwLock> lk(mu_);
    auto it = id_to_node_.find(id);
    if (it == id_to_node_.end()) return ERR_NOT_FOUND;
    VL_HIP(hipSetDevice(device_));
    VL_HIP(hipMemcpy(out, store_->device_master() + (uint64_t)it->second * dim_, dim_ * sizeof(double),
                     hipMemcpyDeviceToHost));
    return OK;
}

void HnswIndex::walk_stats(uint64_t* queries, uint64_t* distance_evals) const
{
    if (queries) *queries = stat_queries_.load();
    if (distance_evals) {
        unsigned long long dev = 0;
        // walks in flight add to the counter with atomics; the copy reads whatever has landed
        if (d_stat_evals_ && hipSetDevice(device_) == hipSuccess)
            (void)hipMemcpy(&dev, d_stat_evals_, sizeof dev, hipMemcpyDeviceToHost);
        *distance_evals = stat_evals_.load() + dev;
    }
}

int HnswIndex::clone(HnswIndex** out) const
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    std::shared_lock<RwLock> lk(mu_);
    VL_HIP(hipSetDevice(device_));
    std::unique_ptr<HnswIndex> c(new HnswIndex(dim_, metric_, params_, device_));
    GpuFlatIndex* st = nullptr;
    VL_TRY(store_->clone(&st));
    c->store_.reset(st);
    VL_HIP(hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking));
    VL_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_stat_evals_), sizeof(unsigned long long)));
    VL_HIP(hipMemset(c->d_stat_evals_, 0, sizeof(unsigned long long)));
    if (n_nodes_) {
        VL_TRY(c->ensure_graph(n_nodes_, n_upper_));
        auto copy = [&](void* dst, const void* src, size_t bytes) -> int {
            if (bytes) VL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream_));
            return OK;
        };
        VL_TRY(copy(c->d_nbr0_, d_nbr0_, n_nodes_ * params_.m0 * sizeof(uint32_t)));
        VL_TRY(copy(c->d_dist0_, d_dist0_, n_nodes_ * params_.m0 * sizeof(unsigned long long)));
        VL_TRY(copy(c->d_cnt0_, d_cnt0_, n_nodes_ * sizeof(uint32_t)));
        VL_TRY(copy(c->d_level_, d_level_, n_nodes_ * sizeof(uint8_t)));
        VL_TRY(copy(c->d_upper_off_, d_upper_off_, n_nodes_ * sizeof(uint32_t)));
        VL_TRY(copy(c->d_nbrU_, d_nbrU_, n_upper_ * params_.m * sizeof(uint32_t)));
        VL_TRY(copy(c->d_distU_, d_distU_, n_upper_ * params_.m * sizeof(unsigned long long)));
        VL_TRY(copy(c->d_cntU_, d_cntU_, n_upper_ * sizeof(uint32_t)));
        VL_TRY(copy(c->d_indeg0_, d_indeg0_, n_nodes_ * sizeof(uint32_t)));
        VL_TRY(copy(c->d_node_id_, d_node_id_, n_nodes_ * sizeof(unsigned long long)));
        VL_TRY(copy(c->d_live_, d_live_, n_nodes_ * sizeof(uint8_t)));
        VL_HIP(hipStreamSynchronize(c->stream_));
    }
    c->level_ = level_;
    c->upper_off_ = upper_off_;
    c->n_upper_ = n_upper_;
    c->entry_ = entry_;
    c->max_level_ = max_level_;
    c->n_nodes_ = n_nodes_;
    c->id_to_node_ = id_to_node_;
    c->node_id_ = node_id_;
    c->live_ = live_;
    c->live_count_ = live_count_;
    *out = c.release();
    return OK;
}

int HnswIndex::export_rows(uint64_t* out_ids, double* out_values) const
{
    std::shared_lock<RwLock> lk(mu_);
    if (live_count_ == 0) return OK;
    if (!out_ids || !out_values) return ERR_INVALID_ARG;
    VL_HIP(hipSetDevice(device_));
    uint64_t w = 0;
    for (uint64_t node = 0; node < n_nodes_;) {  // copy maximal runs of live nodes
        if (!live_[node]) {
            ++node;
            continue;
        }
        uint64_t e = node;
        while (e < n_nodes_ && live_[e]) ++e;
        VL_HIP(hipMemcpy(out_values + w * dim_, store_->device_master() + node * dim_, (e - node) * dim_ * sizeof(double),
                         hipMemcpyDeviceToHost));
        for (uint64_t i = node; i < e; ++i) out_ids[w++] = node_id_[i];
        node = e;
    }
    return OK;
}

void HnswIndex::graph_info(uint64_t* n_nodes, uint32_t* entry, int* max_level, uint32_t* m, uint32_t* m0,
                           uint64_t* upper_slots) const
{
    std::shared_lock<RwLock> lk(mu_);
    if (n_nodes) *n_nodes = n_nodes_;
    if (entry) *entry = entry_;
    if (max_level) *max_level = max_level_;
    if (m) *m = params_.m;
    if (m0) *m0 = params_.m0;
    if (upper_slots) *upper_slots = n_upper_;
}

int HnswIndex::graph_export(uint8_t* level, uint32_t* upper_off, uint32_t* cnt0, uint32_t* nbr0, uint32_t* cntU,
                            uint32_t* nbrU, uint64_t* node_ids, uint8_t* live, double* rows) const
{
    std::shared_lock<RwLock> lk(mu_);
    const uint64_t n = n_nodes_;
    if (n == 0) return OK;
    VL_HIP(hipSetDevice(device_));
    if (level) std::memcpy(level, level_.data(), n);
    if (upper_off) std::memcpy(upper_off, upper_off_.data(), n * sizeof(uint32_t));
    if (node_ids) std::memcpy(node_ids, node_id_.data(), n * sizeof(uint64_t));
    if (live) std::memcpy(live, live_.data(), n);
    if (cnt0) VL_HIP(hipMemcpy(cnt0, d_cnt0_, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (nbr0) VL_HIP(hipMemcpy(nbr0, d_nbr0_, n * params_.m0 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (cntU && n_upper_) VL_HIP(hipMemcpy(cntU, d_cntU_, n_upper_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (nbrU && n_upper_) VL_HIP(hipMemcpy(nbrU, d_nbrU_, n_upper_ * params_.m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (rows) VL_HIP(hipMemcpy(rows, store_->device_master(), n * dim_ * sizeof(double), hipMemcpyDeviceToHost));
    return OK;
}

int HnswIndex::max_id(uint64_t* out) const
{
    std::shared_lock<RwLock> lk(mu_);
    if (id_to_node_.empty()) return ERR_NOT_FOUND;
    uint64_t m = 0;
    for (const auto& kv : id_to_node_) m = std::max(m, kv.first);
    *out = m;
    return OK;
}

// ---------------------------------------------------------------------------------------------
// search (src/index/hnsw.rs:415-496)
// ---------------------------------------------------------------------------------------------
int HnswIndex::ensure_io(WalkScratch* ws, uint64_t nq, uint64_t k) const
{
    const uint64_t qn = nq * dim_;
    if (qn > ws->q_cap) {
        if (ws->d_q) (void)hipFree(ws->d_q);
        ws->d_q = nullptr;
        ws->q_cap = 0;
        VL_HIP(hipMalloc(reinterpret_cast<void**>(&ws->d_q), qn * sizeof(double)));
        ws->q_cap = qn;
    }
    const uint64_t words = nq * (2 * k + 1);
    if (words > ws->out_cap) {
        if (ws->d_out) (void)hipFree(ws->d_out);
        if (ws->h_out) (void)hipHostFree(ws->h_out);
        ws->d_out = nullptr;
        ws->h_out = nullptr;
        ws->out_cap = 0;
        VL_HIP(hipMalloc(reinterpret_cast<void**>(&ws->d_out), words * sizeof(unsigned long long)));
        VL_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_out), words * sizeof(unsigned long long), hipHostMallocDefault));
        ws->out_cap = words;
    }
    return OK;
}

// the caller holds mu_ (shared): search_batch's queries whose beam would exceed the walk kernel's
int HnswIndex::search_exact_fallback(const double* query, uint64_t k, uint64_t* out_ids, double* out_scores,
                                     uint64_t* out_n, uint64_t out_limit) const
{
    *out_n = 0;
    if (live_count_ == 0) return OK;
    // every tombstoned node could sit in front of a live one: ask for that many more
    const uint64_t want = std::min<uint64_t>(n_nodes_, k + (n_nodes_ - live_count_));
    std::vector<uint64_t> pos(want);
    std::vector<double> flat_scores(want);
    uint64_t got = 0;
    // the flat metric score is a decreasing function of Metric::distance's f64 value for all four metrics
    VL_TRY(store_->search(query, dim_, want, metric_, pos.data(), nullptr, flat_scores.data(), &got));
    std::vector<uint64_t> dist(got);
    VL_TRY(store_->hnsw_distances(query, dim_, metric_, pos.data(), got, dist.data()));
    struct Res {
        uint64_t id;
        double score;
    };
    std::vector<Res> res;
    stat_queries_.fetch_add(1, std::memory_order_relaxed);
    stat_evals_.fetch_add(n_nodes_ + got, std::memory_order_relaxed);
    const uint64_t max_candidates = std::min<uint64_t>(k, live_count_);
    for (uint64_t i = 0; i < got && res.size() < max_candidates; ++i) {
        if (pos[i] >= n_nodes_ || !live_[pos[i]]) continue;
        res.push_back({node_id_[pos[i]], hnsw_score(dist[i], metric_)});
    }
    std::stable_sort(res.begin(), res.end(), [](const Res& a, const Res& b) { return a.score > b.score; });  // :493
    const uint64_t n_out = std::min<uint64_t>(res.size(), out_limit);
    for (uint64_t i = 0; i < n_out; ++i) {
        out_ids[i] = res[i].id;
        out_scores[i] = res[i].score;
    }
    *out_n = n_out;
    return OK;
}

int HnswIndex::search(const double* query, uint64_t q_len, uint64_t k, int metric, uint32_t ef, uint64_t* out_ids,
                      double* out_scores, uint64_t* out_n, uint64_t out_stride) const
{
    const uint64_t out_limit = out_stride ? out_stride : k;
    // Walk launches of different callers run side by side (each borrows its own WalkScratch); a launch of one query
    // still leaves most of the chip idle, so callers that arrive together can also share a launch.  Everything that can fail without walking is
    // settled on the calling thread; compatible requests (same k and ef; the metric is the index's) are walked
    // by one search_batch(), each caller receiving exactly what its own search_batch(nq = 1) would return.
    if (!co_.enabled() || !out_n || q_len != dim_ || metric != metric_ || k == 0 || !query || !out_ids || !out_scores ||
        k > (uint64_t)HNSW_MAX_EF || ef > (uint32_t)HNSW_MAX_EF)
        return search_batch(query, 1, q_len, k, metric, ef, out_ids, out_scores, out_n, out_stride);
    CoalesceReq r{query, k, ef, out_ids, out_scores, out_n, out_limit};
    co_.run(
        r, [](const CoalesceReq& a, const CoalesceReq& o) { return a.k == o.k && a.ef == o.ef; },
        [this](std::vector<CoalesceReq*>& batch) {
            const uint64_t nq = batch.size(), kk = batch[0]->k;
            int rc = OK;
            if (nq > 1) try {
                std::vector<double> q(nq * dim_);
                for (uint64_t i = 0; i < nq; ++i) std::memcpy(q.data() + i * dim_, batch[i]->query, dim_ * sizeof(double));
                std::vector<uint64_t> ids(nq * kk), cnt(nq);
                std::vector<double> scores(nq * kk);
                rc = search_batch(q.data(), nq, dim_, kk, metric_, batch[0]->ef, ids.data(), scores.data(), cnt.data());
                if (rc == OK) {
                    for (uint64_t i = 0; i < nq; ++i) {
                        CoalesceReq* o = batch[i];
                        const uint64_t m = std::min<uint64_t>(cnt[i], o->out_limit);
                        for (uint64_t j = 0; j < m; ++j) {
                            o->out_ids[j] = ids[i * kk + j];
                            o->out_scores[j] = scores[i * kk + j];
                        }
                        *o->out_n = m;
                        o->rc = OK;
                    }
                    return;
                }
            } catch (...) {  // host allocation failed: the callers are answered one by one below
            }
            for (CoalesceReq* o : batch) {  // alone, or the batch failed as a whole: per-caller status
                o->rc = search_batch(o->query, 1, dim_, o->k, metric_, o->ef, o->out_ids, o->out_scores, o->out_n, o->out_limit);
                if (o->rc != OK) o->err = last_error();
            }
        });
    if (r.rc != OK) set_last_error(r.err);
    return r.rc;
}

int HnswIndex::search_batch(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric, uint32_t ef,
                            uint64_t* out_ids, double* out_scores, uint64_t* out_n, uint64_t out_stride) const
{
    const uint64_t os = out_stride ? out_stride : k;  // the caller's row stride = the most it takes per query
    if (!out_n && nq) return ERR_INVALID_ARG;
    for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
    if (metric < 0 || metric > 3) return ERR_INVALID_ARG;
    if (ef > (uint32_t)HNSW_MAX_EF) {  // our own knob (the reference has none): refused, never silently narrowed
        set_last_error("HNSW search: ef = " + std::to_string(ef) + " exceeds the walk's beam ceiling of " + std::to_string(HNSW_MAX_EF));
        return ERR_INVALID_ARG;
    }
    std::shared_lock<RwLock> lk(mu_);
    if (q_len != dim_) {  // :416-421, checked even when the index is empty
        set_dim_mismatch(dim_, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(q_len));
        return ERR_DIM_MISMATCH;
    }
    if (metric != metric_) {  // :425-430
        set_last_error("Metric mismatch: the HNSW index was built for metric " + std::to_string(metric_) +
                       ", search requested " + std::to_string(metric));
        return ERR_METRIC_MISMATCH;
    }
    if (nq == 0 || live_count_ == 0) return OK;                      // :432-434
    const uint64_t max_candidates = std::min<uint64_t>(k, live_count_);  // :437
    if (max_candidates == 0) return OK;
    if (!queries || !out_ids || !out_scores) return ERR_INVALID_ARG;
    // hnsw.nearest(&q, ef = max_candidates, ..) (:454); an explicit ef widens the beam, never narrows it
    // ef == 0 (the trait's search): the reference hands ef = min(k, len) to the crate's walk, and so does this one by
    // default (min_beam_ = 0).  A caller that opted in (vl_index_hnsw_set_min_beam) keeps a beam of at least min_beam_
    // entries and gets the best min(k, len) of it -- never fewer results; with a floor of 32 on embedding-like data
    // recall@10 is 0.96 instead of 0.78 at N = 1 M for about the same batch throughput.
    uint64_t ef_walk = ef ? std::max<uint64_t>(max_candidates, ef) : std::max<uint64_t>(max_candidates, min_beam_.load());
    if (ef_walk > (uint64_t)HNSW_MAX_EF) {
        // Only min(k, len) can have pushed the beam past the walk kernel's ceiling (an explicit ef beyond it was
        // refused above, the opt-in floor is capped at it): answer from the row store with the exact scan instead --
        // the true nearest neighbours in Metric::distance order, i.e. what a perfect walk would return,
        // post-processed exactly like a walk's beam below.
        lk.unlock();
        for (uint64_t qi = 0; qi < nq; ++qi)
            VL_TRY(search_exact_fallback(queries + qi * dim_, k, out_ids + qi * os, out_scores + qi * os, out_n + qi, os));
        return OK;
    }

    VL_HIP(hipSetDevice(device_));
    // Device and pinned scratch, and the kernel's output stride, are sized by the beam (kd <= HNSW_MAX_EF entries
    // per query), never by the caller's k: a huge k on a small index must return min(k, len) results, not OOM, and
    // nq * (2k + 1) must not be able to wrap.  k is used only for the caller's own [nq, k] row stride.
    const uint64_t kd = max_candidates;
    WalkScratch* ws = acquire_scratch(nq);
    if (!ws) return ERR_OOM;
    struct Back {
        const HnswIndex* h;
        WalkScratch* w;
        ~Back()
        {
            (void)hipStreamSynchronize(w->stream);
            h->release_scratch(w);
        }
    } back{this, ws};
    hipStream_t st = ws->stream;
    {
        const int rc = ensure_io(ws, nq, kd);
        if (rc != OK) return rc;
    }
    // straight from the caller's buffer (the runtime stages pageable memory itself; an extra copy into a pinned
    // staging area cost 12 % of a 2000-query batch); the stream is synchronised before this call returns
    VL_HIP(hipMemcpyAsync(ws->d_q, queries, nq * dim_ * sizeof(double), hipMemcpyHostToDevice, st));
    const HnswGraphView g = view(ws);
    // The kernel finishes each walk the way HNSWIndex::search does (:468-495): beam in (distance, node) order,
    // the closest max_candidates taken, tombstones among them dropped, distances converted to scores.  Like the
    // reference, tombstones can make fewer than k results come back; only a caller that named its own ef
    // (vl_index_search_ef) gets the freed slots refilled from the rest of the beam.
    unsigned long long* d_ids = ws->d_out;
    double* d_scores = reinterpret_cast<double*>(ws->d_out + nq * kd);
    unsigned long long* d_n = ws->d_out + 2 * nq * kd;
    hipError_t le = launch_hnsw_search(st, metric_, g, ws->d_q, (uint32_t)nq, (uint32_t)ef_walk, entry_, max_level_,
                                       (uint32_t)max_candidates, ef ? 1u : 0u, (uint32_t)kd, d_ids, d_scores, d_n, d_stat_evals_);
    if (le == hipSuccess) le = hipMemcpyAsync(ws->h_out, ws->d_out, nq * (2 * kd + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    const hipError_t se = hipStreamSynchronize(st);  // always: the scratch must be idle before it goes back to the pool
    if (le == hipSuccess) le = se;
    if (le != hipSuccess) {
        set_last_error(std::string("HNSW walk: ") + hipGetErrorString(le));
        return ERR_DEVICE;
    }
    stat_queries_.fetch_add(nq, std::memory_order_relaxed);
    const unsigned long long* h_n = ws->h_out + 2 * nq * kd;
    for (uint64_t qi = 0; qi < nq; ++qi) {
        const uint64_t m = std::min<uint64_t>({(uint64_t)h_n[qi], kd, os});
        std::memcpy(out_ids + qi * os, ws->h_out + qi * kd, m * sizeof(uint64_t));
        std::memcpy(out_scores + qi * os, ws->h_out + nq * kd + qi * kd, m * sizeof(double));
        out_n[qi] = m;
    }
    return OK;
}

}  // namespace vl
