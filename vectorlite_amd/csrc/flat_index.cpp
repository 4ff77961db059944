// flat_index.cpp -- host logic of the GPU flat index (see flat_index.hpp).
//
// Mirrors `impl VectorIndex for FlatIndex` (reference src/index/flat.rs:82-135): same argument
// meaning, same error behaviour, same result order.  All arithmetic on vectors runs in the HIP
// kernels (kernels.hip); the host only stages queries, keeps the position -> id table and maps
// positions back to ids for the k winners (the reference re-attaches text/metadata the same way,
// src/index/flat.rs:106-114).  There is no CPU compute fallback.
#include "flat_index.hpp"
#include "shard.hpp"

#include <chrono>

#include <algorithm>
#include <map>
#include <type_traits>
#include <cmath>
#include <cstring>

namespace vl {

// ---------------------------------------------------------------------------------------------
// thread-local diagnostics
// ---------------------------------------------------------------------------------------------
namespace {
thread_local std::string t_last_error;
thread_local uint64_t t_dim_expected = 0, t_dim_actual = 0;
thread_local int t_last_path = PATH_NONE;
}  // namespace

void set_last_error(const std::string& msg) { t_last_error = msg; }
const char* last_error() { return t_last_error.c_str(); }
void set_dim_mismatch(uint64_t expected, uint64_t actual)
{
    t_dim_expected = expected;
    t_dim_actual = actual;
}
void get_dim_mismatch(uint64_t* expected, uint64_t* actual)
{
    if (expected) *expected = t_dim_expected;
    if (actual) *actual = t_dim_actual;
}
void set_last_path(int p) { t_last_path = p; }
int last_path() { return t_last_path; }

#define VL_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                   \
            (void)hipGetLastError(); /* a failed call (hipMalloc out of memory ...) leaves the thread's sticky error behind: the next launch's hipGetLastError() must not report it */ \
            return (e_ == hipErrorOutOfMemory) ? (int)ERR_OOM : (int)ERR_DEVICE;                 \
        }                                                                                        \
    } while (0)

#define VL_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != OK) return rc_; \
    } while (0)

namespace {
constexpr double DOMAIN_MAX_ABS = 1099511627776.0;          // 2^40
constexpr double DOMAIN_MIN_NORM = 9.094947017729282e-13;   // 2^-40
constexpr size_t BOUNCE_BYTES = 64ull << 20;

template <typename T>
int dev_alloc(T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) return OK;
    VL_HIP(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return OK;
}
template <typename T>
int pinned_alloc(T** p, size_t count)
{
    *p = nullptr;
    VL_HIP(hipHostMalloc(reinterpret_cast<void**>(p), count * sizeof(T), hipHostMallocDefault));
    return OK;
}
}  // namespace

Workspace::~Workspace()
{
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    void* dev[] = {d_q64, d_partials, d_partials64, d_result, d_nan, d_scores, d_okeys,
                   d_opos, d_out_pos, d_out_scores, d_positions, d_dists};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    void* host[] = {h_q64, h_result, h_nan, mf_h_q64, mf_h_result, mf_h_dom, k3_h_q64, k3_h_result};
    for (void* p : host)
        if (p) (void)hipHostFree(p);
    void* mfd[] = {mf.q_bf16, mf.gmax, mf.thr, mf.cand, mf.cnt, mf_d_q64, mf_lists, mf_scores, k3_d_q64};
    for (void* p : mfd)
        if (p) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (hipEvent_t e : mf_ev_done)
        if (e) (void)hipEventDestroy(e);
    if (mf_ev_h2d) (void)hipEventDestroy(mf_ev_h2d);
    if (stream) (void)hipStreamDestroy(stream);
}

// ---------------------------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------------------------
GpuFlatIndex::GpuFlatIndex(uint64_t dim, int device)
    : dim_(dim), ld_((uint32_t)((dim + 3) & ~3ull)), device_(device)
{
    ws_pool_ = attach_pool(device, dim);
}

namespace {
// Copies a query into its staging slot and decides whether it lies in the fast-path domain (finite, |v| <= 2^40,
// norm 0 or >= 2^-40).  The norm only feeds the error bound and that domain test (the kernels recompute every score
// from the values), so it is summed with four partial accumulators the compiler can vectorise; a query outside the
// domain is staged as zeros (keeps the f32 / bf16 scans finite) with norm 0 and answered on the exact path.
bool stage_query(const double* q, double* dst, uint64_t dim, double* norm_out)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
    uint64_t i = 0;
    for (; i + 4 <= dim; i += 4) {
        const double a = q[i], b = q[i + 1], c = q[i + 2], d = q[i + 3];
        dst[i] = a;
        dst[i + 1] = b;
        dst[i + 2] = c;
        dst[i + 3] = d;
        s0 += a * a;
        s1 += b * b;
        s2 += c * c;
        s3 += d * d;
        const double fa = std::fabs(a), fb = std::fabs(b), fc = std::fabs(c), fd = std::fabs(d);
        m0 = fa > m0 ? fa : m0;
        m1 = fb > m1 ? fb : m1;
        m2 = fc > m2 ? fc : m2;
        m3 = fd > m3 ? fd : m3;
    }
    for (; i < dim; ++i) {
        const double a = q[i];
        dst[i] = a;
        s0 += a * a;
        const double fa = std::fabs(a);
        m0 = fa > m0 ? fa : m0;
    }
    const double qq = (s0 + s1) + (s2 + s3);
    const double m01 = m0 > m1 ? m0 : m1, m23 = m2 > m3 ? m2 : m3;
    const double qmax = m01 > m23 ? m01 : m23;
    const double norm = std::sqrt(qq);
    // a NaN component makes qq NaN (the max ignores it); an infinity shows up in qmax (and qq)
    const bool finite = qq == qq && qmax <= 1.797693134862315708e308 && norm <= 1.797693134862315708e308;
    const bool in_domain = finite && qmax <= DOMAIN_MAX_ABS && (norm == 0.0 || norm >= DOMAIN_MIN_NORM);
    if (!in_domain) {
        for (uint64_t j = 0; j < dim; ++j) dst[j] = 0.0;
        *norm_out = 0.0;
    } else {
        *norm_out = norm;
    }
    return in_domain;
}
}  // namespace

int GpuFlatIndex::create(uint64_t dim, int device, GpuFlatIndex** out)
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    if (dim == 0 || dim > 0x0FFFFFFFull) {
        set_last_error("dimension must be in [1, 2^28)");
        return ERR_INVALID_ARG;
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        set_last_error("no HIP device visible: vectorlite_amd has no CPU fallback");
        return ERR_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        set_last_error("device ordinal out of range");
        return ERR_DEVICE;
    }
    VL_HIP(hipSetDevice(device));
    std::unique_ptr<GpuFlatIndex> idx(new GpuFlatIndex(dim, device));
    VL_HIP(hipStreamCreateWithFlags(&idx->mut_stream_, hipStreamNonBlocking));
    VL_TRY(dev_alloc(&idx->d_stats_, 1));
    VL_HIP(hipMemsetAsync(idx->d_stats_, 0, sizeof(IngestStats), idx->mut_stream_));
    VL_HIP(hipStreamSynchronize(idx->mut_stream_));
    if (const char* sf = getenv("VL_SINGLE_FILTER")) {
        if (std::strcmp(sf, "bf16") == 0) idx->set_single_filter(1);
    }
    // Concurrent single searches share slab passes by default (the reference's many-readers model, src/client.rs:398):
    // window 0, so a lone caller leads a pass of one = the plain single-search path.  VL_COALESCE=0 starts handles with it off.
    {
        const char* ce = getenv("VL_COALESCE");
        if (!(ce && ce[0] == '0')) idx->set_coalescing(COALESCE_DEFAULT_BATCH, 0);
    }
    *out = idx.release();
    return OK;
}

// ---------------------------------------------------------------------------------------------
// workspace pools, one per (device, dimension), shared by the handles (flat_index.hpp)
// ---------------------------------------------------------------------------------------------
namespace {
struct PoolKey {
    int device;
    uint64_t dim;
    bool operator<(const PoolKey& o) const { return device != o.device ? device < o.device : dim < o.dim; }
};
std::mutex g_pools_mu;
// never destroyed: a pool that outlives main() (handles leaked by a host that exits without destroying them) must not
// call into a HIP runtime that is already gone from a static destructor
auto& g_pools = *new std::map<PoolKey, std::unique_ptr<WorkspacePool>>();
}  // namespace

WorkspacePool* GpuFlatIndex::attach_pool(int device, uint64_t dim)
{
    std::lock_guard<std::mutex> g(g_pools_mu);
    auto& slot = g_pools[PoolKey{device, dim}];
    if (!slot) slot.reset(new WorkspacePool());
    slot->users += 1;
    return slot.get();
}

void GpuFlatIndex::detach_pool(int device, uint64_t dim)
{
    std::unique_ptr<WorkspacePool> last;
    {
        std::lock_guard<std::mutex> g(g_pools_mu);
        auto it = g_pools.find(PoolKey{device, dim});
        if (it == g_pools.end()) return;
        if (--it->second->users == 0) {
            last = std::move(it->second);
            g_pools.erase(it);
        }
    }
    if (last) {
        (void)hipSetDevice(device);
        last->all.clear();  // ~Workspace frees its buffers
    }
}

GpuFlatIndex::~GpuFlatIndex()
{
    (void)hipSetDevice(device_);
    if (ws_pool_) detach_pool(device_, dim_);
    if (mut_stream_) (void)hipStreamSynchronize(mut_stream_);
    void* dev[] = {d_master_, d_slab_, d_inv_norm_, d_flags_, d_stats_, d_bounce_, d_slab16_, d_sqnorm_, d_norm16_, d_slab16f_, d_ids_};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (mut_stream_) (void)hipStreamDestroy(mut_stream_);
}

uint64_t GpuFlatIndex::len() const
{
    std::shared_lock<RwLock> lk(mu_);
    return ids_.size();
}

int GpuFlatIndex::reserve(uint64_t n_rows)
{
    std::unique_lock<RwLock> lk(mu_);
    VL_HIP(hipSetDevice(device_));
    return ensure_capacity(n_rows);
}

void GpuFlatIndex::truncate(uint64_t n_rows)
{
    std::unique_lock<RwLock> lk(mu_);
    if (n_rows >= ids_.size()) return;
    for (uint64_t p = n_rows; p < row_flags_.size(); ++p)
        if (row_flags_[p] & ROW_OUT_OF_DOMAIN) --n_out_of_domain_;
    ids_.resize(n_rows);
    row_flags_.resize(n_rows);
    id_counts_valid_ = false;  // rebuilt on demand from ids_
    if (slab16_rows_ > n_rows) slab16_rows_ = n_rows;
    if (slab16f_rows_ > n_rows) slab16f_rows_ = n_rows;
    if (d_ids_rows_ > n_rows) d_ids_rows_ = n_rows;
}

int GpuFlatIndex::ensure_capacity(uint64_t rows)
{
    if (rows <= cap_) return OK;
    if (rows >= 0xFFFFFFF0ull) {
        set_last_error("row count exceeds the u32 position space");
        return ERR_INVALID_ARG;
    }
    // The bf16 copies of the slab are rebuilt on demand by the next large batch: they go first -- at tens of millions of
    // rows they are tens of gigabytes the growth below can use.
    if (d_norm16_) {
        if (d_slab16_) (void)hipFree(d_slab16_);
        if (d_slab16f_) (void)hipFree(d_slab16f_);
        (void)hipFree(d_sqnorm_);
        (void)hipFree(d_norm16_);
        d_slab16_ = nullptr;
        d_slab16f_ = nullptr;
        d_sqnorm_ = nullptr;
        d_norm16_ = nullptr;
        slab16_rows_ = 0;
        slab16f_rows_ = 0;
    }
    const uint64_t n = ids_.size();
    // One array at a time: allocate the larger one, copy, free the old one.  The transient need is the index as it stands
    // plus the LARGEST new array (the f64 master), not plus all four -- an index that was never reserve()d can grow to
    // well past half of the card (growing all four at once failed at 34 M x 384 rows with 138 GiB free).  An array that
    // has grown keeps its capacity if a later one fails: cap_ (the minimum) moves only when all four hold new_cap rows.
    auto grow = [&](auto*& ptr, uint64_t& cap_arr, uint64_t new_cap, uint64_t per_row) -> int {
        if (cap_arr >= new_cap) return OK;
        using T = std::remove_reference_t<decltype(*ptr)>;
        T* fresh = nullptr;
        VL_TRY(dev_alloc(&fresh, new_cap * per_row));
        hipError_t ce = hipSuccess;
        if (n && ptr) ce = hipMemcpyAsync(fresh, ptr, n * per_row * sizeof(T), hipMemcpyDeviceToDevice, mut_stream_);
        const hipError_t se = hipStreamSynchronize(mut_stream_);
        if (ce == hipSuccess) ce = se;
        if (ce != hipSuccess) {  // the old array stays the index; the new one is not leaked
            (void)hipGetLastError();
            (void)hipFree(fresh);
            set_last_error(std::string("growing the row store failed: ") + hipGetErrorString(ce));
            return ERR_DEVICE;
        }
        if (ptr) (void)hipFree(ptr);
        ptr = fresh;
        cap_arr = new_cap;
        return OK;
    };
    auto attempt = [&](uint64_t new_cap) -> int {
        VL_TRY(grow(d_master_, cap_master_, new_cap, dim_ ? dim_ : 1));
        VL_TRY(grow(d_slab_, cap_slab_, new_cap, ld_ ? ld_ : 4));
        VL_TRY(grow(d_inv_norm_, cap_inv_, new_cap, 1));
        VL_TRY(grow(d_flags_, cap_flags_, new_cap, 1));
        cap_ = std::min(std::min(cap_master_, cap_slab_), std::min(cap_inv_, cap_flags_));
        return OK;
    };
    const uint64_t grown = cap_ < (1ull << 20) ? cap_ * 2 : cap_ + cap_ / 2;
    const uint64_t geometric = std::max<uint64_t>({rows, grown, 1024});
    int rc = attempt(geometric);
    if (rc == ERR_OOM && rows < geometric) rc = attempt(rows);  // no room for the geometric step: exactly what was asked for
    return rc;
}

// Rows [first, first+n) of d_master_ are in place: derive slab / inv_norm / flags / stats.
int GpuFlatIndex::ingest_range(uint64_t first, uint64_t n)
{
    if (n == 0) return OK;
    VL_HIP(launch_ingest(mut_stream_, d_master_ + first * dim_, d_slab_ + first * ld_, d_inv_norm_ + first,
                         d_flags_ + first, d_stats_, n, (uint32_t)dim_, ld_));
    std::vector<uint8_t> fl(n);
    IngestStats st;
    VL_HIP(hipMemcpyAsync(fl.data(), d_flags_ + first, n, hipMemcpyDeviceToHost, mut_stream_));
    VL_HIP(hipMemcpyAsync(&st, d_stats_, sizeof(st), hipMemcpyDeviceToHost, mut_stream_));
    VL_HIP(hipStreamSynchronize(mut_stream_));
    row_flags_.resize(first + n);
    for (uint64_t i = 0; i < n; ++i) {
        row_flags_[first + i] = fl[i];
        if (fl[i] & ROW_OUT_OF_DOMAIN) ++n_out_of_domain_;
    }
    double mn;
    static_assert(sizeof(mn) == sizeof(st.max_norm_bits), "f64 bits");
    std::memcpy(&mn, &st.max_norm_bits, sizeof(mn));
    if (mn > max_row_norm_) max_row_norm_ = mn;
    return OK;
}

void GpuFlatIndex::rebuild_id_counts() const
{
    id_counts_.clear();
    id_counts_.reserve(ids_.size() * 2);
    for (uint64_t id : ids_) ++id_counts_[id];
    id_counts_valid_ = true;
}

// ---------------------------------------------------------------------------------------------
// add / delete
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::add(uint64_t id, const double* values, uint64_t len)
{
    if (len != dim_) {  // src/index/flat.rs:83-85
        set_dim_mismatch(dim_, len);
        set_last_error("Vector dimension mismatch");
        return ERR_DIM_MISMATCH;
    }
    if (!values && dim_) return ERR_INVALID_ARG;
    return add_bulk(&id, values, 1, /*validate=*/true, /*values_on_device=*/false);
}

int GpuFlatIndex::add_bulk(const uint64_t* ids, const double* values, uint64_t n, bool validate,
                           bool values_on_device, int src_device)
{
    if (n == 0) return OK;
    if (!ids || (!values && dim_)) return ERR_INVALID_ARG;
    std::unique_lock<RwLock> lk(mu_);
    VL_HIP(hipSetDevice(device_));

    uint64_t n_take = n;
    int rc_after = OK;
    if (validate) {  // n sequential add() calls: stop at the first duplicate id (src/index/flat.rs:86-88)
        if (!id_counts_valid_) rebuild_id_counts();
        for (uint64_t i = 0; i < n; ++i) {
            auto it = id_counts_.find(ids[i]);
            if (it != id_counts_.end() && it->second > 0) {
                n_take = i;
                rc_after = ERR_DUP_ID;
                set_last_error("Vector ID " + std::to_string(ids[i]) + " already exists");
                break;
            }
            ++id_counts_[ids[i]];
        }
    } else {
        id_counts_valid_ = false;  // FlatIndex::new validates nothing; counts are rebuilt on demand
    }
    if (n_take == 0) return rc_after;

    const uint64_t first = ids_.size();
    int rc = ensure_capacity(first + n_take);
    if (rc == OK && dim_) {
        hipError_t e;
        if (values_on_device && src_device >= 0 && src_device != device_)
            e = hipMemcpyPeerAsync(d_master_ + first * dim_, device_, values, src_device, n_take * dim_ * sizeof(double),
                                   mut_stream_);
        else
            e = hipMemcpyAsync(d_master_ + first * dim_, values, n_take * dim_ * sizeof(double),
                               values_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, mut_stream_);
        if (e != hipSuccess) {
            set_last_error(std::string("hipMemcpyAsync(rows): ") + hipGetErrorString(e));
            rc = ERR_DEVICE;
        }
    }
    if (rc == OK) rc = ingest_range(first, n_take);
    if (rc != OK) {
        if (validate)  // roll the id bookkeeping back
            for (uint64_t i = 0; i < n_take; ++i) --id_counts_[ids[i]];
        return rc;
    }
    ids_.insert(ids_.end(), ids, ids + n_take);
    return rc_after;
}

// Order-preserving removal of one row (Vec::retain, src/index/flat.rs:94): rows behind it move up
// by one.  Overlapping device ranges are moved through a bounce buffer in ascending chunks.
int GpuFlatIndex::remove_position(uint64_t pos)
{
    const uint64_t n = ids_.size();
    const uint64_t tail = n - pos - 1;
    if (tail) {
        if (!d_bounce_) {
            VL_HIP(hipMalloc(&d_bounce_, BOUNCE_BYTES));
            bounce_bytes_ = BOUNCE_BYTES;
        }
        struct Buf {
            char* base;
            size_t row_bytes;
        } bufs[] = {{reinterpret_cast<char*>(d_master_), dim_ * sizeof(double)},
                    {reinterpret_cast<char*>(d_slab_), ld_ * sizeof(float)},
                    {reinterpret_cast<char*>(d_inv_norm_), sizeof(float)},
                    {reinterpret_cast<char*>(d_flags_), 1}};
        for (const Buf& b : bufs) {
            if (b.row_bytes == 0) continue;
            char* dst = b.base + pos * b.row_bytes;
            const char* src = dst + b.row_bytes;
            const size_t total = tail * b.row_bytes;
            for (size_t off = 0; off < total; off += bounce_bytes_) {
                const size_t c = std::min(bounce_bytes_, total - off);
                VL_HIP(hipMemcpyAsync(d_bounce_, src + off, c, hipMemcpyDeviceToDevice, mut_stream_));
                VL_HIP(hipMemcpyAsync(dst + off, d_bounce_, c, hipMemcpyDeviceToDevice, mut_stream_));
            }
        }
        VL_HIP(hipStreamSynchronize(mut_stream_));
    }
    if (row_flags_[pos] & ROW_OUT_OF_DOMAIN) --n_out_of_domain_;
    if (slab16_rows_ > pos) slab16_rows_ = pos;  // rows behind the hole are re-converted on demand
    if (slab16f_rows_ > pos) slab16f_rows_ = pos;
    if (d_ids_rows_ > pos) d_ids_rows_ = pos;    // ... and the device id table re-uploaded from there
    ids_.erase(ids_.begin() + pos);
    row_flags_.erase(row_flags_.begin() + pos);
    return OK;
}

int GpuFlatIndex::remove(uint64_t id) { return remove_report(id, nullptr); }

bool GpuFlatIndex::contains(uint64_t id) const
{
    std::unique_lock<RwLock> lk(mu_);  // may rebuild the table
    if (!id_counts_valid_) rebuild_id_counts();
    auto it = id_counts_.find(id);
    return it != id_counts_.end() && it->second > 0;
}

int GpuFlatIndex::find_first(uint64_t id, uint64_t* out_pos) const
{
    std::shared_lock<RwLock> lk(mu_);
    auto it = std::find(ids_.begin(), ids_.end(), id);
    if (it == ids_.end()) return ERR_NOT_FOUND;
    if (out_pos) *out_pos = (uint64_t)(it - ids_.begin());
    return OK;
}

int GpuFlatIndex::get_row_at(uint64_t pos, double* out) const
{
    if (!out && dim_) return ERR_INVALID_ARG;
    std::shared_lock<RwLock> lk(mu_);
    if (pos >= ids_.size()) return ERR_NOT_FOUND;
    if (dim_) {
        VL_HIP(hipSetDevice(device_));
        VL_HIP(hipMemcpy(out, d_master_ + pos * dim_, dim_ * sizeof(double), hipMemcpyDeviceToHost));
    }
    return OK;
}

int GpuFlatIndex::remove_report(uint64_t id, std::vector<uint64_t>* removed_positions)
{
    std::unique_lock<RwLock> lk(mu_);
    VL_HIP(hipSetDevice(device_));
    // retain(|e| e.id != id): every matching row goes, highest position first
    for (uint64_t p = ids_.size(); p-- > 0;) {
        if (ids_[p] == id) {
            VL_TRY(remove_position(p));
            if (removed_positions) removed_positions->push_back(p);
            if (id_counts_valid_) {
                auto it = id_counts_.find(id);
                if (it != id_counts_.end() && it->second > 0) --it->second;
            }
        }
    }
    return OK;  // absent id is Ok(()) (src/index/flat.rs:93-96)
}

// ---------------------------------------------------------------------------------------------
// workspaces
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::prepare_ws(Workspace* ws) const
{
    ws->device = device_;
    VL_HIP(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
    // staged queries: up to SCAN_BATCH_QB rows of dim f64, followed by their norms
    const size_t qn = (size_t)SCAN_BATCH_QB * (dim_ + 1) + 8;
    ws->q_cap = qn;
    VL_TRY(dev_alloc(&ws->d_q64, qn));
    VL_TRY(pinned_alloc(&ws->h_q64, qn));
    VL_TRY(dev_alloc(&ws->d_partials, PARTIALS32_ENTRIES));
    VL_TRY(dev_alloc(&ws->d_partials64, PARTIALS64_ENTRIES));
    VL_TRY(dev_alloc(&ws->d_result, SELECT_MAX_ROUNDS));
    VL_TRY(pinned_alloc(&ws->h_result, SELECT_MAX_ROUNDS > SCAN_BATCH_QB ? SELECT_MAX_ROUNDS : SCAN_BATCH_QB));
    VL_TRY(dev_alloc(&ws->d_nan, 1));
    VL_TRY(pinned_alloc(&ws->h_nan, 1));
    VL_HIP(hipEventCreate(&ws->ev0));
    VL_HIP(hipEventCreate(&ws->ev1));
    return OK;
}

Workspace* GpuFlatIndex::acquire_ws() const
{
    WorkspacePool* pool = ws_pool_;
    {
        std::lock_guard<std::mutex> g(pool->mu);
        if (!pool->free_.empty()) {
            Workspace* w = pool->free_.back();
            pool->free_.pop_back();
            return w;
        }
    }
    std::unique_ptr<Workspace> w(new Workspace());
    if (prepare_ws(w.get()) != OK) return nullptr;
    Workspace* raw = w.get();
    std::lock_guard<std::mutex> g(pool->mu);
    pool->all.push_back(std::move(w));
    return raw;
}

void GpuFlatIndex::release_ws(Workspace* ws) const
{
    std::lock_guard<std::mutex> g(ws_pool_->mu);
    ws_pool_->free_.push_back(ws);
}

void GpuFlatIndex::profile_enable(bool on) { profile_.store(on); }

void GpuFlatIndex::profile_read(uint64_t* n, double* ms, uint64_t* bytes)
{
    std::lock_guard<std::mutex> g(prof_mu_);
    if (n) *n = prof_n_;
    if (ms) *ms = prof_ms_;
    if (bytes) *bytes = prof_bytes_;
    prof_n_ = 0;
    prof_ms_ = 0.0;
    prof_bytes_ = 0;
}

// ---------------------------------------------------------------------------------------------
// search (src/index/flat.rs:98-119)
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::search(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                         uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (co_.enabled())
        return search_coalesced(query, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
    return search_direct(query, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
}

// ---------------------------------------------------------------------------------------------
// Coalescing of concurrent single-query calls (SURVEY 8(b) threading row / 8(f) f1).
// The reference serves many searches at once under RwLock::read (src/client.rs:398, one tokio
// worker each); every one of them walks the whole slab.  Here concurrent callers share slab passes:
// a caller that finds no batch in flight becomes the leader, takes every queued request with its own
// (metric, k) -- all that piled up while the previous batch was on the GPU -- and answers them with
// ONE search_batch() pass; the others sleep until their request is marked done.  Every caller still
// receives exactly what search_direct() would have returned (search_batch's contract); a batch that
// fails as a whole (e.g. a NaN score for one member) is redone one by one so errors stay per caller.
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::search_coalesced(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                                   uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (!out_n) return ERR_INVALID_ARG;
    *out_n = 0;
    {   // everything that can fail or finish without touching the slab is settled on the calling thread
        std::shared_lock<RwLock> lk(mu_);
        const uint64_t n = ids_.size();
        if (metric < 0 || metric > 3 || (n != 0 && q_len != dim_) || n == 0 || k == 0 || (!query && dim_) || !out_scores ||
            force_path_.load() != 0)
            goto direct;
    }
    {
        CoalesceReq r{query, k, metric, out_pos, out_ids, out_scores, out_n};
        co_.run(
            r, [](const CoalesceReq& a, const CoalesceReq& o) { return a.metric == o.metric && a.k == o.k; },
            [this](std::vector<CoalesceReq*>& batch) { run_coalesced(batch); });
        set_last_path(r.path);
        if (r.rc != OK) set_last_error(r.err);
        return r.rc;
    }
direct:
    return search_direct(query, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
}

void GpuFlatIndex::run_coalesced(std::vector<CoalesceReq*>& batch) const
{
    auto one = [&](CoalesceReq* o) {
        o->rc = search_direct(o->query, dim_, o->k, o->metric, o->out_pos, o->out_ids, o->out_scores, o->out_n);
        o->path = last_path();
        if (o->rc != OK) o->err = last_error();
    };
    if (batch.size() == 1) {
        one(batch[0]);
        return;
    }
    // scratch and row stride use min(k, len), never the caller's raw k: k = u64::MAX would throw length_error,
    // k = 2^63 with two queries would wrap nq * k to 0 (mutators are exclusive, so len cannot move under a search)
    uint64_t n_rows;
    {
        std::shared_lock<RwLock> lk(mu_);
        n_rows = ids_.size();
    }
    const uint64_t nq = batch.size(), k = std::min<uint64_t>(batch[0]->k, n_rows);
    int rc = OK;
    try {
        std::vector<double> q(nq * dim_);
        for (uint64_t i = 0; i < nq; ++i) std::memcpy(q.data() + i * dim_, batch[i]->query, dim_ * sizeof(double));
        std::vector<uint64_t> pos(nq * k), ids(nq * k), cnt(nq);
        std::vector<double> scores(nq * k);
        rc = search_batch(q.data(), nq, dim_, k, batch[0]->metric, pos.data(), ids.data(), scores.data(), cnt.data());
        if (rc == OK) {
            const int path = last_path();
            for (uint64_t i = 0; i < nq; ++i) {
                CoalesceReq* o = batch[i];
                const uint64_t m = cnt[i];
                for (uint64_t j = 0; j < m; ++j) {
                    if (o->out_pos) o->out_pos[j] = pos[i * k + j];
                    if (o->out_ids) o->out_ids[j] = ids[i * k + j];
                    o->out_scores[j] = scores[i * k + j];
                }
                *o->out_n = m;
                o->rc = OK;
                o->path = path;
            }
            return;
        }
    } catch (...) {  // bad_alloc, length_error: answer the callers one by one instead
        rc = ERR_OOM;
    }
    for (CoalesceReq* o : batch) one(o);  // per-caller errors (src/index/flat.rs:116 panics only the offending search)
}

int GpuFlatIndex::search_direct(const double* query, uint64_t q_len, uint64_t k, int metric, uint64_t* out_pos,
                                uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (!out_n) return ERR_INVALID_ARG;
    *out_n = 0;
    if (metric < 0 || metric > 3) {
        set_last_error("unknown metric");
        return ERR_INVALID_ARG;
    }
    std::shared_lock<RwLock> lk(mu_);
    const uint64_t n = ids_.size();
    if (n != 0 && q_len != dim_) {  // :99-104 (skipped while the index is empty)
        set_dim_mismatch(dim_, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(q_len));
        return ERR_DIM_MISMATCH;
    }
    if (n == 0 || k == 0) return OK;  // truncate(0) / nothing stored
    if ((!query && dim_) || !out_scores) return ERR_INVALID_ARG;
    const uint64_t k_eff = std::min<uint64_t>(k, n);

    VL_HIP(hipSetDevice(device_));
    Workspace* ws = acquire_ws();
    if (!ws) return ERR_DEVICE;
    active_searches_.fetch_add(1, std::memory_order_relaxed);
    const int rc = search_locked(ws, query, k_eff, metric, out_pos, out_ids, out_scores, out_n, false);
    active_searches_.fetch_sub(1, std::memory_order_relaxed);
    if (rc != OK) (void)hipStreamSynchronize(ws->stream);
    release_ws(ws);
    return rc;
}

// nq independent searches sharing slab passes: up to MFMA_MAX_BATCH queries per pass over the bf16 slab
// (k_mfma_scan: cosine / dot / Euclidean, dim <= 768, >= MFMA_MIN_ROWS rows), otherwise groups of SCAN_BATCH_QB
// queries per pass over the f32 slab (k_scan_batch).  Every query still gets its own exact rescoring, bound check
// and, if that fails, its own single-query or exact-path run: each row of the output is exactly what search() returns.
int GpuFlatIndex::search_batch(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric,
                               uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (nq == 0) return OK;
    if (!out_n) return ERR_INVALID_ARG;
    for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
    if (metric < 0 || metric > 3) {
        set_last_error("unknown metric");
        return ERR_INVALID_ARG;
    }
    std::shared_lock<RwLock> lk(mu_);
    return search_batch_locked(queries, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
}

// the caller holds mu_ (shared) and has zeroed out_n
int GpuFlatIndex::search_batch_locked(const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric,
                                      uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    const uint64_t n = ids_.size();
    if (n != 0 && q_len != dim_) {
        set_dim_mismatch(dim_, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(q_len));
        return ERR_DIM_MISMATCH;
    }
    if (n == 0 || k == 0) return OK;
    if (!queries || !out_scores) return ERR_INVALID_ARG;
    const uint64_t k_eff = std::min<uint64_t>(k, n);

    VL_HIP(hipSetDevice(device_));
    Workspace* ws = acquire_ws();
    if (!ws) return ERR_DEVICE;
    struct Releaser {
        const GpuFlatIndex* self;
        Workspace* ws;
        ~Releaser()
        {
            (void)hipStreamSynchronize(ws->stream);
            self->release_ws(ws);
        }
    } rel{this, ws};
    hipStream_t st = ws->stream;

    // row lengths without an 8-query f32 shape can still take the MFMA filter (its bf16 slab is padded to 128)
    const bool f32_batch = scan_batch_supported(ld_);
    const bool batchable = nq > 1 && force_path_.load() == 0 && k_eff <= (uint64_t)KFAST_MAX && n_out_of_domain_ == 0;
    auto out_at = [&](uint64_t* base, uint64_t qi) { return base ? base + qi * k : nullptr; };
    auto single = [&](uint64_t qi, bool skip_fast) -> int {
        return search_locked(ws, queries + qi * dim_, k_eff, metric, out_at(out_pos, qi), out_at(out_ids, qi),
                             out_scores + qi * k, out_n + qi, skip_fast);
    };
    if (!batchable) {
        for (uint64_t qi = 0; qi < nq; ++qi) VL_TRY(single(qi, false));
        return OK;
    }

    // two or more queries: bf16 MFMA candidate filter; whatever it cannot certify is redone below
    std::vector<uint8_t> done(nq, 0);
    const char* mf_env = getenv("VL_MFMA");
    const bool mfma_on = !(mf_env && mf_env[0] == '0');
    const char* mf_min = getenv("VL_MFMA_MIN_BATCH");
    const uint64_t mfma_min = mf_min && *mf_min ? (uint64_t)atoi(mf_min) : (uint64_t)MFMA_MIN_BATCH;
    if (mfma_on && nq >= mfma_min && n >= MFMA_MIN_ROWS && mfma_scan_supported((uint32_t)dim_, metric)) {
        VL_TRY(search_batch_mfma(ws, queries, nullptr, nq, k, k_eff, metric, out_pos, out_ids, out_scores, out_n, &done));
        uint64_t left = 0;
        for (uint64_t qi = 0; qi < nq; ++qi) left += done[qi] ? 0 : 1;
        if (left == 0) {
            set_last_path(PATH_FAST);
            return OK;
        }
        if (left <= 2 || !f32_batch) {  // one or two stragglers (or no 8-query f32 shape): one by one on the f32 path
            for (uint64_t qi = 0; qi < nq; ++qi)
                if (!done[qi]) VL_TRY(single(qi, false));
            return OK;
        }
    }
    if (!f32_batch) {
        for (uint64_t qi = 0; qi < nq; ++qi) VL_TRY(single(qi, false));
        return OK;
    }

    // what is left (everything, or what the MFMA filter could not certify): 8 queries per f32 slab pass
    std::vector<uint64_t> todo;
    todo.reserve(nq);
    for (uint64_t qi = 0; qi < nq; ++qi)
        if (!done[qi]) todo.push_back(qi);
    const uint64_t nt = todo.size();
    const bool prof = profile_.load();
    // Up to K3_PIPE_QUERIES queries are staged at once (one H2D copy) and answered by ONE scan launch -- their groups of 8 as
    // blockIdx.y, each group one pass over the slab -- and ONE finalize launch, then one stream sync: staged, launched and
    // synchronised pass by pass a 50 000-row index answered 70-100 k Manhattan queries per second where cosine batches
    // reach millions (round 3's verdict, item 8).
    if (!ws->k3_d_q64) {
        const size_t words = (size_t)K3_PIPE_QUERIES * (dim_ + 1);
        VL_TRY(dev_alloc(&ws->k3_d_q64, words));
        VL_TRY(pinned_alloc(&ws->k3_h_q64, words));
        VL_TRY(pinned_alloc(&ws->k3_h_result, (size_t)K3_PIPE_QUERIES));
    }
    std::vector<uint8_t> in_domain((size_t)K3_PIPE_QUERIES);
    for (uint64_t base = 0; base < nt; base += K3_PIPE_QUERIES) {
        const uint32_t cnt = (uint32_t)std::min<uint64_t>(K3_PIPE_QUERIES, nt - base);
        double* norms = ws->k3_h_q64 + (size_t)cnt * dim_;
        for (uint32_t j = 0; j < cnt; ++j)
            in_domain[j] = stage_query(queries + todo[base + j] * dim_, ws->k3_h_q64 + (size_t)j * dim_, dim_, &norms[j]) ? 1 : 0;
        VL_HIP(hipMemcpyAsync(ws->k3_d_q64, ws->k3_h_q64, ((size_t)cnt * dim_ + cnt) * sizeof(double), hipMemcpyHostToDevice, st));
        const double* d_norms = ws->k3_d_q64 + (size_t)cnt * dim_;
        const uint32_t passes = (cnt + SCAN_BATCH_QB - 1) / SCAN_BATCH_QB;
        if (prof) VL_HIP(hipEventRecord(ws->ev0, st));
        {   // ONE scan launch (groups of 8 queries as blockIdx.y: each group is one pass over the slab) and ONE finalize launch
            ScanPlan plan;
            VL_HIP(launch_scan_batch(st, metric, d_slab_, d_inv_norm_, ws->k3_d_q64, cnt, n, (uint32_t)dim_, ld_, ws->d_partials, &plan));
            if (prof) VL_HIP(hipEventRecord(ws->ev1, st));
            VL_HIP(launch_merge_finalize(st, metric, ws->d_partials, plan.grid, (int)cnt, d_master_, ws->k3_d_q64, d_norms,
                                         (uint32_t)dim_, n, (uint32_t)k_eff, max_row_norm_, ws->k3_h_result));
        }
        VL_HIP(hipStreamSynchronize(st));
        if (prof) {  // (the events bracket the one scan launch: `passes` slab passes side by side)
            float ms = 0.f;
            VL_HIP(hipEventElapsedTime(&ms, ws->ev0, ws->ev1));
            std::lock_guard<std::mutex> gl(prof_mu_);
            prof_n_ += passes;
            prof_ms_ += ms;
            prof_bytes_ += (uint64_t)passes * n * (uint64_t)ld_ * sizeof(float);
        }
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint64_t qi = todo[base + j];
            const SearchResultBlock& r = ws->k3_h_result[j];  // (a fallback below uses the workspace's other buffers, not these)
            const bool ok = in_domain[j] && !(r.flags & RESULT_NEEDS_EXACT) && r.n_out == k_eff;
            if (ok) {
                for (uint64_t i = 0; i < k_eff; ++i) {
                    const uint32_t p = r.pos[i];
                    if (p >= n) {
                        set_last_error("batch fast path returned an out-of-range position (kernel bug)");
                        return ERR_DEVICE;
                    }
                    if (out_pos) out_pos[qi * k + i] = p;
                    if (out_ids) out_ids[qi * k + i] = ids_[p];
                    out_scores[qi * k + i] = r.score[i];
                }
                out_n[qi] = k_eff;
            } else {
                VL_TRY(single(qi, true));
            }
        }
        set_last_path(PATH_FAST);
    }
    return OK;
}

int GpuFlatIndex::search_locked(Workspace* ws, const double* query, uint64_t k_eff, int metric,
                                uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n,
                                bool skip_fast) const
{
    const uint64_t n = ids_.size();
    hipStream_t st = ws->stream;

    // stage the query in the pinned block: the f64 values, then the norm
    double qq = 0.0, qmax = 0.0;
    bool q_finite = true;
    for (uint64_t i = 0; i < dim_; ++i) {
        const double v = query[i];
        ws->h_q64[i] = v;
        qq += v * v;
        const double av = std::fabs(v);
        if (!(av <= 1.797693134862315708e308)) q_finite = false;
        if (av > qmax) qmax = av;
    }
    const double q_norm = std::sqrt(qq);
    const bool q_in_domain = q_finite && qmax <= DOMAIN_MAX_ABS && (q_norm == 0.0 || q_norm >= DOMAIN_MIN_NORM);
    ws->h_q64[dim_] = q_norm;
    // The f32 scan takes its query in the kernel arguments and the finalize kernel reads the pinned block itself,
    // so the common case needs NO copy in front of the kernels.  Every other kernel (bf16 filter, the k > 60 lists,
    // the exact scan: all workgroups read the query) wants it in device memory: copied on first use.
    bool q_on_device = false;
    auto q_to_device = [&]() -> int {
        if (!q_on_device) {
            VL_HIP(hipMemcpyAsync(ws->d_q64, ws->h_q64, (dim_ + 1) * sizeof(double), hipMemcpyHostToDevice, st));
            q_on_device = true;
        }
        return OK;
    };

    const int forced = force_path_.load();
    const bool fast_ok = !skip_fast && forced == 0 && k_eff <= (uint64_t)KFAST_MAX && n_out_of_domain_ == 0 &&
                         q_in_domain;

    // Opt-in first stage: scan the bf16 copy of the slab (half the HBM bytes).  Its candidates get the
    // same exact f64 rescoring and a bound with the bf16 row-rounding term; if that cannot certify
    // the answer the f32 scan below runs as before.  The stage switches itself off on data where it
    // rarely certifies (dense neighbourhoods).
    if (fast_ok && single_filter_.load() == 1 && scan_bf16_supported((uint32_t)dim_, metric) &&
        !(bf16_tries_.load() >= 64 && bf16_fails_.load() * 3 > bf16_tries_.load())) {
        VL_TRY(ensure_bf16_slab(false));
        VL_TRY(q_to_device());
        const bool prof = profile_.load();
        int grid = 0;
        if (prof) VL_HIP(hipEventRecord(ws->ev0, st));
        VL_HIP(launch_scan_bf16(st, metric, d_slab16_, d_norm16_, d_sqnorm_, ws->d_q64, n, (uint32_t)dim_,
                                ws->d_partials, &grid));
        if (prof) VL_HIP(hipEventRecord(ws->ev1, st));
        // rows are rounded to bf16 (relative 2^-8 after the f32 step), the query is f32: u_b (1 + u) + u
        VL_HIP(launch_merge_finalize(st, metric, ws->d_partials, grid, 1, d_master_, ws->d_q64, ws->d_q64 + dim_,
                                     (uint32_t)dim_, n, (uint32_t)k_eff, max_row_norm_, ws->h_result, 0.00392));
        VL_HIP(hipStreamSynchronize(st));
        if (prof) {
            float ms = 0.f;
            VL_HIP(hipEventElapsedTime(&ms, ws->ev0, ws->ev1));
            std::lock_guard<std::mutex> g(prof_mu_);
            prof_n_ += 1;
            prof_ms_ += ms;
            prof_bytes_ += n * (uint64_t)mfma_ldb((uint32_t)dim_) * 2;
        }
        bf16_tries_.fetch_add(1);
        const SearchResultBlock& r = *ws->h_result;
        if (!(r.flags & RESULT_NEEDS_EXACT) && r.n_out == k_eff) {
            for (uint64_t i = 0; i < k_eff; ++i) {
                const uint32_t p = r.pos[i];
                if (p >= n) {
                    set_last_error("bf16 filter returned an out-of-range position (kernel bug)");
                    return ERR_DEVICE;
                }
                if (out_pos) out_pos[i] = p;
                if (out_ids) out_ids[i] = ids_[p];
                out_scores[i] = r.score[i];
            }
            *out_n = k_eff;
            set_last_path(PATH_FAST);
            return OK;
        }
        bf16_fails_.fetch_add(1);
    }
    if (fast_ok) {
        const bool prof = profile_.load();
        ScanPlan plan;
        // one search = two launches: the scan (query in its kernel arguments) and the finalize kernel, which reads
        // the f64 query from the pinned block, writes the result block into pinned memory and stamps it; the host
        // waits for the stamp, not for the stream
        const bool qarg = scan_takes_qarg(ld_);
        const float* q32 = nullptr;
        if (qarg) {
            if (ws->q32.size() < ld_) ws->q32.assign(ld_, 0.0f);
            for (uint64_t i = 0; i < dim_; ++i) ws->q32[i] = (float)query[i];  // nearest even, like load_q4 on the device
            q32 = ws->q32.data();
        } else {
            VL_TRY(q_to_device());
        }
        const double* fq = q_on_device ? ws->d_q64 : ws->h_q64;
        uint32_t seq = ++ws->seq;
        if (seq == 0) seq = ++ws->seq;
        ws->h_result->seq = 0;
        if (prof) VL_HIP(hipEventRecord(ws->ev0, st));
        VL_HIP(launch_scan(st, metric, d_slab_, d_inv_norm_, ws->d_q64, n, (uint32_t)dim_, ld_, ws->d_partials, &plan, q32));
        if (prof) VL_HIP(hipEventRecord(ws->ev1, st));
        VL_HIP(launch_merge_finalize(st, metric, ws->d_partials, plan.grid, 1, d_master_, fq, fq + dim_, (uint32_t)dim_, n,
                                     (uint32_t)k_eff, max_row_norm_, ws->h_result, 0.0, seq));
        last_scan_variant_.store(plan.variant, std::memory_order_relaxed);
        last_scan_grid_.store(plan.grid, std::memory_order_relaxed);
        last_scan_qarg_.store(qarg ? 1 : 0, std::memory_order_relaxed);
        VL_TRY(wait_result(ws, seq));
        if (prof) {
            float ms = 0.f;
            hipError_t pe = hipEventElapsedTime(&ms, ws->ev0, ws->ev1);
            if (pe == hipErrorNotReady) {
                VL_HIP(hipEventSynchronize(ws->ev1));
                pe = hipEventElapsedTime(&ms, ws->ev0, ws->ev1);
            }
            VL_HIP(pe);
            std::lock_guard<std::mutex> g(prof_mu_);
            prof_n_ += 1;
            prof_ms_ += ms;
            prof_bytes_ += n * (uint64_t)ld_ * sizeof(float);
        }
        const SearchResultBlock& r = *ws->h_result;
        if (!(r.flags & RESULT_NEEDS_EXACT) && r.n_out == k_eff) {
            for (uint64_t i = 0; i < k_eff; ++i) {
                const uint32_t p = r.pos[i];
                if (p >= n) {
                    set_last_error("fast path returned an out-of-range position (kernel bug)");
                    return ERR_DEVICE;
                }
                if (out_pos) out_pos[i] = p;
                if (out_ids) out_ids[i] = ids_[p];
                out_scores[i] = r.score[i];
            }
            *out_n = k_eff;
            set_last_path(PATH_FAST);
            return OK;
        }
        // ties at the cut or a failed bound: fall through to the exact kernels
    }

    // 60 < k <= 220: several 64-entry candidate lists (one per partition of the scan's workgroups) rescored and
    // ranked together; still one f32 scan (2.3 ms at N = 10 M) instead of the exact f64 scan + selection rounds
    if (!skip_fast && forced == 0 && k_eff > (uint64_t)KFAST_MAX && k_eff <= (uint64_t)KMULTI_MAX && n_out_of_domain_ == 0 &&
        q_in_domain && n > 4 * (uint64_t)KP) {
        int parts = (int)std::min<uint64_t>(4, (k_eff + 36 + KP - 1) / KP);
        ScanPlan plan;
        VL_TRY(q_to_device());
        VL_HIP(launch_scan(st, metric, d_slab_, d_inv_norm_, ws->d_q64, n, (uint32_t)dim_, ld_, ws->d_partials, &plan));
        while (parts < 4 && plan.grid % parts != 0) ++parts;  // partitions are equal runs of workgroup lists
        if (plan.grid % parts == 0 && plan.grid >= parts) {
            VL_HIP(launch_merge_finalize_multi(st, metric, ws->d_partials, plan.grid, parts, d_master_, ws->d_q64,
                                               ws->d_q64 + dim_, (uint32_t)dim_, n, (uint32_t)k_eff, max_row_norm_,
                                               ws->h_result));
            VL_HIP(hipStreamSynchronize(st));
            const SearchResultBlock& r0 = ws->h_result[0];
            if (!(r0.flags & RESULT_NEEDS_EXACT) && r0.n_out == k_eff) {
                for (uint64_t i = 0; i < k_eff; ++i) {
                    const uint32_t p = ws->h_result[i / KP].pos[i % KP];
                    if (p >= n) {
                        set_last_error("multi-list fast path returned an out-of-range position (kernel bug)");
                        return ERR_DEVICE;
                    }
                    if (out_pos) out_pos[i] = p;
                    if (out_ids) out_ids[i] = ids_[p];
                    out_scores[i] = ws->h_result[i / KP].score[i % KP];
                }
                *out_n = k_eff;
                set_last_path(PATH_FAST);
                return OK;
            }
        } else {
            VL_HIP(hipStreamSynchronize(st));
        }
    }

    std::vector<uint32_t> pos;
    std::vector<double> scores;
    VL_TRY(q_to_device());
    VL_TRY(run_exact(ws, metric, n, k_eff, &pos, &scores));
    for (uint64_t i = 0; i < k_eff; ++i) {
        if (pos[i] >= n) {
            set_last_error("exact path returned an out-of-range position (kernel bug)");
            return ERR_DEVICE;
        }
        if (out_pos) out_pos[i] = pos[i];
        if (out_ids) out_ids[i] = ids_[pos[i]];
        out_scores[i] = scores[i];
    }
    *out_n = k_eff;
    return OK;
}

// Completion of a single search.  The finalize kernel stores the result block in pinned host memory and then the
// stamp (system-scope release); a lone caller polls the stamp -- the block is readable a PCIe write after the kernel's
// last store, where hipStreamSynchronize adds the runtime's own completion path (signal, interrupt or its polling
// interval) on top.  The poll is bounded: the stream is queried now and then so that a failed launch surfaces as an
// error instead of a hang, and when several searches are in flight the threads sleep in hipStreamSynchronize instead
// of each burning a core.
int GpuFlatIndex::wait_result(Workspace* ws, uint32_t seq) const
{
    const uint32_t* stamp = &ws->h_result->seq;
    auto stamped = [&]() { return __atomic_load_n(stamp, __ATOMIC_ACQUIRE) == seq; };
    if (active_searches_.load(std::memory_order_relaxed) <= SPIN_MAX_SEARCHERS) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 1;; ++it) {
            if (stamped()) return OK;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((it & 0x3FFF) == 0) {  // every ~16 k polls (some hundred microseconds)
                const hipError_t q = hipStreamQuery(ws->stream);
                if (q == hipSuccess) break;            // stream drained: the stamp is there, or the launch was lost
                if (q != hipErrorNotReady) VL_HIP(q);
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(SPIN_MAX_MS)) break;
            }
        }
    }
    VL_HIP(hipStreamSynchronize(ws->stream));
    if (!stamped()) {
        set_last_error("the finalize kernel completed without stamping its result block");
        return ERR_DEVICE;
    }
    return OK;
}

int GpuFlatIndex::run_exact(Workspace* ws, int metric, uint64_t n, uint64_t k_eff, std::vector<uint32_t>* pos,
                            std::vector<double>* scores) const
{
    hipStream_t st = ws->stream;
    if (ws->scores_cap < n) {
        if (ws->d_scores) (void)hipFree(ws->d_scores);
        ws->d_scores = nullptr;
        ws->scores_cap = 0;
        const size_t cap = std::max<size_t>(n, 1024);
        VL_TRY(dev_alloc(&ws->d_scores, cap));
        ws->scores_cap = cap;
    }
    VL_HIP(hipMemsetAsync(ws->d_nan, 0, sizeof(uint32_t), st));
    VL_HIP(launch_exact_scan(st, metric, d_master_, ws->d_q64, n, (uint32_t)dim_, ws->d_scores, ws->d_nan));
    VL_HIP(hipMemcpyAsync(ws->h_nan, ws->d_nan, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    pos->resize(k_eff);
    scores->resize(k_eff);

    if (n == 1) {  // a 1-element sort never calls the comparator: even a NaN score is returned
        VL_HIP(hipMemcpyAsync(scores->data(), ws->d_scores, sizeof(double), hipMemcpyDeviceToHost, st));
        VL_HIP(hipStreamSynchronize(st));
        (*pos)[0] = 0;
        set_last_path(PATH_EXACT_SELECT);
        return OK;
    }

    const int forced = force_path_.load();
    // k <= 64: one selection pass over scores[]; k <= 1024: rounds of 64, each restricted to the rows ranked
    // behind the previous round's last entry (an 80 MB pass per round at N = 10 M instead of a 10 M-element sort)
    const uint64_t rounds = (k_eff + KP - 1) / KP;
    // measured: a round costs ~70 us at N = 1 M and ~140 us at 10 M, the sort 0.34 ms and 6 ms
    const uint64_t max_rounds = std::min<uint64_t>(SELECT_MAX_ROUNDS, std::max<uint64_t>(2, n / 250000));
    const bool use_select = rounds <= max_rounds && forced != PATH_EXACT_SORT;
    if (use_select) {
        for (uint64_t r = 0; r < rounds; ++r) {
            const uint32_t kr = (uint32_t)std::min<uint64_t>(KP, k_eff - r * KP);
            VL_HIP(launch_exact_select(st, ws->d_scores, n, kr, ws->d_partials64, ws->d_nan, ws->d_result + r,
                                       r ? ws->d_result + (r - 1) : nullptr));
        }
        VL_HIP(hipMemcpyAsync(ws->h_result, ws->d_result, rounds * sizeof(SearchResultBlock), hipMemcpyDeviceToHost, st));
        VL_HIP(hipStreamSynchronize(st));
    } else {
        const uint64_t cap = sort_capacity_for(n);
        if (ws->sort_cap < cap) {
            if (ws->d_okeys) (void)hipFree(ws->d_okeys);
            if (ws->d_opos) (void)hipFree(ws->d_opos);
            ws->d_okeys = nullptr;
            ws->d_opos = nullptr;
            ws->sort_cap = 0;
            VL_TRY(dev_alloc(&ws->d_okeys, cap));
            VL_TRY(dev_alloc(&ws->d_opos, cap));
            ws->sort_cap = cap;
        }
        if (ws->out_cap < k_eff) {
            if (ws->d_out_pos) (void)hipFree(ws->d_out_pos);
            if (ws->d_out_scores) (void)hipFree(ws->d_out_scores);
            ws->d_out_pos = nullptr;
            ws->d_out_scores = nullptr;
            ws->out_cap = 0;
            VL_TRY(dev_alloc(&ws->d_out_pos, k_eff));
            VL_TRY(dev_alloc(&ws->d_out_scores, k_eff));
            ws->out_cap = k_eff;
        }
        VL_HIP(launch_exact_sort(st, ws->d_scores, n, k_eff, ws->d_okeys, ws->d_opos, ws->d_out_pos,
                                 ws->d_out_scores));
        VL_HIP(hipMemcpyAsync(pos->data(), ws->d_out_pos, k_eff * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        VL_HIP(hipMemcpyAsync(scores->data(), ws->d_out_scores, k_eff * sizeof(double), hipMemcpyDeviceToHost, st));
        VL_HIP(hipStreamSynchronize(st));
    }
    if (*ws->h_nan) {
        // sort_by(|a, b| b.score.partial_cmp(&a.score).unwrap()) panics on a NaN (src/index/flat.rs:116)
        set_last_error("NaN similarity score: the reference panics in partial_cmp().unwrap()");
        return ERR_NAN_SCORE;
    }
    if (use_select) {
        for (uint64_t r = 0; r < rounds; ++r) {
            const SearchResultBlock& blk = ws->h_result[r];
            const uint64_t kr = std::min<uint64_t>(KP, k_eff - r * KP);
            if (blk.n_out != kr) {
                set_last_error("exact select returned an unexpected result count");
                return ERR_DEVICE;
            }
            for (uint64_t i = 0; i < kr; ++i) {
                (*pos)[r * KP + i] = blk.pos[i];
                (*scores)[r * KP + i] = blk.score[i];
            }
        }
        set_last_path(PATH_EXACT_SELECT);
    } else {
        set_last_path(PATH_EXACT_SORT);
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------
// large batches: bf16 MFMA candidate filter (mfma_scan.hip) + the same exact finalize
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::ensure_bf16_slab(bool frag_major) const
{
    std::lock_guard<std::mutex> g(bf16_mu_);
    const uint64_t n = ids_.size();
    const uint32_t ldb = mfma_ldb((uint32_t)dim_);
    // whole MFMA tiles: the batch kernels read the last, partial tile past the live rows (and mask them)
    const size_t cap16 = (cap_ + MFMA_TILE_ROWS - 1) / MFMA_TILE_ROWS * MFMA_TILE_ROWS;
    auto fail = [&](hipError_t e) {
        (void)hipGetLastError();
        set_last_error(std::string("bf16 slab allocation failed: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? (int)ERR_OOM : (int)ERR_DEVICE;
    };
    if (!d_norm16_) {  // both per-row arrays or neither: a half-made set would be launched with null pointers next time
        float* sq = nullptr;
        float* nr = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&sq), cap16 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&nr), cap16 * sizeof(float));
        if (e != hipSuccess) {
            if (sq) (void)hipFree(sq);
            if (nr) (void)hipFree(nr);
            return fail(e);
        }
        d_sqnorm_ = sq;
        d_norm16_ = nr;
    }
    void*& slab = frag_major ? d_slab16f_ : d_slab16_;
    uint64_t& rows_done = frag_major ? slab16f_rows_ : slab16_rows_;
    if (!slab) {
        void* s16 = nullptr;
        const hipError_t e = hipMalloc(&s16, cap16 * (size_t)ldb * 2);
        if (e != hipSuccess) return fail(e);
        slab = s16;
        rows_done = 0;
    }
    if (rows_done < n) {
        if (frag_major) {
            // the fragment layout interleaves 16 neighbouring rows: conversion restarts at the group boundary
            const uint64_t first = rows_done & ~15ull;
            VL_HIP(launch_rows_bf16_frag(mut_stream_, d_master_ + first * dim_, first, n - first, (uint32_t)dim_, slab, d_norm16_,
                                         d_sqnorm_));
        } else {
            char* dst = reinterpret_cast<char*>(slab) + rows_done * (size_t)ldb * 2;
            VL_HIP(launch_rows_bf16(mut_stream_, d_master_ + rows_done * dim_, n - rows_done, (uint32_t)dim_, dst,
                                    d_norm16_ + rows_done, d_sqnorm_ + rows_done));
        }
        VL_HIP(hipStreamSynchronize(mut_stream_));
        rows_done = n;
    }
    return OK;
}

// position -> id on the device, for exchange records written by the finalize kernel (search_batch_to_record).  Caller holds
// mu_ (shared or unique); concurrent callers serialise on bf16_mu_ like the other lazily built device copies.
int GpuFlatIndex::ensure_device_ids() const
{
    std::lock_guard<std::mutex> g(bf16_mu_);
    const uint64_t n = ids_.size();
    if (d_ids_cap_ < n) {
        unsigned long long* fresh = nullptr;
        const uint64_t want = std::max<uint64_t>(cap_, n);
        VL_HIP(hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(unsigned long long)));
        if (d_ids_) (void)hipFree(d_ids_);
        d_ids_ = fresh;
        d_ids_cap_ = want;
        d_ids_rows_ = 0;
    }
    if (d_ids_rows_ < n) {
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "u64 ids");
        VL_HIP(hipMemcpyAsync(d_ids_ + d_ids_rows_, ids_.data() + d_ids_rows_, (n - d_ids_rows_) * sizeof(uint64_t),
                              hipMemcpyHostToDevice, mut_stream_));
        VL_HIP(hipStreamSynchronize(mut_stream_));
        d_ids_rows_ = n;
    }
    return OK;
}

int GpuFlatIndex::search_batch_to_record(const double* queries, bool queries_on_device, uint64_t nq, uint64_t q_len, uint64_t ks,
                                         int metric, uint64_t row_offset, unsigned long long* d_record, bool* handled) const
{
    if (handled) *handled = false;
    if (!handled || !d_record || nq == 0 || ks == 0 || !queries || metric < 0 || metric > 3) return OK;  // the host path decides
    VL_HIP(hipSetDevice(device_));
    std::shared_lock<RwLock> lk(mu_);
    const uint64_t n = ids_.size();
    if (n == 0 || q_len != dim_) return OK;  // (errors and empty shards are the host path's to report)
    const uint64_t k_eff = std::min<uint64_t>(ks, n);
    const char* mf_env = getenv("VL_MFMA");
    const char* mf_min = getenv("VL_MFMA_MIN_BATCH");
    const uint64_t mfma_min = mf_min && *mf_min ? (uint64_t)atoi(mf_min) : (uint64_t)MFMA_MIN_BATCH;
    const bool direct = nq > 1 && force_path_.load() == 0 && k_eff <= (uint64_t)KFAST_MAX && n_out_of_domain_ == 0 &&
                        !(mf_env && mf_env[0] == '0') && nq >= mfma_min && n >= MFMA_MIN_ROWS &&
                        mfma_scan_supported((uint32_t)dim_, metric) && nq * ks < (1ull << 31);
    if (!direct) return OK;
    VL_TRY(ensure_device_ids());
    ShardRecordSink sink;
    sink.cnt = d_record + SHARD_HDR_WORDS;
    sink.score_bits = sink.cnt + nq;
    sink.gpos = sink.score_bits + nq * ks;
    sink.ids = sink.gpos + nq * ks;
    sink.pos_to_id = d_ids_;
    sink.row_offset = row_offset;
    sink.ks = (uint32_t)ks;
    // host-side answers of the queries the filter certifies are not needed (the record is the answer); the others come back
    // through `done` and are redone below
    std::vector<uint8_t> done(nq, 0);
    std::vector<uint64_t> cnt(nq, 0);
    std::vector<uint64_t> pos, ids;
    std::vector<double> sc;
    try {
        pos.resize(nq * ks);
        ids.resize(nq * ks);
        sc.resize(nq * ks);
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory in the shard search");
        return ERR_OOM;
    }
    {
        Workspace* ws = acquire_ws();
        if (!ws) return ERR_DEVICE;
        struct Releaser {
            const GpuFlatIndex* self;
            Workspace* ws;
            ~Releaser()
            {
                (void)hipStreamSynchronize(ws->stream);
                self->release_ws(ws);
            }
        } rel{this, ws};
        VL_TRY(search_batch_mfma(ws, queries_on_device ? nullptr : queries, queries_on_device ? queries : nullptr, nq, ks, k_eff, metric,
                                 pos.data(), ids.data(), sc.data(), cnt.data(), &done, &sink));
    }
    set_last_path(PATH_FAST);
    // what the filter could not certify: the host paths answer those queries (under the same shared lock: one index state
    // for the whole batch) and their slices of the record are patched; typically none
    std::vector<uint64_t> left;
    for (uint64_t qi = 0; qi < nq; ++qi)
        if (!done[qi]) left.push_back(qi);
    if (!left.empty()) {
        const uint64_t m = left.size();
        std::vector<double> hq(m * dim_);
        for (uint64_t i = 0; i < m; ++i) {
            if (queries_on_device)
                VL_HIP(hipMemcpy(hq.data() + i * dim_, queries + left[i] * dim_, dim_ * sizeof(double), hipMemcpyDeviceToHost));
            else
                std::memcpy(hq.data() + i * dim_, queries + left[i] * dim_, dim_ * sizeof(double));
        }
        std::vector<uint64_t> t_pos(m * ks), t_ids(m * ks), t_n(m, 0);
        std::vector<double> t_sc(m * ks);
        VL_TRY(search_batch_locked(hq.data(), m, dim_, ks, metric, t_pos.data(), t_ids.data(), t_sc.data(), t_n.data()));
        for (uint64_t i = 0; i < m; ++i) {
            const uint64_t qi = left[i], c = std::min<uint64_t>(t_n[i], ks);
            std::vector<unsigned long long> buf(3 * ks, 0ull);
            for (uint64_t j = 0; j < c; ++j) {
                std::memcpy(&buf[j], &t_sc[i * ks + j], 8);
                buf[ks + j] = t_pos[i * ks + j] + row_offset;
                buf[2 * ks + j] = t_ids[i * ks + j];
            }
            const unsigned long long c64 = c;
            VL_HIP(hipMemcpy(sink.cnt + qi, &c64, 8, hipMemcpyHostToDevice));
            VL_HIP(hipMemcpy(sink.score_bits + qi * ks, buf.data(), ks * 8, hipMemcpyHostToDevice));
            VL_HIP(hipMemcpy(sink.gpos + qi * ks, buf.data() + ks, ks * 8, hipMemcpyHostToDevice));
            VL_HIP(hipMemcpy(sink.ids + qi * ks, buf.data() + 2 * ks, ks * 8, hipMemcpyHostToDevice));
        }
    }
    *handled = true;
    return OK;
}

int GpuFlatIndex::ensure_mfma_scratch(Workspace* ws) const
{
    if (ws->mf.nq_cap) return OK;
    const size_t nqc = MFMA_MAX_BATCH;
    const uint32_t ldb = mfma_ldb((uint32_t)dim_);
    const size_t nqp = nqc + 256;  // whole query chunks (128 / 96 / 256 queries) past the last query
    VL_HIP(hipMalloc(&ws->mf.q_bf16, nqp * ldb * 2));
    VL_TRY(dev_alloc(&ws->mf.gmax, nqc * MFMA_GROUPS));
    VL_TRY(dev_alloc(&ws->mf.thr, nqc));
    VL_TRY(dev_alloc(&ws->mf.cand, nqc * MFMA_CAND_CAP));
    VL_TRY(dev_alloc(&ws->mf.cnt, nqp));
    VL_TRY(dev_alloc(&ws->mf_d_q64, nqc * (dim_ + 1)));
    VL_TRY(pinned_alloc(&ws->mf_h_q64, nqc * (dim_ + 1)));
    VL_TRY(dev_alloc(&ws->mf_lists, nqc * KP));
    VL_TRY(dev_alloc(&ws->mf_scores, nqc * KP));
    VL_TRY(pinned_alloc(&ws->mf_h_result, 2 * nqc));  // two launch sequences in flight: one being unpacked, one on the GPU
    VL_TRY(pinned_alloc(&ws->mf_h_dom, 2 * nqc));
    for (auto& e : ws->mf_ev_done) VL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    VL_HIP(hipEventCreateWithFlags(&ws->mf_ev_h2d, hipEventDisableTiming));
    ws->mf.nq_cap = (uint32_t)nqc;
    ws->mf.nq_pad_cap = (uint32_t)nqp;
    return OK;
}

// Cosine / dot / Euclidean batches of >= MFMA_MIN_BATCH queries.  done[qi] is set for every query
// answered here; the caller redoes the others (bound check failed, candidate overflow, ...).
int GpuFlatIndex::search_batch_mfma(Workspace* ws, const double* queries, const double* d_queries, uint64_t nq,
                                    uint64_t k, uint64_t k_eff, int metric, uint64_t* out_pos, uint64_t* out_ids,
                                    double* out_scores, uint64_t* out_n, std::vector<uint8_t>* done,
                                    const ShardRecordSink* sink) const
{
    const uint64_t n = ids_.size();
    const bool frag = mfma_rows_kernel((uint32_t)dim_);  // which kernel, hence which slab layout
    VL_TRY(ensure_bf16_slab(frag));
    const void* slab16 = frag ? d_slab16f_ : d_slab16_;
    VL_TRY(ensure_mfma_scratch(ws));
    hipStream_t st = ws->stream;
    // (2 + u) * u with u = 2^-8 + 2^-23 (f64 -> f32 -> bf16 double rounding), see DESIGN.md
    const double in_extra = 0.0079;
    const bool prof = profile_.load();
    static const bool trace = getenv("VL_TRACE_BATCH") != nullptr;  // diagnostic: host-side phases of a sequence on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
    };
    const uint64_t seq = mfma_sequence_queries((uint32_t)dim_);
    // Launch sequences run as a two-deep pipeline: sequence s + 1 is staged and enqueued BEFORE the host waits for
    // sequence s, so the GPU goes from one sequence's finalize straight into the next one's filter while the host
    // unpacks result blocks (config 5 = two sequences of 2048 queries: rounds 1-3 ran them back to back with a stream
    // sync, the unpacking and the next staging in between).  Device buffers are shared: stream order keeps one sequence's
    // kernels behind the previous one's.  What the host touches comes in two copies (result blocks, in-domain flags), and
    // the one pinned staging area of host queries is reused only after its copies have left (ev_h2d).
    struct SeqState {
        uint64_t q0 = 0;
        uint32_t g = 0;
        bool live = false;
        std::vector<uint8_t> in_domain;
        std::chrono::steady_clock::time_point t0, t1, t2;
    } sq[2];
    const bool pipelined = !prof;  // (the profile's event pair brackets one sequence at a time)
    auto enqueue = [&](int slot, uint64_t q0) -> int {
        SeqState& S = sq[slot];
        S.q0 = q0;
        S.g = (uint32_t)std::min<uint64_t>(seq, nq - q0);
        S.live = true;
        S.t0 = now();
        const uint32_t g = S.g;
        S.in_domain.assign(g, 0);
        SearchResultBlock* res = ws->mf_h_result + (size_t)slot * MFMA_MAX_BATCH;
        unsigned char* dom = ws->mf_h_dom + (size_t)slot * MFMA_MAX_BATCH;
        bool prepared = false;
        if (d_queries) {  // already on this GPU: staged by a kernel; the flags land in pinned memory, read after the wait
            hipError_t pe = hipSuccess;
            prepared = frag && launch_prepare_queries(st, d_queries + q0 * dim_, g, (uint32_t)dim_, DOMAIN_MAX_ABS, DOMAIN_MIN_NORM,
                                                      ws->mf_d_q64, ws->mf_d_q64 + (size_t)g * dim_, dom, ws->mf, &pe);
            VL_HIP(pe);
            if (!prepared)
                VL_HIP(launch_stage_queries(st, d_queries + q0 * dim_, g, (uint32_t)dim_, DOMAIN_MAX_ABS, DOMAIN_MIN_NORM,
                                            ws->mf_d_q64, ws->mf_d_q64 + (size_t)g * dim_, dom));
        } else {
            // Host queries go to pinned memory (domain test, norm) and over PCIe in pieces: the copy of one piece runs
            // while the host stages the next, so a 6.3 MB batch (1024 x 768) costs about its staging time alone
            // (170 us) instead of staging + copy (170 + 115 us) in front of the first kernel.
            if (ws->mf_h2d_pending) {  // the previous sequence's copies out of this staging area
                VL_HIP(hipEventSynchronize(ws->mf_ev_h2d));
                ws->mf_h2d_pending = false;
            }
            double* norms = ws->mf_h_q64 + (size_t)g * dim_;
            const uint32_t piece = (uint32_t)std::max<uint64_t>(16, (1u << 20) / (dim_ * sizeof(double)));  // ~1 MB
            for (uint32_t j0 = 0; j0 < g; j0 += piece) {
                const uint32_t j1 = std::min<uint32_t>(g, j0 + piece);
                for (uint32_t j = j0; j < j1; ++j)
                    S.in_domain[j] = stage_query(queries + (q0 + j) * dim_, ws->mf_h_q64 + (size_t)j * dim_, dim_, &norms[j]) ? 1 : 0;
                VL_HIP(hipMemcpyAsync(ws->mf_d_q64 + (size_t)j0 * dim_, ws->mf_h_q64 + (size_t)j0 * dim_,
                                      (size_t)(j1 - j0) * dim_ * sizeof(double), hipMemcpyHostToDevice, st));
            }
            VL_HIP(hipMemcpyAsync(ws->mf_d_q64 + (size_t)g * dim_, norms, (size_t)g * sizeof(double), hipMemcpyHostToDevice, st));
            VL_HIP(hipEventRecord(ws->mf_ev_h2d, st));
            ws->mf_h2d_pending = true;
        }
        S.t1 = now();
        if (prof) VL_HIP(hipEventRecord(ws->ev0, st));
        MfmaLaunchInfo li;
        VL_HIP(launch_mfma_candidates(st, metric, slab16, d_norm16_, d_sqnorm_, ws->mf_d_q64, g, n, (uint32_t)dim_,
                                      ws->mf, ws->mf_lists, &li, prepared));
        {
            const int v[6] = {li.ksteps, li.metric, li.chunks, li.grid_x, li.stages, li.sample_blocks};
            for (int i = 0; i < 6; ++i) last_filter_[i].store(v[i], std::memory_order_relaxed);
        }
        if (prof) VL_HIP(hipEventRecord(ws->ev1, st));
        ShardRecordSink seq_sink;
        if (sink) {  // this sequence's queries sit at [q0, q0 + g) of the record
            seq_sink = *sink;
            seq_sink.q0 = (uint32_t)q0;
        }
        // batches: the rescoring split over four workgroups per query + a rank / emit launch (launch_batch_finalize); a handful
        // of queries keeps the one-workgroup-per-query kernel (one launch less)
        static const bool split_off = []() { const char* v = getenv("VL_BATCH_FINALIZE"); return v && v[0] == '0'; }();
        if (g >= 8 && !split_off)
            VL_HIP(launch_batch_finalize(st, metric, ws->mf_lists, (int)g, d_master_, ws->mf_d_q64, ws->mf_d_q64 + (size_t)g * dim_,
                                         (uint32_t)dim_, n, (uint32_t)k_eff, max_row_norm_, res, in_extra, ws->mf_scores,
                                         sink ? &seq_sink : nullptr));
        else
            VL_HIP(launch_merge_finalize(st, metric, ws->mf_lists, 1, (int)g, d_master_, ws->mf_d_q64,
                                         ws->mf_d_q64 + (size_t)g * dim_, (uint32_t)dim_, n, (uint32_t)k_eff, max_row_norm_,
                                         res, in_extra, 0, sink ? &seq_sink : nullptr));
        VL_HIP(hipEventRecord(ws->mf_ev_done[slot], st));
        S.t2 = now();
        return OK;
    };
    auto consume = [&](int slot) -> int {
        SeqState& S = sq[slot];
        if (!S.live) return OK;
        S.live = false;
        const uint32_t g = S.g;
        const SearchResultBlock* res = ws->mf_h_result + (size_t)slot * MFMA_MAX_BATCH;
        const unsigned char* dom = ws->mf_h_dom + (size_t)slot * MFMA_MAX_BATCH;
        VL_HIP(hipEventSynchronize(ws->mf_ev_done[slot]));
        if (d_queries)
            for (uint32_t j = 0; j < g; ++j) S.in_domain[j] = dom[j];
        const auto t_3 = now();
        if (prof) {
            float ms = 0.f;
            VL_HIP(hipEventElapsedTime(&ms, ws->ev0, ws->ev1));
            std::lock_guard<std::mutex> gl(prof_mu_);
            prof_n_ += 1;
            prof_ms_ += ms;
            prof_bytes_ += n * (uint64_t)mfma_ldb((uint32_t)dim_) * 2;
        }
        for (uint32_t j = 0; j < g; ++j) {
            const uint64_t qi = S.q0 + j;
            const SearchResultBlock& r = res[j];
            if (!S.in_domain[j] || (r.flags & RESULT_NEEDS_EXACT) || r.n_out != k_eff) continue;
            bool ok = true;
            for (uint64_t i = 0; i < k_eff; ++i) ok = ok && r.pos[i] < n;
            if (!ok) {
                set_last_error("MFMA path returned an out-of-range position (kernel bug)");
                return ERR_DEVICE;
            }
            for (uint64_t i = 0; i < k_eff; ++i) {
                const uint32_t p = r.pos[i];
                if (out_pos) out_pos[qi * k + i] = p;
                if (out_ids) out_ids[qi * k + i] = ids_[p];
                out_scores[qi * k + i] = r.score[i];
            }
            out_n[qi] = k_eff;
            (*done)[qi] = 1;
        }
        if (trace) {
            const auto t4 = now();
            fprintf(stderr, "[vl batch] %u queries: stage %ld us, enqueue %ld us, wait %ld us, unpack %ld us\n", g, us(S.t0, S.t1),
                    us(S.t1, S.t2), us(S.t2, t_3), us(t_3, t4));
        }
        return OK;
    };
    int rc = OK;
    int slot = 0;
    for (uint64_t q0 = 0; q0 < nq && rc == OK; q0 += seq, slot ^= 1) {
        rc = consume(slot);  // the sequence that used this slot's host buffers two steps ago (nothing on the first two turns)
        if (rc == OK) rc = enqueue(slot, q0);
        if (rc == OK && !pipelined) rc = consume(slot);
    }
    // drain (also on an error: nothing of this call may still be writing result blocks when the workspace goes back)
    for (int d = 0; d < 2; ++d, slot ^= 1) {
        const int r2 = rc == OK ? consume(slot) : OK;
        if (rc == OK) rc = r2;
    }
    if (rc != OK) (void)hipStreamSynchronize(st);
    return rc;
}

// Queries in device memory.  The MFMA batch path takes them where they are; everything else goes through the host.
int GpuFlatIndex::search_batch_device(const double* d_queries, uint64_t nq, uint64_t q_len, uint64_t k, int metric,
                                      uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n) const
{
    if (nq == 0) return OK;
    if (!out_n) return ERR_INVALID_ARG;
    for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
    if (metric < 0 || metric > 3) {
        set_last_error("unknown metric");
        return ERR_INVALID_ARG;
    }
    VL_HIP(hipSetDevice(device_));
    auto via_host = [&]() -> int {  // the reference's surface with host queries: every check and path of search_batch()
        std::vector<double> h;
        if (d_queries && q_len) {
            try {
                h.resize((size_t)nq * q_len);
            } catch (const std::bad_alloc&) {
                set_last_error("out of host memory copying the queries");
                return ERR_OOM;
            }
            VL_HIP(hipMemcpy(h.data(), d_queries, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        return search_batch(d_queries ? h.data() : nullptr, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n);
    };
    std::shared_lock<RwLock> lk(mu_);
    const uint64_t n = ids_.size();
    // the checks of search_batch(), before any byte of the queries is touched (src/index/flat.rs:99-104: an empty index
    // accepts any query length)
    if (n != 0 && q_len != dim_) {
        set_dim_mismatch(dim_, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(q_len));
        return ERR_DIM_MISMATCH;
    }
    if (n == 0 || k == 0) return OK;
    if (!d_queries || !out_scores) return ERR_INVALID_ARG;
    const uint64_t k_eff = std::min<uint64_t>(k, n);
    const char* mf_env = getenv("VL_MFMA");
    const char* mf_min = getenv("VL_MFMA_MIN_BATCH");
    const uint64_t mfma_min = mf_min && *mf_min ? (uint64_t)atoi(mf_min) : (uint64_t)MFMA_MIN_BATCH;
    const bool direct = nq > 1 && force_path_.load() == 0 &&
                        k_eff <= (uint64_t)KFAST_MAX && n_out_of_domain_ == 0 && !(mf_env && mf_env[0] == '0') &&
                        nq >= mfma_min && n >= MFMA_MIN_ROWS && mfma_scan_supported((uint32_t)dim_, metric);
    if (!direct) {
        lk.unlock();
        return via_host();
    }
    std::vector<uint8_t> done(nq, 0);
    {
        Workspace* ws = acquire_ws();
        if (!ws) return ERR_DEVICE;
        struct Releaser {
            const GpuFlatIndex* self;
            Workspace* ws;
            ~Releaser()
            {
                (void)hipStreamSynchronize(ws->stream);
                self->release_ws(ws);
            }
        } rel{this, ws};
        VL_TRY(search_batch_mfma(ws, nullptr, d_queries, nq, k, k_eff, metric, out_pos, out_ids, out_scores, out_n, &done));
    }
    set_last_path(PATH_FAST);
    // what the filter could not certify (ties, candidate overflow, a query outside the fast-path domain): those queries
    // go to the host and through search_batch(), results scattered back
    std::vector<uint64_t> left;
    for (uint64_t qi = 0; qi < nq; ++qi)
        if (!done[qi]) left.push_back(qi);
    if (left.empty()) return OK;
    const uint64_t m = left.size();
    std::vector<double> hq;
    std::vector<uint64_t> t_pos, t_ids, t_n;
    std::vector<double> t_sc;
    try {
        hq.resize(m * dim_);
        t_pos.resize(m * k_eff);
        t_ids.resize(m * k_eff);
        t_sc.resize(m * k_eff);
        t_n.resize(m);
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory answering the queries the MFMA filter could not certify");
        return ERR_OOM;
    }
    for (uint64_t i = 0; i < m; ++i)
        VL_HIP(hipMemcpy(hq.data() + i * dim_, d_queries + left[i] * dim_, dim_ * sizeof(double), hipMemcpyDeviceToHost));
    VL_TRY(search_batch_locked(hq.data(), m, dim_, k_eff, metric, t_pos.data(), t_ids.data(), t_sc.data(), t_n.data()));  // still under the caller's shared lock: one index state for the whole batch
    for (uint64_t i = 0; i < m; ++i) {
        const uint64_t qi = left[i];
        for (uint64_t j = 0; j < t_n[i]; ++j) {
            if (out_pos) out_pos[qi * k + j] = t_pos[i * k_eff + j];
            if (out_ids) out_ids[qi * k + j] = t_ids[i * k_eff + j];
            out_scores[qi * k + j] = t_sc[i * k_eff + j];
        }
        out_n[qi] = t_n[i];
    }
    return OK;
}

// ---------------------------------------------------------------------------------------------
// point lookups / export / clone
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::get_vector(uint64_t id, double* out) const
{
    if (!out && dim_) return ERR_INVALID_ARG;
    std::shared_lock<RwLock> lk(mu_);
    auto it = std::find(ids_.begin(), ids_.end(), id);  // first match (src/index/flat.rs:129-131)
    if (it == ids_.end()) return ERR_NOT_FOUND;
    const uint64_t pos = (uint64_t)(it - ids_.begin());
    if (dim_) {
        VL_HIP(hipSetDevice(device_));
        VL_HIP(hipMemcpy(out, d_master_ + pos * dim_, dim_ * sizeof(double), hipMemcpyDeviceToHost));
    }
    return OK;
}

int GpuFlatIndex::max_id(uint64_t* out) const
{
    if (!out) return ERR_INVALID_ARG;
    std::shared_lock<RwLock> lk(mu_);
    if (ids_.empty()) return ERR_NOT_FOUND;
    *out = *std::max_element(ids_.begin(), ids_.end());
    return OK;
}

int GpuFlatIndex::export_rows(uint64_t* out_ids, double* out_values) const
{
    std::shared_lock<RwLock> lk(mu_);
    const uint64_t n = ids_.size();
    if (n == 0) return OK;
    if (!out_ids || (!out_values && dim_)) return ERR_INVALID_ARG;
    std::memcpy(out_ids, ids_.data(), n * sizeof(uint64_t));
    if (dim_) {
        VL_HIP(hipSetDevice(device_));
        VL_HIP(hipMemcpy(out_values, d_master_, n * dim_ * sizeof(double), hipMemcpyDeviceToHost));
    }
    return OK;
}

int GpuFlatIndex::clone(GpuFlatIndex** out) const
{
    if (!out) return ERR_INVALID_ARG;
    *out = nullptr;
    std::shared_lock<RwLock> lk(mu_);
    GpuFlatIndex* c = nullptr;
    VL_TRY(create(dim_, device_, &c));
    std::unique_ptr<GpuFlatIndex> guard(c);
    if (!ids_.empty())
        VL_TRY(c->add_bulk(ids_.data(), d_master_, ids_.size(), /*validate=*/false, /*values_on_device=*/true));
    c->force_path_.store(force_path_.load());
    *out = guard.release();
    return OK;
}

// ---------------------------------------------------------------------------------------------
// HNSW distance callbacks (src/index/hnsw.rs:113-174)
// ---------------------------------------------------------------------------------------------
int GpuFlatIndex::hnsw_distances(const double* query, uint64_t q_len, int metric, const uint64_t* positions,
                                 uint64_t m, uint64_t* out) const
{
    if (metric < 0 || metric > 3) return ERR_INVALID_ARG;
    std::shared_lock<RwLock> lk(mu_);
    if (q_len != dim_) {
        set_dim_mismatch(dim_, q_len);
        set_last_error("Dimension mismatch: expected " + std::to_string(dim_) + ", got " + std::to_string(q_len));
        return ERR_DIM_MISMATCH;
    }
    if (m == 0) return OK;
    if (!positions || !out || (!query && dim_)) return ERR_INVALID_ARG;
    const uint64_t n = ids_.size();
    std::vector<uint32_t> p32(m);
    for (uint64_t i = 0; i < m; ++i) {
        if (positions[i] >= n) {
            set_last_error("position out of range");
            return ERR_NOT_FOUND;
        }
        p32[i] = (uint32_t)positions[i];
    }
    VL_HIP(hipSetDevice(device_));
    Workspace* ws = acquire_ws();
    if (!ws) return ERR_DEVICE;
    struct Releaser {
        const GpuFlatIndex* self;
        Workspace* ws;
        ~Releaser()
        {
            (void)hipStreamSynchronize(ws->stream);
            self->release_ws(ws);
        }
    } rel{this, ws};
    hipStream_t st = ws->stream;
    if (ws->hn_cap < m) {
        if (ws->d_positions) (void)hipFree(ws->d_positions);
        if (ws->d_dists) (void)hipFree(ws->d_dists);
        ws->d_positions = nullptr;
        ws->d_dists = nullptr;
        ws->hn_cap = 0;
        const size_t cap = std::max<size_t>(m, 256);
        VL_TRY(dev_alloc(&ws->d_positions, cap));
        VL_TRY(dev_alloc(&ws->d_dists, cap));
        ws->hn_cap = cap;
    }
    for (uint64_t i = 0; i < dim_; ++i) ws->h_q64[i] = query[i];
    if (dim_) VL_HIP(hipMemcpyAsync(ws->d_q64, ws->h_q64, dim_ * sizeof(double), hipMemcpyHostToDevice, st));
    VL_HIP(hipMemcpyAsync(ws->d_positions, p32.data(), m * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    VL_HIP(launch_hnsw_distances(st, metric, d_master_, ws->d_q64, (uint32_t)dim_, ws->d_positions, (uint32_t)m,
                                 ws->d_dists));
    VL_HIP(hipMemcpyAsync(out, ws->d_dists, m * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    VL_HIP(hipStreamSynchronize(st));
    return OK;
}

// Ingest step in front of add (src/embeddings.rs:169-181): f32 embeddings are widened and
// normalised on the device in chunks (launch_embed_f32) and handed to `append` as device-resident f64 rows, so
// the host never holds an f64 copy and PCIe carries 4 bytes per value instead of 8.
int add_embeddings_f32(int device, uint64_t dim, const uint64_t* ids, const float* emb, uint64_t n, bool normalize,
                       bool emb_on_device, const std::function<int(const uint64_t*, const double*, uint64_t)>& append)
{
    if (n == 0) return OK;
    if (!ids || (!emb && dim)) return ERR_INVALID_ARG;
    VL_HIP(hipSetDevice(device));
    const uint64_t row_bytes = std::max<uint64_t>(dim, 1) * sizeof(double);
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (512ull << 20) / row_bytes));
    struct Scratch {
        hipStream_t st = nullptr;
        double* rows = nullptr;
        float* staged = nullptr;
        ~Scratch()
        {
            if (rows) (void)hipFree(rows);
            if (staged) (void)hipFree(staged);
            if (st) (void)hipStreamDestroy(st);
        }
    } sc;
    VL_HIP(hipStreamCreateWithFlags(&sc.st, hipStreamNonBlocking));
    VL_HIP(hipMalloc(&sc.rows, chunk * row_bytes));
    if (!emb_on_device) VL_HIP(hipMalloc(&sc.staged, chunk * std::max<uint64_t>(dim, 1) * sizeof(float)));
    for (uint64_t done = 0; done < n; done += chunk) {
        const uint64_t c = std::min(chunk, n - done);
        const float* src = emb + done * dim;
        if (!emb_on_device && dim) {
            VL_HIP(hipMemcpyAsync(sc.staged, src, c * dim * sizeof(float), hipMemcpyHostToDevice, sc.st));
            src = sc.staged;
        }
        VL_HIP(launch_embed_f32(sc.st, src, c, (uint32_t)dim, normalize, sc.rows));
        VL_HIP(hipStreamSynchronize(sc.st));
        VL_TRY(append(ids + done, sc.rows, c));  // stops at the first duplicate id like n sequential add() calls
    }
    return OK;
}

int search_embeddings_f32(int device, uint64_t dim, const float* emb, uint64_t nq, bool normalize, bool emb_on_device,
                          const std::function<int(const double*, uint64_t)>& search)
{
    if (nq == 0) return OK;
    if (!emb && dim) return ERR_INVALID_ARG;
    if (dim == 0 || nq > (1ull << 31) / std::max<uint64_t>(dim, 1) * 64) {  // 16 GiB of f64 queries: not a batch
        set_last_error("query batch too large (or dimension 0)");
        return ERR_INVALID_ARG;
    }
    VL_HIP(hipSetDevice(device));
    // conversion scratch (a stream, the f64 queries, the staged f32 embeddings) comes from a small process-wide pool: a
    // hipMalloc / hipFree pair per call would cost more than the conversion kernel and synchronise the device
    struct Scratch {
        int device = 0;
        hipStream_t st = nullptr;
        double* rows = nullptr;
        size_t rows_cap = 0;
        float* staged = nullptr;
        size_t staged_cap = 0;
        ~Scratch()
        {
            (void)hipSetDevice(device);
            if (rows) (void)hipFree(rows);
            if (staged) (void)hipFree(staged);
            if (st) (void)hipStreamDestroy(st);
        }
    };
    static std::mutex pool_mu;
    static std::vector<std::unique_ptr<Scratch>> pool;  // idle scratches (at most 8 are kept)
    std::unique_ptr<Scratch> scp;
    {
        std::lock_guard<std::mutex> g(pool_mu);
        for (size_t i = 0; i < pool.size(); ++i)
            if (pool[i]->device == device) {
                scp = std::move(pool[i]);
                pool.erase(pool.begin() + (long)i);
                break;
            }
    }
    if (!scp) {
        scp.reset(new Scratch());
        scp->device = device;
        VL_HIP(hipStreamCreateWithFlags(&scp->st, hipStreamNonBlocking));
    }
    struct Return {
        std::unique_ptr<Scratch>& p;
        ~Return()
        {
            std::lock_guard<std::mutex> g(pool_mu);
            if (p && pool.size() < 8) pool.push_back(std::move(p));
        }
    } ret{scp};
    Scratch& sc = *scp;
    const size_t need = (size_t)nq * dim;
    if (sc.rows_cap < need) {
        if (sc.rows) (void)hipFree(sc.rows);
        sc.rows = nullptr;
        sc.rows_cap = 0;
        VL_HIP(hipMalloc(&sc.rows, need * sizeof(double)));
        sc.rows_cap = need;
    }
    const float* src = emb;
    if (!emb_on_device) {
        if (sc.staged_cap < need) {
            if (sc.staged) (void)hipFree(sc.staged);
            sc.staged = nullptr;
            sc.staged_cap = 0;
            VL_HIP(hipMalloc(&sc.staged, need * sizeof(float)));
            sc.staged_cap = need;
        }
        VL_HIP(hipMemcpyAsync(sc.staged, emb, need * sizeof(float), hipMemcpyHostToDevice, sc.st));
        src = sc.staged;
    }
    VL_HIP(launch_embed_f32(sc.st, src, nq, (uint32_t)dim, normalize, sc.rows));
    VL_HIP(hipStreamSynchronize(sc.st));
    return search(sc.rows, nq);
}

}  // namespace vl
