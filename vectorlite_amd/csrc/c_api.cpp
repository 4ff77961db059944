// c_api.cpp -- extern "C" boundary (include/vectorlite_amd.h) over vl::GpuFlatIndex.
// No C++ exception crosses it: every entry point catches and maps to a vl_status.
#include "../../include/vectorlite_amd.h"

#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <vector>

#include "flat_index.hpp"
#include "hnsw_index.hpp"
#include "multi_index.hpp"
#include "shard.hpp"
#include "vlc_loader.hpp"

// enum VectorIndexWrapper { Flat(FlatIndex), HNSW(Box<HNSWIndex>) } (src/lib.rs:271-276): exactly one
// of the pointers is set; the vl_index_* entry points dispatch like the wrapper's impl
// (src/lib.rs:278-327).
struct vl_index {
    vl::GpuFlatIndex* flat;
    vl::HnswIndex* hnsw;
    vl::MultiFlatIndex* multi;  // a flat index over several GPUs (multi_index.hpp): behaves as VectorIndexWrapper::Flat
};

// one rank's end of the RCCL communicator of a row-sharded index (shard.hpp)
struct vl_comm {
    vl::ShardComm* c;
};

namespace {
constexpr int VL_ABI_VERSION = 1;

template <typename F>
int guarded(F&& f)
{
    // HIP keeps a sticky per-thread error: a call that failed earlier on this thread -- in this library (a reservation
    // beyond the card, a device ordinal that does not exist) or in the host application -- would be reported by the
    // next kernel launch's hipGetLastError() as that launch's own failure.  Every entry point starts from a clean slate.
    (void)hipGetLastError();
    try {
        return f();
    } catch (const std::bad_alloc&) {
        vl::set_last_error("host allocation failed");
        return VL_ERR_OOM;
    } catch (const std::exception& e) {
        vl::set_last_error(std::string("internal error: ") + e.what());
        return VL_ERR_DEVICE;
    } catch (...) {
        vl::set_last_error("internal error");
        return VL_ERR_DEVICE;
    }
}

// the flat flavours share one method surface (GpuFlatIndex / MultiFlatIndex): f is a generic lambda
template <typename F>
int on_flat(const vl_index* h, F&& f)
{
    return h->multi ? f(h->multi) : f(h->flat);
}

int wrap(vl::GpuFlatIndex* idx, vl_index** out)
{
    vl_index* h = new (std::nothrow) vl_index{idx, nullptr, nullptr};
    if (!h) {
        delete idx;
        return VL_ERR_OOM;
    }
    *out = h;
    return VL_OK;
}
}  // namespace

extern "C" {

int vl_flat_create(uint64_t dim, int device, vl_index** out)
{
    return guarded([&]() -> int {
        if (!out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::GpuFlatIndex* idx = nullptr;
        int rc = vl::GpuFlatIndex::create(dim, device, &idx);
        if (rc != VL_OK) return rc;
        return wrap(idx, out);
    });
}

int vl_flat_from_rows(uint64_t dim, const uint64_t* ids, const double* values, uint64_t n, int device,
                      vl_index** out)
{
    return guarded([&]() -> int {
        if (!out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::GpuFlatIndex* idx = nullptr;
        int rc = vl::GpuFlatIndex::create(dim, device, &idx);
        if (rc != VL_OK) return rc;
        rc = idx->add_bulk(ids, values, n, /*validate=*/false, /*values_on_device=*/false);
        if (rc != VL_OK) {
            delete idx;
            return rc;
        }
        return wrap(idx, out);
    });
}

int vl_flat_create_multi(uint64_t dim, const int* device_ids, int n_dev, int mode, vl_index** out)
{
    return guarded([&]() -> int {
        if (!out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::MultiFlatIndex* m = nullptr;
        const int rc = vl::MultiFlatIndex::create(dim, device_ids, n_dev, mode, &m);
        if (rc != VL_OK) return rc;
        vl_index* h = new (std::nothrow) vl_index{nullptr, nullptr, m};
        if (!h) {
            delete m;
            return VL_ERR_OOM;
        }
        *out = h;
        return VL_OK;
    });
}

int vl_index_parts(const vl_index* h, int* n_parts, int* mode, uint64_t* rows, uint64_t* searches, int capacity)
{
    if (!h) return VL_ERR_INVALID_ARG;
    if (!h->multi) {  // a single-GPU handle is one part
        if (n_parts) *n_parts = 1;
        if (mode) *mode = -1;
        if (capacity >= 1) {
            if (rows) rows[0] = vl_index_len(h);
            if (searches) searches[0] = 0;
        }
        return VL_OK;
    }
    const int P = h->multi->n_parts();
    if (n_parts) *n_parts = P;
    if (mode) *mode = h->multi->mode();
    if (capacity >= P) h->multi->part_stats(rows, searches);
    return VL_OK;
}

int vl_hnsw_create_ex(uint64_t dim, int metric, uint32_t m, uint32_t m0, uint32_t ef_construction, uint64_t seed,
                      int device, vl_index** out)
{
    return guarded([&]() -> int {
        if (!out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::HnswParams p;
        p.m = m;
        p.m0 = m0;
        p.ef_construction = ef_construction;
        p.seed = seed;
        vl::HnswIndex* idx = nullptr;
        int rc = vl::HnswIndex::create(dim, metric, p, device, &idx);
        if (rc != VL_OK) return rc;
        vl_index* h = new (std::nothrow) vl_index{nullptr, idx, nullptr};
        if (!h) {
            delete idx;
            return VL_ERR_OOM;
        }
        *out = h;
        return VL_OK;
    });
}

int vl_hnsw_create(uint64_t dim, int metric, int device, vl_index** out)
{
    const vl::HnswParams d;  // default cargo profile: M = 16, M0 = 32 (src/index/hnsw.rs:95-109)
    return vl_hnsw_create_ex(dim, metric, d.m, d.m0, d.ef_construction, d.seed, device, out);
}

#define VL_FLAT_ONLY(h)                                                         \
    if (!(h) || !(h)->flat) {                                                   \
        vl::set_last_error("this entry point needs a single-GPU flat index handle"); \
        return VL_ERR_INVALID_ARG;                                              \
    }

int vl_index_clone(const vl_index* h, vl_index** out)
{
    return guarded([&]() -> int {
        if (!h || !out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        if (h->hnsw) {
            vl::HnswIndex* hc = nullptr;
            int rc = h->hnsw->clone(&hc);
            if (rc != VL_OK) return rc;
            vl_index* w = new (std::nothrow) vl_index{nullptr, hc, nullptr};
            if (!w) {
                delete hc;
                return VL_ERR_OOM;
            }
            *out = w;
            return VL_OK;
        }
        if (h->multi) {
            vl::MultiFlatIndex* mc = nullptr;
            int rc = h->multi->clone(&mc);
            if (rc != VL_OK) return rc;
            vl_index* w = new (std::nothrow) vl_index{nullptr, nullptr, mc};
            if (!w) {
                delete mc;
                return VL_ERR_OOM;
            }
            *out = w;
            return VL_OK;
        }
        vl::GpuFlatIndex* c = nullptr;
        int rc = h->flat->clone(&c);
        if (rc != VL_OK) return rc;
        return wrap(c, out);
    });
}

void vl_index_destroy(vl_index* h)
{
    if (!h) return;
    try {
        delete h->flat;
        delete h->hnsw;
        delete h->multi;
    } catch (...) {
    }
    delete h;
}

int vl_index_reserve(vl_index* h, uint64_t n_rows)
{
    return guarded([&]() -> int {
        if (h && h->hnsw) return VL_OK;  // the graph grows on demand
        if (!h) return VL_ERR_INVALID_ARG;
        return on_flat(h, [&](auto* f) { return f->reserve(n_rows); });
    });
}

int vl_index_add(vl_index* h, uint64_t id, const double* values, uint64_t len)
{
    return guarded([&]() -> int {
        if (!h) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->add(id, values, len);
        return on_flat(h, [&](auto* f) { return f->add(id, values, len); });
    });
}

int vl_index_add_bulk(vl_index* h, const uint64_t* ids, const double* values, uint64_t n, int validate,
                      int values_on_device)
{
    return guarded([&]() -> int {
        if (!h) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->add_bulk(ids, values, n, values_on_device != 0);  // HNSW add always validates
        return on_flat(h, [&](auto* f) { return f->add_bulk(ids, values, n, validate != 0, values_on_device != 0); });
    });
}

int vl_index_add_embeddings_f32(vl_index* h, const uint64_t* ids, const float* embeddings, uint64_t n, int normalize,
                                int validate, int embeddings_on_device)
{
    return guarded([&]() -> int {
        if (!h) return VL_ERR_INVALID_ARG;
        if (h->hnsw)
            return vl::add_embeddings_f32(h->hnsw->device(), h->hnsw->dimension(), ids, embeddings, n, normalize != 0,
                                          embeddings_on_device != 0,
                                          [&](const uint64_t* i, const double* rows, uint64_t c) {
                                              return h->hnsw->add_bulk(i, rows, c, true);
                                          });
        return on_flat(h, [&](auto* f) {
            return vl::add_embeddings_f32(f->device(), f->dimension(), ids, embeddings, n, normalize != 0, embeddings_on_device != 0,
                                          [&](const uint64_t* i, const double* rows, uint64_t c) {
                                              return f->add_bulk(i, rows, c, validate != 0, true);
                                          });
        });
    });
}

int vl_index_delete(vl_index* h, uint64_t id)
{
    return guarded([&]() -> int {
        if (!h) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->remove(id);
        return on_flat(h, [&](auto* f) { return f->remove(id); });
    });
}

int vl_index_search(const vl_index* h, const double* query, uint64_t q_len, uint64_t k, int metric,
                    uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || !out_n) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->search(query, q_len, k, metric, 0, out_ids, out_scores, out_n);
        if (!out_ids && k != 0 && vl_index_len(h) != 0) return VL_ERR_INVALID_ARG;
        return on_flat(h, [&](auto* f) { return f->search(query, q_len, k, metric, nullptr, out_ids, out_scores, out_n); });
    });
}

int vl_index_search_cap(const vl_index* h, const double* query, uint64_t q_len, uint64_t k, int metric,
                        uint64_t out_capacity, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    // Flat: results for a smaller k are a prefix of those for a larger k (one total order: score desc, insertion
    // order), so truncating k to the capacity IS writing the first `out_capacity` entries of the k results.
    // HNSW: k also sets the walk's beam (ef = min(k, len), src/index/hnsw.rs:437), so the walk keeps the caller's k and
    // only the copy-out is capped.
    if (h && h->hnsw && k > out_capacity)
        return guarded([&]() -> int {
            if (!out_n) return VL_ERR_INVALID_ARG;
            if (out_capacity == 0) {  // nothing may be written: the walk's errors are still the caller's to see
                uint64_t id1 = 0;
                double sc1 = 0.0;
                const int rc = h->hnsw->search(query, q_len, k, metric, 0, &id1, &sc1, out_n, 1);
                *out_n = 0;
                return rc;
            }
            return h->hnsw->search(query, q_len, k, metric, 0, out_ids, out_scores, out_n, out_capacity);
        });
    return vl_index_search(h, query, q_len, k < out_capacity ? k : out_capacity, metric, out_ids, out_scores, out_n);
}

int vl_index_search_batch_cap(const vl_index* h, const double* queries, uint64_t nq, uint64_t q_len, uint64_t k,
                              int metric, uint64_t out_stride, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    if (h && h->hnsw)  // the walk keeps the caller's k (its beam is ef = min(k, len)); rows out_stride apart, capped at out_stride
        return guarded([&]() -> int {
            if (!out_n && nq) return VL_ERR_INVALID_ARG;
            if (nq == 0) return VL_OK;
            if (out_stride == 0) {
                for (uint64_t q = 0; q < nq; ++q) out_n[q] = 0;
                return VL_OK;
            }
            return h->hnsw->search_batch(queries, nq, q_len, k, metric, 0, out_ids, out_scores, out_n, out_stride);
        });
    if (k >= out_stride)  // rows are then exactly out_stride apart, which is the plain call's layout for k = out_stride
        return vl_index_search_batch(h, queries, nq, q_len, out_stride, metric, out_ids, out_scores, out_n);
    return guarded([&]() -> int {
        if (!h || (!out_n && nq)) return VL_ERR_INVALID_ARG;
        if (nq == 0) return VL_OK;
        // k < out_stride: answer into [nq, kk] scratch and spread the rows.  kk = k whenever that scratch is of a sane size:
        // the library bounds what it writes by min(k, len) under its own lock, so no length snapshot (which a concurrent
        // add could outdate) is involved; an absurd k falls back to min(k, len now).
        uint64_t kk = k;
        if (k > (1ull << 26) / nq) {
            const uint64_t len = vl_index_len(h);
            kk = k < len ? k : len;
        }
        std::vector<uint64_t> ids(nq * kk);
        std::vector<double> scores(nq * kk);
        const int rc = on_flat(h, [&](auto* f) { return f->search_batch(queries, nq, q_len, kk, metric, nullptr, ids.data(), scores.data(), out_n); });
        if (rc != VL_OK) return rc;
        for (uint64_t q = 0; q < nq; ++q) {
            const uint64_t m = out_n[q] < kk ? out_n[q] : kk;
            for (uint64_t j = 0; j < m; ++j) {
                if (out_ids) out_ids[q * out_stride + j] = ids[q * kk + j];
                if (out_scores) out_scores[q * out_stride + j] = scores[q * kk + j];
            }
        }
        return VL_OK;
    });
}

int vl_index_search_positions(const vl_index* h, const double* query, uint64_t q_len, uint64_t k, int metric,
                              uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || h->hnsw) {
            vl::set_last_error("storage positions are a flat-index notion");
            return VL_ERR_INVALID_ARG;
        }
        if (!out_n) return VL_ERR_INVALID_ARG;
        return on_flat(h, [&](auto* f) { return f->search(query, q_len, k, metric, out_pos, out_ids, out_scores, out_n); });
    });
}

int vl_index_search_batch(const vl_index* h, const double* queries, uint64_t nq, uint64_t q_len, uint64_t k,
                          int metric, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || (!out_n && nq)) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->search_batch(queries, nq, q_len, k, metric, 0, out_ids, out_scores, out_n);
        return on_flat(h, [&](auto* f) { return f->search_batch(queries, nq, q_len, k, metric, nullptr, out_ids, out_scores, out_n); });
    });
}

int vl_index_search_batch_positions(const vl_index* h, const double* queries, uint64_t nq, uint64_t q_len,
                                    uint64_t k, int metric, uint64_t* out_pos, uint64_t* out_ids,
                                    double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || h->hnsw) {
            vl::set_last_error("storage positions are a flat-index notion");
            return VL_ERR_INVALID_ARG;
        }
        if (!out_n && nq) return VL_ERR_INVALID_ARG;
        return on_flat(h, [&](auto* f) { return f->search_batch(queries, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n); });
    });
}

int vl_index_search_batch_embeddings_f32(const vl_index* h, const float* embeddings, uint64_t nq, uint64_t dim, int normalize,
                                         int embeddings_on_device, uint64_t k, int metric, uint64_t* out_ids,
                                         double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || (!out_n && nq)) return VL_ERR_INVALID_ARG;
        for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
        const uint64_t want = vl_index_dimension(h);
        const bool empty_flat = !h->hnsw && vl_index_len(h) == 0;  // src/index/flat.rs:99: an empty flat index accepts any length
        if (dim != want && !empty_flat) {
            vl::set_dim_mismatch(want, dim);
            vl::set_last_error("Dimension mismatch: expected " + std::to_string(want) + ", got " + std::to_string(dim));
            return VL_ERR_DIM_MISMATCH;
        }
        if (nq == 0 || empty_flat) return VL_OK;
        const int device = h->hnsw ? h->hnsw->device() : (h->multi ? h->multi->device() : h->flat->device());
        return vl::search_embeddings_f32(device, dim, embeddings, nq, normalize != 0, embeddings_on_device != 0,
                                         [&](const double* d_q, uint64_t n) -> int {
                                             return vl_index_search_batch_dev(h, d_q, n, dim, k, metric, nullptr, out_ids, out_scores, out_n);
                                         });
    });
}

int vl_index_search_batch_dev(const vl_index* h, const double* d_queries, uint64_t nq, uint64_t q_len, uint64_t k,
                              int metric, uint64_t* out_pos, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || (!out_n && nq)) return VL_ERR_INVALID_ARG;
        if (h->hnsw) {  // graph walks read their queries from the host staging area: copy, then the usual call
            if (out_pos) {
                vl::set_last_error("storage positions are a flat-index notion");
                return VL_ERR_INVALID_ARG;
            }
            std::vector<double> hq;
            if (d_queries && nq && q_len && q_len == h->hnsw->dimension()) {  // (a wrong q_len is reported before any query is read)
                hq.resize((size_t)nq * q_len);
                if (hipMemcpy(hq.data(), d_queries, hq.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
                    vl::set_last_error("copying the device queries to the host failed");
                    return VL_ERR_DEVICE;
                }
            }
            static const double never_read = 0.0;  // nq == 0 or a wrong q_len: search_batch returns before it reads a query
            const double* hp = !d_queries ? nullptr : (hq.empty() ? &never_read : hq.data());
            return h->hnsw->search_batch(hp, nq, q_len, k, metric, 0, out_ids, out_scores, out_n);
        }
        return on_flat(h, [&](auto* f) { return f->search_batch_device(d_queries, nq, q_len, k, metric, out_pos, out_ids, out_scores, out_n); });
    });
}

// ---- row-sharded batched search over RCCL (shard.hpp) ----------------------------------------------------
int vl_comm_unique_id(uint8_t* out_id)
{
    return guarded([&]() -> int { return vl::ShardComm::unique_id(out_id); });
}

int vl_comm_create(const uint8_t* id, int world, int rank, int device, vl_comm** out)
{
    return guarded([&]() -> int {
        if (!out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::ShardComm* c = nullptr;
        const int rc = vl::ShardComm::create(id, world, rank, device, &c);
        if (rc != VL_OK) return rc;
        vl_comm* h = new (std::nothrow) vl_comm{c};
        if (!h) {
            delete c;
            return VL_ERR_OOM;
        }
        *out = h;
        return VL_OK;
    });
}

void vl_comm_destroy(vl_comm* comm)
{
    if (!comm) return;
    try {
        delete comm->c;
    } catch (...) {
    }
    delete comm;
}

int vl_comm_world(const vl_comm* comm) { return comm && comm->c ? comm->c->world() : 0; }
int vl_comm_rank(const vl_comm* comm) { return comm && comm->c ? comm->c->rank() : -1; }

int vl_comm_profile_enable(vl_comm* comm, int enable)
{
    if (!comm || !comm->c) return VL_ERR_INVALID_ARG;
    comm->c->profile_enable(enable != 0);
    return VL_OK;
}

int vl_comm_profile_read(vl_comm* comm, uint64_t* calls, double* local_ms, double* h2d_ms, double* allgather_ms,
                         double* merge_ms)
{
    if (!comm || !comm->c) return VL_ERR_INVALID_ARG;
    comm->c->profile_read(calls, local_ms, h2d_ms, allgather_ms, merge_ms);
    return VL_OK;
}

int vl_comm_record_paths(const vl_comm* comm, uint64_t* on_device, uint64_t* via_host)
{
    if (!comm || !comm->c) return VL_ERR_INVALID_ARG;
    comm->c->record_paths(on_device, via_host);
    return VL_OK;
}

int vl_shard_sync(const vl_index* shard, vl_comm* comm, uint64_t* out_offset, uint64_t* out_total)
{
    return guarded([&]() -> int {
        VL_FLAT_ONLY(shard);
        if (!comm || !comm->c) return VL_ERR_INVALID_ARG;
        return comm->c->sync(shard->flat, out_offset, out_total);
    });
}

int vl_shard_search_batch(const vl_index* shard, vl_comm* comm, const double* queries, uint64_t nq, uint64_t q_len,
                          uint64_t k, int metric, uint64_t* out_gpos, uint64_t* out_ids, double* out_scores,
                          uint64_t* out_n)
{
    return guarded([&]() -> int {
        VL_FLAT_ONLY(shard);
        if (!comm || !comm->c) return VL_ERR_INVALID_ARG;
        return comm->c->search_batch(shard->flat, queries, nq, q_len, k, metric, out_gpos, out_ids, out_scores, out_n);
    });
}

int vl_shard_search_batch_dev(const vl_index* shard, vl_comm* comm, const double* d_queries, uint64_t nq, uint64_t q_len,
                              uint64_t k, int metric, uint64_t* out_gpos, uint64_t* out_ids, double* out_scores,
                              uint64_t* out_n)
{
    return guarded([&]() -> int {
        VL_FLAT_ONLY(shard);
        if (!comm || !comm->c) return VL_ERR_INVALID_ARG;
        return comm->c->search_batch(shard->flat, d_queries, nq, q_len, k, metric, out_gpos, out_ids, out_scores, out_n, true);
    });
}

uint64_t vl_shard_packed_words(uint64_t nq, uint64_t ks) { return vl::shard_packed_words(nq, ks); }

int vl_shard_search_local(const vl_index* shard, uint64_t row_offset, int corpus_has_rows, const double* queries,
                          uint64_t nq, uint64_t q_len, uint64_t ks, int metric, uint64_t* out_packed)
{
    return guarded([&]() -> int {
        VL_FLAT_ONLY(shard);
        if (!out_packed || metric < 0 || metric > 3) return VL_ERR_INVALID_ARG;
        vl::shard_search_local(shard->flat, row_offset, UINT64_MAX, corpus_has_rows != 0, queries, nq, q_len, ks, metric,
                               reinterpret_cast<unsigned long long*>(out_packed));
        return VL_OK;  // the search's own status is word 0 of the record
    });
}

int vl_shard_search_local_dev(const vl_index* shard, uint64_t row_offset, int corpus_has_rows, const double* d_queries,
                              uint64_t nq, uint64_t q_len, uint64_t ks, int metric, uint64_t* out_packed)
{
    return guarded([&]() -> int {
        VL_FLAT_ONLY(shard);
        if (!out_packed || metric < 0 || metric > 3) return VL_ERR_INVALID_ARG;
        vl::shard_search_local(shard->flat, row_offset, UINT64_MAX, corpus_has_rows != 0, d_queries, nq, q_len, ks, metric,
                               reinterpret_cast<unsigned long long*>(out_packed), true);
        return VL_OK;  // the search's own status is word 0 of the record
    });
}

namespace {
std::mutex& merger_pool_mu()
{
    static std::mutex mu;
    return mu;
}
std::vector<std::unique_ptr<vl::ShardMerger>>& merger_pool()
{
    // idle mergers of vl_shard_merge, any device, at most 8; never destroyed (a static destructor would call hipFree after
    // the HIP runtime has been torn down at process exit)
    static auto* pool = new std::vector<std::unique_ptr<vl::ShardMerger>>();
    return *pool;
}
}  // namespace

int vl_shard_merge(int device, const uint64_t* gathered, uint32_t world, uint64_t nq, uint64_t ks, uint64_t k,
                   uint64_t* out_gpos, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (nq == 0) return VL_OK;
        if (!out_n) return VL_ERR_INVALID_ARG;
        for (uint64_t i = 0; i < nq; ++i) out_n[i] = 0;
        if (k == 0 || ks == 0) return VL_OK;
        if (!gathered || !out_scores || world == 0 || world > (uint32_t)vl::SHARD_MAX_WORLD || nq > 0x7FFFFFFFull ||
            ks > 0x7FFFFFFFull || vl::shard_packed_words(nq, ks) * world > (1ull << 29))
            return VL_ERR_INVALID_ARG;
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
            (void)hipGetLastError();
            vl::set_last_error("no usable HIP device (vectorlite_amd has no CPU fallback)");
            return VL_ERR_DEVICE;
        }
        // the bring-your-own-transport path: mergers (a stream, the gathered-records / output buffers) are kept per device
        // and lent to one call at a time -- making them per call cost ~0.9 ms of allocations against a 25 us merge kernel
        struct Lease {
            std::unique_ptr<vl::ShardMerger> m;
            ~Lease()
            {
                if (!m) return;
                std::lock_guard<std::mutex> g(merger_pool_mu());
                auto& pool = merger_pool();
                if (pool.size() < 8) pool.push_back(std::move(m));
            }
        } lease;
        {
            std::lock_guard<std::mutex> g(merger_pool_mu());
            auto& pool = merger_pool();
            for (size_t i = 0; i < pool.size(); ++i)
                if (pool[i]->device() == device) {
                    lease.m = std::move(pool[i]);
                    pool.erase(pool.begin() + (long)i);
                    break;
                }
        }
        if (!lease.m) lease.m.reset(new vl::ShardMerger(device));
        return lease.m->merge_host(reinterpret_cast<const unsigned long long*>(gathered), world, nq, ks, k, out_gpos, out_ids,
                                   out_scores, out_n);
    });
}

uint64_t vl_index_len(const vl_index* h) { return !h ? 0 : (h->hnsw ? h->hnsw->len() : (h->multi ? h->multi->len() : h->flat->len())); }
int vl_index_is_empty(const vl_index* h) { return vl_index_len(h) == 0; }
uint64_t vl_index_dimension(const vl_index* h)
{
    return !h ? 0 : (h->hnsw ? h->hnsw->dimension() : (h->multi ? h->multi->dimension() : h->flat->dimension()));
}

// VectorIndexWrapper::index_type / ::metric (src/lib.rs:329-346): Flat -> (0, None), HNSW -> (1, Some(m)).
int vl_index_type(const vl_index* h) { return (h && h->hnsw) ? 1 : 0; }
int vl_index_metric(const vl_index* h, int* out_metric)
{
    if (!h || !out_metric) return VL_ERR_INVALID_ARG;
    if (!h->hnsw) return VL_ERR_NOT_FOUND;  // None
    *out_metric = h->hnsw->metric();
    return VL_OK;
}

int vl_index_search_ef(const vl_index* h, const double* queries, uint64_t nq, uint64_t q_len, uint64_t k, uint32_t ef,
                       int metric, uint64_t* out_ids, double* out_scores, uint64_t* out_n)
{
    return guarded([&]() -> int {
        if (!h || !h->hnsw) {
            vl::set_last_error("vl_index_search_ef needs an HNSW handle");
            return VL_ERR_INVALID_ARG;
        }
        return h->hnsw->search_batch(queries, nq, q_len, k, metric, ef, out_ids, out_scores, out_n);
    });
}

int vl_index_get_vector(const vl_index* h, uint64_t id, double* out_values)
{
    return guarded([&]() -> int {
        if (!h || !out_values) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->get_vector(id, out_values);
        return on_flat(h, [&](auto* f) { return f->get_vector(id, out_values); });
    });
}

int vl_index_max_id(const vl_index* h, uint64_t* out_id)
{
    return guarded([&]() -> int {
        if (!h || !out_id) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->max_id(out_id);
        return on_flat(h, [&](auto* f) { return f->max_id(out_id); });
    });
}

int vl_index_export(const vl_index* h, uint64_t* out_ids, double* out_values)
{
    return guarded([&]() -> int {
        if (!h) return VL_ERR_INVALID_ARG;
        if (h->hnsw) return h->hnsw->export_rows(out_ids, out_values);
        return on_flat(h, [&](auto* f) { return f->export_rows(out_ids, out_values); });
    });
}

int vl_index_hnsw_distances(const vl_index* h, const double* query, uint64_t q_len, int metric,
                            const uint64_t* positions, uint64_t m, uint64_t* out_dist)
{
    return guarded([&]() -> int {
        VL_FLAT_ONLY(h);
        return h->flat->hnsw_distances(query, q_len, metric, positions, m, out_dist);
    });
}

// convert_distance_to_similarity(d as f64 / 1000.0, metric): src/index/hnsw.rs:51-75, :478-479.
// Four scalar operations on the k winners' u64 distances; no vector data is touched here.
double vl_hnsw_score(uint64_t d_u64, int metric) { return vl::hnsw_score(d_u64, metric); }

struct vl_vlc_doc {
    vl::VlcDoc* d;
};

int vl_vlc_open(const char* path, vl_vlc_doc** out)
{
    return guarded([&]() -> int {
        if (!path || !out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::VlcDoc* d = nullptr;
        const int rc = vl::vlc_open(path, &d);
        if (rc != VL_OK) return rc;
        vl_vlc_doc* h = new (std::nothrow) vl_vlc_doc{d};
        if (!h) {
            vl::vlc_close(d);
            return VL_ERR_OOM;
        }
        *out = h;
        return VL_OK;
    });
}

void vl_vlc_close(vl_vlc_doc* doc)
{
    if (!doc) return;
    vl::vlc_close(doc->d);
    delete doc;
}

const char* vl_vlc_name(const vl_vlc_doc* doc) { return doc ? vl::vlc_name(doc->d) : ""; }

int vl_vlc_info(const vl_vlc_doc* doc, int* index_type, int* metric, uint64_t* dim, uint64_t* rows, uint64_t* vector_count,
                uint64_t* dimension)
{
    if (!doc) return VL_ERR_INVALID_ARG;
    vl::vlc_info(doc->d, index_type, metric, dim, rows, vector_count, dimension);
    return VL_OK;
}

int vl_vlc_side_table(const vl_vlc_doc* doc, uint64_t* ids, uint64_t* text_off, uint64_t* text_len, uint64_t* meta_off,
                      uint64_t* meta_len)
{
    if (!doc) return VL_ERR_INVALID_ARG;
    return vl::vlc_side_table(doc->d, ids, text_off, text_len, meta_off, meta_len);
}

int vl_vlc_read_values(const vl_vlc_doc* doc, uint64_t first, uint64_t n, double* out_values)
{
    return guarded([&]() -> int {
        if (!doc) return VL_ERR_INVALID_ARG;
        return vl::vlc_read_values(doc->d, first, n, out_values);
    });
}

int vl_vlc_build_index(const vl_vlc_doc* doc, int device, vl_index** out)
{
    return guarded([&]() -> int {
        if (!doc || !out) return VL_ERR_INVALID_ARG;
        *out = nullptr;
        vl::GpuFlatIndex* f = nullptr;
        vl::HnswIndex* hn = nullptr;
        const int rc = vl::vlc_build_index(doc->d, device, &f, &hn);
        if (rc != VL_OK) return rc;
        vl_index* h = new (std::nothrow) vl_index{f, hn, nullptr};
        if (!h) {
            delete f;
            delete hn;
            return VL_ERR_OOM;
        }
        *out = h;
        return VL_OK;
    });
}

const char* vl_last_error(void) { return vl::last_error(); }
void vl_last_dim_mismatch(uint64_t* expected, uint64_t* actual) { vl::get_dim_mismatch(expected, actual); }
int vl_last_path(void) { return vl::last_path(); }

int vl_index_force_path(vl_index* h, int path)
{
    if (!h || h->hnsw || (path != 0 && path != VL_PATH_EXACT_SELECT && path != VL_PATH_EXACT_SORT)) return VL_ERR_INVALID_ARG;
    return on_flat(h, [&](auto* f) { f->force_path(path); return (int)VL_OK; });
}

int vl_index_set_single_filter(vl_index* h, int mode)
{
    if (!h || h->hnsw || (mode != 0 && mode != 1)) return VL_ERR_INVALID_ARG;
    return on_flat(h, [&](auto* f) { f->set_single_filter(mode); return (int)VL_OK; });
}

int vl_index_set_coalescing(vl_index* h, int max_batch, int window_us)
{
    if (!h || max_batch < 0 || window_us < 0) return VL_ERR_INVALID_ARG;
    if (h->hnsw) h->hnsw->set_coalescing(max_batch, window_us);
    else on_flat(h, [&](auto* f) { f->set_coalescing(max_batch, window_us); return (int)VL_OK; });
    return VL_OK;
}

int vl_index_coalesce_stats(const vl_index* h, uint64_t* batches, uint64_t* queries)
{
    if (!h) return VL_ERR_INVALID_ARG;
    if (h->hnsw) h->hnsw->coalesce_stats(batches, queries);
    else on_flat(h, [&](auto* f) { f->coalesce_stats(batches, queries); return (int)VL_OK; });
    return VL_OK;
}

int vl_index_coalesce_gather(vl_index* h, int adaptive, uint64_t* waits, uint64_t* waited_us)
{
    if (!h || adaptive < -1 || adaptive > 1) return VL_ERR_INVALID_ARG;
    if (h->hnsw) h->hnsw->coalesce_gather(adaptive, waits, waited_us);
    else on_flat(h, [&](auto* f) { f->coalesce_gather(adaptive, waits, waited_us); return (int)VL_OK; });
    return VL_OK;
}

int vl_index_hnsw_graph_info(const vl_index* h, uint64_t* n_nodes, uint32_t* entry, int* max_level, uint32_t* m,
                             uint32_t* m0, uint64_t* upper_slots)
{
    if (!h || !h->hnsw) {
        vl::set_last_error("this entry point needs an HNSW index handle");
        return VL_ERR_INVALID_ARG;
    }
    h->hnsw->graph_info(n_nodes, entry, max_level, m, m0, upper_slots);
    return VL_OK;
}

int vl_index_hnsw_graph_export(const vl_index* h, uint8_t* level, uint32_t* upper_off, uint32_t* cnt0, uint32_t* nbr0,
                               uint32_t* cntU, uint32_t* nbrU, uint64_t* node_ids, uint8_t* live, double* rows)
{
    return guarded([&]() -> int {
        if (!h || !h->hnsw) {
            vl::set_last_error("this entry point needs an HNSW index handle");
            return VL_ERR_INVALID_ARG;
        }
        return h->hnsw->graph_export(level, upper_off, cnt0, nbr0, cntU, nbrU, node_ids, live, rows);
    });
}

int vl_index_hnsw_set_min_beam(vl_index* h, uint32_t min_beam)
{
    if (!h || !h->hnsw) {
        vl::set_last_error("this entry point needs an HNSW index handle");
        return VL_ERR_INVALID_ARG;
    }
    h->hnsw->set_min_beam(min_beam);
    return VL_OK;
}

int vl_index_hnsw_walk_stats(const vl_index* h, uint64_t* queries, uint64_t* distance_evals)
{
    if (!h || !h->hnsw) return VL_ERR_INVALID_ARG;
    h->hnsw->walk_stats(queries, distance_evals);
    return VL_OK;
}

int vl_index_last_scan(const vl_index* h, int* variant, int* grid, int* query_in_kernarg)
{
    if (!h || h->hnsw) return VL_ERR_INVALID_ARG;
    return on_flat(h, [&](auto* f) { f->last_scan(variant, grid, query_in_kernarg); return (int)VL_OK; });
}

int vl_index_last_filter(const vl_index* h, int* out6)
{
    if (!h || h->hnsw || !out6) return VL_ERR_INVALID_ARG;
    return on_flat(h, [&](auto* f) { f->last_filter(out6); return (int)VL_OK; });
}

int vl_index_profile_enable(vl_index* h, int enable)
{
    if (!h || h->hnsw) return VL_ERR_INVALID_ARG;
    return on_flat(h, [&](auto* f) { f->profile_enable(enable != 0); return (int)VL_OK; });
}

int vl_index_profile_read(vl_index* h, uint64_t* n_scan_launches, double* scan_ms_total, uint64_t* scan_bytes_total)
{
    if (!h || h->hnsw) return VL_ERR_INVALID_ARG;
    return on_flat(h, [&](auto* f) { f->profile_read(n_scan_launches, scan_ms_total, scan_bytes_total); return (int)VL_OK; });
}

int vl_runtime_info(int* n_devices, int* abi_version)
{
    if (abi_version) *abi_version = VL_ABI_VERSION;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (n_devices) *n_devices = n;
    return VL_OK;
}

}  // extern "C"
