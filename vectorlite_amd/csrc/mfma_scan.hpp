// mfma_scan.hpp -- launch interface of the bf16 MFMA batched scan (mfma_scan.hip, K4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.hpp"

namespace vl {

constexpr int MFMA_GROUPS = 256;       // workgroups (= row groups) of the sampling pass
constexpr int MFMA_CAND_CAP = 4096;    // candidate buffer entries per query
constexpr int MFMA_MAX_BATCH = 2048;   // queries per launch sequence (scratch is sized for this: ~70 MB per workspace);
                                       // config 5's 4096 queries are two sequences: half the fixed launches of four.
                                       // Not 4096: 32 query chunks leave the sampling pass 8 workgroups per chunk = 64 groups,
                                       // and the 64th largest of 64 group maxima is far too loose a threshold (every
                                       // candidate buffer overflowed and all queries fell back: measured)
// Fewer rows than this (MFMA_GROUPS tiles of 32) leave the sampling pass with fewer than 64 groups: no
// threshold, every score a candidate, guaranteed buffer overflow -- such indexes take the f32 batch path.
constexpr uint64_t MFMA_MIN_ROWS = (uint64_t)MFMA_GROUPS * 32;
constexpr int MFMA_MIN_BATCH = 2;      // measured at N = 10 M x 384: one bf16 pass answers 2..256 queries in 1.9-2.0 ms,
                                       // the f32 batch kernel needs 2.7-2.9 ms per pass of up to 8 (it stays for Manhattan,
                                       // row lengths without an MFMA shape, and as the fallback)

// bf16 slab row stride in elements: dim rounded up to the MFMA K step (16)
// Row stride of the bf16 slab in elements: the next length the MFMA kernel has a shape for (8 / 16 / 24 / 32 / 48
// K steps of 16), zero padded -- a 100- or 300-dimensional index takes the batch filter too, at the price of the
// padding; beyond 768 dimensions there is no shape and the stride is just the dimension rounded up to 16.
inline uint32_t mfma_ldb(uint32_t dim)
{
    const uint32_t shapes[] = {128u, 256u, 384u, 512u, 768u};
    for (uint32_t s : shapes)
        if (dim <= s) return s;
    return (dim + 15u) & ~15u;
}

struct MfmaScratch {
    void* q_bf16 = nullptr;     // [nq_pad_cap, ldb] bf16
    int* gmax = nullptr;        // [nq_cap, MFMA_GROUPS]
    float* thr = nullptr;       // [nq_cap]
    Cand32* cand = nullptr;     // [nq_cap, MFMA_CAND_CAP]
    uint32_t* cnt = nullptr;    // [nq_pad_cap]
    uint32_t nq_cap = 0;        // queries per launch sequence the buffers hold
    uint32_t nq_pad_cap = 0;    // ... rounded up to whole query chunks: rows of q_bf16 and cnt (the padding queries never produce candidates)
};

bool mfma_scan_supported(uint32_t dim, int metric);

// Queries one launch sequence answers for this dimension (<= MFMA_MAX_BATCH): the sampling pass must hand k_thresholds
// at least 128 groups, i.e. 16 workgroups of 8 waves per query chunk -- at most 16 co-resident chunks on 256 CUs:
// 16 x 128 queries up to stride 512, 16 x 64 at stride 768.
uint32_t mfma_sequence_queries(uint32_t dim);
// Queries already in device memory -> the staging area of a launch sequence: d_dst [nq, dim] (zeros for a query outside
// the fast-path domain), d_norms [nq], in_domain [nq] (device-visible, e.g. pinned host memory).
hipError_t launch_stage_queries(hipStream_t s, const double* d_src, uint32_t nq, uint32_t dim, double max_abs,
                                double min_norm, double* d_dst, double* d_norms, unsigned char* in_domain);

// launch_stage_queries, the bf16 conversion of the staged queries and the clearing of the candidate counters as ONE kernel,
// for a sequence the row-stationary kernel will answer.  Returns false (nothing launched) when that kernel does not apply --
// the caller then stages with launch_stage_queries and launch_mfma_candidates converts as before; true: pass
// queries_prepared = true to launch_mfma_candidates.  *err carries the launch status.
bool launch_prepare_queries(hipStream_t s, const double* d_src, uint32_t nq, uint32_t dim, double max_abs, double min_norm,
                            double* d_dst, double* d_norms, unsigned char* in_domain, const MfmaScratch& w, hipError_t* err);

// f64 master rows [n, dim] -> UNIT-NORMALISED bf16 rows [n, ldb] (x/|x| in f64 -> f32 -> bf16, round to
// nearest even), |row| and |row|^2 (f64, rounded once to f32)
hipError_t launch_rows_bf16(hipStream_t s, const double* master, uint64_t n, uint32_t dim, void* out_bf16,
                            float* out_norm, float* out_sqnorm);

// Every bf16 row stride with an MFMA shape (128 / 256 / 384 / 512 / 768) takes the row-stationary kernel k_mfma_rows, which reads a FRAGMENT-MAJOR
// bf16 slab (launch_rows_bf16_frag); VL_MFMA_KERNEL=tile keeps the LDS-tile kernel k_mfma_scan and its row-major slab.
bool mfma_rows_kernel(uint32_t dim);
// rows [row0, row0 + n) of the index (master_rows = their f64 rows) -> the fragment-major slab and the two per-row arrays,
// all three addressed from their BASE (row 0): the layout interleaves 16 neighbouring rows
hipError_t launch_rows_bf16_frag(hipStream_t s, const double* master_rows, uint64_t row0, uint64_t n, uint32_t dim,
                                 void* slab_frag_base, float* norm_base, float* sqnorm_base);

// rows per k_mfma_scan tile (its largest shape).  The bf16 slab and the two per-row arrays must be ALLOCATED in whole tiles
// (ceil(n_rows / MFMA_TILE_ROWS) * MFMA_TILE_ROWS rows): the kernel fetches a tile as one contiguous block and masks
// the rows past n_rows afterwards, whatever they hold.
constexpr uint32_t MFMA_TILE_ROWS = 64;

// nq queries (f64 [nq, dim]) against the bf16 slab: writes one sorted top-64 candidate list per query
// (out_lists[nq][64], the layout k_merge_finalize takes with n_lists = 1).
// slab_bf16: the row-major slab, or the fragment-major one when mfma_rows_kernel(dim).
// row_norm / row_sqnorm: the arrays launch_rows_bf16 / launch_rows_bf16_frag wrote (dot and Euclidean keys need them).
// q64: [nq, dim] f64 queries followed by their [nq] f64 norms (0 = answer on the exact path: no candidates are collected).
// What one launch sequence of the filter looked like (for diagnostics: bench.py ties its PMC traffic figures to it).
struct MfmaLaunchInfo {
    int ksteps = 0;        // K steps of 16 = bf16 row stride / 16 (k_mfma_rows<KSTEPS, ...>); 0 = the LDS-tile kernel ran
    int metric = 0;
    int chunks = 0;        // query chunks (gridDim.y)
    int grid_x = 0;        // workgroups per chunk of the LAST pass-1 stage
    int stages = 0;        // pass-1 launches
    int sample_blocks = 0; // 32-row blocks the sampling pass covered
};
hipError_t launch_mfma_candidates(hipStream_t s, int metric, const void* slab_bf16, const float* row_norm,
                                  const float* row_sqnorm,
                                  const double* q64, uint32_t nq, uint64_t n_rows, uint32_t dim,
                                  const MfmaScratch& w, Cand32* out_lists, MfmaLaunchInfo* info = nullptr,
                                  bool queries_prepared = false);

// Single-query scan of the bf16 slab (opt-in filter): per-workgroup top-64 lists like launch_scan.
bool scan_bf16_supported(uint32_t dim, int metric);
hipError_t launch_scan_bf16(hipStream_t s, int metric, const void* slab_bf16, const float* row_norm,
                            const float* row_sqnorm, const double* q64, uint64_t n, uint32_t dim, Cand32* partials,
                            int* grid_out);

}  // namespace vl
