// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the distance-scan hot path.
//
// Reference semantics being reproduced (paths relative to /root/reference):
//   FlatIndex::search            src/index/flat.rs:98-119   (score every row, stable sort desc, truncate k)
//   SimilarityMetric::calculate  src/lib.rs:380-391, :425-572 (f64, strict index order, no FMA)
//   HNSW Metric::distance -> u64 src/index/hnsw.rs:113-174
//
// Design (DESIGN.md has the full argument):
//   * K1 k_scan: streams the [N, ld] f32 slab once. G lanes share a row, every lane issues 16-byte
//     non-temporal loads straight to VGPRs (no LDS round trip: nothing is reused), several row groups
//     in flight per wave; partial sums are reduced with cross-lane shuffles; each wave keeps a sorted
//     top-64 candidate list, ONE ENTRY PER LANE, updated by ballot/shuffle only when a row beats the
//     wave's current 64th key.  HBM-bound: 4*ld algorithmic bytes per row.
//   * K2 k_merge_finalize: merges the per-workgroup lists (bitonic merges in registers/LDS), then
//     RE-SCORES the 64 candidates from the f64 master rows in the reference's exact operation order
//     (separate multiply and add, index order, this file is compiled with -ffp-contract=off), ranks
//     them by (score desc, position asc) and proves -- with a rigorous f32 error bound on every row
//     that was NOT kept -- that no other row can reach the k-th score.  If the proof fails (ties at
//     the cut, adversarial data) the host re-runs the query on the exact kernels below.
//   * Exact path: k_exact_scan (reference-order f64 score of every row, LDS-transposed so global
//     reads stay coalesced) + k_select64 / bitonic sort on (score desc, position asc).
//   * k_hnsw_dist: the four HNSW distance callbacks, f64 reference order, Rust `as u64` semantics.
#include "kernels.hpp"
#include "device_common.hpp"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

namespace vl {
using namespace dev;
namespace {

// Merge level: workgroup (b, q) folds lists [64b, 64b+64) of query q (16 waves x 4 lists, then a
// 4-level tree) into one sorted list out[q][b].
template <typename K, typename C>
__global__ __launch_bounds__(1024) void k_merge_lists(const C* __restrict__ lists, int n_lists, size_t in_stride_q,
                                                      C* __restrict__ out, size_t out_stride_q)
{
    constexpr int NW = 16;
    __shared__ C sh[NW * WAVE];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    lists += (size_t)blockIdx.y * in_stride_q;
    out += (size_t)blockIdx.y * out_stride_q;
    const int first = blockIdx.x * 64 + wave * 4;
    int count = n_lists - first;
    count = count < 0 ? 0 : (count > 4 ? 4 : count);
    TopList<K> L;
    fold_lists4<K, C>(L, lists, first, count);
    block_merge<K, C, NW>(L, sh);
    if (wave == 0) {
        C e = {};
        e.key = L.key;
        e.pos = L.pos;
        out[(size_t)blockIdx.x * KP + lane] = e;
    }
}

// ---------------------------------------------------------------------------------------------
// K1: f32 slab scan
// ---------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ float acc4(float a, const f32x4 x, const f32x4 q)
{
    if (METRIC == COSINE || METRIC == DOT) {
        a = fmaf(x.x, q.x, a);
        a = fmaf(x.y, q.y, a);
        a = fmaf(x.z, q.z, a);
        a = fmaf(x.w, q.w, a);
    } else if (METRIC == EUCLIDEAN) {
        const float d0 = x.x - q.x, d1 = x.y - q.y, d2 = x.z - q.z, d3 = x.w - q.w;
        a = fmaf(d0, d0, a);
        a = fmaf(d1, d1, a);
        a = fmaf(d2, d2, a);
        a = fmaf(d3, d3, a);
    } else {
        a += fabsf(x.x - q.x);
        a += fabsf(x.y - q.y);
        a += fabsf(x.z - q.z);
        a += fabsf(x.w - q.w);
    }
    return a;
}

// Scan key (larger = better) from the reduced row sum.  Cosine is ranked by dot * (1/|row|): the
// 1/|query| factor is the same positive number for every row.  Euclidean/Manhattan are ranked by
// the negated sum: 1/(1+sqrt(s)) and 1/(1+s) are monotone in s, so the scan needs no sqrt/division.
template <int METRIC>
__device__ __forceinline__ float scan_key(float sum, float inv_norm)
{
    if (METRIC == COSINE) return sum * inv_norm;
    if (METRIC == DOT) return sum;
    return -sum;
}

// The query arrives as f64 (the reference's `search(&[f64])`); each lane rounds its own slice to
// f32 (round to nearest even, the same rounding the slab rows got at ingest).  Columns past `dim`
// (slab padding) read as zero.
__device__ __forceinline__ f32x4 load_q4(const double* __restrict__ q64, uint32_t j4, uint32_t dim)
{
    // Clamped, never predicated: `i < dim ? q64[i] : 0` makes hipcc branch around every load and wait for
    // each in turn (48 dependent L2 round trips in the prologue of every wave of the dim-384 scan).
    const uint32_t i = j4 * 4, last = dim - 1;
    const double v0 = q64[i + 0 < dim ? i + 0 : last];
    const double v1 = q64[i + 1 < dim ? i + 1 : last];
    const double v2 = q64[i + 2 < dim ? i + 2 : last];
    const double v3 = q64[i + 3 < dim ? i + 3 : last];
    f32x4 r;
    r.x = i + 0 < dim ? (float)v0 : 0.0f;
    r.y = i + 1 < dim ? (float)v1 : 0.0f;
    r.z = i + 2 < dim ? (float)v2 : 0.0f;
    r.w = i + 3 < dim ? (float)v3 : 0.0f;
    return r;
}

template <int G>
__device__ __forceinline__ float group_reduce(float a)
{
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) a += __shfl_xor(a, o);
    return a;
}

// Specialised: ld4 == G * VPL float4 per row, U row groups in flight per wave.  `qv` is the lane's slice of the
// f32 query (columns c + G j), loaded by the kernel entry that wraps this body.
template <int METRIC, int G, int VPL, int U>
__device__ __forceinline__ void scan_body(const f32x4* __restrict__ slab, const float* __restrict__ inv_norm,
                                          const f32x4 (&qv)[VPL], uint32_t n, Cand32* __restrict__ out)
{
    constexpr int RPS = WAVE / G;  // rows per step of one wave
    constexpr uint32_t LD4 = G * VPL;
    __shared__ Cand32 sh[4 * WAVE];

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int g = lane / G, c = lane % G;

    const uint32_t n_steps = (n + RPS - 1) / RPS;
    const uint32_t n_waves = gridDim.x * 4;
    const uint32_t wave_global = blockIdx.x * 4 + wave;

    TopList<float> L;
    L.init();

    for (uint32_t s0 = wave_global; s0 < n_steps; s0 += n_waves * U) {
        f32x4 x[U][VPL];
        uint32_t row[U];
        float inv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t s = s0 + (uint32_t)u * n_waves;
            row[u] = s < n_steps ? s * RPS + g : n;  // n marks "no row"
            const uint32_t r = row[u] < n ? row[u] : n - 1;  // clamp: loads stay in bounds
            const f32x4* p = slab + (size_t)r * LD4 + c;
#pragma unroll
            for (int j = 0; j < VPL; ++j) x[u][j] = __builtin_nontemporal_load(p + G * j);
            inv[u] = 1.0f;
            if (METRIC == COSINE) inv[u] = inv_norm[r];  // issued with the row loads, not after them
        }
        // every load of this iteration is issued before the first FMA: left alone, the scheduler
        // trades memory-level parallelism for registers and serialises the loads two at a time
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = 0.0f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) a = acc4<METRIC>(a, x[u][j], qv[j]);
            a = group_reduce<G>(a);
            const float key = scan_key<METRIC>(a, inv[u]);
            L.offer(key, row[u], row[u] < n && c == 0);
        }
    }

    block_merge<float, Cand32, 4>(L, sh);
    if (wave == 0) {
        Cand32 e;
        e.key = L.key;
        e.pos = L.pos;
        out[(size_t)blockIdx.x * KP + lane] = e;
    }
}

// K1, the single-query form: the f32 query travels IN THE KERNEL ARGUMENTS (rows of up to SCAN_QARG_FLOATS padded
// columns = 3 KB of the 4 KB kernarg segment), rounded on the host exactly as load_q4 rounds it here (f64 -> f32,
// nearest even).  No H2D copy precedes the scan: a search is this launch + the finalize kernel.
struct alignas(16) ScanQArg {
    float v[SCAN_QARG_FLOATS];
};

template <int METRIC, int G, int VPL, int U>
__global__ __launch_bounds__(256) void k_scan(const f32x4* __restrict__ slab, const float* __restrict__ inv_norm,
                                              uint32_t n, Cand32* __restrict__ out, const ScanQArg qa)
{
    static_assert(G * VPL * 4 <= SCAN_QARG_FLOATS, "row too long for the kernarg query");
    const int c = lane_id() % G;
    f32x4 qv[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) qv[j] = *reinterpret_cast<const f32x4*>(&qa.v[4 * (c + G * j)]);
    scan_body<METRIC, G, VPL, U>(slab, inv_norm, qv, n, out);
}

// The same scan with the f64 query in device memory (rows longer than the kernarg form holds; the multi-list
// k > 60 path, which stages the query on the device anyway).
template <int METRIC, int G, int VPL, int U>
__global__ __launch_bounds__(256) void k_scan_q64(const f32x4* __restrict__ slab, const float* __restrict__ inv_norm,
                                                  const double* __restrict__ q64, uint32_t dim, uint32_t n,
                                                  Cand32* __restrict__ out)
{
    const int c = lane_id() % G;
    f32x4 qv[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) qv[j] = load_q4(q64, c + G * j, dim);
    scan_body<METRIC, G, VPL, U>(slab, inv_norm, qv, n, out);
}

// Generic: any ld4 (float4 per row), G = lanes per row (power of two <= 64).
template <int METRIC, int G>
__global__ __launch_bounds__(256) void k_scan_generic(const f32x4* __restrict__ slab,
                                                      const float* __restrict__ inv_norm,
                                                      const double* __restrict__ q64, uint32_t dim, uint32_t n,
                                                      uint32_t ld4, Cand32* __restrict__ out)
{
    constexpr int RPS = WAVE / G;
    __shared__ Cand32 sh[4 * WAVE];

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int g = lane / G, c = lane % G;

    const uint32_t n_steps = (n + RPS - 1) / RPS;
    const uint32_t n_waves = gridDim.x * 4;

    TopList<float> L;
    L.init();

    for (uint32_t s = blockIdx.x * 4 + wave; s < n_steps; s += n_waves) {
        const uint32_t row = s * RPS + g;
        const bool valid = row < n;
        const uint32_t r = valid ? row : n - 1;
        const f32x4* p = slab + (size_t)r * ld4;
        float a = 0.0f;
        for (uint32_t j = c; j < ld4; j += G) a = acc4<METRIC>(a, p[j], load_q4(q64, j, dim));
        a = group_reduce<G>(a);
        float inv = 1.0f;
        if (METRIC == COSINE) inv = inv_norm[r];
        L.offer(scan_key<METRIC>(a, inv), row, valid && c == 0);
    }

    block_merge<float, Cand32, 4>(L, sh);
    if (wave == 0) {
        Cand32 e;
        e.key = L.key;
        e.pos = L.pos;
        out[(size_t)blockIdx.x * KP + lane] = e;
    }
}

// K3: small-batch scan.  One pass over the slab serves QB queries: each row group is loaded ONCE into
// registers and scored against QB queries whose f32 copies sit in LDS (lanes that share a column read
// the same 16 bytes: an LDS broadcast).  Per wave, QB independent top-64 lists.  Still streams
// N*ld*4 bytes per pass; with QB = 8 the f32 VALU work (2*QB flop per 4 bytes) is about level with
// the HBM time, beyond that the GEMM/MFMA form takes over.
template <int METRIC, int G, int VPL, int QB>
__global__ __launch_bounds__(256) void k_scan_batch(const f32x4* __restrict__ slab, const float* __restrict__ inv_norm,
                                                    const double* __restrict__ q64, uint32_t n_q, uint32_t dim,
                                                    uint32_t n, Cand32* __restrict__ out)
{
    constexpr int RPS = WAVE / G;
    constexpr uint32_t LD4 = G * VPL;
    __shared__ f32x4 qs[QB][LD4];
    __shared__ Cand32 sh[4 * WAVE];
    // Lazy insertion (round 4).  A row that beats a list's 64th entry is not inserted at once (ballot, rank, two DPP shifts,
    // selects: ~25 dependent instructions per entry, and on a SMALL index every wave builds its lists from the ~1 500 rows it
    // sees -- 64 (1 + ln(rows / 64)) ~ 270 insertions per list: 69 % of this kernel on a 50 000-row index, measured with the
    // insertions compiled out): it is appended to a 64-entry buffer of its (wave, query) in LDS, and a full buffer is sorted
    // (bitonic, registers) and merged into the list in one go -- ~4 instructions per entry.  The threshold the rows are held
    // against is a little stale in between (more rows pass than would one by one); nothing that belongs is ever dropped, and
    // the list after a merge is the top 64 of everything the wave has seen: the same lists as before, bit for bit.
    __shared__ Cand32 cbuf[4][QB][WAVE];

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int g = lane / G, c = lane % G;
    // blockIdx.y = a group of QB queries: groups share nothing but the slab (one launch answers up to gridDim.y * QB queries;
    // on a small index that is what keeps the chip busy -- one group's pass is a few workgroups)
    {
        const uint32_t first = blockIdx.y * QB;
        q64 += (size_t)first * dim;
        out += (size_t)first * gridDim.x * KP;
        n_q = n_q - first < (uint32_t)QB ? n_q - first : (uint32_t)QB;
    }

    for (uint32_t idx = threadIdx.x; idx < QB * LD4; idx += 256) {
        const uint32_t qi = idx / LD4, j = idx % LD4;
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        qs[qi][j] = qi < n_q ? load_q4(q64 + (size_t)qi * dim, j, dim) : z;
    }
    __syncthreads();

    const uint32_t n_steps = (n + RPS - 1) / RPS;
    const uint32_t n_waves = gridDim.x * 4;

    TopList<float> L[QB];
    uint32_t n_buf[QB];  // wave-uniform: entries waiting in cbuf[wave][qi]
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        L[qi].init();
        n_buf[qi] = 0;
    }
    auto flush = [&](int qi) {  // cbuf[wave][qi][0 .. n_buf) -> sorted -> merged into L[qi]
        float fk = -INFINITY;
        uint32_t fp = POS_SENTINEL;
        asm volatile("" : "+v"(fk));  // (keeps the -inf out of constant propagation, as in TopList::init)
        if ((uint32_t)lane < n_buf[qi]) {
            const Cand32 e = cbuf[wave][qi][lane];
            fk = e.key;
            fp = e.pos;
        }
        sort64_reversed<float>(fk, fp);
        L[qi].merge_reversed(fk, fp);
        n_buf[qi] = 0;
    };

    for (uint32_t s = blockIdx.x * 4 + wave; s < n_steps; s += n_waves) {
        const uint32_t row = s * RPS + g;
        const bool valid = row < n;
        const uint32_t r = valid ? row : n - 1;
        const f32x4* p = slab + (size_t)r * LD4 + c;
        f32x4 x[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) x[j] = __builtin_nontemporal_load(p + G * j);
        float inv = 1.0f;
        if (METRIC == COSINE) inv = inv_norm[r];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float a = 0.0f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) a = acc4<METRIC>(a, x[j], qs[qi][c + G * j]);
            a = group_reduce<G>(a);
#ifndef VL_DBG_K3_NOINSERT
            {
                const float key = scan_key<METRIC>(a, inv);
                const bool cand = valid && c == 0 && (uint32_t)qi < n_q && better<float>(key, row, L[qi].thr_key, L[qi].thr_pos);
                const unsigned long long m = __ballot(cand);
                if (m != 0ull) {  // wave-uniform; at most RPS lanes offer a row per step
                    const uint32_t slot = n_buf[qi] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (cand) {
                        Cand32 e;
                        e.key = key;
                        e.pos = row;
                        cbuf[wave][qi][slot] = e;
                    }
                    n_buf[qi] += (uint32_t)__popcll(m);
                    if (n_buf[qi] > (uint32_t)(WAVE - RPS)) flush(qi);  // the next step could overflow the buffer
                }
            }
#else  // diagnostic build (tools/build_variant_k.sh): what does the scan cost WITHOUT the top-64 list maintenance?
            if (scan_key<METRIC>(a, inv) == 12345.678f) L[qi].pos = row;
#endif
            // one query's LDS reads at a time: without this the scheduler hoists every query's
            // 48 registers of LDS loads above the first FMA and the kernel needs 460+ VGPRs
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        if ((uint32_t)qi < n_q) {  // uniform
            if (n_buf[qi]) flush(qi);
            block_merge<float, Cand32, 4>(L[qi], sh);
            if (wave == 0) {
                Cand32 e;
                e.key = L[qi].key;
                e.pos = L[qi].pos;
                out[((size_t)qi * gridDim.x + blockIdx.x) * KP + lane] = e;
            }
            __syncthreads();
        }
    }
}

// Rescore up to 64 rows (positions in LDS) against the query: global reads are coalesced along the
// row (all threads load a [64][CH] f64 tile into LDS), then lanes 0..63 of wave 0 each walk one
// row's chunk in index order.  Row stride CH+1 doubles keeps the per-lane ds_read_b64 conflict-free.
constexpr int RESCORE_CH = 96;

template <int METRIC, int NTHREADS>
__device__ __forceinline__ void rescore_rows(const double* __restrict__ master, const double* __restrict__ q64,
                                             uint32_t dim, const uint32_t* sh_pos, int n_rows,
                                             double (*tile)[RESCORE_CH + 1], double* qtile, Acc64<METRIC>& A)
{
    // PF chunks of 96 columns are fetched from HBM together (one memory latency for 384 columns
    // instead of four); they then pass through the LDS tile one after the other, in column order.
    constexpr int PER = (KP * RESCORE_CH + NTHREADS - 1) / NTHREADS;  // tile elements per thread
    constexpr int PF = (NTHREADS >= 1024) ? 4 : 1;
    const int tid = threadIdx.x;
    A.init();
    if (n_rows <= 0) return;  // workgroup-uniform (an all-sentinel candidate list)
    for (uint32_t g0 = 0; g0 < dim; g0 += PF * RESCORE_CH) {
        double pre[PF][PER];
        double qpre[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const uint32_t c0 = g0 + p * RESCORE_CH;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                // clamped, never predicated (a load under a condition is waited for before the next one issues):
                // rows >= n_rows are not walked and columns >= dim not stepped, so the duplicates are unused
                int idx = tid + i * NTHREADS;
                idx = idx < KP * RESCORE_CH ? idx : KP * RESCORE_CH - 1;
                int r = idx / RESCORE_CH;
                r = r < n_rows ? r : n_rows - 1;
                uint32_t col = c0 + (uint32_t)(idx % RESCORE_CH);
                col = col < dim ? col : dim - 1;
                pre[p][i] = master[(size_t)sh_pos[r] * dim + col];
            }
            {
                uint32_t col = c0 + (uint32_t)(tid % RESCORE_CH);
                col = col < dim ? col : dim - 1;
                qpre[p] = q64[col];
            }
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const uint32_t c0 = g0 + p * RESCORE_CH;
            if (c0 >= dim) break;  // workgroup-uniform
            const uint32_t cw = (dim - c0) < (uint32_t)RESCORE_CH ? (dim - c0) : (uint32_t)RESCORE_CH;
            __syncthreads();  // previous chunk fully consumed
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = tid + i * NTHREADS;
                if (idx < KP * RESCORE_CH) tile[idx / RESCORE_CH][idx % RESCORE_CH] = pre[p][i];
            }
            if (tid < RESCORE_CH) qtile[tid] = qpre[p];
            __syncthreads();
            if (tid < n_rows) {
                uint32_t cc = 0;
                for (; cc + 8 <= cw; cc += 8) {  // 16 LDS reads in flight, then 8 steps in index order
                    double xv[8], yv[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        xv[t] = tile[tid][cc + t];
                        yv[t] = qtile[cc + t];
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) A.step(xv[t], yv[t]);
                }
                for (; cc < cw; ++cc) A.step(tile[tid][cc], qtile[cc]);
            }
        }
    }
}

#ifdef VL_DBG_STAMPS
// diagnostic build only (tools/build_variant_k.sh): 100 MHz timestamps of the phases of one finalize launch
__device__ unsigned long long vl_dbg_stamps[32];
#define VL_STAMP(i)                                                                  \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                   \
            vl_dbg_stamps[i] = __builtin_amdgcn_s_memrealtime();                     \
            vl_dbg_stamps[16 + (i)] = __builtin_amdgcn_s_memtime(); /* shader clock */ \
        }                                                                            \
    } while (0)
#else
#define VL_STAMP(i) do {} while (0)
#endif

// The finalize kernels' rescoring (1024 threads): the same reference arithmetic with the serial part cut down to
// what the reference makes serial.  `a += x * y` under -ffp-contract=off is t = fl(x * y); a = fl(a + t): the
// products do not depend on the running sum, so ALL threads compute them while they move the rows through LDS
// (64 rows x 48 columns per tile), and one lane per row then only ADDS them in index order.  Cosine's three sums
// (x.y, x.x, y.y) are walked by three different waves at the same time.  Per row that leaves `dim` dependent f64
// adds (~2 us at dim 384) where rescore_rows walks dim x (2 LDS reads + 6 dependent f64 operations).
constexpr int RP_CH = 48;               // columns per LDS tile (row stride 49 doubles: conflict-free ds_read_b64)
constexpr int RP_NCH = 8;               // tiles per block of columns fetched from HBM together
constexpr int RP_BW = RP_CH * RP_NCH;   // 384 columns: one memory round trip for a dim-384 row

template <int METRIC, int ROWS = KP, int NCH = RP_NCH>
struct RescoreLds {
    double tA[ROWS][RP_CH + 1];                                 // the summands of `a`
    double tB[METRIC == COSINE ? ROWS : 1][RP_CH + 1];          // cosine: x * x
    double qblk[RP_CH * NCH];                                   // the query's columns of the current block
    double qq[METRIC == COSINE ? RP_CH * NCH : 1];              // cosine: y * y
    double b[ROWS];
    double c;
};

// ROWS candidate rows rescored by a workgroup of NTHREADS threads (ROWS * RP_CH tile elements, RP_PER per thread):
// 64 rows x 1024 threads for a single search's finalize, 16 rows x 256 threads for a quarter of a batch query's list.
template <int METRIC, int ROWS, int NTHREADS, int NCH = RP_NCH>
__device__ __forceinline__ void rescore_rows_par(const double* __restrict__ master, const double* __restrict__ q64,
                                                 uint32_t dim, const uint32_t* sh_pos, int n_rows,
                                                 RescoreLds<METRIC, ROWS, NCH>& S, Acc64<METRIC>& A,
                                                 const double (&q_first)[(RP_CH * NCH + NTHREADS - 1) / NTHREADS])
{
    // q_first[j]: q64[min(tid + j NTHREADS, dim - 1)], fetched by the caller at kernel entry (a single search reads
    // its query from pinned host memory: that PCIe round trip then hides behind the list merge)
    constexpr int RP_PER = ROWS * RP_CH / NTHREADS;
    constexpr int BW = RP_CH * NCH;  // columns fetched from HBM together (one memory round trip)
    constexpr int QN = (BW + NTHREADS - 1) / NTHREADS;
    static_assert(ROWS * RP_CH % NTHREADS == 0 && ROWS <= WAVE, "tile elements divide evenly; one lane per row");
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wave = tid >> 6;
    A.init();
    if (n_rows <= 0) return;  // workgroup-uniform (an all-sentinel candidate list)
    // what this thread accumulates: wave 0 the rows' `a`, wave 1 their `b`, wave 2 the query's `c` (cosine)
    double acc = (METRIC == COSINE) ? 0.0 : -0.0;
    for (uint32_t g0 = 0; g0 < dim; g0 += BW) {
        double pre[NCH][RP_PER];
#pragma unroll
        for (int p = 0; p < NCH; ++p) {
            const uint32_t c0 = g0 + p * RP_CH;
#pragma unroll
            for (int i = 0; i < RP_PER; ++i) {
                // clamped, never predicated (a load under a condition is waited for before the next one issues):
                // rows >= n_rows are not walked and columns >= dim not summed, so the duplicates are unused
                const int idx = tid + i * NTHREADS;
                int r = idx / RP_CH;
                r = r < n_rows ? r : n_rows - 1;
                uint32_t col = c0 + (uint32_t)(idx % RP_CH);
                col = col < dim ? col : dim - 1;
                pre[p][i] = master[(size_t)sh_pos[r] * dim + col];
            }
        }
        double qv[QN];
#pragma unroll
        for (int j = 0; j < QN; ++j) {
            qv[j] = q_first[j];
            if (g0 != 0 && tid + j * NTHREADS < BW) {
                const uint32_t col = g0 + (uint32_t)(tid + j * NTHREADS);
                qv[j] = q64[col < dim ? col : dim - 1];
            }
        }
        VL_STAMP(4);
        __syncthreads();  // the previous block's qblk / tiles are consumed
#pragma unroll
        for (int j = 0; j < QN; ++j)
            if (tid + j * NTHREADS < BW) {
                S.qblk[tid + j * NTHREADS] = qv[j];
                if (METRIC == COSINE) S.qq[tid + j * NTHREADS] = qv[j] * qv[j];
            }
        __syncthreads();
        VL_STAMP(5);
#pragma unroll
        for (int p = 0; p < NCH; ++p) {
            const uint32_t c0 = g0 + p * RP_CH;
            if (c0 >= dim) break;  // workgroup-uniform
            const uint32_t cw = (dim - c0) < (uint32_t)RP_CH ? (dim - c0) : (uint32_t)RP_CH;
            if (p) __syncthreads();  // the previous tile is consumed
#pragma unroll
            for (int i = 0; i < RP_PER; ++i) {
                const int idx = tid + i * NTHREADS;
                const int r = idx / RP_CH, cc = idx % RP_CH;
                const double x = pre[p][i], y = S.qblk[p * RP_CH + cc];
                if (METRIC == COSINE) {
                    S.tA[r][cc] = x * y;
                    S.tB[r][cc] = x * x;
                } else if (METRIC == EUCLIDEAN) {
                    const double d = x - y;
                    S.tA[r][cc] = d * d;
                } else if (METRIC == MANHATTAN) {
                    S.tA[r][cc] = fabs(x - y);
                } else {
                    S.tA[r][cc] = x * y;
                }
            }
            __syncthreads();
            if (wave == 0 || (METRIC == COSINE && wave == 1)) {
                if (lane < n_rows) {
                    const double* t = (wave == 0) ? &S.tA[lane][0] : &S.tB[METRIC == COSINE ? lane : 0][0];
                    uint32_t cc = 0;
                    for (; cc + 16 <= cw; cc += 16) {  // 16 LDS reads in flight, then 16 adds in index order
                        double v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = t[cc + u];
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc += v[u];
                    }
                    for (; cc < cw; ++cc) acc += t[cc];
                }
            } else if (METRIC == COSINE && wave == 2) {
                // the query's own sum of squares, in index order like the rows': batched the same way -- one element per
                // LDS round trip made THIS wave the slowest of the three (0.84 -> 0.73 us per tile; the rest is barriers and the products)
                const double* t = &S.qq[p * RP_CH];
                uint32_t cc = 0;
                for (; cc + 16 <= cw; cc += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = t[cc + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc += v[u];
                }
                for (; cc < cw; ++cc) acc += t[cc];
            }
            if (p == 0) VL_STAMP(6);
        }
    }
    VL_STAMP(7);
    __syncthreads();
    if (METRIC == COSINE) {
        if (wave == 1 && lane < ROWS) S.b[lane] = acc;
        if (wave == 2 && lane == 0) S.c = acc;
        __syncthreads();
        if (wave == 0) {
            A.a = acc;
            A.b = S.b[lane < ROWS ? lane : 0];
            A.c = S.c;
        }
    } else if (wave == 0) {
        A.a = acc;
    }
}

// Upper bound B on the REFERENCE f64 score of any row whose f32 scan key is <= t, for in-domain data
// (finite, |v| <= 2^40, row norms 0 or >= 2^-40).  u = 2^-24, n = padded dim, R = max row norm,
// Q = |query|.  Derivation in DESIGN.md ("Exactness bound"); every u-term carries a 2x safety factor.
// `in_extra` is the additional relative input-rounding term of a lower-precision candidate filter
// (bf16 MFMA path: (2 + 2^-8) * 2^-8 per product, rigorous, no safety factor needed); 0 for the f32 scan.
template <int METRIC>
__device__ __forceinline__ double bound_for_key(float t_key, uint32_t n, double R, double Q, double in_extra)
{
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double nn = (double)n;
    const double t = (double)t_key;
    if (METRIC == COSINE) {
        if (!(Q > 0.0)) return (double)INFINITY;
        return t / Q + 2.0 * (nn + 4.0) * u + in_extra + 1e-12;
    }
    if (METRIC == DOT) {
        return t + (2.0 * (nn + 2.0) * u + in_extra) * R * Q + 1e-12 * (1.0 + R * Q);
    }
    if (METRIC == EUCLIDEAN && in_extra > 0.0) {
        // GEMM-form key of the MFMA path: key = 2 x.q - |x|^2 = |q|^2 - |x - q|^2 (real numbers).
        // |key32 - key| <= 2 (in_extra + (n+2)u) R Q + u R^2 + u (2 R Q + R^2); u-terms doubled.
        const double err = 2.0 * in_extra * R * Q + 4.0 * (nn + 4.0) * u * (R * Q + R * R);
        double s_lo = Q * Q - t - err;
        if (!(s_lo > 0.0)) s_lo = 0.0;
        return (1.0 / (1.0 + sqrt(s_lo) * (1.0 - 1e-12))) * (1.0 + 1e-15);
    }
    const double ts = t < 0.0 ? -t : 0.0;  // key = -sum
    double d_lo;
    if (METRIC == EUCLIDEAN) {
        d_lo = sqrt(ts) * (1.0 - 2.0 * (nn + 2.0) * u) - 4.0 * u * (R + Q);
    } else {
        d_lo = ts * (1.0 - 2.0 * (nn + 2.0) * u) - 4.0 * u * sqrt(nn) * (R + Q);
    }
    if (!(d_lo > 0.0)) d_lo = 0.0;
    return (1.0 / (1.0 + d_lo * (1.0 - 1e-12))) * (1.0 + 1e-15);
}

// Phase 3 of a finalize, one WAVE per query: lane j holds candidate j's reference score and position (lanes >= n_cand hold
// nothing); key64 = the candidate list's 64th scan key.  Rank by (score desc, pos asc) = the reference's stable sort, run the
// exactness bound check, write the result block (and, for a row shard, its slice of the exchange record).
template <int METRIC>
__device__ __forceinline__ void rank_check_emit(double sc, uint32_t my_pos, float key64, int n_cand, uint32_t k, uint64_t n_rows,
                                                uint32_t ld, double R, double Q, double in_extra,
                                                SearchResultBlock* __restrict__ out, uint32_t seq, const ShardRecordSink& sink,
                                                uint32_t q_in_launch)
{
    const int lane = lane_id();
    const bool valid = lane < n_cand;
    const bool any_nan = __ballot(valid && sc != sc) != 0ull;
    // rank = how many candidates stand in front of this one: every candidate sits in a lane of this wave, so
    // entry j is read with v_readlane (a scalar operand of the compares), no LDS round trip per entry
    int rank = 0;
    for (int j = 0; j < n_cand; ++j) {
        const double sj = read_lane(sc, j);
        const uint32_t pj = read_lane(my_pos, j);
        rank += (sj > sc || (sj == sc && pj < my_pos)) ? 1 : 0;
    }
    const uint32_t k_eff = (uint64_t)k < n_rows ? k : (uint32_t)n_rows;
    uint32_t flags = 0;
    if (any_nan) flags |= RESULT_HAS_NAN | RESULT_NEEDS_EXACT;
    if (n_rows <= (uint64_t)KP) {
        // the list must hold EVERY row; a shorter one (a candidate stage that overflowed or dropped rows)
        // proves nothing
        if ((uint64_t)n_cand < n_rows) flags |= RESULT_NEEDS_EXACT;
    } else {
        // rows outside the candidate list exist: all of them have scan key <= the 64th key
        if (n_cand < KP) {
            flags |= RESULT_NEEDS_EXACT;  // keys were not finite: outside the fast-path domain
        } else {
            const double B = bound_for_key<METRIC>(key64, ld, R, Q, in_extra);
            // score of the entry ranked k_eff-1
            const unsigned long long at_cut = __ballot(valid && rank == (int)k_eff - 1);
            double s_cut = 0.0;
            if (at_cut) s_cut = __shfl(sc, __ffsll((long long)at_cut) - 1);
            if (!at_cut || !(s_cut > B)) flags |= RESULT_NEEDS_EXACT;
        }
    }
    if (valid && rank < (int)k_eff) {
        out->pos[rank] = my_pos;
        out->score[rank] = sc;
    }
    if (lane == 0) {
        out->n_out = k_eff;
        out->flags = flags;
    }
    if (sink.cnt) {  // a row shard's exchange record, written where the all-gather reads it (shard.hpp)
        const uint32_t q = sink.q0 + q_in_launch;
        const uint32_t offered = (flags == 0u) ? (k_eff < sink.ks ? k_eff : sink.ks) : 0u;  // not certified: the host redoes it
        if (valid && (uint32_t)rank < offered) {
            const size_t e = (size_t)q * sink.ks + (uint32_t)rank;
            sink.score_bits[e] = (unsigned long long)__double_as_longlong(sc);
            sink.gpos[e] = sink.row_offset + (unsigned long long)my_pos;
            sink.ids[e] = sink.pos_to_id[my_pos];
        }
        if (lane == 0) sink.cnt[q] = offered;
    }
    VL_STAMP(9);
    if (seq) {
        // the host spins on out->seq (pinned memory): every lane's stores of the block above are complete at
        // system scope before the stamp leaves (this wave is the only writer of the block)
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    VL_STAMP(10);
}

// A BATCH's finalize (n_lists = 1: the MFMA filter's one sorted top-64 list per query) as two launches.  k_merge_finalize
// spends a 1024-thread workgroup per query -- right for ONE search, where latency is everything -- but in a batch every
// workgroup then issues its 196 KB of row loads at once, waits for the memory system to serve all of them, and computes with
// one wave while 15 wait (1024 queries x 64 rows x 768 dimensions: 124 us = 3.2 TB/s of master rows).  Here a query's 64
// candidates are rescored by FOUR workgroups of 256 threads (16 rows each: 8 workgroups per CU, out of step with each other,
// so some load while others add), and a second launch of one wave per query ranks, checks the bound and emits.  Same
// arithmetic, same order: per row, products by all threads, one lane adds them in index order.
constexpr int BF_ROWS = 16;
#ifndef VL_BF_NCH
#define VL_BF_NCH 4
#endif
constexpr int BF_NCH = VL_BF_NCH;  // 48-column tiles fetched together: 4 = 192 columns in flight per thread (24 registers of row
                                   // data, 8 workgroups per CU); 8 = a dim-384 row in one round trip but 104 registers = 4 per CU
template <int METRIC>
__global__ __launch_bounds__(256) void k_batch_rescore(const Cand32* __restrict__ lists, const double* __restrict__ master,
                                                       const double* __restrict__ q64, uint32_t dim, double* __restrict__ scores)
{
    const uint32_t q = blockIdx.x >> 2, quarter = blockIdx.x & 3u;
    __shared__ RescoreLds<METRIC, BF_ROWS, BF_NCH> rs;
    __shared__ uint32_t sh_pos[BF_ROWS];
    __shared__ int sh_n;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const Cand32* mine = lists + (size_t)q * KP + quarter * BF_ROWS;
    q64 += (size_t)q * dim;
    constexpr int QN = (RP_CH * BF_NCH + 255) / 256;
    double q_first[QN];
#pragma unroll
    for (int j = 0; j < QN; ++j) {
        const uint32_t c = (uint32_t)(tid + j * 256);
        q_first[j] = q64[c < dim ? c : dim - 1];
    }
    if (wave == 0) {
        const uint32_t p = lane < BF_ROWS ? mine[lane].pos : POS_SENTINEL;
        if (lane < BF_ROWS) sh_pos[lane] = p;
        const unsigned long long real = __ballot(p != POS_SENTINEL);  // sorted list: the real entries come first
        if (lane == 0) sh_n = __popcll(real);
    }
    __syncthreads();
    const int n_mine = sh_n;
    Acc64<METRIC> A;
    rescore_rows_par<METRIC, BF_ROWS, 256, BF_NCH>(master, q64, dim, sh_pos, n_mine, rs, A, q_first);
    if (wave == 0 && lane < BF_ROWS) scores[(size_t)q * KP + quarter * BF_ROWS + lane] = lane < n_mine ? A.score() : 0.0;
}

template <int METRIC>
__global__ __launch_bounds__(256) void k_batch_rank_emit(const Cand32* __restrict__ lists, const double* __restrict__ scores,
                                                         const double* __restrict__ q_norms, uint32_t nq, uint32_t ld,
                                                         uint64_t n_rows, uint32_t k, double R, double in_extra,
                                                         SearchResultBlock* __restrict__ out, ShardRecordSink sink)
{
    const int lane = lane_id();
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const Cand32 e = lists[(size_t)q * KP + lane];
    const int n_cand = __popcll(__ballot(e.pos != POS_SENTINEL));
    const double sc = lane < n_cand ? scores[(size_t)q * KP + lane] : 0.0;
    rank_check_emit<METRIC>(sc, e.pos, read_lane(e.key, KP - 1), n_cand, k, n_rows, ld, R, q_norms[q], in_extra, out + q, 0u, sink, q);
}

// K2: merge partial lists, rescore, rank, bound-check.  One workgroup of 1024 threads.
template <int METRIC>
__global__ __launch_bounds__(1024) void k_merge_finalize(const Cand32* __restrict__ partials, int n_lists,
                                                         size_t list_stride_q, const double* __restrict__ master,
                                                         const double* __restrict__ q64,
                                                         const double* __restrict__ q_norms, uint32_t dim,
                                                         uint32_t ld, uint64_t n_rows, uint32_t k, double R,
                                                         double in_extra, SearchResultBlock* __restrict__ out,
                                                         uint32_t seq, ShardRecordSink sink)
{
    // one workgroup per query of the batch
    partials += (size_t)blockIdx.x * list_stride_q;
    q64 += (size_t)blockIdx.x * dim;
    out += blockIdx.x;
    const double Q = q_norms[blockIdx.x];
    constexpr int NW = 16;
    __shared__ Cand32 sh_lists[NW * WAVE];
    __shared__ RescoreLds<METRIC> rs;
    __shared__ uint32_t sh_pos[KP];
    __shared__ float sh_key[KP];
    __shared__ int sh_ncand;

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;

    VL_STAMP(0);
    // phase 1: n_lists <= 64 sorted lists: up to 16 waves x 4 bitonic folds, then a tree merge over the waves that hold one
    TopList<float> L;
    double q_first[1] = {0.0};
    {
        const int first = wave * 4;
        int count = n_lists - first;
        count = count < 0 ? 0 : (count > 4 ? 4 : count);
        Cand32 e[4];
        fold_lists4_load<Cand32>(e, partials, first, count);
        // the query's first block is requested BEHIND the list loads (vector loads return in order: in front of
        // them the wave would sit out the pinned-memory round trip before it can merge) and is consumed only in phase 2
        if (threadIdx.x < RP_BW) q_first[0] = q64[threadIdx.x < dim ? threadIdx.x : dim - 1];
        fold_lists4_merge<float, Cand32>(L, e, count);
    }
    VL_STAMP(1);
    block_merge_n<float, Cand32>(L, sh_lists, (n_lists + 3) / 4);
    VL_STAMP(2);
    if (wave == 0) {
        sh_pos[lane] = L.pos;
        sh_key[lane] = L.key;
        const unsigned long long real = __ballot(L.pos != POS_SENTINEL);
        if (lane == 0) sh_ncand = __popcll(real);
    }
    __syncthreads();
    const int n_cand = sh_ncand;

    // phase 2: exact f64 rescoring of the candidates
    Acc64<METRIC> A;
    VL_STAMP(3);
    rescore_rows_par<METRIC, KP, 1024>(master, q64, dim, sh_pos, n_cand, rs, A, q_first);
    VL_STAMP(8);

    // phase 3: rank by (score desc, pos asc), bound check, emit
    if (wave == 0) {
        const bool valid = lane < n_cand;
        const double sc = valid ? A.score() : 0.0;
        rank_check_emit<METRIC>(sc, sh_pos[lane], sh_key[KP - 1], n_cand, k, n_rows, ld, R, Q, in_extra, out, seq, sink, blockIdx.x);
    }
}

// K2 for 60 < k <= 220: the workgroup lists are cut into P partitions (any partition of the rows will do), each
// keeps its own top 64, and all 64 P candidates are rescored and ranked together.  A row left out of partition
// p's list has scan key <= that list's 64th key t_p, so the answer stands iff score[k-1] > max_p B(t_p) -- the
// same proof as K2's, per partition.  On random data the 64th key of one of P partitions sits near global rank
// 64 P, so k up to ~ 64 P - 30 is certified; anything else falls back to the exact kernels as before.
constexpr int KMULTI_PARTS = 4;

template <int METRIC>
__global__ __launch_bounds__(1024) void k_merge_finalize_multi(const Cand32* __restrict__ lists, int n_lists,
                                                               size_t part_stride, int n_parts,
                                                               const double* __restrict__ master,
                                                               const double* __restrict__ q64,
                                                               const double* __restrict__ q_norms, uint32_t dim,
                                                               uint32_t ld, uint64_t n_rows, uint32_t k, double R,
                                                               SearchResultBlock* __restrict__ out)
{
    constexpr int NW = 16;
    __shared__ Cand32 sh_lists[NW * WAVE];
    __shared__ RescoreLds<METRIC> rs;
    __shared__ uint32_t sh_pos[KMULTI_PARTS * KP];
    __shared__ float sh_key[KMULTI_PARTS * KP];
    __shared__ double sh_score[KMULTI_PARTS * KP];
    __shared__ int sh_ncand[KMULTI_PARTS];
    __shared__ int sh_flags;
    __shared__ double sh_cut;
    __shared__ int sh_has_cut;
    const double Q = q_norms[0];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    double q_first[1] = {0.0};
    if (threadIdx.x < RP_BW) q_first[0] = q64[threadIdx.x < dim ? threadIdx.x : dim - 1];
    if (threadIdx.x == 0) {
        sh_flags = 0;
        sh_has_cut = 0;
        sh_cut = 0.0;
    }
    // phase 1: per partition, n_lists <= 64 sorted lists -> its top 64
    for (int p = 0; p < n_parts; ++p) {
        TopList<float> L;
        const int first = wave * 4;
        int count = n_lists - first;
        count = count < 0 ? 0 : (count > 4 ? 4 : count);
        fold_lists4<float, Cand32>(L, lists + (size_t)p * part_stride, first, count);
        block_merge<float, Cand32, NW>(L, sh_lists);
        if (wave == 0) {
            sh_pos[p * KP + lane] = L.pos;
            sh_key[p * KP + lane] = L.key;
            const unsigned long long real = __ballot(L.pos != POS_SENTINEL);
            if (lane == 0) sh_ncand[p] = __popcll(real);
        }
        __syncthreads();
    }
    // phase 2: exact f64 rescoring, 64 rows at a time through the same LDS tile
    for (int p = 0; p < n_parts; ++p) {
        Acc64<METRIC> A;
        rescore_rows_par<METRIC, KP, 1024>(master, q64, dim, sh_pos + p * KP, sh_ncand[p], rs, A, q_first);
        if (wave == 0) sh_score[p * KP + lane] = lane < sh_ncand[p] ? A.score() : 0.0;
        __syncthreads();
    }
    // phase 3: rank all candidates by (score desc, pos asc), bound check, emit
    const int t = threadIdx.x;
    const int total_slots = n_parts * KP;
    const bool valid = t < total_slots && (t % KP) < sh_ncand[t / KP];
    const double sc = valid ? sh_score[t] : 0.0;
    const uint32_t my_pos = t < total_slots ? sh_pos[t] : POS_SENTINEL;
    if (valid && sc != sc) atomicOr(&sh_flags, (int)(RESULT_HAS_NAN | RESULT_NEEDS_EXACT));
    int rank = 0;
    for (int j = 0; j < total_slots; ++j) {
        if ((j % KP) >= sh_ncand[j / KP]) continue;
        const double sj = sh_score[j];
        const uint32_t pj = sh_pos[j];
        rank += (sj > sc || (sj == sc && pj < my_pos)) ? 1 : 0;
    }
    const uint32_t k_eff = (uint64_t)k < n_rows ? k : (uint32_t)n_rows;
    if (valid && rank == (int)k_eff - 1) {
        sh_cut = sc;
        sh_has_cut = 1;
    }
    __syncthreads();
    if (valid && rank < (int)k_eff) {
        out[rank / KP].pos[rank % KP] = my_pos;
        out[rank / KP].score[rank % KP] = sc;
    }
    if (t == 0) {
        uint32_t flags = (uint32_t)sh_flags;
        int total = 0;
        double B = -1.0e300;
        bool excluded_rows = false;
        for (int p = 0; p < n_parts; ++p) {
            total += sh_ncand[p];
            if (sh_ncand[p] == KP) {  // a full list: rows behind its 64th key exist (or may exist)
                excluded_rows = true;
                const double bp = bound_for_key<METRIC>(sh_key[p * KP + KP - 1], ld, R, Q, 0.0);
                B = bp > B ? bp : B;
            }
        }
        if (!sh_has_cut || (uint32_t)total < k_eff) flags |= RESULT_NEEDS_EXACT;
        else if (excluded_rows && !(sh_cut > B)) flags |= RESULT_NEEDS_EXACT;
        if ((uint64_t)total < n_rows && !excluded_rows) flags |= RESULT_NEEDS_EXACT;  // short lists that do not cover the index
        out[0].n_out = k_eff;
        out[0].flags = flags;
    }
}

// ---------------------------------------------------------------------------------------------
// Exact path
// ---------------------------------------------------------------------------------------------
constexpr int EX_ROWS = 256;  // rows per workgroup tile (one per thread)
constexpr int EX_CH = 16;     // columns per chunk: 128 B of each row = one full line

template <int METRIC>
__global__ __launch_bounds__(256) void k_exact_scan(const double* __restrict__ master,
                                                    const double* __restrict__ q64, uint64_t n, uint32_t dim,
                                                    double* __restrict__ scores, uint32_t* __restrict__ nan_flag)
{
    // One thread owns one row and walks it in index order (the reference's serial f64 chains), so the rows
    // of a tile are transposed through LDS: 16 consecutive threads fetch one row's 128-byte line, then each
    // thread reads its own row from the padded tile.  The next chunk's loads are issued into registers
    // BEFORE the current chunk's arithmetic, so HBM latency hides behind the 16 dependent steps.
    __shared__ double tile[EX_ROWS][EX_CH + 1];
    __shared__ double qtile[EX_CH];
    const int tid = threadIdx.x;
    const uint64_t n_tiles = (n + EX_ROWS - 1) / EX_ROWS;
    const uint32_t n_chunks = (dim + EX_CH - 1) / EX_CH;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t row0 = t * EX_ROWS;
        const uint64_t rows_here = (n - row0) < (uint64_t)EX_ROWS ? (n - row0) : (uint64_t)EX_ROWS;
        Acc64<METRIC> A;
        A.init();
        double pre[EX_CH];
        double qpre = 0.0;
        auto fetch = [&](uint32_t c0) {
            const uint32_t cw = (dim - c0) < (uint32_t)EX_CH ? (dim - c0) : (uint32_t)EX_CH;
#pragma unroll
            for (int i = 0; i < EX_CH; ++i) {
                // clamped, never predicated: a load under a condition makes hipcc wait for each one in turn;
                // rows >= rows_here are never walked and columns >= cw never stepped, so duplicates are harmless
                const int idx = tid + i * 256;
                uint64_t r = (uint64_t)(idx / EX_CH);
                uint32_t cc = (uint32_t)(idx % EX_CH);
                r = r < rows_here ? r : rows_here - 1;
                cc = cc < cw ? cc : cw - 1;
                pre[i] = master[(row0 + r) * dim + c0 + cc];
            }
            {
                uint32_t qc = (uint32_t)(tid % EX_CH);
                qc = qc < cw ? qc : cw - 1;
                qpre = q64[c0 + qc];
            }
        };
        fetch(0);
        for (uint32_t ch = 0; ch < n_chunks; ++ch) {
            const uint32_t c0 = ch * EX_CH;
            const uint32_t cw = (dim - c0) < (uint32_t)EX_CH ? (dim - c0) : (uint32_t)EX_CH;
            __syncthreads();  // the previous chunk's readers are done with the tile
#pragma unroll
            for (int i = 0; i < EX_CH; ++i) {
                const int idx = tid + i * 256;
                tile[idx / EX_CH][idx % EX_CH] = pre[i];
            }
            if (tid < EX_CH) qtile[tid] = qpre;
            __syncthreads();
            if (ch + 1 < n_chunks) fetch(c0 + EX_CH);  // in flight during the arithmetic below
            if ((uint64_t)tid < rows_here) {
                if (cw == (uint32_t)EX_CH) {
#pragma unroll
                    for (uint32_t cc = 0; cc < (uint32_t)EX_CH; ++cc) A.step(tile[tid][cc], qtile[cc]);
                } else {
                    for (uint32_t cc = 0; cc < cw; ++cc) A.step(tile[tid][cc], qtile[cc]);
                }
            }
        }
        if ((uint64_t)tid < rows_here) {
            const double sc = A.score();
            scores[row0 + tid] = sc;
            if (sc != sc) atomicOr(nan_flag, 1u);
        }
    }
}

// top-64 of scores[] by (score desc, pos asc): per-wave lists, then per-workgroup merge
// `after` (optional): the result block of the previous round; only rows ranked strictly behind its last
// entry are offered, so round r yields ranks 64 r .. 64 r + 63 of the full (score desc, pos asc) order.
__global__ __launch_bounds__(256) void k_select64(const double* __restrict__ scores, uint32_t n,
                                                  const SearchResultBlock* __restrict__ after,
                                                  Cand64* __restrict__ out)
{
    __shared__ Cand64 sh[4 * WAVE];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const uint32_t n_steps = (n + WAVE - 1) / WAVE;
    const uint32_t n_waves = gridDim.x * 4;
    TopList<double> L;
    L.init();
    const bool cut = after != nullptr;
    const double cut_key = cut ? after->score[KP - 1] : 0.0;
    const uint32_t cut_pos = cut ? after->pos[KP - 1] : 0u;
    for (uint32_t s = blockIdx.x * 4 + wave; s < n_steps; s += n_waves) {
        const uint32_t row = s * WAVE + lane;
        bool valid = row < n;
        const double sc = valid ? scores[row] : 0.0;
        if (cut) valid = valid && better<double>(cut_key, cut_pos, sc, row);
        L.offer(sc, row, valid);
    }
    block_merge<double, Cand64, 4>(L, sh);
    if (wave == 0) {
        Cand64 e;
        e.key = L.key;
        e.pos = L.pos;
        e.pad = 0;
        out[(size_t)blockIdx.x * KP + lane] = e;
    }
}

__global__ __launch_bounds__(1024) void k_merge64_emit(const Cand64* __restrict__ partials, int n_lists,
                                                       uint64_t n_rows, uint32_t k,
                                                       const uint32_t* __restrict__ nan_flag,
                                                       SearchResultBlock* __restrict__ out)
{
    constexpr int NW = 16;
    __shared__ Cand64 sh[NW * WAVE];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    TopList<double> L;
    {
        const int first = wave * 4;
        int count = n_lists - first;
        count = count < 0 ? 0 : (count > 4 ? 4 : count);
        fold_lists4<double, Cand64>(L, partials, first, count);
    }
    block_merge<double, Cand64, NW>(L, sh);
    if (wave == 0) {
        const uint32_t k_eff = (uint64_t)k < n_rows ? k : (uint32_t)n_rows;
        if ((uint32_t)lane < k_eff || k == (uint32_t)KP) {  // a full round also carries its cut entry for the next one
            out->pos[lane] = L.pos;
            out->score[lane] = L.key;
        }
        if (lane == 0) {
            out->n_out = k_eff;
            out->flags = (*nan_flag) ? RESULT_HAS_NAN : 0u;
        }
    }
}

// ---- device-wide bitonic sort on (score desc, pos asc) for k > 64 ------------------------------
// okey: u64 whose ASCENDING unsigned order is the DESCENDING order of the score (-0.0 == +0.0).
__device__ __forceinline__ unsigned long long desc_key(double s)
{
    s = s + 0.0;  // -0.0 -> +0.0: partial_cmp treats them as equal
    unsigned long long b = (unsigned long long)__double_as_longlong(s);
    const unsigned long long asc = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return ~asc;
}

__global__ void k_sort_init(const double* __restrict__ scores, uint64_t n, uint64_t cap,
                            unsigned long long* __restrict__ okeys, uint32_t* __restrict__ opos)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < cap;
         i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < n) {
            okeys[i] = desc_key(scores[i]);
            opos[i] = (uint32_t)i;
        } else {
            okeys[i] = ~0ull;
            opos[i] = POS_SENTINEL;
        }
    }
}

__device__ __forceinline__ bool pair_greater(unsigned long long ka, uint32_t pa, unsigned long long kb, uint32_t pb)
{
    return ka > kb || (ka == kb && pa > pb);
}

constexpr int SORT_LOCAL = 2048;  // elements sorted in LDS by one 1024-thread workgroup

// All stages with j < SORT_LOCAL for k in [k_lo, k_hi] (k_lo == k_hi > SORT_LOCAL: tail of one k).
__global__ __launch_bounds__(1024) void k_bitonic_local(unsigned long long* __restrict__ okeys,
                                                        uint32_t* __restrict__ opos, uint64_t k_lo, uint64_t k_hi)
{
    __shared__ unsigned long long sk[SORT_LOCAL];
    __shared__ uint32_t sp[SORT_LOCAL];
    const uint64_t base = (uint64_t)blockIdx.x * SORT_LOCAL;
    const int tid = threadIdx.x;
    sk[tid] = okeys[base + tid];
    sp[tid] = opos[base + tid];
    sk[tid + 1024] = okeys[base + tid + 1024];
    sp[tid + 1024] = opos[base + tid + 1024];
    __syncthreads();
    for (uint64_t k = k_lo; k <= k_hi; k <<= 1) {
        uint64_t j = (k >> 1) < (uint64_t)(SORT_LOCAL >> 1) ? (k >> 1) : (uint64_t)(SORT_LOCAL >> 1);
        for (; j >= 1; j >>= 1) {
            // thread t handles the pair (i, i ^ j) with i the t-th index whose bit j is clear
            const uint32_t jj = (uint32_t)j;
            const uint32_t i = ((tid & ~(jj - 1)) << 1) | (tid & (jj - 1));
            const uint32_t l = i | jj;
            const bool up = (((base + i) & k) == 0);
            const unsigned long long ka = sk[i], kb = sk[l];
            const uint32_t pa = sp[i], pb = sp[l];
            const bool gt = pair_greater(ka, pa, kb, pb);
            if (gt == up) {
                sk[i] = kb;
                sp[i] = pb;
                sk[l] = ka;
                sp[l] = pa;
            }
            __syncthreads();
        }
    }
    okeys[base + tid] = sk[tid];
    opos[base + tid] = sp[tid];
    okeys[base + tid + 1024] = sk[tid + 1024];
    opos[base + tid + 1024] = sp[tid + 1024];
}

__global__ void k_bitonic_global(unsigned long long* __restrict__ okeys, uint32_t* __restrict__ opos,
                                 uint64_t cap, uint64_t j, uint64_t k)
{
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < (cap >> 1);
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint64_t l = i | j;
        const bool up = ((i & k) == 0);
        const unsigned long long ka = okeys[i], kb = okeys[l];
        const uint32_t pa = opos[i], pb = opos[l];
        if (pair_greater(ka, pa, kb, pb) == up) {
            okeys[i] = kb;
            opos[i] = pb;
            okeys[l] = ka;
            opos[l] = pa;
        }
    }
}

__global__ void k_sort_emit(const uint32_t* __restrict__ opos, const double* __restrict__ scores, uint64_t k,
                            uint32_t* __restrict__ out_pos, double* __restrict__ out_scores)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < k;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = opos[i];
        out_pos[i] = p;
        out_scores[i] = scores[p];
    }
}

// ---------------------------------------------------------------------------------------------
// HNSW distance callbacks
// ---------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(256) void k_hnsw_dist(const double* __restrict__ master,
                                                   const double* __restrict__ q64, uint32_t dim,
                                                   const uint32_t* __restrict__ positions, uint32_t m,
                                                   unsigned long long* __restrict__ out)
{
    __shared__ double tile[KP][RESCORE_CH + 1];
    __shared__ double qtile[RESCORE_CH];
    __shared__ uint32_t sh_pos[KP];
    const uint32_t base = blockIdx.x * KP;
    const int n_here = (m - base) < (uint32_t)KP ? (int)(m - base) : KP;
    if (threadIdx.x < (unsigned)n_here) sh_pos[threadIdx.x] = positions[base + threadIdx.x];
    __syncthreads();
    Acc64<METRIC> A;
    // the reference calls distance(query, stored) (a = query); every callback is symmetric in
    // (a, b) bit for bit: x*y and |x-y| commute, (x-y)^2 == (y-x)^2.
    rescore_rows<METRIC, 256>(master, q64, dim, sh_pos, n_here, tile, qtile, A);
    if (threadIdx.x < (unsigned)n_here) out[base + threadIdx.x] = hnsw_quantise<METRIC>(A);
}

// ---------------------------------------------------------------------------------------------
// Ingest: f64 master rows -> f32 slab + 1/|row| cache + domain flags
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ingest(const double* __restrict__ master, float* __restrict__ slab,
                                                float* __restrict__ inv_norm, uint8_t* __restrict__ flags,
                                                IngestStats* __restrict__ stats, uint64_t n, uint32_t dim,
                                                uint32_t ld)
{
    const int lane = lane_id();
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += n_waves) {
        const double* src = master + row * dim;
        float* dst = slab + row * ld;
        double ss = 0.0, mx = 0.0;
        bool bad = false;
        for (uint32_t c = lane; c < ld; c += WAVE) {
            const double v = c < dim ? src[c] : 0.0;
            dst[c] = (float)v;  // round to nearest even
            ss += v * v;
            const double av = fabs(v);
            mx = av > mx ? av : mx;
            bad |= !(av <= 1.797693134862315708e308);  // inf or NaN
        }
#pragma unroll
        for (int o = WAVE / 2; o >= 1; o >>= 1) {
            ss += __shfl_xor(ss, o);
            const double m2 = __shfl_xor(mx, o);
            mx = m2 > mx ? m2 : mx;
        }
        const bool any_bad = __ballot(bad) != 0ull;
        if (lane == 0) {
            const double norm = sqrt(ss);
            const bool ood = any_bad || !(mx <= 1099511627776.0 /* 2^40 */) || !(norm <= 1.0e300) ||
                             (norm != 0.0 && norm < 9.094947017729282e-13 /* 2^-40 */);
            inv_norm[row] = (norm > 0.0 && !ood) ? (float)(1.0 / norm) : 0.0f;
            flags[row] = ood ? ROW_OUT_OF_DOMAIN : 0;
            if (ood) {
                atomicAdd(&stats->n_out_of_domain, 1u);
            } else {
                atomicMax(&stats->max_norm_bits, (unsigned long long)__double_as_longlong(norm));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Embedding ingest: the model's f32 output -> the f64 row the index stores, with the
// arithmetic of src/embeddings.rs:171-179: x as f64; norm = sqrt(sum of x*x in index order);
// x / norm when norm > 0, the widened values unchanged otherwise.
// One wave owns 64 rows per trip: the rows' columns go through a padded LDS tile 64 at a time (coalesced
// 256-byte reads), lane r then adds row r's squares in index order; the scaled rows are written
// coalesced, each lane fetching its row's norm by shuffle.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_f32(const float* __restrict__ emb, uint64_t n, uint32_t dim,
                                                   int normalize, double* __restrict__ out)
{
    __shared__ float tile[4][WAVE][WAVE + 1];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t n_groups = (n + WAVE - 1) / WAVE;
    // every wave of a workgroup runs the same trips (the barriers below), rows past n are skipped
    for (uint64_t g0 = (uint64_t)blockIdx.x * 4; g0 < n_groups; g0 += (uint64_t)gridDim.x * 4) {
        const uint64_t base = (g0 + wave) * WAVE;
        double ss = -0.0;  // `.sum::<f64>()` folds from -0.0
        if (normalize) {
            for (uint32_t c0 = 0; c0 < dim; c0 += WAVE) {
                const uint32_t c = c0 + lane;
                __syncthreads();
                for (int rr = 0; rr < WAVE; ++rr) {
                    const uint64_t row = base + rr;
                    tile[wave][rr][lane] = (row < n && c < dim) ? emb[row * dim + c] : 0.f;
                }
                __syncthreads();
                const uint32_t w = dim - c0 < (uint32_t)WAVE ? dim - c0 : (uint32_t)WAVE;
                for (uint32_t j = 0; j < w; ++j) {
                    const double v = (double)tile[wave][lane][j];
                    ss += v * v;
                }
            }
        }
        const double norm = sqrt(ss);
        const int scale = (normalize && norm > 0.0) ? 1 : 0;
        for (int rr = 0; rr < WAVE; ++rr) {
            const uint64_t row = base + rr;
            if (row >= n) break;  // wave-uniform
            const double nr = __shfl(norm, rr);
            const int sc = __shfl(scale, rr);
            for (uint32_t c = lane; c < dim; c += WAVE) {
                const double v = (double)emb[row * dim + c];
                out[row * dim + c] = sc ? v / nr : v;
            }
        }
    }
}

// The same arithmetic for a FEW rows (a batch of query embeddings): one wave per row instead of one lane per row, so
// that 1024 rows fill the chip (k_embed_f32 keeps 16 waves busy on them: 484 us at dim 768).  The lanes square the row's
// values in parallel into LDS; ONE lane adds them in index order (the reference's `.sum::<f64>()` is sequential, and so
// is this chain: dim dependent f64 adds); the scaled row is written coalesced.  Rows of at most EMBED_ROW_MAX values.
constexpr uint32_t EMBED_ROW_MAX = 2048;

__global__ __launch_bounds__(256) void k_embed_f32_rows(const float* __restrict__ emb, uint64_t n, uint32_t dim,
                                                        int normalize, double* __restrict__ out)
{
    __shared__ double sq[4][EMBED_ROW_MAX];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;  // wave-uniform; no workgroup barrier below
    const float* x = emb + row * dim;
    double ss = -0.0;  // `.sum::<f64>()` folds from -0.0
    if (normalize) {
        for (uint32_t c = lane; c < dim; c += WAVE) {
            const double v = (double)x[c];
            sq[wave][c] = v * v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed
        if (lane == 0) {
            uint32_t c = 0;
            for (; c + 8 <= dim; c += 8) {  // 8 LDS reads in flight, then 8 adds in index order
                double t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = sq[wave][c + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += t[j];
            }
            for (; c < dim; ++c) ss += sq[wave][c];
        }
        ss = __shfl(ss, 0);
    }
    const double norm = sqrt(ss);
    const bool scale = normalize && norm > 0.0;
    for (uint32_t c = lane; c < dim; c += WAVE) {
        const double v = (double)x[c];
        out[row * dim + c] = scale ? v / norm : v;
    }
}

template <typename F>
hipError_t dispatch_metric(int metric, F&& f)
{
    switch (metric) {
    case COSINE: return f(std::integral_constant<int, COSINE>{});
    case EUCLIDEAN: return f(std::integral_constant<int, EUCLIDEAN>{});
    case MANHATTAN: return f(std::integral_constant<int, MANHATTAN>{});
    case DOT: return f(std::integral_constant<int, DOT>{});
    default: return hipErrorInvalidValue;
    }
}

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_embed_f32(hipStream_t s, const float* emb, uint64_t n, uint32_t dim, bool normalize, double* out)
{
    if (n == 0 || dim == 0) return hipSuccess;
    if (n <= 16384 && dim <= EMBED_ROW_MAX) {  // a batch of queries, a small add: one wave per row
        hipLaunchKernelGGL(k_embed_f32_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, emb, n, dim, normalize ? 1 : 0, out);
        return hipGetLastError();
    }
    const uint64_t blocks = ((n + WAVE - 1) / WAVE + 3) / 4;
    const int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(k_embed_f32, dim3(grid), dim3(256), 0, s, emb, n, dim, normalize ? 1 : 0, out);
    return hipGetLastError();
}

hipError_t launch_ingest(hipStream_t s, const double* master, float* slab, float* inv_norm, uint8_t* flags,
                         IngestStats* stats, uint64_t n, uint32_t dim, uint32_t ld)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 3) / 4;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(k_ingest, dim3(grid), dim3(256), 0, s, master, slab, inv_norm, flags, stats, n, dim, ld);
    return hipGetLastError();
}

namespace {
// lanes per row for a row of ld4 float4 (generic kernel)
int lanes_per_row(uint32_t ld4)
{
    int g = 1;
    while (g < 64 && (uint32_t)(g * 2) <= ld4) g *= 2;
    return g;
}

// Compiled (G lanes per row, VPL float4 per lane, U row groups in flight, BPC workgroups per CU)
// instantiations of k_scan.  The first entry whose G*VPL equals ld4 is the default; VL_SCAN_G /
// VL_SCAN_U / VL_SCAN_GRID pick another shape (tuning only).  Measured on MI355X at N=10M, dim=384
// (profiles/r01_tune_scan.txt): 12 x 16-byte loads in flight per lane and 3 workgroups per CU
// (12 waves, ~144 KB in flight per CU) stream at 7.17 TB/s; more resident waves were slower.
#define VL_SCAN_VARIANTS(X)                                                                         \
    X(8, 4, 3, 3) X(8, 8, 2, 3) X(8, 12, 1, 3) X(8, 16, 1, 3) X(16, 12, 1, 3) X(16, 16, 1, 3)       \
    X(16, 24, 1, 2) X(32, 1, 8, 4) X(32, 2, 4, 4) X(32, 4, 2, 4) X(32, 8, 1, 4) X(32, 12, 1, 3)      \
    X(32, 3, 2, 4) X(32, 3, 4, 3) X(16, 6, 1, 4) X(16, 6, 2, 1) X(8, 12, 2, 1) X(4, 24, 1, 2)        \
    X(32, 6, 1, 4) X(32, 6, 2, 3) X(16, 12, 2, 1) X(8, 24, 1, 2) X(64, 3, 2, 4) X(64, 3, 4, 3)       \
    X(16, 2, 4, 4) X(8, 4, 4, 3) X(16, 4, 2, 4) X(16, 4, 3, 3) X(8, 8, 1, 4) X(16, 8, 1, 4)          \
    X(16, 8, 2, 2)

struct ScanShape {
    bool special;
    int g, vpl, u, bpc;
};

ScanShape scan_shape(uint32_t ld4)
{
    static const int table[][4] = {
#define VL_ROW(G, VPL, U, BPC) {G, VPL, U, BPC},
        VL_SCAN_VARIANTS(VL_ROW)
#undef VL_ROW
    };
    const int want_g = env_int("VL_SCAN_G", 0), want_u = env_int("VL_SCAN_U", 0);
    const int n = (int)(sizeof(table) / sizeof(table[0]));
    int pick = -1;
    for (int i = 0; i < n; ++i) {
        if ((uint32_t)(table[i][0] * table[i][1]) != ld4) continue;
        if (pick < 0) pick = i;
        if (want_g && table[i][0] == want_g && (!want_u || table[i][2] == want_u)) {
            pick = i;
            break;
        }
        if (!want_g && want_u && table[i][2] == want_u && table[i][0] == table[pick][0]) {
            pick = i;
            break;
        }
    }
    if (pick < 0) return {false, lanes_per_row(ld4), 0, 1, 4};
    return {true, table[pick][0], table[pick][1], table[pick][2], table[pick][3]};
}
}  // namespace

namespace {
// Workgroups of `kernel` (256 threads, no dynamic LDS) that are resident at once on this device:
// the scan is one persistent wave of workgroups that grid-stride over the rows, so every CU streams
// for the whole launch and no second, partially filled round of workgroups trails behind.
int resident_blocks(const void* kernel, int* n_cus)
{
    struct Entry {
        const void* k;
        int dev;
        int blocks;
        int cus;
    };
    static Entry cache[256];
    static int n_cache = 0;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < n_cache; ++i)
        if (cache[i].k == kernel && cache[i].dev == dev) {
            *n_cus = cache[i].cus;
            return cache[i].blocks;
        }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    // the occupancy API can over-report by one block per CU (MI355X_MICROARCH.md): stay one below 8
    if (per_cu > 7) per_cu = 7;
    const int blocks = per_cu * cus;
    if (n_cache < 256) cache[n_cache++] = {kernel, dev, blocks, cus};
    *n_cus = cus;
    return blocks;
}

int scan_grid(uint64_t n, const ScanShape& sh, const void* kernel)
{
    const uint64_t rps = 64 / sh.g;
    const uint64_t steps = (n + rps - 1) / rps;
    const uint64_t per_block = 4ull * sh.u;
    uint64_t blocks = (steps + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)env_int("VL_SCAN_GRID", 0);
    if (cap == 0) {
        int cus = 0;
        const uint64_t resident = (uint64_t)resident_blocks(kernel, &cus);
        cap = (uint64_t)sh.bpc * (uint64_t)cus;
        if (cap > resident) cap = resident;
    }
    if (blocks > cap) blocks = cap;
    if (blocks > (uint64_t)SCAN_MAX_GRID) blocks = SCAN_MAX_GRID;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
}  // namespace

#ifdef VL_DBG_STAMPS
extern "C" int vl_dbg_read_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vl_dbg_stamps), sizeof(unsigned long long) * 32);
}
#endif

bool scan_takes_qarg(uint32_t ld)
{
    if ((ld & 3) || ld > (uint32_t)SCAN_QARG_FLOATS) return false;
    return scan_shape(ld / 4).special;
}

hipError_t launch_scan(hipStream_t s, int metric, const float* slab, const float* inv_norm, const double* q64,
                       uint64_t n, uint32_t dim, uint32_t ld, Cand32* partials, ScanPlan* plan, const float* q32_host)
{
    if (n == 0 || n >= 0xFFFFFFFFull || (ld & 3)) return hipErrorInvalidValue;
    const uint32_t ld4 = ld / 4;
    const ScanShape sh = scan_shape(ld4);
    const f32x4* slab4 = reinterpret_cast<const f32x4*>(slab);
    const uint32_t n32 = (uint32_t)n;
    const bool qarg = q32_host != nullptr && sh.special && ld <= (uint32_t)SCAN_QARG_FLOATS;
    if (!qarg && !q64) return hipErrorInvalidValue;
    int grid = 0;
    hipError_t rc = dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        if (sh.special) {
            bool launched = false;
#define VL_TRY_VARIANT(G, VPL, U, BPC)                                                                          \
    if (!launched && sh.g == G && sh.vpl == VPL && sh.u == U) {                                             \
        if constexpr (G * VPL * 4 <= SCAN_QARG_FLOATS) {                                                     \
            if (qarg) {                                                                                      \
                auto kern = k_scan<MM, G, VPL, U>;                                                           \
                grid = scan_grid(n, sh, reinterpret_cast<const void*>(kern));                                \
                ScanQArg qa;                                                                                 \
                memcpy(qa.v, q32_host, (size_t)ld * sizeof(float));                                          \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, slab4, inv_norm, n32, partials, qa);   \
                launched = true;                                                                             \
            }                                                                                                \
        }                                                                                                    \
        if (!launched) {                                                                                     \
            auto kern = k_scan_q64<MM, G, VPL, U>;                                                           \
            grid = scan_grid(n, sh, reinterpret_cast<const void*>(kern));                                    \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, slab4, inv_norm, q64, dim, n32, partials); \
            launched = true;                                                                                 \
        }                                                                                                    \
    }
            VL_SCAN_VARIANTS(VL_TRY_VARIANT)
#undef VL_TRY_VARIANT
            if (!launched) return hipErrorInvalidValue;
        } else {
#define VL_SCAN_GEN(G)                                                                                        \
    case G: {                                                                                                 \
        auto kern = k_scan_generic<MM, G>;                                                                    \
        grid = scan_grid(n, sh, reinterpret_cast<const void*>(kern));                                         \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, slab4, inv_norm, q64, dim, n32, ld4, partials);       \
    } break;
            switch (sh.g) {
                VL_SCAN_GEN(1)
                VL_SCAN_GEN(2)
                VL_SCAN_GEN(4)
                VL_SCAN_GEN(8)
                VL_SCAN_GEN(16)
                VL_SCAN_GEN(32)
                VL_SCAN_GEN(64)
            default: return hipErrorInvalidValue;
            }
#undef VL_SCAN_GEN
        }
        return hipGetLastError();
    });
    if (plan) {
        plan->grid = grid;
        plan->variant = sh.special ? (sh.g * 10000 + sh.vpl * 100 + sh.u) : -sh.g;
    }
    return rc;
}

// Batch shapes: one (G, VPL) per supported row length, QB = 8 queries per pass.
// (lanes per row, 16-byte chunks per lane): row strides 32, 64, 96, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1536 floats.
// The first shape whose G * VPL matches the stride runs.  Strides 32 and 64 take FOUR lanes per row (16 rows per wave step):
// at such lengths a (row, query) pair is ~35 instructions of which 19 are the cross-lane reduction, the key compare, the
// ballot and the branch -- per ROW costs that 16 rows per step halve (round 4, `tools/manhattan_probe.py`).
#define VL_BATCH_SHAPES(X) X(4, 2) X(4, 4) X(8, 1) X(8, 2) X(8, 3) X(8, 4) X(8, 6) X(8, 8) X(8, 10) X(8, 12) X(8, 16) X(16, 10) X(16, 12) X(16, 16) X(16, 24)

bool scan_batch_supported(uint32_t ld)
{
    const uint32_t ld4 = ld / 4;
    bool ok = false;
#define VL_CHK(G, VPL) ok = ok || ((uint32_t)(G * VPL) == ld4);
    VL_BATCH_SHAPES(VL_CHK)
#undef VL_CHK
    return ok && (ld % 4) == 0;
}

hipError_t launch_scan_batch(hipStream_t s, int metric, const float* slab, const float* inv_norm, const double* q64,
                             uint32_t nq, uint64_t n, uint32_t dim, uint32_t ld, Cand32* partials, ScanPlan* plan)
{
    if (n == 0 || n >= 0xFFFFFFFFull || (ld & 3) || nq == 0 || nq > (uint32_t)SCAN_BATCH_MAX_QUERIES) return hipErrorInvalidValue;
    const uint32_t ld4 = ld / 4;
    const f32x4* slab4 = reinterpret_cast<const f32x4*>(slab);
    const uint32_t n32 = (uint32_t)n;
    const uint32_t groups = (nq + SCAN_BATCH_QB - 1) / SCAN_BATCH_QB;
    int grid = 0;
    hipError_t rc = dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        bool launched = false;
#define VL_TRY_BATCH(G, VPL)                                                                                  \
    if (!launched && (uint32_t)(G * VPL) == ld4) {                                                            \
        auto kern = k_scan_batch<MM, G, VPL, SCAN_BATCH_QB>;                                                  \
        int cus = 0;                                                                                          \
        const int resident = resident_blocks(reinterpret_cast<const void*>(kern), &cus);                      \
        const uint64_t steps = (n + (64 / G) - 1) / (64 / G);                                                 \
        uint64_t blocks = (steps + 3) / 4;                                                                    \
        uint64_t cap = (uint64_t)env_int("VL_BATCH_GRID", 0);                                                 \
        if (cap == 0) cap = (uint64_t)resident;                                                               \
        if (cap > (uint64_t)SCAN_BATCH_MAX_GRID) cap = SCAN_BATCH_MAX_GRID;                                   \
        if (groups > 1) {                                                                                     \
            /* several groups in one launch: the resident workgroups are shared out over them, every query's */ \
            /* lists must fit the finalize kernel without a merge level (<= 64) and the partial-list buffer */  \
            cap = std::max<uint64_t>(1, cap / groups);                                                        \
            cap = std::min<uint64_t>(cap, 64);                                                                \
            cap = std::min<uint64_t>(cap, std::max<uint64_t>(1, PARTIALS32_LISTS / ((uint64_t)groups * SCAN_BATCH_QB))); \
        }                                                                                                     \
        if (blocks > cap) blocks = cap;                                                                       \
        grid = (int)(blocks < 1 ? 1 : blocks);                                                                \
        hipLaunchKernelGGL(kern, dim3(grid, groups), dim3(256), 0, s, slab4, inv_norm, q64, nq, dim, n32, partials); \
        launched = true;                                                                                      \
    }
        VL_BATCH_SHAPES(VL_TRY_BATCH)
#undef VL_TRY_BATCH
        if (!launched) return hipErrorInvalidValue;
        return hipGetLastError();
    });
    if (plan) {
        plan->grid = grid;
        plan->variant = 0;
    }
    return rc;
}

namespace {
// Reduce n_lists sorted lists per query to <= 64 with merge levels.  `scratch` holds two ping-pong
// regions of nq x 64 lists.  Returns the final list array, count and per-query stride.
template <typename K, typename C>
const C* reduce_lists(hipStream_t s, const C* lists, int* n_lists, size_t* stride_q, int nq, C* scratch)
{
    int n = *n_lists;
    size_t in_stride = *stride_q;
    int ping = 0;
    while (n > 64) {
        const int blocks = (n + 63) / 64;
        const size_t out_stride = (size_t)64 * KP;
        C* out = scratch + (size_t)ping * nq * out_stride;
        hipLaunchKernelGGL((k_merge_lists<K, C>), dim3(blocks, nq), dim3(1024), 0, s, lists, n, in_stride, out,
                           out_stride);
        lists = out;
        in_stride = out_stride;
        n = blocks;
        ping ^= 1;
    }
    *n_lists = n;
    *stride_q = in_stride;
    return lists;
}
}  // namespace

hipError_t launch_merge_finalize(hipStream_t s, int metric, Cand32* partials, int n_lists, int nq,
                                 const double* master, const double* q64, const double* q_norms, uint32_t dim,
                                 uint64_t n_rows, uint32_t k, double max_row_norm, SearchResultBlock* out,
                                 double in_extra, uint32_t seq, const ShardRecordSink* sink)
{
    if (seq && nq != 1) return hipErrorInvalidValue;
    const ShardRecordSink sk = sink ? *sink : ShardRecordSink{};
    const uint32_t ld = (dim + 3u) & ~3u;
    size_t stride = (size_t)n_lists * KP;
    const Cand32* lists = reduce_lists<float, Cand32>(s, partials, &n_lists, &stride, nq, partials + PARTIALS32_LISTS * KP);
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        hipLaunchKernelGGL((k_merge_finalize<MM>), dim3(nq), dim3(1024), 0, s, lists, n_lists, stride, master, q64,
                           q_norms, dim, ld, n_rows, k, max_row_norm, in_extra, out, seq, sk);
        return hipGetLastError();
    });
}

hipError_t launch_batch_finalize(hipStream_t s, int metric, const Cand32* lists, int nq, const double* master, const double* q64,
                                 const double* q_norms, uint32_t dim, uint64_t n_rows, uint32_t k, double max_row_norm,
                                 SearchResultBlock* out, double in_extra, double* score_scratch, const ShardRecordSink* sink)
{
    if (nq <= 0 || !score_scratch) return hipErrorInvalidValue;
    const uint32_t ld = (dim + 3u) & ~3u;
    const ShardRecordSink sk = sink ? *sink : ShardRecordSink{};
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        hipLaunchKernelGGL((k_batch_rescore<MM>), dim3((unsigned)nq * 4u), dim3(256), 0, s, lists, master, q64, dim, score_scratch);
        hipLaunchKernelGGL((k_batch_rank_emit<MM>), dim3(((unsigned)nq + 3u) / 4u), dim3(256), 0, s, lists, (const double*)score_scratch,
                           q_norms, (uint32_t)nq, ld, n_rows, k, max_row_norm, in_extra, out, sk);
        return hipGetLastError();
    });
}

hipError_t launch_merge_finalize_multi(hipStream_t s, int metric, Cand32* partials, int n_lists_total, int n_parts,
                                       const double* master, const double* q64, const double* q_norm, uint32_t dim,
                                       uint64_t n_rows, uint32_t k, double max_row_norm, SearchResultBlock* out)
{
    if (n_parts < 2 || n_parts > KMULTI_PARTS || n_lists_total % n_parts != 0) return hipErrorInvalidValue;
    const uint32_t ld = (dim + 3u) & ~3u;
    int n_lists = n_lists_total / n_parts;  // consecutive workgroup lists form one partition
    size_t stride = (size_t)n_lists * KP;
    const Cand32* lists = reduce_lists<float, Cand32>(s, partials, &n_lists, &stride, n_parts, partials + PARTIALS32_LISTS * KP);
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        hipLaunchKernelGGL((k_merge_finalize_multi<MM>), dim3(1), dim3(1024), 0, s, lists, n_lists, stride, n_parts, master,
                           q64, q_norm, dim, ld, n_rows, k, max_row_norm, out);
        return hipGetLastError();
    });
}

hipError_t launch_exact_scan(hipStream_t s, int metric, const double* master, const double* q64, uint64_t n,
                             uint32_t dim, double* scores, uint32_t* nan_flag)
{
    if (n == 0) return hipSuccess;
    const uint64_t tiles = (n + EX_ROWS - 1) / EX_ROWS;
    const int grid = (int)(tiles < 4096 ? tiles : 4096);
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        hipLaunchKernelGGL((k_exact_scan<MM>), dim3(grid), dim3(256), 0, s, master, q64, n, dim, scores, nan_flag);
        return hipGetLastError();
    });
}

int select_grid_for(uint64_t n)
{
    const uint64_t steps = (n + 63) / 64;
    uint64_t blocks = (steps + 3) / 4;
    if (blocks > (uint64_t)SELECT_MAX_GRID) blocks = SELECT_MAX_GRID;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

hipError_t launch_exact_select(hipStream_t s, const double* scores, uint64_t n, uint32_t k, Cand64* partials,
                               const uint32_t* nan_flag, SearchResultBlock* out, const SearchResultBlock* after)
{
    if (n == 0 || n >= 0xFFFFFFFFull || k > (uint32_t)KP) return hipErrorInvalidValue;
    const int grid = select_grid_for(n);
    hipLaunchKernelGGL(k_select64, dim3(grid), dim3(256), 0, s, scores, (uint32_t)n, after, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int n_lists = grid;
    size_t stride = (size_t)n_lists * KP;
    const Cand64* lists = reduce_lists<double, Cand64>(s, partials, &n_lists, &stride, 1, partials + (size_t)SELECT_MAX_GRID * KP);
    hipLaunchKernelGGL(k_merge64_emit, dim3(1), dim3(1024), 0, s, lists, n_lists, n, k, nan_flag, out);
    return hipGetLastError();
}

uint64_t sort_capacity_for(uint64_t n)
{
    uint64_t cap = SORT_LOCAL;
    while (cap < n) cap <<= 1;
    return cap;
}

hipError_t launch_exact_sort(hipStream_t s, const double* scores, uint64_t n, uint64_t k, uint64_t* okeys,
                             uint32_t* opos, uint32_t* out_pos, double* out_scores)
{
    if (n == 0 || n >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint64_t cap = sort_capacity_for(n);
    unsigned long long* ok = reinterpret_cast<unsigned long long*>(okeys);
    const int tgrid = (int)((cap / 256) < 8192 ? (cap / 256) : 8192);
    hipLaunchKernelGGL(k_sort_init, dim3(tgrid), dim3(256), 0, s, scores, n, cap, ok, opos);
    const int lgrid = (int)(cap / SORT_LOCAL);
    // k = 2 .. SORT_LOCAL entirely in LDS
    hipLaunchKernelGGL(k_bitonic_local, dim3(lgrid), dim3(1024), 0, s, ok, opos, (uint64_t)2, (uint64_t)SORT_LOCAL);
    for (uint64_t kk = (uint64_t)SORT_LOCAL << 1; kk <= cap; kk <<= 1) {
        for (uint64_t j = kk >> 1; j >= (uint64_t)SORT_LOCAL; j >>= 1) {
            const uint64_t pairs = cap >> 1;
            const int g = (int)((pairs / 256) < 16384 ? (pairs / 256) : 16384);
            hipLaunchKernelGGL(k_bitonic_global, dim3(g), dim3(256), 0, s, ok, opos, cap, j, kk);
        }
        hipLaunchKernelGGL(k_bitonic_local, dim3(lgrid), dim3(1024), 0, s, ok, opos, kk, kk);
    }
    const uint64_t kk = k < n ? k : n;
    if (kk > 0) {
        const int egrid = (int)(((kk + 255) / 256) < 4096 ? ((kk + 255) / 256) : 4096);
        hipLaunchKernelGGL(k_sort_emit, dim3(egrid), dim3(256), 0, s, opos, scores, kk, out_pos, out_scores);
    }
    return hipGetLastError();
}

hipError_t launch_hnsw_distances(hipStream_t s, int metric, const double* master, const double* q64,
                                 uint32_t dim, const uint32_t* positions, uint32_t m, uint64_t* out)
{
    if (m == 0) return hipSuccess;
    const int grid = (int)((m + KP - 1) / KP);
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        hipLaunchKernelGGL((k_hnsw_dist<MM>), dim3(grid), dim3(256), 0, s, master, q64, dim, positions, m,
                           reinterpret_cast<unsigned long long*>(out));
        return hipGetLastError();
    });
}

}  // namespace vl
