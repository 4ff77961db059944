// mfma_scan.hip -- K4: large-batch flat scan as a bf16 MFMA GEMM with a fused top-k candidate filter.
//
// No reference counterpart (the reference has no batch search, src/lib.rs:224-245).  With hundreds of
// queries per slab pass the scan stops being HBM-bound, so scores[row, query] are
// computed on the matrix cores: v_mfma_f32_32x32x16_bf16, rows as the A operand (streamed through
// LDS, one contiguous 32-row tile at a time), 32 queries per wave as the B operand held in registers
// for the whole launch (128 queries per workgroup share every row tile).
//
// bf16 scores are only a CANDIDATE FILTER.  The Q x N score matrix is never written:
//   pass 0 (a 1/16 sample of the rows): every workgroup reports, per query, the best key of its
//     contiguous row range; the 64th largest of those maxima is a valid lower bound T_q of the
//     query's 64th best key (64 distinct rows reach it);
//   pass 1 (all rows): keys >= T_q are appended to the query's candidate buffer (about 10^3 of 10^7);
//   then per query: top-64 of the buffer -> the same finalize kernel as the f32 path: exact f64
//     rescoring from the master rows, (score desc, position asc) ranking and the bound check, now
//     with the bf16 input-rounding term (2^-8 relative per operand) in the bound.
// A query whose check fails (or whose buffer overflows) is redone on the f32 path by the host.
#include "mfma_scan.hpp"

#include <stdlib.h>

#include <type_traits>

#include "device_common.hpp"

namespace vl {
using namespace dev;
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


constexpr int MF_ROWS = 32;  // rows per tile (MFMA M)

// order-preserving float <-> int (for max over possibly negative keys)
__device__ __forceinline__ int enc_f(float f)
{
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec_f(int e) { return __int_as_float(e >= 0 ? e : e ^ 0x7FFFFFFF); }

template <int KSTEPS, int MODE, int METRIC, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_mfma_scan(const __bf16* __restrict__ slab16,
                                                   const float* __restrict__ row_aux,
                                                   const __bf16* __restrict__ q16, uint32_t nq, uint32_t n_tiles,
                                                   uint32_t n_rows, int* __restrict__ gmax, uint32_t n_groups,
                                                   const float* __restrict__ thr, Cand32* __restrict__ cand,
                                                   uint32_t* __restrict__ cnt, uint32_t cap)
{
    constexpr int LDB = KSTEPS * 16;            // bf16 elements per row
    constexpr int ROW_BYTES = LDB * 2;
    constexpr int LDS_ROW = ROW_BYTES + 16;     // +16 B: 32 rows land on 16 distinct 4-bank slots
    constexpr int CHUNKS = MF_ROWS * ROW_BYTES / 16;  // 16-byte pieces per tile
    constexpr int NT = NWAVES * 64;                   // threads per workgroup
    constexpr int MF_QPB = NWAVES * 32;               // queries per workgroup (MFMA N = 32 per wave)
    constexpr int CPT = (CHUNKS + NT - 1) / NT;       // pieces per thread
    constexpr int CPR = ROW_BYTES / 16;               // pieces per row
    constexpr int NBUF = (2 * MF_ROWS * LDS_ROW <= 60000) ? 2 : 1;  // static LDS stays under 64 KB
    __shared__ __attribute__((aligned(16))) unsigned char a_lds[NBUF][MF_ROWS * LDS_ROW];
    __shared__ __attribute__((aligned(16))) float inv_lds[NBUF][MF_ROWS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const uint32_t q = blockIdx.y * MF_QPB + wave * 32 + col;  // this lane's query (B column)
    const bool q_valid = q < nq;

    bf16x8 bfrag[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
        bfrag[s] = *reinterpret_cast<const bf16x8*>(q16 + (size_t)q * LDB + 16 * s + 8 * half);

    // tile schedule: MODE 0 gives every workgroup ONE contiguous range (its maxima describe distinct
    // rows); MODE 1 grid-strides so that all workgroups stream neighbouring tiles
    uint32_t t, t_end, t_step;
    if (MODE == 0) {
        const uint32_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
        t = blockIdx.x * per;
        t_end = t + per < n_tiles ? t + per : n_tiles;
        t_step = 1;
    } else {
        t = blockIdx.x;
        t_end = n_tiles;
        t_step = gridDim.x;
    }

    float thr_q = INFINITY;  // padding queries never pass
    if (MODE == 1 && q_valid) thr_q = thr[q];
    float run_max = -INFINITY;

    // Register prefetch ring: DEPTH tiles are in flight per workgroup.  One workgroup per CU leaves
    // only the loop itself to hide the ~2 us HBM latency, and one tile's MFMAs cover ~0.7 us of it.
    constexpr int DEPTH = (CPT <= 3) ? 4 : 2;
    u32x4 stage[DEPTH][CPT];
    float stage_inv[DEPTH];
    auto issue_loads = [&](uint32_t tile, u32x4(&st)[CPT], float& st_inv) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(slab16) + (size_t)tile * MF_ROWS * ROW_BYTES;
        const uint32_t row0 = tile * MF_ROWS;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < CHUNKS && row0 + (uint32_t)(c / CPR) < n_rows)
                v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (size_t)c * 16));
            st[i] = v;
        }
        // per-row scalar of the key: cosine 1/|x| (key = dot * inv), Euclidean |x|^2 (key = 2 dot - |x|^2)
        if (tid < MF_ROWS) st_inv = (METRIC != DOT && row0 + tid < n_rows) ? row_aux[row0 + tid] : 1.0f;
    };
    auto write_lds = [&](int buf, const u32x4(&st)[CPT], float st_inv) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            if (c < CHUNKS) {
                const int r = c / CPR, cc = c % CPR;
                *reinterpret_cast<u32x4*>(&a_lds[buf][r * LDS_ROW + cc * 16]) = st[i];
            }
        }
        if (tid < MF_ROWS) inv_lds[buf][tid] = st_inv;
    };

#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
        stage_inv[j] = 1.0f;
        if (t + (uint32_t)j * t_step < t_end) issue_loads(t + (uint32_t)j * t_step, stage[j], stage_inv[j]);
    }
    int buf = 0;
    for (uint32_t base = t; base < t_end; base += DEPTH * t_step) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const uint32_t t = base + (uint32_t)j * t_step;
        if (t >= t_end) break;  // wave-uniform
        write_lds(buf, stage[j], stage_inv[j]);
        __syncthreads();
        if (t + DEPTH * t_step < t_end) issue_loads(t + DEPTH * t_step, stage[j], stage_inv[j]);

        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // A fragments are read from LDS one GROUP ahead of the MFMAs that consume them: left alone the
        // compiler reuses one register quad and every MFMA waits out a full LDS round trip
        const unsigned char* arow = &a_lds[buf][col * LDS_ROW + half * 16];
        constexpr int GS = 4;
        constexpr int NG = KSTEPS / GS;
        static_assert(KSTEPS % GS == 0, "K steps come in whole groups");
        bf16x8 afrag[2][GS];
#pragma unroll
        for (int j = 0; j < GS; ++j) afrag[0][j] = *reinterpret_cast<const bf16x8*>(arow + j * 32);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int j = 0; j < GS; ++j)
                    afrag[(g + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(arow + ((g + 1) * GS + j) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < GS; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[g & 1][j], bfrag[g * GS + j], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // C layout: column = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5):
        // registers 4g..4g+3 are the four consecutive rows 8g + 4*half + {0..3}.
        // Epilogue on the common path = 16 multiplies, a max tree and ONE compare against the
        // query's threshold; the per-row work only runs for the rare tile that holds a candidate.
        const uint32_t row0 = t * MF_ROWS;
        float keys[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 aux = *reinterpret_cast<const f32x4*>(&inv_lds[buf][8 * g4 + 4 * half]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float key = acc[4 * g4 + j];
                if (METRIC == COSINE) key *= aux[j];
                if (METRIC == EUCLIDEAN) key = 2.0f * key - aux[j];  // = |q|^2 - |x - q|^2, |q|^2 is per query
                keys[4 * g4 + j] = key;
            }
        }
        if (row0 + MF_ROWS > n_rows) {  // last, partial tile (wave-uniform)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (row0 + (uint32_t)((reg & 3) + 8 * (reg >> 2) + 4 * half) >= n_rows) keys[reg] = -INFINITY;
        }
        float m = fmaxf(keys[0], keys[1]);
#pragma unroll
        for (int reg = 2; reg < 16; ++reg) m = fmaxf(m, keys[reg]);
        if (MODE == 0) {
            run_max = fmaxf(run_max, m);
        } else if (m >= thr_q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (keys[reg] >= thr_q) {
                    const uint32_t slot = atomicAdd(&cnt[q], 1u);
                    if (slot < cap) {
                        Cand32 e;
                        e.key = keys[reg];
                        e.pos = row0 + (uint32_t)((reg & 3) + 8 * (reg >> 2) + 4 * half);
                        cand[(size_t)q * cap + slot] = e;
                    }
                }
            }
        }
        if (NBUF == 1) __syncthreads();  // single buffer: everyone is done reading before the next write
        buf = (NBUF == 2) ? (buf ^ 1) : 0;
      }
    }
    if (MODE == 0) {
        const float other = __shfl_xor(run_max, 32);  // the two half-waves saw different rows of the same query
        run_max = fmaxf(run_max, other);
        if (q_valid && half == 0 && blockIdx.x < n_groups) gmax[(size_t)q * n_groups + blockIdx.x] = enc_f(run_max);
    }
}

// T_q = the 64th largest group maximum (a lower bound of the query's 64th best key); -inf when fewer
// than 64 groups exist.  One wave per query.
__global__ __launch_bounds__(256) void k_thresholds(const int* __restrict__ gmax, uint32_t n_groups, uint32_t nq,
                                                    float* __restrict__ thr)
{
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    TopList<float> L;
    L.init();
    for (uint32_t g0 = 0; g0 < n_groups; g0 += 64) {
        const uint32_t g = g0 + lane;
        const bool ok = g < n_groups;
        const float v = ok ? dec_f(gmax[(size_t)q * n_groups + g]) : -INFINITY;
        L.offer(v, g, ok && v > -INFINITY);
    }
    // lane 63 holds the 64th largest (or the sentinel when there are fewer than 64 finite maxima)
    const float t = read_lane(L.key, 63);
    const uint32_t p = read_lane(L.pos, 63);
    if (lane == 0) thr[q] = (p == POS_SENTINEL) ? -INFINITY : t;
}

// Per query: top-64 of its candidate buffer by (key desc, position asc) -> one sorted list.
__global__ __launch_bounds__(256) void k_select_candidates(const Cand32* __restrict__ cand,
                                                           const uint32_t* __restrict__ cnt, uint32_t cap,
                                                           Cand32* __restrict__ lists)
{
    __shared__ Cand32 sh[4 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t n = cnt[q];
    TopList<float> L;
    L.init();
    if (n <= cap) {  // an overflowed buffer yields an all-sentinel list: the host redoes that query
        for (uint32_t i0 = wave * 64; i0 < n; i0 += 256) {
            const uint32_t i = i0 + lane;
            const bool ok = i < n;
            Cand32 e;
            e.key = 0.f;
            e.pos = 0;
            if (ok) e = cand[(size_t)q * cap + i];
            L.offer(e.key, e.pos, ok);
        }
    }
    block_merge<float, Cand32, 4>(L, sh);
    if (wave == 0) {
        Cand32 e;
        e.key = L.key;
        e.pos = L.pos;
        lists[(size_t)q * KP + lane] = e;
    }
}

// f64 queries -> bf16 [nq_pad, ldb] (zero padded rows and columns), rounded f64 -> f32 -> bf16 (RNE).
__global__ void k_queries_bf16(const double* __restrict__ q64, uint32_t nq, uint32_t nq_pad, uint32_t dim,
                               uint32_t ldb, __bf16* __restrict__ out)
{
    const size_t total = (size_t)nq_pad * ldb;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t qi = (uint32_t)(i / ldb), c = (uint32_t)(i % ldb);
        const float v = (qi < nq && c < dim) ? (float)q64[(size_t)qi * dim + c] : 0.0f;
        out[i] = (__bf16)v;
    }
}

// f64 master rows -> bf16 slab rows [n, ldb]
__global__ void k_rows_bf16(const double* __restrict__ master, uint64_t n, uint32_t dim, uint32_t ldb,
                            __bf16* __restrict__ out)
{
    const size_t total = (size_t)n * ldb;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ldb;
        const uint32_t c = (uint32_t)(i % ldb);
        const float v = c < dim ? (float)master[r * dim + c] : 0.0f;
        out[i] = (__bf16)v;
    }
}

// |row|^2 in f64 (wave per row), rounded once to f32
__global__ __launch_bounds__(256) void k_rows_sqnorm(const double* __restrict__ master, uint64_t n, uint32_t dim,
                                                     float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += n_waves) {
        double ss = 0.0;
        for (uint32_t c = lane; c < dim; c += 64) {
            const double v = master[row * dim + c];
            ss += v * v;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) out[row] = (float)ss;
    }
}

}  // namespace

#define VL_MFMA_KSTEPS(X) X(8) X(16) X(24) X(32) X(48)

namespace {
// 8 waves (256 queries per workgroup) while the query fragments leave room for two waves per SIMD;
// dim >= 768 keeps 192 registers of fragments per lane, so it runs 4 waves with the whole register file
int env_waves(uint32_t ldb)
{
    const char* v = getenv("VL_MFMA_WAVES");
    const int w = v && *v ? atoi(v) : (ldb >= 768 ? 4 : 8);
    return w == 4 ? 4 : 8;
}
int env_grid()
{
    const char* v = getenv("VL_MFMA_GRID");
    const int g = v && *v ? atoi(v) : 512;
    return g < 1 ? 1 : g;
}
}  // namespace

bool mfma_scan_supported(uint32_t dim, int metric)
{
    if (metric != COSINE && metric != DOT && metric != EUCLIDEAN) return false;
    const uint32_t ldb = mfma_ldb(dim);
    bool ok = false;
#define VL_CHK(K) ok = ok || (ldb == (uint32_t)(K * 16));
    VL_MFMA_KSTEPS(VL_CHK)
#undef VL_CHK
    return ok;
}

hipError_t launch_rows_bf16(hipStream_t s, const double* master, uint64_t n, uint32_t dim, void* out_bf16,
                            float* out_sqnorm)
{
    if (n == 0) return hipSuccess;
    const uint32_t ldb = mfma_ldb(dim);
    const size_t total = (size_t)n * ldb;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(k_rows_bf16, dim3(grid), dim3(256), 0, s, master, n, dim, ldb, reinterpret_cast<__bf16*>(out_bf16));
    const int g2 = (int)std::min<uint64_t>((n + 3) / 4, 8192);
    hipLaunchKernelGGL(k_rows_sqnorm, dim3(g2), dim3(256), 0, s, master, n, dim, out_sqnorm);
    return hipGetLastError();
}

hipError_t launch_mfma_candidates(hipStream_t s, int metric, const void* slab_bf16, const float* row_aux,
                                  const double* q64, uint32_t nq, uint64_t n_rows, uint32_t dim,
                                  const MfmaScratch& w, Cand32* out_lists)
{
    if (nq == 0 || n_rows == 0 || n_rows >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (!mfma_scan_supported(dim, metric) || nq > w.nq_cap) return hipErrorInvalidValue;
    const uint32_t ldb = mfma_ldb(dim);
    const int nwaves = env_waves(ldb);
    const uint32_t qpb = (uint32_t)nwaves * 32;
    const uint32_t nq_pad = (nq + qpb - 1) / qpb * qpb;
    if (nq_pad > w.nq_cap) return hipErrorInvalidValue;
    const uint32_t n_tiles = (uint32_t)((n_rows + MF_ROWS - 1) / MF_ROWS);
    __bf16* q16 = reinterpret_cast<__bf16*>(w.q_bf16);
    const __bf16* slab = reinterpret_cast<const __bf16*>(slab_bf16);
    {
        const size_t total = (size_t)nq_pad * ldb;
        const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(k_queries_bf16, dim3(grid), dim3(256), 0, s, q64, nq, nq_pad, dim, ldb, q16);
    }
    hipError_t e = hipMemsetAsync(w.cnt, 0, (size_t)nq_pad * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;

    // pass 0: a sample of the tiles, one contiguous range per workgroup = one "group" per workgroup
    uint32_t sample_tiles = n_tiles / 16;
    if (sample_tiles < (uint32_t)MFMA_GROUPS) sample_tiles = n_tiles < (uint32_t)MFMA_GROUPS ? n_tiles : (uint32_t)MFMA_GROUPS;
    const uint32_t n_groups = sample_tiles < (uint32_t)MFMA_GROUPS ? sample_tiles : (uint32_t)MFMA_GROUPS;
    const dim3 grid0(n_groups, nq_pad / qpb);
    const uint64_t sample_rows = std::min<uint64_t>((uint64_t)sample_tiles * MF_ROWS, n_rows);
    const int pass1_blocks = (int)std::min<uint32_t>(n_tiles, (uint32_t)env_grid());
    const dim3 grid1(pass1_blocks, nq_pad / qpb);

    bool launched = false;
#define VL_LAUNCH3(K, MET, NW)                                                                                          \
    {                                                                                                                   \
        hipLaunchKernelGGL((k_mfma_scan<K, 0, MET, NW>), grid0, dim3(NW * 64), 0, s, slab, row_aux, q16, nq,           \
                           sample_tiles, (uint32_t)sample_rows, w.gmax, n_groups, (const float*)nullptr,               \
                           (Cand32*)nullptr, (uint32_t*)nullptr, 0u);                                                   \
        hipLaunchKernelGGL(k_thresholds, dim3((nq + 3) / 4), dim3(256), 0, s, w.gmax, n_groups, nq, w.thr);             \
        hipLaunchKernelGGL((k_mfma_scan<K, 1, MET, NW>), grid1, dim3(NW * 64), 0, s, slab, row_aux, q16, nq, n_tiles,  \
                           (uint32_t)n_rows, (int*)nullptr, 0u, w.thr, w.cand, w.cnt, (uint32_t)MFMA_CAND_CAP);         \
        launched = true;                                                                                                \
    }
#define VL_LAUNCH(K)                                                            \
    if (!launched && ldb == (uint32_t)(K * 16)) {                               \
        if (metric == COSINE) {                                                 \
            if (nwaves == 8) VL_LAUNCH3(K, COSINE, 8) else VL_LAUNCH3(K, COSINE, 4) \
        } else if (metric == EUCLIDEAN) {                                       \
            if (nwaves == 8) VL_LAUNCH3(K, EUCLIDEAN, 8) else VL_LAUNCH3(K, EUCLIDEAN, 4) \
        } else {                                                                \
            if (nwaves == 8) VL_LAUNCH3(K, DOT, 8) else VL_LAUNCH3(K, DOT, 4)   \
        }                                                                       \
    }
    VL_MFMA_KSTEPS(VL_LAUNCH)
#undef VL_LAUNCH
#undef VL_LAUNCH3
    if (!launched) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_select_candidates, dim3(nq), dim3(256), 0, s, w.cand, w.cnt, (uint32_t)MFMA_CAND_CAP, out_lists);
    return hipGetLastError();
}

}  // namespace vl
