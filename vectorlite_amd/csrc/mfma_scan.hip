// mfma_scan.hip -- K4: large-batch flat scan as a bf16 MFMA GEMM with a fused top-k candidate filter.
//
// No reference counterpart (the reference has no batch search, src/lib.rs:224-245).  With hundreds of
// queries per slab pass the scan stops being HBM-bound, so scores[row, query] are
// computed on the matrix cores (v_mfma_f32_16x16x32_bf16).  Two kernels:
//   k_mfma_rows (K4r, the shipped one, every row stride 128 .. 768): row-stationary waves -- a workgroup's queries
//     sit in LDS, every wave streams its own 32-row blocks from the fragment-major slab into A-fragment registers;
//     workgroups are placed XCD-aware (xcd_map.hpp), the launch plan is filter_plan.hpp;
//   k_mfma_scan (round 1's LDS-tile kernel, only with VL_MFMA_KERNEL=tile): rows streamed through an LDS tile, 32
//     queries per wave held in registers.
//
// bf16 scores are only a CANDIDATE FILTER.  The Q x N score matrix is never written:
//   pass 0 (a sample: 1/32 .. 1/64 of the rows, at least 32768 of them): every group of row blocks reports, per query,
//     its best key; the 64th largest of those maxima is a valid lower bound T_q of the query's 64th best key
//     (64 distinct rows reach it);
//   pass 1 (all rows, in up to four stages of growing size; between stages T_q is raised to the 64th best
//     candidate found so far): keys >= T_q are appended to the query's candidate buffer (a few hundred of 10^7);
//   then per query: top-64 of the buffer -> the same finalize kernel as the f32 path: exact f64
//     rescoring from the master rows, (score desc, position asc) ranking and the bound check, now
//     with the bf16 input-rounding term (2^-8 relative per operand) in the bound.
// A query whose check fails (or whose buffer overflows) is redone on the f32 path by the host.
#include "mfma_scan.hpp"
#include "filter_plan.hpp"
#include "xcd_map.hpp"

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
static_assert(2 * MF_ROWS == (int)MFMA_TILE_ROWS, "the slab is allocated in whole tiles of the largest shape (SUB = 2)");

// order-preserving float <-> int (for max over possibly negative keys)
__device__ __forceinline__ int enc_f(float f)
{
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec_f(int e) { return __int_as_float(e >= 0 ? e : e ^ 0x7FFFFFFF); }

// The 64th largest of the 64 * VALS values a wave holds (VALS per lane; pad with -inf): a bit-by-bit search on the
// order-preserving encoding, 32 rounds of VALS ballots -- no insertion loop, no dependence on the order of the input.
// Fewer than 64 finite values give -inf (the padding is then the 64th largest).
template <int VALS>
__device__ __forceinline__ float kth64_of_wave(const float (&v)[VALS])
{
    uint32_t u[VALS];
#pragma unroll
    for (int i = 0; i < VALS; ++i) u[i] = (uint32_t)enc_f(v[i]) ^ 0x80000000u;  // unsigned order = float order
    uint32_t t = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t c = t | (1u << b);
        uint32_t n_ge = 0;
#pragma unroll
        for (int i = 0; i < VALS; ++i) n_ge += (uint32_t)__popcll(__ballot(u[i] >= c));
        if (n_ge >= 64u) t = c;  // wave-uniform
    }
    return dec_f((int)(t ^ 0x80000000u));
}

// max of three finite floats in one instruction (fmaxf() costs a canonicalising v_max x, x per operand in IEEE mode;
// the operands here are MFMA sums of finite bf16 products, or -inf)
__device__ __forceinline__ float max3f(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int KSTEPS, int MODE, int METRIC, int NWAVES, int QT, int KSPLIT = 1, int SUB = 1>
__global__ __launch_bounds__(NWAVES * 64) void k_mfma_scan(const __bf16* __restrict__ slab16,
                                                   const float* __restrict__ row_nrm,
                                                   const float* __restrict__ row_sqn,
                                                   const __bf16* __restrict__ q16, uint32_t nq, uint32_t n_tiles,
                                                   uint32_t n_rows, int* __restrict__ gmax, uint32_t n_groups,
                                                   const float* __restrict__ thr, Cand32* __restrict__ cand,
                                                   uint32_t* __restrict__ cnt, uint32_t cap, uint32_t tile_begin)
{
    constexpr int LDB = KSTEPS * 16;            // bf16 elements per row
    constexpr int ROW_BYTES = LDB * 2;
    // +32 B: a 16x16x32 A fragment is read as row (lane & 15) + 16 rb, 16 bytes at k-group (lane >> 4); with two
    // 16-byte slots of padding per row the 16 lanes of every ds_read_b128 group hit 16 distinct slots
    constexpr int LDS_ROW = ROW_BYTES + 32;
    // SUB = 2: a tile is 64 rows = two 32-row MFMA blocks worked off one after the other between the same pair of
    // barriers, which halves the workgroup barriers (and the per-tile staging code) per flop
    static_assert(SUB == 1 || (SUB == 2 && KSPLIT == 1), "two sub-tiles only without the K split");
    constexpr int TR = MF_ROWS * SUB;                 // rows per tile
    constexpr int CHUNKS = TR * ROW_BYTES / 16;       // 16-byte pieces per tile
    constexpr int NT = NWAVES * 64;                   // threads per workgroup
    // KSPLIT = 2 (dim 768): waves w and w + QWAVES serve the SAME 32 QT queries, each over one half of K, so a wave
    // keeps half of the query fragments (96 instead of 192 VGPRs) and two waves fit on a SIMD; the partial sums
    // meet in LDS once per tile and each wave of the pair finishes one of the two 16-row blocks.
    static_assert(KSPLIT == 1 || KSPLIT == 2, "K is whole or halved");
    constexpr int QWAVES = NWAVES / KSPLIT;           // waves with distinct query tiles
    constexpr int NRB = 2 / KSPLIT;                   // 16-row blocks a wave finishes
    constexpr int MF_QPB = QWAVES * 32 * QT;          // queries per workgroup: QT 32-column MFMA tiles per wave
    constexpr int CPT = (CHUNKS + NT - 1) / NT;       // pieces per thread
    constexpr int CPR = ROW_BYTES / 16;               // pieces per row
    // Candidates found in the loop go to a workgroup ring in LDS and are flushed to the per-query
    // global buffers AFTER the loop: a (rare, conditional) global atomic inside the loop would make
    // the loop's vmcnt bookkeeping path-dependent and collapse the prefetch ring to depth 1.
    constexpr int RING = (MODE == 1) ? 1024 : 1;
    constexpr int XCH = (KSPLIT == 2) ? 2 * NWAVES * QT * 2 * 64 * 16 : 0;  // bytes of the partial-sum exchange
    constexpr int NBUF = (2 * TR * LDS_ROW + RING * 10 + XCH <= 150000) ? 2 : 1;  // gfx950: 160 KB of LDS per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char a_lds[NBUF][TR * LDS_ROW];
    __shared__ __attribute__((aligned(16))) float inv_lds[NBUF][TR];  // |x|   (dot, Euclidean)
    __shared__ __attribute__((aligned(16))) float sqn_lds[NBUF][TR];  // |x|^2 (Euclidean)
    __shared__ f32x4 xch[KSPLIT == 2 ? 2 : 1][KSPLIT == 2 ? NWAVES : 1][QT][2][KSPLIT == 2 ? 64 : 1];
    __shared__ float rm_lds[KSPLIT == 2 ? NWAVES * QT * 2 * 16 : 1];
    __shared__ float ring_key[RING];
    __shared__ uint32_t ring_pos[RING];
    __shared__ unsigned short ring_q[RING];
    // Every wave owns one segment of the ring and counts its entries in a wave-uniform register:
    // appending a candidate needs no atomic and no other wave (see the epilogue).
    constexpr int SEG = RING / NWAVES > 0 ? RING / NWAVES : 1;
    __shared__ uint32_t wave_cnt[NWAVES];
    uint32_t my_cnt = 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int qwave = wave % QWAVES, khalf = wave / QWAVES;  // query tile owner, K half (0 when KSPLIT == 1)
    // v_mfma_f32_16x16x32_bf16 (holds a higher clock than the 32x32x16 form on this chip: 1.76 vs 1.27-1.45 PF in
    // the bare fragment loop, tools/micro/mfma_loop.hip).  A wave still owns 32 rows x 32 queries per QT tile,
    // as 2 row blocks x 2 query blocks of 16: lane = (c16, kg) holds row/query c16 of its block and the 8 values
    // k = 32 s + 8 kg .. + 7 of K step s.
    const int c16 = lane & 15, kg = lane >> 4;
    constexpr int KS32 = KSTEPS / 2;  // K = 32 per MFMA
    static_assert(KSTEPS % 2 == 0, "whole 32-deep K steps");
    constexpr int KSW = KS32 / KSPLIT;  // K steps of this wave
    static_assert(KS32 % KSPLIT == 0, "K halves are whole steps");
    uint32_t q[QT][2];
    bool q_valid[QT][2];
    bf16x8 bfrag[QT][2][KSW];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            q[qt][qb] = blockIdx.y * MF_QPB + (qwave * QT + qt) * 32 + qb * 16 + c16;
            q_valid[qt][qb] = q[qt][qb] < nq;
#pragma unroll
            for (int s = 0; s < KSW; ++s)
                bfrag[qt][qb][s] =
                    *reinterpret_cast<const bf16x8*>(q16 + (size_t)q[qt][qb] * LDB + 32 * (s + khalf * KSW) + 8 * kg);
        }

    // tile schedule: workgroup x takes tiles x, x + gridDim.x, ... in both modes, so neighbouring
    // workgroups stream neighbouring tiles.  In MODE 0 that residue class is the workgroup's "group":
    // its maximum belongs to rows no other group holds.
    uint32_t t = tile_begin + blockIdx.x;  // MODE 1 runs in stages over [tile_begin, n_tiles)
    const uint32_t t_end = n_tiles, t_step = gridDim.x;

    float thr_q[QT][2], run_max[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            thr_q[qt][qb] = INFINITY;  // padding queries never pass
            if (MODE == 1 && q_valid[qt][qb]) thr_q[qt][qb] = thr[q[qt][qb]];
            run_max[qt][qb] = -INFINITY;
        }

    // Register prefetch ring: DEPTH tiles are in flight per workgroup.  One workgroup per CU leaves
    // only the loop itself to hide the ~2 us HBM latency, and one tile's MFMAs cover a fraction of it.
    constexpr int DEPTH = (CPT <= 3) ? 4 : 2;  // (a third tile in flight at dim 768 changed nothing: 3.09 vs 3.07 ms)
    u32x4 stage[DEPTH][CPT];
    float stage_inv[DEPTH], stage_sqn[DEPTH];
    // Loads are UNCONDITIONAL (addresses are clamped instead of predicated): a load under a branch
    // makes the vmcnt bookkeeping path-dependent and hipcc then drains the whole ring (vmcnt(0))
    // before every use, which exposes one full HBM latency per tile.
    static_assert(CHUNKS % NT == 0, "a tile is a whole number of 16-byte pieces per thread");
    auto issue_loads = [&](uint32_t tile, u32x4(&st)[CPT], float& st_inv, float& st_sqn) {
        // Rows are dense (ROW_BYTES = CPR * 16), so a tile is ONE contiguous block: a workgroup-uniform base plus
        // 16 * (tid + i NT) per piece -- no per-piece row arithmetic, no clamp.  The slab is allocated in whole
        // tiles (MFMA_TILE_ROWS), so the last, partial tile reads rows past n_rows; the epilogue masks them.
        const unsigned char* tbase = reinterpret_cast<const unsigned char*>(slab16) + (size_t)tile * (TR * ROW_BYTES);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            // plain (cacheable) loads on purpose: the workgroups of the other query chunks read the same
            // tile at about the same time and are served by the XCD's L2 / the Infinity Cache
            st[i] = *reinterpret_cast<const u32x4*>(tbase + (uint32_t)(tid + i * NT) * 16u);
        }
        // The slab rows are unit-normalised (cosine needs no per-row scalar at all); dot restores
        // x.q = (x^.q) |x|, Euclidean uses key = 2 (x^.q) |x| - |x|^2.
        const uint32_t arow_i = tile * TR + (uint32_t)(tid & (TR - 1));
        st_inv = (METRIC != COSINE) ? row_nrm[arow_i] : 1.0f;
        st_sqn = (METRIC == EUCLIDEAN) ? row_sqn[arow_i] : 0.0f;
    };
    auto write_lds = [&](int buf, const u32x4(&st)[CPT], float st_inv, float st_sqn) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * NT;
            const int r = c / CPR, cc = c % CPR;
            *reinterpret_cast<u32x4*>(&a_lds[buf][r * LDS_ROW + cc * 16]) = st[i];
        }
        if (METRIC != COSINE && tid < TR) inv_lds[buf][tid] = st_inv;
        if (METRIC == EUCLIDEAN && tid < TR) sqn_lds[buf][tid] = st_sqn;
    };

    if (t >= t_end) {  // nothing to do for this workgroup (uniform); MODE 0 still reports -inf maxima
        if (MODE == 0) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    if (q_valid[qt][qb] && kg == 0 && blockIdx.x < n_groups)
                        gmax[(size_t)q[qt][qb] * n_groups + blockIdx.x] = enc_f(-INFINITY);
        }
        return;
    }
    const uint32_t t_last = t + ((t_end - 1 - t) / t_step) * t_step;  // last tile of this workgroup
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
        const uint32_t pt = t + (uint32_t)j * t_step;
        issue_loads(pt < t_end ? pt : t_last, stage[j], stage_inv[j], stage_sqn[j]);
    }
    int buf = 0;
    int xpar = 0;
    // Ring flush: the only global atomics of the kernel.  It runs at trip boundaries (where hipcc
    // drains vmcnt anyway) when the ring is half full, and once after the loop.
    auto flush_ring = [&]() {  // caller has published wave_cnt[] and passed a barrier
        for (uint32_t e = tid; e < (uint32_t)(NWAVES * SEG); e += NT) {
            const uint32_t w = e / SEG, i = e % SEG;
            if (i < wave_cnt[w]) {
                const uint32_t qq = blockIdx.y * MF_QPB + ring_q[e];
                const uint32_t slot = atomicAdd(&cnt[qq], 1u);
                if (slot < cap) {
                    Cand32 c;
                    c.key = ring_key[e];
                    c.pos = ring_pos[e];
                    cand[(size_t)qq * cap + slot] = c;
                }
            }
        }
        // a wave whose segment overflowed lost candidates of ITS 32 QT queries only: those are redone by the host
        if (tid < MF_QPB && (wave_cnt[tid / (32 * QT)] > (uint32_t)SEG ||
                             (KSPLIT == 2 && wave_cnt[tid / (32 * QT) + (KSPLIT - 1) * QWAVES] > (uint32_t)SEG))) {
            const uint32_t qq = blockIdx.y * MF_QPB + tid;
            if (qq < nq) atomicAdd(&cnt[qq], cap + 1u);
        }
        my_cnt = 0;
        __syncthreads();  // everyone is done reading wave_cnt[] and the ring before they are reused
    };
    // hipcc drains the ring (vmcnt(0)) at the loop header but uses counted waits inside straight-line
    // code, so one trip covers UNROLL * DEPTH tiles: one drain per 16 tiles instead of one per 4.
    constexpr int UNROLL = (MODE == 0) ? 1 : 4;  // the sampling pass is 1/16 of the work: keep it small
    for (uint32_t base = t; base < t_end; base += UNROLL * DEPTH * t_step) {
#pragma unroll
        for (int jj2 = 0; jj2 < UNROLL * DEPTH; ++jj2) {
            const int j = jj2 % DEPTH;
            // No early exit: the tail repeats the last tile with its results masked, so that every
            // trip issues the same loads and hipcc can use counted vmcnt waits (a ring, not a drain).
            const uint32_t tile_raw = base + (uint32_t)jj2 * t_step;
            const bool tile_live = tile_raw < t_end;
            const uint32_t tile = tile_live ? tile_raw : t_last;
            write_lds(buf, stage[j], stage_inv[j], stage_sqn[j]);
            __syncthreads();
            {
                const uint32_t nt = tile_raw + DEPTH * t_step;  // refill the slot just consumed (clamped at the tail)
                issue_loads(nt < t_end ? nt : t_last, stage[j], stage_inv[j], stage_sqn[j]);
            }

#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
            f32x4 acc[QT][2][2];  // [row block][query block]: rows 16 rb + 4 kg + reg, query 16 qb + c16
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) acc[qt][rb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            // A fragments are read from LDS one GROUP ahead of the MFMAs that consume them (left alone
            // the compiler reuses one register quad and every MFMA waits out an LDS round trip); each
            // fragment feeds 2 QT MFMAs.
            const unsigned char* arow = &a_lds[buf][(sub * MF_ROWS + c16) * LDS_ROW + kg * 16 + khalf * (KSW * 64)];
            constexpr int GS = 2;            // K = 32 steps per group: 4 fragment reads, 8 QT MFMAs of 16 cycles
            constexpr int NG = KSW / GS;
            static_assert(KSW % GS == 0, "K steps come in whole groups");
            bf16x8 afrag[2][GS][2];
#pragma unroll
            for (int jj = 0; jj < GS; ++jj)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    afrag[0][jj][rb] = *reinterpret_cast<const bf16x8*>(arow + rb * 16 * LDS_ROW + jj * 64);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int jj = 0; jj < GS; ++jj)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
                            afrag[(g + 1) & 1][jj][rb] =
                                *reinterpret_cast<const bf16x8*>(arow + rb * 16 * LDS_ROW + ((g + 1) * GS + jj) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < GS; ++jj)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                            for (int qb = 0; qb < 2; ++qb)
                                acc[qt][rb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    afrag[g & 1][jj][rb], bfrag[qt][qb][g * GS + jj], acc[qt][rb][qb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KSPLIT == 2) {
                // each wave of a pair hands over the row block the other one finishes (khalf 0 keeps rows 0-15, khalf 1
                // rows 16-31); the slot alternates with the tile, so the barrier at the top of the next tile is all that
                // separates this tile's reads from the writes two tiles on
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) xch[xpar][wave][qt][qb][lane] = khalf ? acc[qt][0][qb] : acc[qt][1][qb];
                __syncthreads();
                const int partner = khalf ? wave - QWAVES : wave + QWAVES;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        const f32x4 other = xch[xpar][partner][qt][qb][lane];
                        const f32x4 mine = khalf ? acc[qt][1][qb] : acc[qt][0][qb];
                        acc[qt][0][qb] = mine + other;
                    }
                xpar ^= 1;
            }
            const int rbase = (KSPLIT == 2) ? 16 * khalf : sub * MF_ROWS;  // first row (in the tile) of the block(s) this wave finishes
            // C layout of a 16x16 block: column = lane & 15 (query), row = 4 (lane >> 4) + reg.
            // Epilogue on the common path = the key arithmetic, a max tree and ONE compare per query
            // against its threshold; the per-row work only runs for the rare tile that holds a candidate.
            const uint32_t row0 = tile * TR;
            f32x4 aux[2], aux2[2];
            if (METRIC != COSINE) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) aux[rb] = *reinterpret_cast<const f32x4*>(&inv_lds[buf][rbase + 16 * rb + 4 * kg]);
            }
            if (METRIC == EUCLIDEAN) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) aux2[rb] = *reinterpret_cast<const f32x4*>(&sqn_lds[buf][rbase + 16 * rb + 4 * kg]);
            }
            const bool partial = row0 + TR > n_rows || !tile_live;  // partial last tile / repeated tail tile (wave-uniform)
            // Two copies of the epilogue behind ONE branch: written as a predicate the row mask costs a compare and two
            // selects per key in every tile, and only the index's last tile (or the repeated tail tile) needs it.
            auto epilogue = [&](auto partial_tag) {
                constexpr bool PARTIAL = decltype(partial_tag)::value;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        float keys[4 * NRB];  // rows rbase + 16 rb + 4 kg + jj
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                float key = acc[qt][rb][qb][jj];                                       // cosine: x^.q
                                if (METRIC == DOT) key *= aux[rb][jj];                                 // x.q
                                if (METRIC == EUCLIDEAN) key = 2.0f * key * aux[rb][jj] - aux2[rb][jj];  // |q|^2 - |x - q|^2
                                if (PARTIAL && (!tile_live || row0 + (uint32_t)(rbase + 16 * rb + 4 * kg + jj) >= n_rows)) key = -INFINITY;
                                keys[4 * rb + jj] = key;
                            }
                        }
                        float m2[NRB];
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) m2[rb] = max3f(max3f(keys[4 * rb], keys[4 * rb + 1], keys[4 * rb + 2]), keys[4 * rb + 3], keys[4 * rb + 3]);
                        const float m = NRB == 2 ? max3f(m2[0], m2[NRB - 1], m2[0]) : m2[0];
                        const float tq = thr_q[qt][qb];
                        if (MODE == 0) {
                            run_max[qt][qb] = max3f(run_max[qt][qb], m, m);
                        } else if (__builtin_amdgcn_ballot_w64(m >= tq) != 0ull) {
                            // Rare per wave-tile, but the WHOLE workgroup waits for the slowest wave at the next
                            // barrier, and with 8 waves some wave takes this branch on most tiles: it must be
                            // short.  No atomic (the wave appends to its own ring segment), register groups without a
                            // candidate are skipped with one ballot, slots come from ballot + mbcnt.
#pragma unroll
                            for (int rb = 0; rb < NRB; ++rb) {
                                if (__builtin_amdgcn_ballot_w64(m2[rb] >= tq) == 0ull) continue;  // wave-uniform
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    const float key = keys[4 * rb + jj];
                                    const bool is_cand = key >= tq && key > -INFINITY;  // masked rows are -inf; T_q may be too
                                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(is_cand);
                                    if (mk != 0ull) {  // wave-uniform
                                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32),
                                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                                        const uint32_t slot = my_cnt + rank;
                                        if (is_cand && slot < (uint32_t)SEG) {
                                            const uint32_t e = (uint32_t)wave * SEG + slot;
                                            ring_key[e] = key;
                                            ring_pos[e] = row0 + (uint32_t)(rbase + 16 * rb + 4 * kg + jj);
                                            ring_q[e] = (unsigned short)((qwave * QT + qt) * 32 + qb * 16 + c16);
                                        }
                                        my_cnt += (uint32_t)__popcll(mk);
                                    }
                                }
                            }
                        }
                    }
                }
            };
            if (partial)
                epilogue(std::true_type{});
            else
                epilogue(std::false_type{});
            __builtin_amdgcn_sched_barrier(0);  // keep the sub-tiles' epilogues apart (see below)
            }  // sub
            if (NBUF == 1) __syncthreads();  // single buffer: everyone is done reading before the next write
            buf = (NBUF == 2) ? (buf ^ 1) : 0;
            // one scheduling region per tile: across the 16 unrolled tiles the scheduler otherwise
            // hoists the epilogues' LDS reads and spills hundreds of registers (MODE 0 has no branch
            // in its epilogue to stop it)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {  // trip boundary: flush when some wave's segment is half full (workgroup-uniform decision)
            if (lane == 0) wave_cnt[wave] = my_cnt;
            __syncthreads();
            uint32_t fullest = 0;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) fullest = wave_cnt[w] > fullest ? wave_cnt[w] : fullest;
            if (fullest >= (uint32_t)(SEG / 2)) flush_ring();
            else __syncthreads();  // wave_cnt[] is rewritten at the next boundary
        }
    }
    if (MODE == 1) {  // whatever the last trips left behind
        if (lane == 0) wave_cnt[wave] = my_cnt;
        __syncthreads();
        flush_ring();
    }
    if (MODE == 0) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float mx = run_max[qt][qb];  // the four k-groups of lanes saw different rows of one query
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                run_max[qt][qb] = mx;
            }
        if (KSPLIT == 2) {  // the pair saw different rows of the same queries: khalf 1 hands its maxima to khalf 0
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    if (khalf == 1 && kg == 0) rm_lds[((qwave * QT + qt) * 2 + qb) * 16 + c16] = run_max[qt][qb];
            __syncthreads();
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    if (khalf == 0) run_max[qt][qb] = fmaxf(run_max[qt][qb], rm_lds[((qwave * QT + qt) * 2 + qb) * 16 + c16]);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (q_valid[qt][qb] && kg == 0 && khalf == 0 && blockIdx.x < n_groups)
                    gmax[(size_t)q[qt][qb] * n_groups + blockIdx.x] = enc_f(run_max[qt][qb]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// K4r: the same filter with the operand roles of the memories swapped -- "row-stationary waves".
//
// k_mfma_scan above shares every ROW tile between its 8 waves through LDS and keeps the QUERIES in registers:
// that costs a workgroup barrier, an LDS staging write and a common epilogue phase per tile, and the 8 waves
// meet at every barrier -- both waves of a SIMD run their epilogue at the same time and the matrix pipe idles
// (round 1: MFMA busy 48 %, waves parked 46 % of their cycles).  Here the 128 QUERIES of a workgroup sit in LDS
// for the whole launch (one barrier, at the start) and every wave streams ITS OWN 32-row blocks straight from
// global memory into MFMA A-fragment registers.  Waves never wait for each other: no per-tile barrier, no staging
// write, no partial-sum exchange, and the two waves of a SIMD drift apart, so one wave's epilogue (VALU) runs under
// the other's MFMAs.
//
//   per wave and 32-row block: A = 2 row blocks of 16 x K, held in 2 * K/32 fragment registers (bf16x8 each),
//   loaded by one 16-byte global load per fragment; the block's two HALVES (64 queries = 4 query blocks each) reuse
//   them; in the second half every fragment is reloaded for the wave's NEXT block right after its last use, so a
//   load has a whole half (~1500-3000 cycles) to land: a rolling register window, no second buffer.
//   B = query fragments, ds_read_b128 from the padded LDS image (conflict-free like the row tiles above), read
//   B_AHEAD K-steps ahead of the MFMAs that consume them; each feeds 2 MFMAs (the two row blocks).
//   LDS read traffic per flop equals k_mfma_scan's (one 1 KB fragment per 2 MFMAs = half the LDS rate).
//   L2 -> CU traffic per flop doubles (128 instead of 256 queries per workgroup share a row): 1/128 B/flop,
//   ~11 TB/s chip-wide at 1.4 PFLOP/s, served by the XCD L2s (the query chunks of one row range are co-resident
//   on one XCD and walk the same blocks); HBM still sees every row once per launch.
// Candidates go to the wave's own LDS ring segment (ballot + mbcnt, as above) and are flushed by the wave itself.
// ---------------------------------------------------------------------------------------------
#ifndef RS_NWAVES_OVERRIDE
#define RS_NWAVES_OVERRIDE 8
#endif
constexpr int RS_NWAVES = RS_NWAVES_OVERRIDE;
// queries per workgroup: 128 (two halves of 4 query blocks) while a query row fits LDS 128 times (strides <= 512);
// 96 at stride 768 (147 of the 160 KB: all that fits), where a row block's K is walked in phases (rs_qpb(), k_mfma_rows).
// Every query block more is one more pair of MFMAs per row fragment a wave pulls in, and fewer query chunks re-reading
// the slab from L2.  Measured on the 1.25 M x 768 shard (1024 queries, Euclidean, the two pass-1 stages, same box):
//   64 queries, 2 phases of 12 K-steps (round 2's first shape) ... 1845 us
//   80 queries, 2 phases ........................................ 1710-1770 us
//   96 queries, 3 phases of 8 ................................... 1700 us
//   96 queries, 4 phases of 6 ................................... 1605-1630 us   <- shipped
//   96 queries, 6 phases of 4 ................................... 1577-1607 us
// (96 queries with 2 phases spill in the Euclidean epilogue: 96 fragment + 48 accumulator + 48 query-fragment registers.)
// The K-steps of row fragments a wave holds (= its loads in flight: 24 KB at 2 phases, 8 KB at 6) make no difference.
#ifndef RS_HQB768
#define RS_HQB768 6
#endif
#ifndef RS_PH768
#define RS_PH768 4
#endif
#ifndef RS_HQB1
#define RS_HQB1 4   // query blocks per half while the whole K stays in registers (strides <= 384 could hold 6: see the anatomy file)
#endif
constexpr int rs_qpb(uint32_t ldb) { return ldb <= 384 ? 32 * RS_HQB1 : (ldb <= 512 ? 128 : 16 * RS_HQB768); }
// ring entries per wave: 192, or 128 where the queries leave less LDS
constexpr int rs_seg(uint32_t ldb) { return (size_t)rs_qpb(ldb) * (ldb * 2 + 32) + 8 * 192 * 10 + 1024 <= 160 * 1024 ? 192 : 128; }


// Two shapes of the same kernel (round 4):
//   RBN = 2, NW = 8  two waves per SIMD, 32-row wave blocks (rounds 2-3; still the sampling pass, MODE 0);
//   RBN = 4, NW = 4  ONE wave per SIMD with the whole 512-register file (accumulators spill over into the AccVGPRs),
//                    64-row wave blocks, every query block of the workgroup in one sub-iteration (4 x 8 wave tile at
//                    strides <= 512, 4 x 6 at 768), K walked in phases of <= 6 K-steps with the row fragments refilled
//                    in place: each ds_read_b128 of a query fragment feeds 4 MFMAs instead of 2 (half the LDS bytes per
//                    flop), and nothing is duplicated between two waves that ran in lockstep anyway.
// STREAM: the row fragments as NONTEMPORAL loads.  With one query chunk (a pass of <= 128 queries, <= 96 at stride 768: what the
// coalescer hands over) the rows are a stream nobody reads twice: 1.43 -> 1.32 ms for a 16-query pass over 10 M x 384
// (6.05 -> 6.6 TB/s of bf16 rows).  With several chunks the other chunks of the XCD re-read a block from L2 and the same
// hint costs 15-25 % (256 queries: 1.82 -> 2.15 ms), so only single-chunk launches of pass 1 take it.
#define RS_LOAD_FRAG(ptr) (STREAM ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(ptr)) : *reinterpret_cast<const bf16x8*>(ptr))
template <int KSTEPS, int MODE, int METRIC, int RBN = 2, int NW = RS_NWAVES, bool STREAM = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, (NW == 4 ? 1 : 2))))
void k_mfma_rows(const __bf16* __restrict__ slab16,
                                                              const float* __restrict__ row_nrm,
                                                              const float* __restrict__ row_sqn,
                                                              const __bf16* __restrict__ q16, uint32_t nq,
                                                              uint32_t blk_begin, uint32_t blk_end, uint32_t n_rows,
                                                              int* __restrict__ gmax, uint32_t n_groups, uint32_t gpw,
                                                              const float* __restrict__ thr, Cand32* __restrict__ cand,
                                                              uint32_t* __restrict__ cnt, uint32_t cap)
{
    constexpr int LDB = KSTEPS * 16;
    constexpr int ROW_BYTES = LDB * 2;
    constexpr int LDS_ROW = ROW_BYTES + 32;   // same padding as k_mfma_scan's row tiles: conflict-free ds_read_b128
    constexpr int KS32 = KSTEPS / 2;          // K = 32 per MFMA
    constexpr int NT = NW * 64;
    constexpr int BR = 16 * RBN;              // rows per wave block
    static_assert((RBN == 2 && NW == 8) || (RBN == 4 && NW == 4), "two waves per SIMD on 32-row blocks, or one on 64-row blocks");
    static_assert(BR <= (int)MFMA_TILE_ROWS, "the slab is allocated in whole tiles of MFMA_TILE_ROWS rows");
    // A row block is worked off in NU sub-iterations that all reuse the KSP row fragments a wave holds:
    //   RBN = 2, dims <= 384: the whole K stays in registers (PH = 1) and the 128 queries come in QH = 2 halves of 4 query blocks;
    //   RBN = 2, dim 768:     K comes in PH phases of KSP K-steps (the fragment registers are refilled for the next phase while
    //                         one runs), the accumulators live through all of them, 16 RS_HQB768 queries (QH = 1);
    //   RBN = 4:              always phases (QH = 1: every query block of the workgroup at once), KSP = 6 K-steps where 6 divides
    //                         the row (strides 384, 768), else 4.
    constexpr int PH = (RBN == 4) ? ((KS32 % 6 == 0) ? KS32 / 6 : (KS32 > 4 ? KS32 / 4 : 1)) : ((KSTEPS > 32) ? RS_PH768 : 1);
    constexpr int QH = (RBN == 4) ? 1 : ((PH == 1) ? 2 : 1);
    constexpr int HQB = (RBN == 4) ? rs_qpb(LDB) / 16 : ((PH > 1) ? RS_HQB768 : (LDB <= 384 ? RS_HQB1 : 4));  // query blocks (of 16) per sub-iteration
    constexpr int QPB = QH * HQB * 16;        // queries per workgroup
    constexpr int QB = QPB / 16;
    constexpr int KSP = KS32 / PH;            // K-steps per phase = row fragments held per 16-row block
    constexpr int NU = QH * PH;               // sub-iterations per row block
    constexpr int NPOS = NU * KSP;            // (sub-iteration, K-step) positions per row block
    static_assert(KS32 % PH == 0 && QPB == rs_qpb(LDB), "K phases are whole steps; host and kernel agree on the chunk");
    // K-steps the query fragments are read ahead of their MFMAs.  The fragment buffers rotate with the position,
    // and the rotation must close over a row block (the loop over blocks re-enters at position 0): NB | NPOS.
    // (16 K-steps of row fragments leave registers for one step ahead only; with 5 or 6 query blocks one step is 10-12
    // MFMAs = 160-192 cycles, more than the LDS latency already)
#ifdef RS_BAHEAD
    constexpr int B_AHEAD = RS_BAHEAD;
#else
    constexpr int B_AHEAD = (KSP >= 16 || HQB >= 5) ? 1 : ((NPOS % 3 == 0) ? 2 : 3);
#endif
    constexpr int NB = B_AHEAD + 1;
    static_assert(NPOS % NB == 0, "the fragment-buffer rotation closes over one row block");
    static_assert(KSTEPS % 2 == 0, "whole 32-deep K steps");
    constexpr int RS_SEG = (NW == 4) ? 2 * rs_seg(LDB) : rs_seg(LDB);  // the ring's LDS is shared out over half as many waves
    static_assert(NPOS % 2 == 0, "two query-fragment buffers rotate over a row block");
    constexpr int RING = (MODE == 1) ? NW * RS_SEG : 1;

    __shared__ __attribute__((aligned(16))) unsigned char q_lds[QPB * LDS_ROW];
    __shared__ float ring_key[RING];
    __shared__ uint32_t ring_pos[RING];
    __shared__ unsigned short ring_q[RING];
    // MODE 0: a workgroup reports gpw groups (1, 2, 4 or 8: its waves in equal shares), so that the sampling pass can run
    // on the same full-chip grid as pass 1 whatever the number of query chunks and still hand k_thresholds 64..256 groups
    __shared__ int gmax_lds[MODE == 0 ? NW * QPB : 1];
    // Euclidean pass 1 with the whole K in registers (strides <= 512) is the one shape that did not fit 256 registers: the
    // 8 per-query-block thresholds of a lane went to scratch and came back through 131-238 scratch loads per row block --
    // on the SAME counter (vmcnt) as the row-fragment loads.  There they live in LDS instead: one ds_read_b32 per use.
#if defined(RS_THR_LDS_OFF)  // diagnostic build: the thresholds in registers everywhere, as before
    constexpr bool THR_LDS = false;
#elif defined(RS_THR_LDS_ALL)  // diagnostic build: in LDS for every metric and stride
    constexpr bool THR_LDS = (MODE == 1 && RBN == 2);
#else
    constexpr bool THR_LDS = (MODE == 1 && METRIC == EUCLIDEAN && PH == 1 && RBN == 2);
#endif
    __shared__ float thr_lds[THR_LDS ? QPB : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: block numbers, row bases
                                                                // and the partial-block branch live in scalar registers
    const int c16 = lane & 15, kg = lane >> 4;
    // ---- which (row-block lane x, query chunk y) this workgroup is: XCD-aware (xcd_map.hpp) ----
    // All the chunks of one x stream the SAME row blocks, so they belong on ONE XCD (its L2 then fetches a block from HBM
    // once and serves the other chunks); with id = x + gridDim.x y that only happens when 8 divides gridDim.x (11 chunks
    // x 23: every XCD fetched every block, 2.7 instead of 1.5 ms).
#ifndef RS_NO_XCD_REMAP
    uint32_t bx, by;
    xcd_pair(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, bx, by);
#else
    const uint32_t bx = blockIdx.x, by = blockIdx.y;
#endif
    const uint32_t chunk_base = by * QPB;

    // ---- the workgroup's queries -> LDS (the only barrier of the kernel besides MODE 0's final combine) ----
    {
        constexpr int CPR = ROW_BYTES / 16;
        constexpr int PIECES = QPB * CPR;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(q16) + (size_t)chunk_base * ROW_BYTES;
        for (int c = tid; c < PIECES; c += NT) {
            const int r = c / CPR, cc = c % CPR;
            *reinterpret_cast<u32x4*>(&q_lds[r * LDS_ROW + cc * 16]) = *reinterpret_cast<const u32x4*>(src + (size_t)c * 16);
        }
        if (MODE == 0)
            for (int c = tid; c < NW * QPB; c += NT) gmax_lds[c] = enc_f(-INFINITY);
        if (THR_LDS)
            for (int c = tid; c < QPB; c += NT) thr_lds[c] = (chunk_base + (uint32_t)c < nq) ? thr[chunk_base + (uint32_t)c] : INFINITY;
    }
    __syncthreads();

    float thr_q[THR_LDS ? 1 : QB], run_max[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const uint32_t qq = chunk_base + qb * 16 + c16;
        if (!THR_LDS) {
            thr_q[qb] = INFINITY;  // padding queries never pass
            if (MODE == 1 && qq < nq) thr_q[qb] = thr[qq];
#ifdef RS_DBG_NOCAND
            thr_q[qb] = INFINITY;  // diagnostic: nothing ever passes
#endif
        }
        run_max[qb] = -INFINITY;
    }
    // this lane's threshold for query block qb: a register, or (THR_LDS) one ds_read_b32 per use
    auto thr_of = [&](int qb) -> float { return THR_LDS ? thr_lds[qb * 16 + c16] : thr_q[THR_LDS ? 0 : qb]; };
    uint32_t my_cnt = 0;

    // block schedule: neighbouring WORKGROUPS stream neighbouring BR-row blocks (block = begin + x + gridDim.x * (wave + NW i)),
    // so in MODE 0 every workgroup (= group) owns rows as soon as there are gridDim.x blocks
#ifdef RS_DBG_PAIRLOAD  // diagnostic: waves w and w + 4 (SIMD partners) stream the SAME blocks -- do their loads share L1?
    const uint32_t stride = gridDim.x * (NW / 2);
    uint32_t b = blk_begin + bx + gridDim.x * (uint32_t)(wave & 3);
#else
    const uint32_t stride = gridDim.x * NW;
    uint32_t b = blk_begin + bx + gridDim.x * (uint32_t)wave;
#endif
    const bool has_work = b < blk_end;
    const uint32_t b_last = has_work ? b + ((blk_end - 1 - b) / stride) * stride : blk_begin;  // a valid block to re-read at the tail

    // The slab this kernel reads is FRAGMENT-MAJOR (k_rows_bf16_frag): per 16-row group and K-step one 1 KB chunk in
    // lane order, so fragment (rb, s) of block blk is ONE contiguous 1 KB wave load (8 whole 128-byte lines) at
    // blk * 32 rows + rb * 16 rows + s KB + 16 lane.  (With row-major rows every fragment load touched 16 half
    // lines; at 128 queries per workgroup the L2 request rate, not the matrix pipe, then set the pace.)
    const uint32_t lane_off = (uint32_t)lane * 16u;
#ifdef RS_DBG_HOTLOAD  // diagnostic: every load hits L2 (the waves re-read the same RS_DBG_HOTLOAD blocks): what do the MISSES cost?
    auto a_ptr = [&](uint32_t blk) { return reinterpret_cast<const unsigned char*>(slab16) + (size_t)(blk % (uint32_t)(RS_DBG_HOTLOAD)) * (BR * ROW_BYTES) + lane_off; };
#else
    auto a_ptr = [&](uint32_t blk) { return reinterpret_cast<const unsigned char*>(slab16) + (size_t)blk * (BR * ROW_BYTES) + lane_off; };
#endif
    bf16x8 afrag[RBN][KSP];
    f32x4 aux[RBN], aux2[RBN];
#pragma unroll
    for (int rb = 0; rb < RBN; ++rb) {
        aux[rb] = f32x4{1.f, 1.f, 1.f, 1.f};
        aux2[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_aux = [&](uint32_t blk) {
#pragma unroll
        for (int rb = 0; rb < RBN; ++rb) {
            const uint32_t r0 = blk * BR + 16 * rb + 4 * kg;
            if (METRIC != COSINE) aux[rb] = *reinterpret_cast<const f32x4*>(row_nrm + r0);
            if (METRIC == EUCLIDEAN) aux2[rb] = *reinterpret_cast<const f32x4*>(row_sqn + r0);
        }
    };
    // per-row scalars: PH = 1 loads a block's with the last fragments of the block before (they are needed after the
    // first half already) and keeps a copy while the next ones arrive; with K phases they are fetched when the block's
    // LAST phase starts -- one set of registers instead of two, half the registers, next to 6 query blocks of accumulators
    constexpr bool AUX_JIT = PH > 1;
    if (has_work) {
        const unsigned char* p = a_ptr(b);
#pragma unroll
        for (int s = 0; s < KSP; ++s)
#pragma unroll
            for (int rb = 0; rb < RBN; ++rb) afrag[rb][s] = RS_LOAD_FRAG(p + rb * (16 * ROW_BYTES) + s * 1024);
        if (!AUX_JIT) load_aux(b);
    }

    // query fragments: position i = (sub-iteration u, step s) reads 4 query blocks at one K-step --
    // PH = 1: the blocks of half u at K-step s; PH = 2: the (only) 4 blocks at K-step u KSP + s
    const unsigned char* qrow = &q_lds[c16 * LDS_ROW + kg * 16];
    bf16x8 bq[NB][HQB];
    auto read_b = [&](int i, bf16x8(&dst)[HQB]) {
        const int u = (i / KSP) % NU, s = i % KSP;
        const int qb0 = (QH == 2) ? u * HQB : 0, kstep = (PH > 1) ? u * KSP + s : s;
#pragma unroll
        for (int j = 0; j < HQB; ++j)
            dst[j] = *reinterpret_cast<const bf16x8*>(qrow + (qb0 + j) * 16 * LDS_ROW + kstep * 64);
    };
#pragma unroll
    for (int i = 0; i < B_AHEAD; ++i) read_b(i, bq[i % NB]);

    auto flush_wave = [&]() {  // this wave's ring segment -> the per-query global buffers (the kernel's only atomics)
        const uint32_t n_e = my_cnt < (uint32_t)RS_SEG ? my_cnt : (uint32_t)RS_SEG;
        for (uint32_t e = lane; e < n_e; e += 64) {
            const uint32_t idx = (uint32_t)wave * RS_SEG + e;
            const uint32_t qq = chunk_base + ring_q[idx];
            const uint32_t slot = atomicAdd(&cnt[qq], 1u);
            if (slot < cap) {
                Cand32 c;
                c.key = ring_key[idx];
                c.pos = ring_pos[idx];
                cand[(size_t)qq * cap + slot] = c;
            }
        }
        if (my_cnt > (uint32_t)RS_SEG) {  // the segment overflowed: candidates of any of the workgroup's queries may be lost -> host redoes them
            for (uint32_t j = lane; j < (uint32_t)QPB; j += 64)
                if (chunk_base + j < nq) atomicAdd(&cnt[chunk_base + j], cap + 1u);
        }
        my_cnt = 0;
    };

#ifdef RS_STAGGER  // diagnostic: the second-dispatched half of the workgroup starts late (s_sleep counts 64 cycles)
    if (wave >= NW / 2) __builtin_amdgcn_s_sleep(RS_STAGGER);
#endif
#ifdef RS_PRIO
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    for (; b < blk_end; b += stride) {
        const uint32_t nb_raw = b + stride;
        const uint32_t nb = nb_raw < blk_end ? nb_raw : b_last;  // the tail re-reads a valid block; its values are never used
        const unsigned char* pn = a_ptr(nb);
        const unsigned char* pc = a_ptr(b);
        const uint32_t row0 = b * BR;
        const bool partial = row0 + BR > n_rows;  // wave-uniform
#if defined(RS_DBG_NOLDS) || defined(RS_DBG_NOLOAD)  // diagnostic: the fragments are opaque per block (no hoisting of the MFMAs)
#pragma unroll
        for (int i2 = 0; i2 < NB; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < HQB; ++j2) asm volatile("" : "+v"(bq[i2][j2]));
#pragma unroll
        for (int s2 = 0; s2 < KSP; ++s2)
#pragma unroll
            for (int rb2 = 0; rb2 < RBN; ++rb2) asm volatile("" : "+v"(afrag[rb2][s2]));
#endif
        f32x4 aux_cp[RBN], aux2_cp[RBN];           // PH = 1: this block's per-row scalars (the registers are reloaded below)
        if (!AUX_JIT) {
#pragma unroll
            for (int rb = 0; rb < RBN; ++rb) {
                aux_cp[rb] = (METRIC != COSINE) ? aux[rb] : f32x4{1.f, 1.f, 1.f, 1.f};
                aux2_cp[rb] = (METRIC == EUCLIDEAN) ? aux2[rb] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        auto& aux_c = AUX_JIT ? aux : aux_cp;
        auto& aux2_c = AUX_JIT ? aux2 : aux2_cp;
        f32x4 acc[RBN][HQB];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int half = (QH == 2) ? u : 0;  // which 4 query blocks this sub-iteration serves
            if (PH == 1 || u == 0) {
#pragma unroll
                for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
                    for (int j = 0; j < HQB; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // last use of the fragments a wave holds: PH = 1 in the last half (refill for the next block); PH = 2 in
            // BOTH phases (phase 0 refills with this block's second K half, phase 1 with the next block's first)
            constexpr bool ALWAYS_RELOAD = PH > 1;
            const bool reload = ALWAYS_RELOAD || u == NU - 1;
            const unsigned char* psrc = (PH > 1 && u < PH - 1) ? pc + (size_t)(u + 1) * KSP * 1024 : pn;
#pragma unroll
            for (int s = 0; s < KSP; ++s) {
                const int i = u * KSP + s;
#ifndef RS_DBG_NOLDS
                read_b((i + B_AHEAD) % NPOS, bq[(i + B_AHEAD) % NB]);
#endif
#ifndef RS_INTERLEAVE_OFF
                // An in-order wave pays the full issue time of every instruction that sits BETWEEN two MFMAs once the pipe
                // has drained (measured: ~16 cycles per ds_read_b128, ~50 per global load, and the partner wave of the
                // SIMD runs in lockstep, so nobody fills the hole).  One memory instruction right behind each MFMA hides
                // in that MFMA's 16 cycles instead: MFMA, ds_read, MFMA, ds_read, ... MFMA, global_load, ...
#else
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int j = 0; j < HQB; ++j)
#pragma unroll
                    for (int rb = 0; rb < RBN; ++rb)
                        acc[rb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[rb][s], bq[i % NB][j], acc[rb][j], 0, 0, 0);
#ifndef RS_DBG_NOLOAD
                if (reload) {  // last use of this K-step's row fragments: refill the registers
#pragma unroll
                    for (int rb = 0; rb < RBN; ++rb)
                        afrag[rb][s] = RS_LOAD_FRAG(psrc + rb * (16 * ROW_BYTES) + s * 1024);
                    if (!AUX_JIT && u == NU - 1 && s == KSP - 1) load_aux(nb);
                }
                if (AUX_JIT && u == NU - 1 && s == 0) load_aux(b);
#endif
#ifndef RS_INTERLEAVE_OFF
#pragma unroll
                for (int g = 0; g < HQB; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
                if (reload) {
#pragma unroll
                    for (int g = 0; g < RBN; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, RBN * HQB - HQB - RBN, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, RBN * HQB - HQB, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef RS_DBG_NOEPI  // diagnostic build: keep the accumulators alive with one add per half, nothing else
            {
                float keep = 0.f;
#pragma unroll
                for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
                    for (int j = 0; j < HQB; ++j) keep += acc[rb][j][0] + acc[rb][j][1] + acc[rb][j][2] + acc[rb][j][3];
                run_max[u] += keep;
            }
#endif
            // ---- epilogue of this half: C layout of a 16x16 block: column = lane & 15 (query), row = 4 (lane >> 4) + reg ----
            // Two copies selected by ONE wave-uniform branch: only the index's last block can hold rows past n_rows, and
            // masking them costs a compare and two selects per key -- in every block, if it is written as a predicate.
            auto epilogue = [&](auto partial_tag) {
                constexpr bool PARTIAL = decltype(partial_tag)::value;
                // key of row (rb, jj) for query block j, from the finished sums.  The common path only needs each query
                // block's maximum; the rare candidate path recomputes the keys it looks at (the same arithmetic on the same
                // registers: the accumulators stay live until the next block zeroes them), so no key array stays live across
                // the branch -- with the 4 x 8 wave tile that array alone would be 128 registers.
                auto key_of = [&](int rb, int j, int jj) -> float {
                    float key = acc[rb][j][jj];                                                   // cosine: x^.q
                    if (METRIC == DOT) key *= aux_c[rb][jj];                                      // x.q
                    if (METRIC == EUCLIDEAN) key = 2.0f * key * aux_c[rb][jj] - aux2_c[rb][jj];   // |q|^2 - |x - q|^2
                    if (PARTIAL && row0 + (uint32_t)(16 * rb + 4 * kg + jj) >= n_rows) key = -INFINITY;
                    return key;
                };
                float mj[HQB];
#pragma unroll
                for (int j = 0; j < HQB; ++j) {
                    // max of the 4 RBN keys, three at a time (fmaxf() costs a canonicalising v_max x, x per operand in IEEE
                    // mode; the keys are MFMA sums of finite bf16 products)
                    float m = max3f(key_of(0, j, 0), key_of(0, j, 1), key_of(0, j, 2));
                    m = max3f(m, key_of(0, j, 3), key_of(1, j, 0));
                    m = max3f(m, key_of(1, j, 1), key_of(1, j, 2));
                    if (RBN == 2) {
                        m = max3f(m, key_of(1, j, 3), key_of(1, j, 3));
                    } else {
                        m = max3f(m, key_of(1, j, 3), key_of(2, j, 0));
                        m = max3f(m, key_of(2, j, 1), key_of(2, j, 2));
                        m = max3f(m, key_of(2, j, 3), key_of(RBN - 1, j, 0));
                        m = max3f(m, key_of(RBN - 1, j, 1), key_of(RBN - 1, j, 2));
                        m = max3f(m, key_of(RBN - 1, j, 3), key_of(RBN - 1, j, 3));
                    }
                    mj[j] = m;
                    // one query block at a time: left alone the scheduler computes every key of the sub-iteration first
                    // (16 RBN x HQB live values -- the 4 x 8 Euclidean tile then needs more than the 512 registers there are)
                    if (RBN == 4 && METRIC != COSINE) __builtin_amdgcn_sched_barrier(0);
                }
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < HQB; ++j) run_max[half * HQB + j] = max3f(run_max[half * HQB + j], mj[j], mj[j]);
                    return;
                }
                // ONE test and ONE branch for the HQB query blocks of the sub-iteration: does any key reach its query's
                // threshold?  (m - T >= 0; T = +inf for padding queries gives -inf, a fully masked block gives -inf or NaN)
                float ex[HQB];
#pragma unroll
                for (int j = 0; j < HQB; ++j) ex[j] = mj[j] - thr_of(half * HQB + j);
                float any = ex[0];
#pragma unroll
                for (int j = 1; j + 1 < HQB; j += 2) any = max3f(any, ex[j], ex[j + 1]);
                if (HQB % 2 == 0) any = max3f(any, ex[HQB - 1], ex[HQB - 1]);
                if (__builtin_amdgcn_ballot_w64(any >= 0.0f) == 0ull) return;
                // rare; only THIS wave pays for it
#pragma unroll
                for (int j = 0; j < HQB; ++j) {
                    const int qb = half * HQB + j;
                    const float tq = thr_of(qb);
                    if (__builtin_amdgcn_ballot_w64(mj[j] >= tq) == 0ull) continue;  // wave-uniform
#pragma unroll
                    for (int rb = 0; rb < RBN; ++rb) {
                        const float k0 = key_of(rb, j, 0), k1 = key_of(rb, j, 1), k2 = key_of(rb, j, 2), k3 = key_of(rb, j, 3);
                        const float mrb = fmaxf(fmaxf(k0, k1), fmaxf(k2, k3));
                        if (__builtin_amdgcn_ballot_w64(mrb >= tq) == 0ull) continue;  // wave-uniform
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const float key = jj == 0 ? k0 : (jj == 1 ? k1 : (jj == 2 ? k2 : k3));
                            const bool is_cand = key >= tq && key > -INFINITY;  // masked rows are -inf; T_q may be too
                            const unsigned long long mk = __builtin_amdgcn_ballot_w64(is_cand);
                            if (mk != 0ull) {  // wave-uniform
                                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32),
                                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                                const uint32_t slot = my_cnt + rank;
                                if (is_cand && slot < (uint32_t)RS_SEG) {
                                    const uint32_t e = (uint32_t)wave * RS_SEG + slot;
                                    ring_key[e] = key;
                                    ring_pos[e] = row0 + (uint32_t)(16 * rb + 4 * kg + jj);
                                    ring_q[e] = (unsigned short)(qb * 16 + c16);
                                }
                                my_cnt += (uint32_t)__popcll(mk);
                            }
                        }
                    }
                }
            };
#ifndef RS_DBG_NOEPI
            if (PH == 1 || u == NU - 1) {  // the sums are complete
                if (partial)
                    epilogue(std::true_type{});
                else
                    epilogue(std::false_type{});
            }
#else
            (void)epilogue;
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 && my_cnt >= (uint32_t)(RS_SEG / 2)) flush_wave();  // wave-uniform
    }
#ifdef RS_DBG_NOEPI
    if (run_max[0] + run_max[1] == 12345.678f) my_cnt = 1;  // diagnostic build: the sums above stay live
#endif
    if (MODE == 1) {
        if (my_cnt) flush_wave();
    } else {
        // group maximum = max over every wave of the workgroup (its rows are held by no other group)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = run_max[qb];  // the four k-groups of lanes saw different rows of one query
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const uint32_t gl = (uint32_t)wave * gpw / NW;  // this wave's group within the workgroup
            if (kg == 0) atomicMax(&gmax_lds[gl * QPB + qb * 16 + c16], enc_f(mx));
        }
        __syncthreads();
        for (uint32_t c = tid; c < gpw * (uint32_t)QPB; c += NT) {
            const uint32_t gl = c / QPB, ql = c % QPB, gidx = bx * gpw + gl;
            if (chunk_base + ql < nq && gidx < n_groups) gmax[(size_t)(chunk_base + ql) * n_groups + gidx] = gmax_lds[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-query scan of the bf16 slab (opt-in candidate filter: half the HBM bytes of the f32 scan).
// Same structure as k_scan (kernels.hip): G lanes share a row, 16-byte non-temporal loads straight to
// registers (8 bf16 each), shuffle reduction, one sorted top-64 list per wave.  The query stays f32
// (only the rows carry bf16 rounding), bf16 -> f32 is a shift, the products accumulate with fmaf.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8_bf16(float a, const u32x4 x, const f32x4 q0, const f32x4 q1)
{
    a = fmaf(__uint_as_float(x.x << 16), q0.x, a);
    a = fmaf(__uint_as_float(x.x & 0xFFFF0000u), q0.y, a);
    a = fmaf(__uint_as_float(x.y << 16), q0.z, a);
    a = fmaf(__uint_as_float(x.y & 0xFFFF0000u), q0.w, a);
    a = fmaf(__uint_as_float(x.z << 16), q1.x, a);
    a = fmaf(__uint_as_float(x.z & 0xFFFF0000u), q1.y, a);
    a = fmaf(__uint_as_float(x.w << 16), q1.z, a);
    a = fmaf(__uint_as_float(x.w & 0xFFFF0000u), q1.w, a);
    return a;
}

template <int METRIC, int G, int VPL>
__global__ __launch_bounds__(256) void k_scan_bf16(const u32x4* __restrict__ slab16, const float* __restrict__ row_nrm,
                                                   const float* __restrict__ row_sqn,
                                                   const double* __restrict__ q64, uint32_t dim, uint32_t n,
                                                   Cand32* __restrict__ out)
{
    constexpr int RPS = WAVE / G;
    constexpr uint32_t LD8 = G * VPL;  // 16-byte chunks (8 bf16) per row
    __shared__ Cand32 sh[4 * WAVE];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int g = lane / G, c = lane % G;

    f32x4 qv[VPL][2];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const uint32_t e0 = 8u * (uint32_t)(c + G * j);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 v;
            // clamped, never predicated (see load_q4 in kernels.hip)
            const uint32_t b0 = e0 + 4 * h, last = dim - 1;
            const double d0 = q64[b0 + 0 < dim ? b0 + 0 : last], d1 = q64[b0 + 1 < dim ? b0 + 1 : last];
            const double d2 = q64[b0 + 2 < dim ? b0 + 2 : last], d3 = q64[b0 + 3 < dim ? b0 + 3 : last];
            v.x = b0 + 0 < dim ? (float)d0 : 0.0f;
            v.y = b0 + 1 < dim ? (float)d1 : 0.0f;
            v.z = b0 + 2 < dim ? (float)d2 : 0.0f;
            v.w = b0 + 3 < dim ? (float)d3 : 0.0f;
            qv[j][h] = v;
        }
    }
    const uint32_t n_steps = (n + RPS - 1) / RPS;
    const uint32_t n_waves = gridDim.x * 4;
    TopList<float> L;
    L.init();
    for (uint32_t s = blockIdx.x * 4 + wave; s < n_steps; s += n_waves) {
        const uint32_t row = s * RPS + g;
        const bool valid = row < n;
        const uint32_t r = valid ? row : n - 1;
        const u32x4* p = slab16 + (size_t)r * LD8 + c;
        u32x4 x[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) x[j] = __builtin_nontemporal_load(p + G * j);
        float nr = 1.0f, sq = 0.0f;
        if (METRIC != COSINE) nr = row_nrm[r];
        if (METRIC == EUCLIDEAN) sq = row_sqn[r];
        __builtin_amdgcn_sched_barrier(0);
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) a = dot8_bf16(a, x[j], qv[j][0], qv[j][1]);
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) a += __shfl_xor(a, o);
        float key = a;                                            // cosine: x^.q
        if (METRIC == DOT) key = a * nr;                          // x.q
        if (METRIC == EUCLIDEAN) key = 2.0f * a * nr - sq;        // |q|^2 - |x - q|^2
        L.offer(key, row, valid && c == 0);
    }
    block_merge<float, Cand32, 4>(L, sh);
    if (wave == 0) {
        Cand32 e;
        e.key = L.key;
        e.pos = L.pos;
        out[(size_t)blockIdx.x * KP + lane] = e;
    }
}

// T_q = the 64th largest group maximum (a lower bound of the query's 64th best key); -inf when fewer
// than 64 groups exist.  One wave per query.
// A query of norm 0 (a zero query, or one the host zeroed because it is outside the fast-path domain) scores 0 on every
// row: every row would be a candidate and its wave's ring segment would overflow, taking the workgroup's other 255
// queries down with it.  It gets T_q = +inf -- no candidates at all; the host answers it on the exact path anyway.
__global__ __launch_bounds__(256) void k_thresholds(const int* __restrict__ gmax, uint32_t n_groups, uint32_t nq,
                                                    const double* __restrict__ q_norms, float* __restrict__ thr)
{
    const int lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    constexpr int VALS = MFMA_GROUPS / 64;  // all loads first, then one selection: no load -> insert -> load chain
    float v[VALS];
#pragma unroll
    for (int i = 0; i < VALS; ++i) {
        const uint32_t g = (uint32_t)(lane + 64 * i);
        v[i] = g < n_groups ? dec_f(gmax[(size_t)q * n_groups + g]) : -INFINITY;
    }
    const float t = kth64_of_wave<VALS>(v);  // -inf when fewer than 64 groups hold rows
    if (lane == 0) thr[q] = (q_norms[q] == 0.0) ? INFINITY : t;
}

// Composite of a candidate: order-preserving key bits, then ~position -- unsigned order = (key desc, position asc) rank.
__device__ __forceinline__ unsigned long long cand_composite(const Cand32 e)
{
    return ((unsigned long long)((uint32_t)enc_f(e.key) ^ 0x80000000u) << 32) | (unsigned long long)(0xFFFFFFFFu - e.pos);
}

// Top-64 of the 64 * VALS composites a wave holds (0 = padding, below every real composite: a real one has ~pos >= 1 in
// its low word): a bit-by-bit search finds the 64th largest, the <= 64 entries at or above it are compacted through
// `row` (64 LDS slots of this wave) and ranked by counting -- no insertion loop.  Writes the sorted list, sentinels behind.
template <int VALS>
__device__ __forceinline__ void top64_of_composites(const unsigned long long (&c)[VALS], unsigned long long* row, int lane,
                                                    Cand32* __restrict__ out)
{
    unsigned long long t = 0ull;  // the 64th largest composite (0 when there are fewer than 64)
    for (int b = 63; b >= 0; --b) {
        const unsigned long long tc = t | (1ull << b);
        uint32_t n_ge = 0;
#pragma unroll
        for (int i = 0; i < VALS; ++i) n_ge += (uint32_t)__popcll(__ballot(c[i] >= tc));
        if (n_ge >= 64u) t = tc;  // wave-uniform
    }
    __builtin_amdgcn_wave_barrier();  // `row` may still be read as the wave's candidate buffer by a slower lane
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < VALS; ++i) {
        const bool sel = c[i] != 0ull && c[i] >= t;
        const unsigned long long mk = __ballot(sel);
        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (sel && slot < 64u) row[slot] = c[i];
        base += (uint32_t)__popcll(mk);
    }
    const uint32_t kept = base < 64u ? base : 64u;  // composites are distinct (positions are), so base <= 64
    __builtin_amdgcn_wave_barrier();
    const unsigned long long me = (uint32_t)lane < kept ? row[lane] : 0ull;
    uint32_t rank = 0;
    for (int jx = 0; jx < 64; ++jx) rank += read_lane(me, jx) > me ? 1u : 0u;
    Cand32 res;
    res.key = -INFINITY;
    res.pos = POS_SENTINEL;
    if ((uint32_t)lane < kept) {
        res.key = dec_f((int)((uint32_t)(me >> 32) ^ 0x80000000u));
        res.pos = 0xFFFFFFFFu - (uint32_t)me;
        out[rank] = res;
    } else {
        out[lane] = res;  // sentinels behind the real entries
    }
}

// The 64th largest 32-bit key (high word of a composite) among the 64 * VALS composites a wave holds; 0 when fewer than 64
// are real.  32 rounds of VALS one-instruction compares -- half the rounds and a third of the work of the 64-bit search.
template <int VALS>
__device__ __forceinline__ uint32_t kth64_key_of_composites(const unsigned long long (&c)[VALS])
{
    uint32_t hi[VALS];  // a real entry's key word is never 0 (that would be the encoding of a NaN): padding (0) is never counted
#pragma unroll
    for (int i = 0; i < VALS; ++i) hi[i] = (uint32_t)(c[i] >> 32);
    uint32_t t = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t tc = t | (1u << b);
        uint32_t n_ge = 0;
#pragma unroll
        for (int i = 0; i < VALS; ++i) n_ge += (uint32_t)__popcll(__ballot(hi[i] >= tc));
        if (n_ge >= 64u) t = tc;  // wave-uniform
    }
    return t;
}

// Per query: top-64 of its candidate buffer by (key desc, position asc) -> one sorted list.  One wave per query.
// The buffer holds every row that beat the threshold of ITS stage; what can still be in the top 64 at the end is what
// beats the FINAL threshold thr[q] (a valid lower bound of the 64th best key: at least 64 rows reach it and all rows have
// been through pass 1), i.e. the 64 that set it plus the last stage's finds -- a few hundred of the one to four thousand
// in the buffer.  So: (A) stream the buffer once (8 loads per lane in flight) and compact the composites at or above
// thr[q] into LDS (<= 1024); (B) find the 64th largest KEY among them (32-bit search) and compact again what reaches it:
// the top 64 plus whatever ties the 64th key, normally under 128; (C) rank those by the full composite (2 values per lane).
// Fallbacks: no usable filter -> every candidate goes through (B) (n <= 1024); more than 128 tie the 64th key -> the
// full-composite search over the first set; a buffer that fits neither -> the sorted-list walk.
constexpr int SEL_KEEP = 1024;
constexpr int SEL_TIES = 128;
__global__ __launch_bounds__(256) void k_select_candidates(const Cand32* __restrict__ cand,
                                                           const uint32_t* __restrict__ cnt, uint32_t cap, uint32_t nq,
                                                           const float* __restrict__ thr, Cand32* __restrict__ lists)
{
    __shared__ unsigned long long sh[4][SEL_KEEP];
    __shared__ unsigned long long sh2[4][SEL_TIES];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const uint32_t n = cnt[q];
    const Cand32* mine = cand + (size_t)q * cap;
    Cand32* out = lists + (size_t)q * KP;
    Cand32 res;
    res.key = -INFINITY;
    res.pos = POS_SENTINEL;
    if (n > cap) {  // an overflowed buffer yields an all-sentinel list: the host redoes that query
        out[lane] = res;
        return;
    }
    // (A) composites at or above t -> sh[wave]; m = how many there are (stored: the first SEL_KEEP)
    auto gather = [&](float t) {
        uint32_t m = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 512u) {
            Cand32 e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t jx = i0 + (uint32_t)(lane + 64 * u);
                e[u] = mine[jx < n ? jx : (n ? n - 1u : 0u)];  // clamped, never predicated: the loads go out together
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t jx = i0 + (uint32_t)(lane + 64 * u);
                const bool sel = jx < n && e[u].key >= t;
                const unsigned long long mk = __ballot(sel);
                const uint32_t slot = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                if (sel && slot < (uint32_t)SEL_KEEP) sh[wave][slot] = cand_composite(e[u]);
                m += (uint32_t)__popcll(mk);
            }
        }
        __builtin_amdgcn_wave_barrier();
        return m;
    };
    uint32_t m = 0;
    bool have = false;
    if (n == 0) {
        out[lane] = res;
        return;
    }
    if (thr != nullptr && n > 128u) {
        m = gather(thr[q]);
        have = m >= 64u && m <= (uint32_t)SEL_KEEP;  // wave-uniform
    }
    if (!have && n <= (uint32_t)SEL_KEEP) {
        m = gather(-INFINITY);  // keys are finite: everything
        have = true;
    }
    if (have) {
#define VL_SEL_FROM_LDS(VALS)                                                                                              \
    {                                                                                                                      \
        unsigned long long c[VALS];                                                                                        \
        _Pragma("unroll") for (int i = 0; i < VALS; ++i)                                                                   \
        {                                                                                                                  \
            const uint32_t jx = (uint32_t)(lane + 64 * i);                                                                 \
            c[i] = jx < m ? sh[wave][jx] : 0ull;                                                                           \
        }                                                                                                                  \
        const uint32_t t64 = kth64_key_of_composites<VALS>(c); /* 0: fewer than 64 entries, all of them stay */            \
        uint32_t m2 = 0;                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < VALS; ++i)                                                                   \
        {                                                                                                                  \
            const bool sel = c[i] != 0ull && (uint32_t)(c[i] >> 32) >= t64;                                                \
            const unsigned long long mk = __ballot(sel);                                                                   \
            const uint32_t slot = m2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u)); \
            if (sel && slot < (uint32_t)SEL_TIES) sh2[wave][slot] = c[i];                                                  \
            m2 += (uint32_t)__popcll(mk);                                                                                  \
        }                                                                                                                  \
        __builtin_amdgcn_wave_barrier();                                                                                   \
        if (m2 <= (uint32_t)SEL_TIES) { /* wave-uniform */                                                                 \
            unsigned long long d[2];                                                                                       \
            d[0] = (uint32_t)lane < m2 ? sh2[wave][lane] : 0ull;                                                           \
            d[1] = (uint32_t)lane + 64u < m2 ? sh2[wave][lane + 64] : 0ull;                                                \
            top64_of_composites<2>(d, sh[wave], lane, out);                                                                \
        } else {                                                                                                           \
            top64_of_composites<VALS>(c, sh[wave], lane, out);                                                             \
        }                                                                                                                  \
    }
        if (m <= 128u) VL_SEL_FROM_LDS(2)
        else if (m <= 256u) VL_SEL_FROM_LDS(4)
        else if (m <= 512u) VL_SEL_FROM_LDS(8)
        else VL_SEL_FROM_LDS(16)
#undef VL_SEL_FROM_LDS
        return;
    }
    TopList<float> L;
    L.init();
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t jx = i0 + lane;
        Cand32 e;
        e.key = 0.f;
        e.pos = 0;
        if (jx < n) e = mine[jx];
        L.offer(e.key, e.pos, jx < n);
    }
    res.key = L.key;
    res.pos = L.pos;
    out[lane] = res;
}

// Between two stages of pass 1: T_q = max(T_q, 64th largest key among the candidates found so far).
// Those are 64 distinct rows, so the value is still a lower bound of the query's 64th best key, and it
// is much tighter than the sampling bound: the later stages (most of the rows) take the candidate branch
// of k_mfma_scan's epilogue a few times less often.  One wave per query.
// A buffer of more than 512 (the second and third refinement of a 10 M-row pass: one to four thousand) is streamed once,
// 8 loads per lane in flight, and only the keys at or above the CURRENT T_q are kept (compacted into LDS): keys below it
// cannot raise it -- if fewer than 64 candidates reach T_q the 64th largest is below T_q and T_q stays -- and those at
// or above it are the 64 that set it plus the finds since, a few hundred.  (The sorted-list walk this replaces took
// 27-100 us per launch: one dependent load per 64 candidates.)
constexpr int REF_KEEP = 1024;
__global__ __launch_bounds__(256) void k_refine_thresholds(const Cand32* __restrict__ cand, const uint32_t* __restrict__ cnt,
                                                          uint32_t cap, uint32_t nq, float* __restrict__ thr)
{
    __shared__ float keep[4][REF_KEEP];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * 4 + wave;
    if (q >= nq) return;
    const uint32_t n = cnt[q];
    if (n < 64u || n > cap) return;  // too few to say anything / overflowed (the host redoes that query)
    const Cand32* mine = cand + (size_t)q * cap;
    if (n <= 512u) {  // the usual case after the first stage: every candidate key in registers, one selection
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t j = (uint32_t)(lane + 64 * i);
            v[i] = j < n ? mine[j].key : -INFINITY;
        }
        const float t64 = kth64_of_wave<8>(v);
        if (lane == 0 && t64 > thr[q]) thr[q] = t64;
        return;
    }
    const float t_old = thr[q];
    uint32_t m = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 512u) {
        float kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t jx = i0 + (uint32_t)(lane + 64 * u);
            kv[u] = mine[jx < n ? jx : n - 1u].key;  // clamped, never predicated: the loads go out together
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t jx = i0 + (uint32_t)(lane + 64 * u);
            const bool sel = jx < n && kv[u] >= t_old;
            const unsigned long long mk = __ballot(sel);
            const uint32_t slot = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (sel && slot < (uint32_t)REF_KEEP) keep[wave][slot] = kv[u];
            m += (uint32_t)__popcll(mk);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (m < 64u) return;  // the 64th largest candidate is below T_q: T_q stays (wave-uniform)
    if (m <= (uint32_t)REF_KEEP) {
        float t64;
        if (m <= 512u) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t j = (uint32_t)(lane + 64 * i);
                v[i] = j < m ? keep[wave][j] : -INFINITY;
            }
            t64 = kth64_of_wave<8>(v);
        } else {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t j = (uint32_t)(lane + 64 * i);
                v[i] = j < m ? keep[wave][j] : -INFINITY;
            }
            t64 = kth64_of_wave<16>(v);
        }
        if (lane == 0 && t64 > t_old) thr[q] = t64;
        return;
    }
    TopList<float> L;  // more than REF_KEEP candidates at or above T_q (a first refinement after a loose sampling bound)
    L.init();
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool ok = i < n;
        Cand32 e;
        e.key = 0.f;
        e.pos = 0;
        if (ok) e = mine[i];
        L.offer(e.key, e.pos, ok);
    }
    const float t64 = read_lane(L.key, 63);
    if (lane == 0 && t64 > thr[q]) thr[q] = t64;
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

// Queries that are already in device memory: what stage_query() (flat_index.cpp) does on the host -- copy into the
// staging area, norm, and the fast-path domain test (finite, |v| <= max_abs, norm 0 or >= min_norm; a query outside it is
// staged as zeros with norm 0 and its flag cleared: the caller answers it on the exact path).  One wave per query.
// The norm only feeds the error bound and that test (the kernels recompute every score from the values).
__global__ __launch_bounds__(256) void k_stage_queries(const double* __restrict__ src, uint32_t nq, uint32_t dim,
                                                       double max_abs, double min_norm, double* __restrict__ dst,
                                                       double* __restrict__ norms, unsigned char* __restrict__ in_domain)
{
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const double* q = src + (size_t)qi * dim;
    double ss = 0.0, mx = 0.0;
    for (uint32_t c = lane; c < dim; c += 64) {
        const double v = q[c];
        ss += v * v;
        const double av = fabs(v);
        mx = av > mx ? av : mx;  // ignores NaN (ss carries it)
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        const double o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    const double norm = sqrt(ss);
    const bool finite = ss == ss && mx <= 1.797693134862315708e308 && norm <= 1.797693134862315708e308;
    const bool ok = finite && mx <= max_abs && (norm == 0.0 || norm >= min_norm);
    for (uint32_t c = lane; c < dim; c += 64) dst[(size_t)qi * dim + c] = ok ? q[c] : 0.0;
    if (lane == 0) {
        norms[qi] = ok ? norm : 0.0;
        in_domain[qi] = ok ? 1 : 0;
    }
}

// k_stage_queries + k_queries_bf16 + the clearing of the candidate counters in ONE launch (device-resident batches on the
// row-stationary kernel): wave w < nq stages query w (copy, norm, domain test) and writes its bf16 row (zero padded to ldb;
// a query outside the domain is zeros there too); waves nq .. nq_pad write the zero rows of the padding queries; every wave
// clears its query's counter.  Three launches and two gaps less in front of the sampling pass (~15 us of a 1.9 ms batch).
__global__ __launch_bounds__(256) void k_prepare_queries(const double* __restrict__ src, uint32_t nq, uint32_t nq_pad, uint32_t dim,
                                                         uint32_t ldb, double max_abs, double min_norm, double* __restrict__ dst,
                                                         double* __restrict__ norms, unsigned char* __restrict__ in_domain,
                                                         __bf16* __restrict__ q16, uint32_t* __restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq_pad) return;
    if (lane == 0) cnt[qi] = 0u;
    if (qi >= nq) {
        for (uint32_t c = lane; c < ldb; c += 64) q16[(size_t)qi * ldb + c] = (__bf16)0.0f;
        return;
    }
    const double* q = src + (size_t)qi * dim;
    // the row-stationary filter takes rows of at most 768 columns: a lane's 12 values are loaded together (clamped, never
    // predicated -- one memory round trip instead of twelve) and stay in registers for the norm, the copy and the bf16 row
    constexpr int PER = 12;
    double v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t c = (uint32_t)(lane + 64 * i);
        v[i] = q[c < dim ? c : dim - 1u];
    }
    double ss = 0.0, mx = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t c = (uint32_t)(lane + 64 * i);
        if (c >= dim) v[i] = 0.0;
        ss += v[i] * v[i];
        const double av = fabs(v[i]);
        mx = av > mx ? av : mx;  // ignores NaN (ss carries it)
    }
    for (uint32_t c = (uint32_t)(lane + 64 * PER); c < dim; c += 64) {  // longer rows (not reached through the filter's own launcher)
        const double w = q[c];
        ss += w * w;
        const double av = fabs(w);
        mx = av > mx ? av : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        const double o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    const double norm = sqrt(ss);
    const bool finite = ss == ss && mx <= 1.797693134862315708e308 && norm <= 1.797693134862315708e308;
    const bool ok = finite && mx <= max_abs && (norm == 0.0 || norm >= min_norm);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t c = (uint32_t)(lane + 64 * i);
        const double w = ok ? v[i] : 0.0;
        if (c < dim) dst[(size_t)qi * dim + c] = w;
        if (c < ldb) q16[(size_t)qi * ldb + c] = (__bf16)(float)w;  // f64 -> f32 -> bf16 (RNE), as k_queries_bf16 rounds the staged value
    }
    for (uint32_t c = (uint32_t)(lane + 64 * PER); c < ldb; c += 64) {
        const double w = (ok && c < dim) ? q[c] : 0.0;
        if (c < dim) dst[(size_t)qi * dim + c] = w;
        q16[(size_t)qi * ldb + c] = (__bf16)(float)w;
    }
    if (lane == 0) {
        norms[qi] = ok ? norm : 0.0;
        in_domain[qi] = ok ? 1 : 0;
    }
}

// f64 master rows -> UNIT-NORMALISED bf16 slab rows [n, ldb] (x/|x| in f64, then f32, then bf16 RNE;
// zero rows stay zero), plus |row| and |row|^2 rounded once to f32.  One wave per row.
__global__ __launch_bounds__(256) void k_rows_bf16(const double* __restrict__ master, uint64_t n, uint32_t dim,
                                                   uint32_t ldb, __bf16* __restrict__ out, float* __restrict__ out_nrm,
                                                   float* __restrict__ out_sqn)
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
        const double nrm = sqrt(ss);
        const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
        for (uint32_t c = lane; c < ldb; c += 64) {
            const float v = c < dim ? (float)(master[row * dim + c] * inv) : 0.0f;
            out[row * ldb + c] = (__bf16)v;
        }
        if (lane == 0) {
            out_nrm[row] = (float)nrm;
            out_sqn[row] = (float)ss;
        }
    }
}

// The same rows in FRAGMENT-MAJOR order for k_mfma_rows: row r, 16-byte piece p (8 bf16: columns 8 p .. 8 p + 7) goes to
// byte (((r / 16) * (ldb / 32) + p / 4) * 64 + (p % 4) * 16 + r % 16) * 16 -- per 16-row group and 32-column K-step
// one 1 KB chunk whose 64 pieces are in MFMA lane order (lane = 16 * (p % 4) + r % 16).  One wave per row.
__global__ __launch_bounds__(256) void k_rows_bf16_frag(const double* __restrict__ master, uint64_t row0, uint64_t n,
                                                        uint32_t dim, uint32_t ldb, unsigned char* __restrict__ out,
                                                        float* __restrict__ out_nrm, float* __restrict__ out_sqn)
{
    const int lane = threadIdx.x & 63;
    const uint64_t n_waves = (uint64_t)gridDim.x * 4;
    const uint32_t ks32 = ldb / 32;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += n_waves) {
        const double* src = master + i * dim;
        double ss = 0.0;
        for (uint32_t c = lane; c < dim; c += 64) {
            const double v = src[c];
            ss += v * v;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        const double nrm = sqrt(ss);
        const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
        const uint64_t r = row0 + i;
        for (uint32_t p = lane; p < ldb / 8; p += 64) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t c = 8 * p + e;
                v[e] = (__bf16)(c < dim ? (float)(src[c] * inv) : 0.0f);
            }
            const size_t off = ((((size_t)(r >> 4) * ks32 + (p >> 2)) * 64) + (size_t)((p & 3) * 16 + (uint32_t)(r & 15))) * 16;
            *reinterpret_cast<bf16x8*>(out + off) = v;
        }
        if (lane == 0) {
            out_nrm[r] = (float)nrm;
            out_sqn[r] = (float)ss;
        }
    }
}

}  // namespace

#define VL_MFMA_KSTEPS(X) X(8) X(16) X(24) X(32) X(48)

namespace {
// Launch shape of k_mfma_scan: 8 waves x 1 query tile (two waves share a SIMD), or 4 waves x 2 query
// tiles (one wave per SIMD with the whole register file, every A fragment feeds two MFMAs).
// dim 768 would keep 192 registers of query fragments per wave (one wave per SIMD: shape 41, 3.07 ms per
// 1024-query pass over 1.25 M rows); its default is shape 82: 8 waves, each pair splitting K (2.85 ms).
int env_shape(uint32_t ldb)
{
    const char* v = getenv("VL_MFMA_SHAPE");
    const int want = v && *v ? atoi(v) : 0;
    if (ldb >= 768) return want == 41 ? 41 : 82;  // 82 = 8 waves, K split over wave pairs (k_mfma_scan: KSPLIT)
    if (want == 81 || want == 42 || want == 41) return want;
    return 81;
}
int env_grid(uint32_t n_chunks)
{
    const char* v = getenv("VL_MFMA_GRID");
    int g = v && *v ? atoi(v) : 0;
    if (g <= 0) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        g = cus / (int)(n_chunks ? n_chunks : 1);
    }
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
                            float* out_norm, float* out_sqnorm)
{
    if (n == 0) return hipSuccess;
    const uint32_t ldb = mfma_ldb(dim);
    const int grid = (int)std::min<uint64_t>((n + 3) / 4, 16384);
    hipLaunchKernelGGL(k_rows_bf16, dim3(grid), dim3(256), 0, s, master, n, dim, ldb, reinterpret_cast<__bf16*>(out_bf16),
                       out_norm, out_sqnorm);
    return hipGetLastError();
}

hipError_t launch_stage_queries(hipStream_t s, const double* d_src, uint32_t nq, uint32_t dim, double max_abs,
                                double min_norm, double* d_dst, double* d_norms, unsigned char* in_domain)
{
    if (nq == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stage_queries, dim3((nq + 3) / 4), dim3(256), 0, s, d_src, nq, dim, max_abs, min_norm, d_dst, d_norms,
                       in_domain);
    return hipGetLastError();
}

bool launch_prepare_queries(hipStream_t s, const double* d_src, uint32_t nq, uint32_t dim, double max_abs, double min_norm,
                            double* d_dst, double* d_norms, unsigned char* in_domain, const MfmaScratch& w, hipError_t* err)
{
    *err = hipSuccess;
    if (nq == 0 || !mfma_rows_kernel(dim) || !mfma_scan_supported(dim, COSINE)) return false;  // the caller stages the old way
    const uint32_t ldb = mfma_ldb(dim);
    const uint32_t rq = (uint32_t)rs_qpb(ldb);
    const uint32_t nq_pad = (nq + rq - 1) / rq * rq;
    if (nq > w.nq_cap || nq_pad > w.nq_pad_cap) return false;
    hipLaunchKernelGGL(k_prepare_queries, dim3((nq_pad + 3) / 4), dim3(256), 0, s, d_src, nq, nq_pad, dim, ldb, max_abs, min_norm,
                       d_dst, d_norms, in_domain, reinterpret_cast<__bf16*>(w.q_bf16), w.cnt);
    *err = hipGetLastError();
    return true;
}

uint32_t mfma_sequence_queries(uint32_t dim)
{
    if (!mfma_rows_kernel(dim)) return 1024;  // the LDS-tile kernel: 4 chunks of 256
    return std::min<uint32_t>((uint32_t)MFMA_MAX_BATCH, 16u * (uint32_t)rs_qpb(mfma_ldb(dim)));
}

bool mfma_rows_kernel(uint32_t dim)
{
    const char* kv = getenv("VL_MFMA_KERNEL");
    const uint32_t ldb = mfma_ldb(dim);
    return (ldb <= 512 || ldb == 768) && !(kv && kv[0] == 't');
}

hipError_t launch_rows_bf16_frag(hipStream_t s, const double* master_rows, uint64_t row0, uint64_t n, uint32_t dim,
                                 void* slab_frag_base, float* norm_base, float* sqnorm_base)
{
    if (n == 0) return hipSuccess;
    const uint32_t ldb = mfma_ldb(dim);
    if (ldb % 32 != 0) return hipErrorInvalidValue;
    const int grid = (int)std::min<uint64_t>((n + 3) / 4, 16384);
    hipLaunchKernelGGL(k_rows_bf16_frag, dim3(grid), dim3(256), 0, s, master_rows, row0, n, dim, ldb,
                       reinterpret_cast<unsigned char*>(slab_frag_base), norm_base, sqnorm_base);
    return hipGetLastError();
}

hipError_t launch_mfma_candidates(hipStream_t s, int metric, const void* slab_bf16, const float* row_norm,
                                  const float* row_sqnorm,
                                  const double* q64, uint32_t nq, uint64_t n_rows, uint32_t dim,
                                  const MfmaScratch& w, Cand32* out_lists, MfmaLaunchInfo* info, bool queries_prepared)
{
    if (nq == 0 || n_rows == 0 || n_rows >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (!mfma_scan_supported(dim, metric) || nq > w.nq_cap) return hipErrorInvalidValue;
    const uint32_t ldb = mfma_ldb(dim);
    __bf16* q16 = reinterpret_cast<__bf16*>(w.q_bf16);
    const __bf16* slab = reinterpret_cast<const __bf16*>(slab_bf16);
    // ---- row-stationary kernel (k_mfma_rows): every stride the index pads to (128 / 256 / 384 / 512 / 768);
    //      VL_MFMA_KERNEL=tile keeps the LDS-tile kernel.  Each path stages its own queries and clears cnt once. ----
    {
        if (mfma_rows_kernel(dim)) {  // slab_bf16 is then the fragment-major slab (launch_rows_bf16_frag)
            const uint32_t rq = (uint32_t)rs_qpb(ldb);
            const uint32_t rnq_pad = (nq + rq - 1) / rq * rq;
            if (rnq_pad > w.nq_pad_cap) return hipErrorInvalidValue;
            const uint32_t r_chunks = rnq_pad / rq;
            if (!queries_prepared) {  // (launch_prepare_queries wrote the bf16 queries and cleared the counters already)
                const size_t total = (size_t)rnq_pad * ldb;
                const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
                hipLaunchKernelGGL(k_queries_bf16, dim3(grid), dim3(256), 0, s, q64, nq, rnq_pad, dim, ldb, q16);
                hipError_t e2 = hipMemsetAsync(w.cnt, 0, (size_t)rnq_pad * sizeof(uint32_t), s);
                if (e2 != hipSuccess) return e2;
            }
            // the launch plan (filter_plan.hpp): sample size and sampling grid, where the pass-1 stages end
            const uint32_t wg_cap = (uint32_t)env_grid(r_chunks);  // co-resident workgroups per query chunk
            FilterKnobs kn;
            auto env_u = [](const char* name) -> uint32_t {
                const char* v = getenv(name);
                return v && *v && atoi(v) > 0 ? (uint32_t)atoi(v) : 0u;
            };
            kn.sample_div = env_u("VL_MFMA_SAMPLE_DIV");
            kn.sample_min_rows = env_u("VL_MFMA_SAMPLE_MIN");
            kn.stages = (int)env_u("VL_MFMA_STAGES");
            kn.stage_end[0] = env_u("VL_MFMA_STAGE1");
            kn.stage_end[1] = env_u("VL_MFMA_STAGE2");
            kn.stage_end[2] = env_u("VL_MFMA_STAGE3");
            const FilterPlan fp = filter_plan(n_rows, wg_cap, (uint32_t)RS_NWAVES, (uint32_t)MFMA_GROUPS, kn);
            const uint32_t sample_blocks = fp.sample_blocks, gx0 = fp.gx0, r_gpw = fp.gpw, r_groups = fp.groups;
            const uint64_t r_sample_rows = std::min<uint64_t>((uint64_t)sample_blocks * 32, n_rows);
            const int r_stages = fp.stages;
            const uint32_t* st_end = fp.st_end;
            bool r_launched = false;
            // Shape of pass 1 (MODE 1): "4x64" = one wave per SIMD on 64-row blocks (round 4), "8x32" = two waves per SIMD on
            // 32-row blocks (rounds 2-3).  The sampling pass (MODE 0, 1/32 .. 1/64 of the rows) keeps the 8-wave shape: its group
            // bookkeeping (gpw groups per workgroup) is written for it.  VL_MFMA_ROWS_SHAPE picks; the default is the measured faster one.
            // Shape of pass 1 (MODE 1).  Shipped: "8x32", two waves per SIMD on 32-row blocks.  A build with -DRS_WIDE_SHAPES also
            // holds "4x64" (one wave per SIMD on 64-row blocks with the 512-register file; VL_MFMA_ROWS_SHAPE=4x64 picks it per
            // launch sequence, tools/rows_shape_ab.py alternates the two in one process).  Measured in round 4 and NOT shipped:
            // 8-28 % slower than 8x32 on every stride (profiles/r04_k4r_shapes_anatomy.txt) -- one wave per SIMD has nobody to
            // hide its epilogue and its waits behind.  The sampling pass (MODE 0) keeps the 8-wave shape either way.
#ifdef RS_WIDE_SHAPES
            const int wide = []() {
                const char* v = getenv("VL_MFMA_ROWS_SHAPE");
                return (v && v[0] == '4') ? 1 : 0;
            }();
#endif
            const uint32_t n_blk64 = (uint32_t)((n_rows + 63) / 64);
            (void)n_blk64;
            const bool r_stream = r_chunks == 1 && []() {
                const char* v = getenv("VL_MFMA_STREAM_LOADS");  // 0: plain loads everywhere (A/B)
                return !(v && v[0] == '0');
            }();  // one chunk: nobody re-reads a row block
#ifdef RS_WIDE_SHAPES  /* stage ends are planned in 32-row blocks: halved (floor) at both ends, the last one ends the index */
#define VL_RLAUNCH_WIDE(K, MET)                                                                                                 \
    if (wide == 1) {                                                                                                            \
        const uint32_t tb = st_end[st] / 2, te = (st + 1 == r_stages) ? n_blk64 : st_end[st + 1] / 2;                            \
        const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>((te - tb + 3) / 4, wg_cap));                               \
        hipLaunchKernelGGL((k_mfma_rows<K, 1, MET, 4, 4>), dim3(gx, r_chunks), dim3(4 * 64), 0, s, slab, row_norm,               \
                           row_sqnorm, q16, nq, tb, te, (uint32_t)n_rows, (int*)nullptr, 0u, 1u, w.thr, w.cand, w.cnt,          \
                           (uint32_t)MFMA_CAND_CAP);                                                                            \
    } else
#else
#define VL_RLAUNCH_WIDE(K, MET)
#endif
#define VL_RLAUNCH2(K, MET)                                                                                                     \
    {                                                                                                                           \
        hipLaunchKernelGGL((k_mfma_rows<K, 0, MET>), dim3(gx0, r_chunks), dim3(RS_NWAVES * 64), 0, s, slab, row_norm,           \
                           row_sqnorm, q16, nq, 0u, sample_blocks, (uint32_t)r_sample_rows, w.gmax, r_groups, r_gpw,            \
                           (const float*)nullptr, (Cand32*)nullptr, (uint32_t*)nullptr, 0u);                                    \
        hipLaunchKernelGGL(k_thresholds, dim3((nq + 3) / 4), dim3(256), 0, s, w.gmax, r_groups, nq, q64 + (size_t)nq * dim,     \
                           w.thr);                                                                                              \
        for (int st = 0; st < r_stages; ++st) {                                                                                 \
            VL_RLAUNCH_WIDE(K, MET)                                                                                             \
            {                                                                                                                   \
                const uint32_t tb = st_end[st], te = st_end[st + 1];                                                            \
                const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>((te - tb + RS_NWAVES - 1) / RS_NWAVES, wg_cap));  \
                if (r_stream)                                                                                                   \
                    hipLaunchKernelGGL((k_mfma_rows<K, 1, MET, 2, RS_NWAVES, true>), dim3(gx, r_chunks), dim3(RS_NWAVES * 64),  \
                                       0, s, slab, row_norm, row_sqnorm, q16, nq, tb, te, (uint32_t)n_rows, (int*)nullptr, 0u,  \
                                       1u, w.thr, w.cand, w.cnt, (uint32_t)MFMA_CAND_CAP);                                      \
                else                                                                                                            \
                hipLaunchKernelGGL((k_mfma_rows<K, 1, MET>), dim3(gx, r_chunks), dim3(RS_NWAVES * 64), 0, s, slab, row_norm,    \
                                   row_sqnorm, q16, nq, tb, te, (uint32_t)n_rows, (int*)nullptr, 0u, 1u, w.thr, w.cand, w.cnt,  \
                                   (uint32_t)MFMA_CAND_CAP);                                                                    \
            }                                                                                                                   \
            if (st + 1 < r_stages)                                                                                              \
                hipLaunchKernelGGL(k_refine_thresholds, dim3((nq + 3) / 4), dim3(256), 0, s, w.cand, w.cnt,                     \
                                   (uint32_t)MFMA_CAND_CAP, nq, w.thr);                                                         \
        }                                                                                                                       \
        r_launched = true;                                                                                                      \
    }
#define VL_RLAUNCH(K)                                             \
    if (!r_launched && ldb == (uint32_t)(K * 16)) {               \
        if (metric == COSINE) VL_RLAUNCH2(K, COSINE)              \
        else if (metric == EUCLIDEAN) VL_RLAUNCH2(K, EUCLIDEAN)   \
        else VL_RLAUNCH2(K, DOT)                                  \
    }
            VL_RLAUNCH(8) VL_RLAUNCH(16) VL_RLAUNCH(24) VL_RLAUNCH(32) VL_RLAUNCH(48)
#undef VL_RLAUNCH
#undef VL_RLAUNCH2
            if (!r_launched) return hipErrorInvalidValue;
            if (info) {
                info->ksteps = (int)(ldb / 16);
                info->metric = metric;
                info->chunks = (int)r_chunks;
                const uint32_t lb = st_end[r_stages - 1], le = st_end[r_stages];
                info->grid_x = (int)std::max<uint32_t>(1u, std::min<uint32_t>((le - lb + RS_NWAVES - 1) / RS_NWAVES, wg_cap));
                info->stages = r_stages;
                info->sample_blocks = (int)sample_blocks;
            }
            hipLaunchKernelGGL(k_select_candidates, dim3((nq + 3) / 4), dim3(256), 0, s, w.cand, w.cnt, (uint32_t)MFMA_CAND_CAP, nq,
                               (const float*)w.thr, out_lists);
            return hipGetLastError();
        }
    }

    // ---- LDS-tile kernel (k_mfma_scan) ----
    // launch shape: (waves per workgroup, 32-query tiles per wave); 256 queries per workgroup
    const int shape = env_shape(ldb);  // 81 = 8 waves x 1 tile, 42 = 4 waves x 2 tiles, 41 = 4 waves x 1 tile, 82 = 8 waves, K split over pairs
    const int nwaves = shape / 10, qt = shape == 82 ? 1 : shape % 10;
    const uint32_t qpb = (uint32_t)(shape == 82 ? nwaves / 2 : nwaves) * 32 * (uint32_t)qt;
    const uint32_t nq_pad = (nq + qpb - 1) / qpb * qpb;
    if (nq_pad > w.nq_pad_cap) return hipErrorInvalidValue;
    // the 8-wave shape works on 64-row tiles (two 32-row MFMA blocks per barrier), the others on 32-row tiles
    // (up to d = 384; at 512 the eight 16-byte pieces per thread of a 64-row tile spill)
    const uint32_t tile_rows = (shape == 81 && ldb <= 384) ? 2u * MF_ROWS : (uint32_t)MF_ROWS;
    const uint32_t n_tiles = (uint32_t)((n_rows + tile_rows - 1) / tile_rows);
    {
        const size_t total = (size_t)nq_pad * ldb;
        const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(k_queries_bf16, dim3(grid), dim3(256), 0, s, q64, nq, nq_pad, dim, ldb, q16);
    }
    hipError_t e = hipMemsetAsync(w.cnt, 0, (size_t)nq_pad * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;

    // pass 0: a sample of the tiles, one contiguous range per workgroup = one "group" per workgroup
    // The threshold is exceeded by about 64 rows of the sample, i.e. by a fraction 64 / sample_rows of all
    // rows: a wave then appends 32 queries x 512 rows x that fraction candidates per trip to its 128-entry
    // ring segment.  A sample of at least 65536 rows (or the whole index) keeps that near 16; with only
    // 8192 sampled rows an index of 10^5 rows overflowed the rings and every query fell back.
    uint32_t sample_tiles = n_tiles / 16;
    const uint32_t min_tiles = 65536u / tile_rows;
    const uint32_t min_sample = n_tiles < min_tiles ? n_tiles : min_tiles;
    if (sample_tiles < min_sample) sample_tiles = min_sample;
    const uint32_t n_groups = sample_tiles < (uint32_t)MFMA_GROUPS ? sample_tiles : (uint32_t)MFMA_GROUPS;
    const dim3 grid0(n_groups, nq_pad / qpb);
    const uint64_t sample_rows = std::min<uint64_t>((uint64_t)sample_tiles * tile_rows, n_rows);
    // all query chunks of a launch are co-resident (one workgroup per CU) and walk the same tile
    // sequence, so a tile is fetched from HBM once and served to the other chunks by L2 / Infinity Cache
    const uint32_t n_chunks = nq_pad / qpb;
    const int pass1_blocks = (int)std::min<uint32_t>(n_tiles, (uint32_t)env_grid(n_chunks));
    // Pass 1 runs in stages of growing size (3/16, 4/16, 9/16 of the tiles); between stages the thresholds
    // are tightened from the candidates found so far (k_refine_thresholds).  Short scans keep one stage.
    uint32_t stage_end[4] = {0, n_tiles, n_tiles, n_tiles};
    int n_stages = 1;
    {
        const char* se = getenv("VL_MFMA_STAGES");
        const int want = se && *se ? atoi(se) : 3;
        if (want >= 2 && n_tiles >= 64u * (uint32_t)pass1_blocks) {
            n_stages = want >= 3 ? 3 : 2;
            stage_end[1] = (uint32_t)((uint64_t)n_tiles * 3 / 16);
            stage_end[2] = n_stages == 3 ? (uint32_t)((uint64_t)n_tiles * 7 / 16) : n_tiles;
            stage_end[3] = n_tiles;
        }
    }

    bool launched = false;
#define VL_LAUNCH3(K, MET, NW, QTT, KSP, SUBP)                                                                                   \
    {                                                                                                                   \
        hipLaunchKernelGGL((k_mfma_scan<K, 0, MET, NW, QTT, KSP, SUBP>), grid0, dim3(NW * 64), 0, s, slab, row_norm, row_sqnorm, q16, nq,       \
                           sample_tiles, (uint32_t)sample_rows, w.gmax, n_groups, (const float*)nullptr,               \
                           (Cand32*)nullptr, (uint32_t*)nullptr, 0u, 0u);                                               \
        hipLaunchKernelGGL(k_thresholds, dim3((nq + 3) / 4), dim3(256), 0, s, w.gmax, n_groups, nq,                     \
                           q64 + (size_t)nq * dim, w.thr);                                                             \
        for (int st = 0; st < n_stages; ++st) {                                                                         \
            const uint32_t tb = stage_end[st], te = stage_end[st + 1];                                                  \
            const dim3 grid1((uint32_t)std::min<uint32_t>(te - tb, (uint32_t)pass1_blocks), nq_pad / qpb);              \
            hipLaunchKernelGGL((k_mfma_scan<K, 1, MET, NW, QTT, KSP, SUBP>), grid1, dim3(NW * 64), 0, s, slab, row_norm, row_sqnorm, q16, nq,   \
                               te, (uint32_t)n_rows, (int*)nullptr, 0u, w.thr, w.cand, w.cnt, (uint32_t)MFMA_CAND_CAP, tb); \
            if (st + 1 < n_stages)                                                                                      \
                hipLaunchKernelGGL(k_refine_thresholds, dim3((nq + 3) / 4), dim3(256), 0, s, w.cand, w.cnt,             \
                                   (uint32_t)MFMA_CAND_CAP, nq, w.thr);                                                 \
        }                                                                                                               \
        launched = true;                                                                                                \
    }
#define VL_LAUNCH2(K, MET)                                        \
    {                                                             \
        if (shape == 81) VL_LAUNCH3(K, MET, 8, 1, 1, (K <= 24 ? 2 : 1)) \
        else if (shape == 42) VL_LAUNCH3(K, MET, 4, 2, 1, 1)      \
        else VL_LAUNCH3(K, MET, 4, 1, 1, 1)                       \
    }
#define VL_LAUNCH(K)                                              \
    if (!launched && ldb == (uint32_t)(K * 16)) {                 \
        if (metric == COSINE) VL_LAUNCH2(K, COSINE)               \
        else if (metric == EUCLIDEAN) VL_LAUNCH2(K, EUCLIDEAN)    \
        else VL_LAUNCH2(K, DOT)                                   \
    }
    if (shape == 82 && ldb == 768) {  // dim 768 with K split over wave pairs
        if (metric == COSINE) VL_LAUNCH3(48, COSINE, 8, 1, 2, 1)
        else if (metric == EUCLIDEAN) VL_LAUNCH3(48, EUCLIDEAN, 8, 1, 2, 1)
        else VL_LAUNCH3(48, DOT, 8, 1, 2, 1)
    }
    VL_MFMA_KSTEPS(VL_LAUNCH)
#undef VL_LAUNCH
#undef VL_LAUNCH2
#undef VL_LAUNCH3
    if (!launched) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_select_candidates, dim3((nq + 3) / 4), dim3(256), 0, s, w.cand, w.cnt, (uint32_t)MFMA_CAND_CAP, nq,
                               (const float*)w.thr, out_lists);
    return hipGetLastError();
}

#define VL_BF16_SCAN_SHAPES(X) X(8, 2) X(8, 4) X(8, 6) X(8, 8) X(16, 6)

bool scan_bf16_supported(uint32_t dim, int metric)
{
    if (metric != COSINE && metric != DOT && metric != EUCLIDEAN) return false;
    const uint32_t ld8 = mfma_ldb(dim) / 8;
    bool ok = false;
#define VL_CHK(G, VPL) ok = ok || (ld8 == (uint32_t)(G * VPL));
    VL_BF16_SCAN_SHAPES(VL_CHK)
#undef VL_CHK
    return ok;
}

hipError_t launch_scan_bf16(hipStream_t s, int metric, const void* slab_bf16, const float* row_norm,
                            const float* row_sqnorm, const double* q64, uint64_t n, uint32_t dim, Cand32* partials,
                            int* grid_out)
{
    if (n == 0 || n >= 0xFFFFFFFFull || !scan_bf16_supported(dim, metric)) return hipErrorInvalidValue;
    const uint32_t ld8 = mfma_ldb(dim) / 8;
    const u32x4* slab = reinterpret_cast<const u32x4*>(slab_bf16);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const char* ge = getenv("VL_SCAN16_BPC");
    const int bpc = ge && *ge ? atoi(ge) : 3;  // measured: 2-3 workgroups per CU stream fastest (like k_scan)
    bool launched = false;
    int grid = 1;
#define VL_L3(MET, G, VPL)                                                                                     \
    {                                                                                                          \
        const uint64_t steps = (n + (64 / G) - 1) / (64 / G);                                                  \
        uint64_t blocks = (steps + 3) / 4;                                                                     \
        if (blocks > (uint64_t)(cus * bpc)) blocks = (uint64_t)(cus * bpc);                                    \
        if (blocks > (uint64_t)SCAN_MAX_GRID) blocks = SCAN_MAX_GRID;                                          \
        grid = (int)(blocks < 1 ? 1 : blocks);                                                                 \
        hipLaunchKernelGGL((k_scan_bf16<MET, G, VPL>), dim3(grid), dim3(256), 0, s, slab, row_norm, row_sqnorm, \
                           q64, dim, (uint32_t)n, partials);                                                   \
        launched = true;                                                                                       \
    }
#define VL_L(G, VPL)                                               \
    if (!launched && ld8 == (uint32_t)(G * VPL)) {                 \
        if (metric == COSINE) VL_L3(COSINE, G, VPL)                \
        else if (metric == EUCLIDEAN) VL_L3(EUCLIDEAN, G, VPL)     \
        else VL_L3(DOT, G, VPL)                                    \
    }
    VL_BF16_SCAN_SHAPES(VL_L)
#undef VL_L
#undef VL_L3
    if (!launched) return hipErrorInvalidValue;
    if (grid_out) *grid_out = grid;
    return hipGetLastError();
}

}  // namespace vl
