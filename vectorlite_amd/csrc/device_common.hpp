// device_common.hpp -- device helpers shared by kernels.hip (flat scan) and hnsw.hip (graph walk).
// Reference paths are relative to /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "kernels.hpp"

namespace vl {
namespace dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Total order of candidates: higher key first, then lower storage position (the reference's
// stable sort keeps insertion order on ties: src/index/flat.rs:116, src/client.rs:665-667).
template <typename K>
__device__ __forceinline__ bool better(K ka, uint32_t pa, K kb, uint32_t pb)
{
    return ka > kb || (ka == kb && pa < pb);
}

// ---- cross-lane moves without LDS traffic ------------------------------------------------------
// lane i <- lane i-1 (lane 0 keeps its own value): one DPP wave_shr:1 move per dword.
__device__ __forceinline__ int wave_shr1_dw(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ float wave_shr1(float v) { return __int_as_float(wave_shr1_dw(__float_as_int(v))); }
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return (uint32_t)wave_shr1_dw((int)v); }
__device__ __forceinline__ double wave_shr1(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = wave_shr1_dw((int)(b & 0xFFFFFFFFll)), hi = wave_shr1_dw((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ unsigned long long wave_shr1(unsigned long long b)
{
    const int lo = wave_shr1_dw((int)(b & 0xFFFFFFFFull)), hi = wave_shr1_dw((int)(b >> 32));
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
// value of a wave-uniform lane (v_readlane_b32 into an SGPR)
__device__ __forceinline__ unsigned long long read_lane(unsigned long long b, int lane)
{
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFull), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ float read_lane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ double read_lane(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}


// ---------------------------------------------------------------------------------------------
// Reference-order f64 arithmetic (this TU is built with -ffp-contract=off: `a += x * y` is one
// rounded multiply followed by one rounded add, like rustc's output for src/lib.rs:425-572).
// ---------------------------------------------------------------------------------------------
template <int METRIC>
struct Acc64 {
    double a, b, c;
    __device__ __forceinline__ void init()
    {
        // cosine folds from (0.0, 0.0, 0.0) (src/lib.rs:428); the `.sum::<f64>()` metrics fold from
        // the float additive identity -0.0 (core::iter::Sum, rustc >= 1.83; crate edition 2024).
        a = (METRIC == COSINE) ? 0.0 : -0.0;
        b = 0.0;
        c = 0.0;
    }
    __device__ __forceinline__ void step(double x, double y)
    {
        if (METRIC == COSINE) {
            a += x * y;
            b += x * x;
            c += y * y;
        } else if (METRIC == EUCLIDEAN) {
            const double d = x - y;
            a += d * d;
        } else if (METRIC == MANHATTAN) {
            a += fabs(x - y);
        } else {
            a += x * y;
        }
    }
    // SimilarityMetric::calculate's return value
    __device__ __forceinline__ double score() const
    {
        if (METRIC == COSINE) {
            const double na = sqrt(b), nb = sqrt(c);
            if (na == 0.0 || nb == 0.0) return 0.0;
            return a / (na * nb);
        }
        if (METRIC == EUCLIDEAN) return 1.0 / (1.0 + sqrt(a));
        if (METRIC == MANHATTAN) return 1.0 / (1.0 + a);
        return a;
    }
};

// Rust `f64 as u64`: truncation toward zero, saturating, NaN -> 0.
__device__ __forceinline__ unsigned long long rust_as_u64(double v)
{
    if (!(v > 0.0)) return 0ull;
    if (v >= 18446744073709551616.0) return ~0ull;
    return (unsigned long long)v;
}

// impl Metric<Vec<f64>>::distance (src/index/hnsw.rs:113-174), split in two steps:
// hnsw_scaled() is the f64 value the reference hands to `as u64`; hnsw_quantise() applies the cast.
template <int METRIC>
__device__ __forceinline__ double hnsw_scaled(const Acc64<METRIC>& A)
{
    if (METRIC == EUCLIDEAN) return sqrt(A.a) * 1000.0;
    if (METRIC == COSINE) {
        const double na = sqrt(A.b), nb = sqrt(A.c);
        if (na == 0.0 || nb == 0.0) return 1000.0;  // `return 1000` (:139-141)
        const double cosine_sim = A.a / (na * nb);
        return (1.0 - cosine_sim) * 1000.0;
    }
    if (METRIC == MANHATTAN) return A.a * 1000.0;
    double d = A.a;  // f64::clamp(-1000, 1000): NaN stays NaN
    if (d < -1000.0) d = -1000.0;
    if (d > 1000.0) d = 1000.0;
    return 1000.0 - d;
}
template <int METRIC>
__device__ __forceinline__ unsigned long long hnsw_quantise(const Acc64<METRIC>& A)
{
    return rust_as_u64(hnsw_scaled<METRIC>(A));
}
// Walk key: a u64 whose unsigned order refines the reference's u64 distance order (it is the f64
// bit pattern of the scaled distance, clamped like `as u64` clamps: negative / NaN -> 0).  Two
// nodes the reference's truncation would tie are still told apart, which keeps a greedy walk from
// stalling on plateaus; rust_as_u64(key_to_scaled(key)) is exactly the reference's distance.
__device__ __forceinline__ unsigned long long walk_key(double scaled)
{
    if (!(scaled > 0.0)) return 0ull;
    return (unsigned long long)__double_as_longlong(scaled);
}
__device__ __forceinline__ unsigned long long walk_key_to_u64(unsigned long long key)
{
    return rust_as_u64(__longlong_as_double((long long)key));
}


template <typename K>
__device__ __forceinline__ K neg_inf();
template <>
__device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <>
__device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

// ---------------------------------------------------------------------------------------------
// Sorted top-64 list held one entry per lane (lane 0 = best).
// ---------------------------------------------------------------------------------------------
template <typename K>
struct TopList {
    K key;
    uint32_t pos;
    K thr_key;  // wave-uniform copy of lane 63's entry
    uint32_t thr_pos;

    __device__ __forceinline__ void init()
    {
        key = neg_inf<K>();
        pos = POS_SENTINEL;
        // Keep the f64 -inf out of constant propagation: hipcc (ROCm 7.2) otherwise materialises the
        // wave-uniform threshold with `s_mov_b64 s[..], 0xfff0000000000000`, a 64-bit literal gfx950
        // cannot encode (it is truncated to 32 bits: the threshold silently becomes +0.0).
        asm volatile("" : "+v"(key));
        thr_key = read_lane(key, WAVE - 1);
        thr_pos = POS_SENTINEL;
    }

    __device__ __forceinline__ void insert(K k, uint32_t p)
    {
        // entries that stay in front of (k, p): a prefix of the lanes because the list is sorted
        const unsigned long long ahead = __ballot(better<K>(key, pos, k, p));
        const int idx = __popcll(ahead);
        const K upk = wave_shr1(key);
        const uint32_t upp = wave_shr1(pos);
        const int lane = lane_id();
        if (lane == idx) {
            key = k;
            pos = p;
        } else if (lane > idx) {
            key = upk;
            pos = upp;
        }
        thr_key = read_lane(key, WAVE - 1);
        thr_pos = read_lane(pos, WAVE - 1);
    }

    // Every lane may offer one (key, pos); lanes are drained in lane order.
    __device__ __forceinline__ void offer(K k, uint32_t p, bool active)
    {
        unsigned long long m = __ballot(active && better<K>(k, p, thr_key, thr_pos));
        while (m) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            m &= m - 1;
            insert(read_lane(k, src), read_lane(p, src));
        }
    }

    // Merge with another sorted list handed over REVERSED (lane i holds its entry 63-i):
    // the element-wise best is a bitonic sequence holding the 64 best of the union.
    __device__ __forceinline__ void merge_reversed(K ok, uint32_t op)
    {
        if (better<K>(ok, op, key, pos)) {
            key = ok;
            pos = op;
        }
        const int lane = lane_id();
#pragma unroll
        for (int o = WAVE / 2; o >= 1; o >>= 1) {
            const K k2 = __shfl_xor(key, o);
            const uint32_t p2 = __shfl_xor(pos, o);
            const bool lower = (lane & o) == 0;
            const bool other_better = better<K>(k2, p2, key, pos);
            if (lower == other_better) {
                key = k2;
                pos = p2;
            }
        }
        thr_key = read_lane(key, WAVE - 1);
        thr_pos = read_lane(pos, WAVE - 1);
    }
};

// Sort the 64 (key, pos) pairs a wave holds one per lane so that lane i ends with the (63 - i)-th best: the REVERSED order
// TopList::merge_reversed() takes.  Bitonic network, 21 compare-exchange stages of cross-lane shuffles.
template <typename K>
__device__ __forceinline__ void sort64_reversed(K& key, uint32_t& pos)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 2; k <= WAVE; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const K k2 = __shfl_xor(key, j);
            const uint32_t p2 = __shfl_xor(pos, j);
            const bool up = (lane & k) == 0;                 // this block sorts worst-first (k = 64: the whole wave)
            const bool keep_worse = ((lane & j) == 0) == up; // the lower lane of an ascending pair keeps the worse entry
            const bool other_worse = better<K>(key, pos, k2, p2);
            if (keep_worse == other_worse) {
                key = k2;
                pos = p2;
            }
        }
    }
}

// Tree-merge the NW sorted wave lists of a workgroup through LDS; wave 0 ends with the result.
template <typename K, typename C, int NW>
__device__ __forceinline__ void block_merge(TopList<K>& L, C* sh /* [NW][64] */)
{
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    sh[wave * WAVE + lane].key = L.key;
    sh[wave * WAVE + lane].pos = L.pos;
    __syncthreads();
#pragma unroll
    for (int s = NW / 2; s >= 1; s >>= 1) {
        if (wave < s) {
            const C o = sh[(wave + s) * WAVE + (WAVE - 1 - lane)];
            L.merge_reversed(o.key, o.pos);
            sh[wave * WAVE + lane].key = L.key;
            sh[wave * WAVE + lane].pos = L.pos;
        }
        __syncthreads();
    }
}

// The same for a workgroup in which only waves [0, active) hold a list (the others' are empty): the tree spans the
// next power of two of `active`, so 12 scan lists (3 waves) cost 2 levels, not 4.  `active` is workgroup-uniform.
template <typename K, typename C>
__device__ __forceinline__ void block_merge_n(TopList<K>& L, C* sh, int active)
{
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    int top = 1;
    while (top < active) top <<= 1;
    if (wave < top) {
        sh[wave * WAVE + lane].key = L.key;
        sh[wave * WAVE + lane].pos = L.pos;
    }
    __syncthreads();
    for (int s = top / 2; s >= 1; s >>= 1) {
        if (wave < s) {
            const C o = sh[(wave + s) * WAVE + (WAVE - 1 - lane)];
            L.merge_reversed(o.key, o.pos);
            sh[wave * WAVE + lane].key = L.key;
            sh[wave * WAVE + lane].pos = L.pos;
        }
        __syncthreads();
    }
}

// Fold up to 4 consecutive sorted lists (global memory) into L with bitonic merges: the loads (all in flight
// together) and the merges are separate steps so that a caller can issue other loads in between.
template <typename C>
__device__ __forceinline__ void fold_lists4_load(C (&e)[4], const C* __restrict__ lists, int first, int count)
{
    const int lane = lane_id();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int li = first + (i < count ? i : 0);
        const int slot = i == 0 ? lane : (KP - 1 - lane);
        if (count > 0) e[i] = lists[(size_t)li * KP + slot];
    }
}
template <typename K, typename C>
__device__ __forceinline__ void fold_lists4_merge(TopList<K>& L, const C (&e)[4], int count)
{
    L.init();
    if (count > 0) {
        L.key = e[0].key;
        L.pos = e[0].pos;
    }
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < count) L.merge_reversed(e[i].key, e[i].pos);
    L.thr_key = read_lane(L.key, WAVE - 1);
    L.thr_pos = read_lane(L.pos, WAVE - 1);
}
template <typename K, typename C>
__device__ __forceinline__ void fold_lists4(TopList<K>& L, const C* __restrict__ lists, int first, int count)
{
    C e[4];
    fold_lists4_load<C>(e, lists, first, count);
    fold_lists4_merge<K, C>(L, e, count);
}


}  // namespace dev
}  // namespace vl
