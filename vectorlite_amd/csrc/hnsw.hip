// hnsw.hip -- GPU graph walk and batched graph build behind the reference's HNSWIndex.
// See hnsw.hpp for what is the reference's (the u64 distance callbacks, reproduced bit for bit)
// and what is our own (the walk: parity with crate hnsw 0.11.0 is unpinned).
//
// One WAVE per query / per inserted node:
//   * a hop loads the <= 32 neighbour ids of the expanded node with one coalesced load and filters them
//     through the walk's own visited set: a bitmap over the nodes (one returning atomic-or per neighbour) plus a
//     log of the nodes it set, so that the set is cleared in time proportional to the walk, not to the graph;
//   * walks navigate by f32 distances (two lanes per neighbour on the flat scan's f32 slab: half the bytes
//     and half the serial length of the f64 row; 3x lower single-query latency, 2x faster build);
//   * a QUERY walk then gives every node of its final beam the reference's exact callback value -- one lane
//     walks one f64 row in index order, separate multiply and add: the per-hop distance kernel of
//     north_star -- from which the returned distances and scores are computed; the BUILD keeps f32-derived
//     edge distances (used only to pick the farthest edge in the link phase);
//   * the beam (result list + frontier in one) is a sorted list held one or two entries per lane,
//     updated with ballot / DPP shifts like the flat scan's top-k list.
#include "hnsw.hpp"

#include <mutex>
#include <type_traits>

#include "device_common.hpp"

namespace vl {
using namespace dev;
namespace {

constexpr uint32_t EXPANDED = 0x80000000u;  // flag bit in the node word (node ids < 2^31)

// Sorted beam: ascending by (dist, node); capacity 64*S, logical width `ef` <= 64*S.
template <int S>
struct BeamList {
    unsigned long long d[S];
    uint32_t v[S];

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            d[s] = ~0ull;
            v[s] = HNSW_NONE;  // carries the EXPANDED bit: never picked for expansion
        }
    }
    __device__ __forceinline__ int rank_of(unsigned long long dist, uint32_t node) const
    {
        int c = 0;
#pragma unroll
        for (int s = 0; s < S; ++s)
            c += __popcll(__ballot(d[s] < dist || (d[s] == dist && (v[s] & ~EXPANDED) < node)));
        return c;
    }
    __device__ __forceinline__ void insert(unsigned long long dist, uint32_t node, int ef)
    {
        const int idx = rank_of(dist, node);
        if (idx >= ef) return;  // wave-uniform
        const int lane = lane_id();
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {  // top slot first: it reads the slot below before that moves
            if ((s + 1) * 64 <= idx) break;  // wave-uniform: this slot and the ones below it hold entries in front of idx
            const int j = s * 64 + lane;
            unsigned long long pd = wave_shr1(d[s]);
            uint32_t pv = wave_shr1(v[s]);
            if (s > 0) {
                const unsigned long long cd = read_lane(d[s > 0 ? s - 1 : 0], 63);
                const uint32_t cv = read_lane(v[s > 0 ? s - 1 : 0], 63);
                if (lane == 0) {
                    pd = cd;
                    pv = cv;
                }
            }
            if (j == idx) {
                d[s] = dist;
                v[s] = node;
            } else if (j > idx) {
                d[s] = pd;
                v[s] = pv;
            }
            if (j >= ef) {
                d[s] = ~0ull;
                v[s] = HNSW_NONE;
            }
        }
    }
    __device__ __forceinline__ void get(int idx, unsigned long long& dist, uint32_t& node) const
    {
        const int si = idx >> 6, l = idx & 63;  // idx is wave-uniform
        dist = read_lane(d[0], l);
        node = read_lane(v[0], l);
#pragma unroll
        for (int s = 1; s < S; ++s)
            if (si == s) {
                dist = read_lane(d[s], l);
                node = read_lane(v[s], l);
            }
    }
    __device__ __forceinline__ int next_unexpanded() const
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const unsigned long long m = __ballot((v[s] & EXPANDED) == 0u);
            if (m) return s * 64 + __ffsll((long long)m) - 1;
        }
        return -1;
    }
    __device__ __forceinline__ void mark_expanded(int idx)
    {
        const int lane = lane_id();
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s * 64 + lane == idx) v[s] |= EXPANDED;
    }
    // every real entry becomes an unexpanded entry point again (next layer)
    __device__ __forceinline__ void reopen()
    {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (v[s] != HNSW_NONE) v[s] &= ~EXPANDED;
    }
    __device__ __forceinline__ void truncate(int ef)
    {
        const int lane = lane_id();
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s * 64 + lane >= ef) {
                d[s] = ~0ull;
                v[s] = HNSW_NONE;
            }
    }
};

// Metric::distance(query, row) (src/index/hnsw.rs:113-174) as a walk key (see walk_key): one lane
// walks one row in the reference's f64 operation order.
template <int METRIC>
__device__ __forceinline__ unsigned long long row_distance(const double* __restrict__ row, const double* q,
                                                           uint32_t dim)
{
    Acc64<METRIC> A;
    A.init();
    uint32_t i = 0;
    if ((dim & 1u) == 0u) {  // rows are 16-byte aligned: two values per load, 8 loads in flight
        const double2* r2 = reinterpret_cast<const double2*>(row);
        for (; i + 16 <= dim; i += 16) {
            double2 x[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x[t] = r2[i / 2 + t];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                A.step(x[t].x, q[i + 2 * t]);
                A.step(x[t].y, q[i + 2 * t + 1]);
            }
        }
    }
    for (; i < dim; ++i) A.step(row[i], q[i]);
    return walk_key(hnsw_scaled<METRIC>(A));
}

struct LayerView {
    const uint32_t* nbr;
    uint32_t cnt;
};

__device__ __forceinline__ LayerView layer_of(const HnswGraphView& g, uint32_t node, int layer)
{
    LayerView lv;
    if (layer == 0) {
        lv.nbr = g.nbr0 + (size_t)node * g.m0;
        lv.cnt = g.cnt0[node];
    } else {
        const uint32_t slot = g.upper_off[node] + (uint32_t)(layer - 1);
        lv.nbr = g.nbrU + (size_t)slot * g.m;
        lv.cnt = g.cntU[slot];
    }
    return lv;
}

// Visited set of ONE walk (one wave): N / 8 bytes of bitmap + a log of the nodes set, both in the walk scratch the
// launch borrowed (hnsw_index.cpp: WalkScratch).  Nothing in it is shared with another walk, so any number of
// walk kernels run at the same time, and it costs 1/32 of a u32 stamp per node.  All accesses to the bitmap are
// L2 atomics (or / and), so their order is the wave's program order whichever lane issues them.
struct Visited {
    uint32_t* bits;
    uint32_t* log;
    uint32_t words, log_cap;
    uint32_t n_log;  // wave-uniform; > log_cap = the log overflowed (clear() then wipes the whole bitmap)

    __device__ __forceinline__ void attach(const HnswGraphView& g, uint32_t slot)
    {
        words = g.vis_words;
        log_cap = g.vis_log_cap;
        bits = g.vis_bits + (size_t)slot * words;
        log = g.vis_log + (size_t)slot * log_cap;
        n_log = 0;
    }
    // every lane may offer one node; true = this call marked it (it had not been visited)
    __device__ __forceinline__ bool mark(uint32_t e, bool act)
    {
        const uint32_t bit = 1u << (e & 31u);
        uint32_t old = bit;
        if (act) old = atomicOr(&bits[e >> 5], bit);
        const bool fresh = act && (old & bit) == 0u;
        const unsigned long long mk = __ballot(fresh);
        if (mk != 0ull) {  // wave-uniform
            const uint32_t idx = n_log + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (fresh && idx < log_cap) log[idx] = e;
            n_log += (uint32_t)__popcll(mk);
        }
        return fresh;
    }
    __device__ __forceinline__ void clear()
    {
        const int lane = lane_id();
        if (n_log <= log_cap) {
            for (uint32_t i = lane; i < n_log; i += 64) {
                // agent-scope load: the entry was stored by another lane of this wave, the line may sit stale in this CU's L1
                const uint32_t e = __hip_atomic_load(&log[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAnd(&bits[e >> 5], 0u);
            }
        } else {
            for (uint32_t w = lane; w < words; w += 64) atomicAnd(&bits[w], 0u);
        }
        n_log = 0;
    }
};

// Beam search of width ef on one layer.  Precondition: L holds the entry points (unexpanded) and
// they are marked in the visited set.
template <int METRIC, int S>
__device__ __forceinline__ void beam_layer(const HnswGraphView& g, const double* q, int layer, Visited& vis,
                                           BeamList<S>& L, int ef, uint32_t* evals = nullptr)
{
    const int lane = lane_id();
    for (;;) {
        const int idx = L.next_unexpanded();
        if (idx < 0) break;
        unsigned long long dc;
        uint32_t c;
        L.get(idx, dc, c);
        L.mark_expanded(idx);
        const LayerView lv = layer_of(g, c, layer);
        uint32_t e = (uint32_t)lane < lv.cnt ? lv.nbr[lane] : HNSW_NONE;
        const bool act = vis.mark(e, e != HNSW_NONE);
        unsigned long long de = ~0ull;
        if (act) de = row_distance<METRIC>(g.master + (size_t)e * g.dim, q, g.dim);
        if (evals) *evals += (uint32_t)__popcll(__ballot(act));  // wave-uniform count of rows walked
        // the current worst only shrinks while we insert: a stale value lets a few extra lanes through,
        // insert() re-checks the rank
        unsigned long long w;
        uint32_t wn;
        L.get(ef - 1, w, wn);
        unsigned long long m = __ballot(act && (wn == HNSW_NONE || de <= w));
        while (m) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            m &= m - 1;
            L.insert(read_lane(de, src), read_lane(e, src), ef);
        }
    }
}

// Navigation distance of the QUERY walk: two lanes share one neighbour's f32 slab row (alternating
// 16-byte chunks, the f32 query in LDS), partial sums joined across the half-waves.  Half the bytes of
// the f64 row and 1/2 of its serial length per lane; the result orders candidates far more finely than
// the reference's own distance does (it truncates to 1e-3), and every node that reaches the final beam
// gets the exact callback value afterwards (k_hnsw_search).
template <int METRIC>
__device__ __forceinline__ unsigned long long row_distance_f32(const HnswGraphView& g, uint32_t node, const float* q32,
                                                               float q_inv, int half)
{
    const f32x4* row = reinterpret_cast<const f32x4*>(g.slab + (size_t)node * g.ld);
    const f32x4* q4 = reinterpret_cast<const f32x4*>(q32);
    const uint32_t ld4 = g.ld / 4;
    float s = 0.f;
    auto step = [&](const f32x4 x, const f32x4 y) {
        if (METRIC == COSINE || METRIC == DOT) {
            s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
        } else if (METRIC == EUCLIDEAN) {
            const float a = x.x - y.x, b = x.y - y.y, c = x.z - y.z, d = x.w - y.w;
            s = fmaf(a, a, s); s = fmaf(b, b, s); s = fmaf(c, c, s); s = fmaf(d, d, s);
        } else {
            s += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
        }
    };
    uint32_t c = (uint32_t)half;
    for (; c + 30 < ld4; c += 32) {  // 16 loads in flight per lane: a 384-float row is 3 round trips, not 6
        f32x4 x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = row[c + 2 * t];
#pragma unroll
        for (int t = 0; t < 16; ++t) step(x[t], q4[c + 2 * t]);
    }
    for (; c + 14 < ld4; c += 16) {
        f32x4 x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = row[c + 2 * t];
#pragma unroll
        for (int t = 0; t < 8; ++t) step(x[t], q4[c + 2 * t]);
    }
    for (; c < ld4; c += 2) step(row[c], q4[c]);
    s += __shfl_xor(s, 32);
    double scaled;
    if (METRIC == COSINE) {
        const float rinv = g.inv_norm[node];
        scaled = (rinv == 0.f || q_inv == 0.f) ? 1000.0 : (1.0 - (double)(s * rinv * q_inv)) * 1000.0;
    } else if (METRIC == EUCLIDEAN) {
        scaled = (double)sqrtf(s) * 1000.0;
    } else if (METRIC == MANHATTAN) {
        scaled = (double)s * 1000.0;
    } else {
        float d = s;
        d = d < -1000.f ? -1000.f : (d > 1000.f ? 1000.f : d);
        scaled = 1000.0 - (double)d;
    }
    return walk_key(scaled);
}

// beam_layer with f32 navigation distances: lanes l and l + 32 serve neighbour l of the expanded node
// (rounds of 32 neighbours when a list is longer).
template <int METRIC, int S>
__device__ __forceinline__ void beam_layer_f32(const HnswGraphView& g, const float* q32, float q_inv, int layer,
                                               Visited& vis, BeamList<S>& L, int ef, uint32_t* evals)
{
    const int lane = lane_id();
    const int half = lane >> 5, nl = lane & 31;
    for (;;) {
        const int idx = L.next_unexpanded();
        if (idx < 0) break;
        unsigned long long dc;
        uint32_t c;
        L.get(idx, dc, c);
        L.mark_expanded(idx);
        const LayerView lv = layer_of(g, c, layer);
        for (uint32_t base = 0; base < lv.cnt; base += 32) {
            uint32_t e = base + (uint32_t)nl < lv.cnt ? lv.nbr[base + nl] : HNSW_NONE;
            bool act = vis.mark(e, e != HNSW_NONE && half == 0);
            act = __shfl((int)act, nl) != 0;  // the upper half-wave follows its partner's visited check
            unsigned long long de = ~0ull;
            if (act) de = row_distance_f32<METRIC>(g, e, q32, q_inv, half);
            *evals += (uint32_t)__popcll(__ballot(act && half == 0));
            unsigned long long w;
            uint32_t wn;
            L.get(ef - 1, w, wn);
            unsigned long long m = __ballot(act && half == 0 && (wn == HNSW_NONE || de <= w));
            while (m) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                m &= m - 1;
                L.insert(read_lane(de, src), read_lane(e, src), ef);
            }
        }
    }
}

// Move to the next layer: keep the entries as entry points, fresh visited set (only they are marked).
template <int S>
__device__ __forceinline__ void next_layer(BeamList<S>& L, Visited& vis)
{
    vis.clear();
    L.reopen();
#pragma unroll
    for (int s = 0; s < S; ++s) (void)vis.mark(L.v[s] & ~EXPANDED, L.v[s] != HNSW_NONE);
}

// convert_distance_to_similarity(d_u64 as f64 / 1000.0, metric) (src/index/hnsw.rs:51-75, :478-479)
template <int METRIC>
__device__ __forceinline__ double hnsw_score_dev(unsigned long long d_u64)
{
    const double distance = (double)d_u64 / 1000.0;
    if (METRIC == EUCLIDEAN || METRIC == MANHATTAN) return 1.0 / (1.0 + distance);
    if (METRIC == COSINE) return 1.0 - distance / 1000.0;
    double v = (1000.0 - distance) / 1000.0;
    if (v < 0.0) v = 0.0;
    if (v > 1.0) v = 1.0;
    return v;
}

template <int METRIC, int S>
__global__ __launch_bounds__(256) void k_hnsw_search(HnswGraphView g, const double* __restrict__ queries, uint32_t nq,
                                                     uint32_t ef, uint32_t entry, int max_level, uint32_t max_candidates,
                                                     uint32_t refill, uint32_t k_stride, unsigned long long* __restrict__ out_ids,
                                                     double* __restrict__ out_scores, unsigned long long* __restrict__ out_n,
                                                     unsigned long long* __restrict__ stat_evals)
{
    // per wave: [ld] f32 query (zero padded), [dim] f64 query, then two (key, node) tables of CAP = 64 S entries
    extern __shared__ double q_lds[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * 4 + wave;
    if (slot >= g.n_slots) return;
    const size_t q_words = ((size_t)g.ld / 2 + g.dim + 1) & ~(size_t)1;  // in doubles, 16-byte granules
    constexpr uint32_t CAP = 64u * (uint32_t)S;                           // entries the beam list holds (ef <= CAP)
    const size_t per_wave = q_words + 3 * (size_t)CAP;                    // + 2 x (8 B key) + 2 x (4 B node) per entry
    double* base = q_lds + (size_t)wave * per_wave;
    float* q32 = reinterpret_cast<float*>(base);
    double* q = base + g.ld / 2;
    unsigned long long* t_key = reinterpret_cast<unsigned long long*>(base + q_words);  // [2][CAP]
    uint32_t* t_node = reinterpret_cast<uint32_t*>(t_key + 2 * CAP);                     // [2][CAP]
    Visited vis;
    vis.attach(g, slot);

    for (uint32_t qi = slot; qi < nq; qi += gridDim.x * 4) {
        float qq = 0.f;
        for (uint32_t i = lane; i < g.ld; i += 64) {
            const double v = i < g.dim ? queries[(size_t)qi * g.dim + i] : 0.0;
            if (i < g.dim) q[i] = v;
            q32[i] = (float)v;
            qq = fmaf((float)v, (float)v, qq);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) qq += __shfl_xor(qq, o);
        const float q_inv = qq > 0.f ? 1.0f / sqrtf(qq) : 0.f;
        __builtin_amdgcn_wave_barrier();
        BeamList<S> L;
        L.init();
        uint32_t evals = 1;  // the entry point
        {
            unsigned long long d0 = row_distance_f32<METRIC>(g, entry, q32, q_inv, lane >> 5);
            d0 = read_lane(d0, 0);
            L.insert(d0, entry, 1);
            (void)vis.mark(entry, lane == 0);
        }
        for (int layer = max_level; layer >= 1; --layer) {
            beam_layer_f32<METRIC, S>(g, q32, q_inv, layer, vis, L, 1, &evals);
            next_layer(L, vis);
        }
        beam_layer_f32<METRIC, S>(g, q32, q_inv, 0, vis, L, (int)ef, &evals);
        vis.clear();  // the set is empty again for this slot's next query (and for the next launch that borrows the scratch)

        // (1) the final beam gets the reference's own callback value: one lane walks one f64 row in index order
        //     (src/index/hnsw.rs:113-174)
        uint32_t n_real = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t j = s * 64 + lane;
            const uint32_t node = L.v[s] == HNSW_NONE ? HNSW_NONE : (L.v[s] & ~EXPANDED);
            const bool real = j < ef && node != HNSW_NONE;
            unsigned long long exact = ~0ull;
            if (real) exact = row_distance<METRIC>(g.master + (size_t)node * g.dim, q, g.dim);
            n_real += (uint32_t)__popcll(__ballot(real));
            t_key[j] = exact;  // j < CAP by construction
            t_node[j] = node;
        }
        evals += n_real;
        __builtin_amdgcn_wave_barrier();
        // (2) the beam in (distance, node) order -- `neighbors` after hnsw.nearest (:454-466), ties of the truncated
        //     u64 broken by the true distance: every entry counts the entries in front of it
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t j = s * 64 + lane;
            const unsigned long long kj = t_key[j];
            const uint32_t nj = t_node[j];
            if (j < ef && nj != HNSW_NONE) {
                uint32_t rank = 0;
                for (uint32_t i = 0; i < ef; ++i) {
                    const unsigned long long ki = t_key[i];
                    const uint32_t ni = t_node[i];
                    rank += (ni != HNSW_NONE && (ki < kj || (ki == kj && ni < nj))) ? 1u : 0u;
                }
                t_key[CAP + rank] = kj;
                t_node[CAP + rank] = nj;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // (3) the closest max_candidates are what hnsw.nearest hands back (`neighbors`, :442-448); tombstoned nodes among
        //     THEM are dropped (:475) -- fewer than k results, like the reference.  refill != 0 (a caller that named its own
        //     ef, vl_index_search_ef): the freed slots are filled from the rest of the beam instead.  Distances -> scores
        //     (:478-479); the score never increases with the distance, so this order IS the stable sort by score (:493)
        const uint32_t n_take = refill ? n_real : (n_real < max_candidates ? n_real : max_candidates);
        uint32_t kept = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t r = s * 64 + lane;
            const bool in = r < n_take;
            const uint32_t node = in ? t_node[CAP + r] : 0u;
            const bool alive = in && g.live[node] != 0;
            const unsigned long long mk = __ballot(alive);
            const uint32_t pos = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (alive && pos < max_candidates) {
                const double scaled = __longlong_as_double((long long)t_key[CAP + r]);
                out_ids[(size_t)qi * k_stride + pos] = g.node_id[node];
                out_scores[(size_t)qi * k_stride + pos] = hnsw_score_dev<METRIC>(rust_as_u64(scaled));
            }
            kept += (uint32_t)__popcll(mk);
        }
        if (lane == 0) {
            out_n[qi] = kept < max_candidates ? kept : max_candidates;
            if (stat_evals) atomicAdd(stat_evals, (unsigned long long)evals);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Build phase A: node p = first + i searches the graph of the nodes < first and fills its own lists.
template <int METRIC, int S>
__global__ __launch_bounds__(256) void k_hnsw_insert_search(HnswGraphView g, uint32_t first, uint32_t n, uint32_t efc,
                                                            uint32_t entry, int max_level, uint32_t flags)
{
    // The whole build works on f32 distances (row_distance_f32: two lanes per row of the f32 slab): walks,
    // neighbour selection and the edge distances kept for the link phase.  Nothing of it reaches a caller --
    // query walks re-evaluate their final beam with the reference's exact f64 callback.
    extern __shared__ double q_lds[];  // per wave: [ld] f32 row of the node, then [ld] f32 row of a candidate
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, nl = lane & 31;
    const uint32_t slot = blockIdx.x * 4 + wave;
    if (slot >= g.n_slots) return;
    float* q32 = reinterpret_cast<float*>(q_lds) + (size_t)wave * 2 * g.ld;
    float* cand32 = q32 + g.ld;
    Visited vis;
    vis.attach(g, slot);
    uint32_t evals = 0;

    for (uint32_t i = slot; i < n; i += gridDim.x * 4) {
        const uint32_t p = first + i;
        const int lp = g.level[p];
        for (uint32_t c = lane; c < g.ld; c += 64) q32[c] = g.slab[(size_t)p * g.ld + c];
        const float q_inv = g.inv_norm[p];
        __builtin_amdgcn_wave_barrier();
        // distance of p to itself in this arithmetic: exact copies of p among the candidates show exactly this key
        const unsigned long long d_self = read_lane(row_distance_f32<METRIC>(g, p, q32, q_inv, half), 0);
        BeamList<S> L;
        L.init();
        {
            unsigned long long d0 = row_distance_f32<METRIC>(g, entry, q32, q_inv, half);
            d0 = read_lane(d0, 0);
            L.insert(d0, entry, 1);
            (void)vis.mark(entry, lane == 0);
        }
        for (int layer = max_level; layer > lp; --layer) {  // greedy descent above the node's own level
            beam_layer_f32<METRIC, S>(g, q32, q_inv, layer, vis, L, 1, &evals);
            next_layer(L, vis);
        }
        for (int layer = lp < max_level ? lp : max_level; layer >= 0; --layer) {
            beam_layer_f32<METRIC, S>(g, q32, q_inv, layer, vis, L, (int)efc, &evals);
            if (layer == 0 && (flags & 4u)) {
                // The walk saw the graph of the nodes < first.  The EARLIER nodes of this batch are not in it yet, so
                // they are offered to the beam directly (their rows are complete; phase B links p into the lists of
                // the ones it picks): node p then chooses among everything a one-by-one insertion would have shown
                // it.  Without this a batch of rows from a region the graph does not cover yet links only to far-away
                // old nodes, gets no incoming edge, and stays invisible to every later walk.
                for (uint32_t base = first; base < p; base += 32) {
                    const uint32_t e = base + (uint32_t)nl < p ? base + (uint32_t)nl : HNSW_NONE;
                    const bool act = e != HNSW_NONE;
                    unsigned long long de = ~0ull;
                    if (act) de = row_distance_f32<METRIC>(g, e, q32, q_inv, half);
                    unsigned long long w;
                    uint32_t wn;
                    L.get((int)efc - 1, w, wn);
                    unsigned long long mm = __ballot(act && half == 0 && (wn == HNSW_NONE || de <= w));
                    while (mm) {
                        const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
                        mm &= mm - 1;
                        L.insert(read_lane(de, src), read_lane(e, src), (int)efc);
                    }
                }
            }
            // neighbour selection (the HNSW diversity heuristic): walk the beam from the closest
            // candidate outwards and keep a candidate only if it is closer to p than to every
            // neighbour kept so far -- this is what gives the graph its long edges.  Lane j holds
            // kept neighbour j; lanes j and j + 32 measure the candidate against it together.
            const uint32_t cap = layer == 0 ? g.m0 : g.m;
            uint32_t* nb;
            unsigned long long* nd;
            uint32_t* cnt;
            if (layer == 0) {
                nb = g.nbr0 + (size_t)p * g.m0;
                nd = g.dist0 + (size_t)p * g.m0;
                cnt = g.cnt0 + p;
            } else {
                const uint32_t us = g.upper_off[p] + (uint32_t)(layer - 1);
                nb = g.nbrU + (size_t)us * g.m;
                nd = g.distU + (size_t)us * g.m;
                cnt = g.cntU + us;
            }
            int total = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) total += __popcll(__ballot(L.v[s] != HNSW_NONE));
            uint32_t selv = HNSW_NONE;
            unsigned long long seld = 0;
            uint32_t nsel = 0;
            unsigned long long kept_m[S];  // which beam entries were kept (bit ci & 63 of word ci >> 6; static indexing only)
#pragma unroll
            for (int s = 0; s < S; ++s) kept_m[s] = 0ull;
            // Exact copies of p (more than M0 of them would otherwise fill every slot and cut the copies off from
            // the rest of the graph) get at most a quarter of the list; the other slots go to distinct rows.
            const uint32_t dup_cap = cap / 4 > 0 ? cap / 4 : 1;
            uint32_t ndup = 0;
            for (int ci = 0; ci < total && nsel < cap; ++ci) {
                unsigned long long dc;
                uint32_t cv;
                L.get(ci, dc, cv);
                cv &= ~EXPANDED;
                const bool is_dup = dc == d_self;
                if (is_dup && ndup >= dup_cap) continue;  // wave-uniform
                bool bad = false;
                if (nsel > 0 && (flags & 1u) && !is_dup) {
                    for (uint32_t c = lane; c < g.ld; c += 64) cand32[c] = g.slab[(size_t)cv * g.ld + c];
                    const float c_inv = g.inv_norm[cv];
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t base = 0; base < nsel; base += 32) {  // kept neighbours in rounds of 32 lane pairs
                        const uint32_t kj = base + (uint32_t)nl;
                        const uint32_t kn = (uint32_t)__shfl((int)selv, (int)(kj & 63u));
                        // kept neighbours that are copies of p itself (kd == d_self) say nothing about direction:
                        // every candidate is exactly as far from them as from p, so they take no part in the test
                        const unsigned long long kd =
                            ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(seld >> 32), (int)(kj & 63u)) << 32) |
                            (uint32_t)__shfl((int)(uint32_t)seld, (int)(kj & 63u));
                        if (kj < nsel && kd != d_self) bad = bad || row_distance_f32<METRIC>(g, kn, cand32, c_inv, half) < dc;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (__ballot(bad) == 0ull) {
                    if ((uint32_t)lane == nsel) {
                        selv = cv;
                        seld = dc;
                    }
                    ++nsel;
                    ndup += is_dup ? 1u : 0u;
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        if ((ci >> 6) == s) kept_m[s] |= 1ull << (ci & 63);
                }
            }
            if (flags & 2u) {  // back-fill with the closest pruned candidates: keeps the degree at cap
                for (int ci = 0; ci < total && nsel < cap; ++ci) {
                    bool kept = false;
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        if ((ci >> 6) == s) kept = (kept_m[s] >> (ci & 63)) & 1ull;
                    if (kept) continue;
                    unsigned long long dc;
                    uint32_t cv;
                    L.get(ci, dc, cv);
                    if (dc == d_self) {  // copies of p stay capped in the back-fill too
                        if (ndup >= dup_cap) continue;
                        ++ndup;
                    }
                    if ((uint32_t)lane == nsel) {
                        selv = cv & ~EXPANDED;
                        seld = dc;
                    }
                    ++nsel;
                }
            }
            if ((uint32_t)lane < nsel) {
                nb[lane] = selv;
                nd[lane] = seld;
                if (layer == 0) atomicAdd(&g.indeg0[selv], 1u);  // p -> selv is an incoming edge of selv
            }
            if (lane == 0) *cnt = nsel;
            if (layer > 0) next_layer(L, vis);
        }
        vis.clear();
    }
}

// Build phase B: add p to the lists of the neighbours it chose (replace the farthest if full and
// p is closer).  One wave per new node; a neighbour's list is updated under its spin lock with
// agent-scope fences (other XCDs' L2s are not coherent: MI355X_MICROARCH.md).
__global__ __launch_bounds__(256) void k_hnsw_insert_link(HnswGraphView g, uint32_t first, uint32_t n, int max_level)
{
    const int lane = lane_id();
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t i = w; i < n; i += gridDim.x * 4) {
        const uint32_t p = first + i;
        const int lp = g.level[p];
        const int top = lp < max_level ? lp : max_level;
        for (int layer = 0; layer <= top; ++layer) {
            const uint32_t cap = layer == 0 ? g.m0 : g.m;
            const uint32_t* pnb;
            const unsigned long long* pnd;
            uint32_t pcnt;
            if (layer == 0) {
                pnb = g.nbr0 + (size_t)p * g.m0;
                pnd = g.dist0 + (size_t)p * g.m0;
                pcnt = g.cnt0[p];
            } else {
                const uint32_t us = g.upper_off[p] + (uint32_t)(layer - 1);
                pnb = g.nbrU + (size_t)us * g.m;
                pnd = g.distU + (size_t)us * g.m;
                pcnt = g.cntU[us];
            }
            // returns whether p entered qn's list (wave-uniform)
            auto link_into = [&](uint32_t qn, unsigned long long d, bool force) -> bool {
                bool entered = false;
                if (lane == 0) {
                    while (atomicCAS(&g.lock[qn], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(4);
                }
                __threadfence();  // acquire: see the previous holder's list
                uint32_t* qb;
                unsigned long long* qd;
                uint32_t* qc;
                if (layer == 0) {
                    qb = g.nbr0 + (size_t)qn * g.m0;
                    qd = g.dist0 + (size_t)qn * g.m0;
                    qc = g.cnt0 + qn;
                } else {
                    const uint32_t us = g.upper_off[qn] + (uint32_t)(layer - 1);
                    qb = g.nbrU + (size_t)us * g.m;
                    qd = g.distU + (size_t)us * g.m;
                    qc = g.cntU + us;
                }
                const uint32_t cq = __hip_atomic_load(qc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cq < cap) {
                    if (lane == 0) {
                        qb[cq] = p;
                        qd[cq] = d;
                        __hip_atomic_store(qc, cq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (layer == 0) atomicAdd(&g.indeg0[p], 1u);
                    }
                    entered = true;
                } else {
                    // farthest entry by (dist, node) -- on layer 0 among the entries that have another incoming edge:
                    // evicting a node's LAST incoming edge would make it unreachable for good
                    unsigned long long md = (uint32_t)lane < cap ? __hip_atomic_load(qd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    uint32_t mv = (uint32_t)lane < cap ? __hip_atomic_load(qb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    int ml = lane;
                    if ((uint32_t)lane >= cap) ml = -1;
                    if (layer == 0 && ml >= 0 && __hip_atomic_load(&g.indeg0[mv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 1u) ml = -1;
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) {
                        const unsigned long long od = __shfl_xor(md, o);
                        const uint32_t ov = __shfl_xor(mv, o);
                        const int ol = __shfl_xor(ml, o);
                        const bool take = ol >= 0 && (ml < 0 || od > md || (od == md && ov > mv));
                        if (take) {
                            md = od;
                            mv = ov;
                            ml = ol;
                        }
                    }
                    if (ml >= 0 && (force || d < md || (d == md && p < mv))) {  // wave-uniform after the reduction
                        bool evict = true;
                        if (layer == 0) {
                            // another wave may be evicting the same node from another list right now: take the edge
                            // away only if one remains
                            uint32_t old = 2u;
                            if (lane == 0) {
                                old = atomicSub(&g.indeg0[mv], 1u);
                                if (old <= 1u) atomicAdd(&g.indeg0[mv], 1u);
                            }
                            evict = (uint32_t)__builtin_amdgcn_readfirstlane((int)old) > 1u;
                        }
                        if (evict) {
                            if (lane == ml) {
                                qb[lane] = p;
                                qd[lane] = d;
                            }
                            if (layer == 0 && lane == 0) atomicAdd(&g.indeg0[p], 1u);
                            entered = true;
                        }
                    }
                }
                __threadfence();  // release: the list is complete before the lock opens
                if (lane == 0) atomicExch(&g.lock[qn], 0u);
                return entered;
            };
            bool any = false;
            for (uint32_t t = 0; t < pcnt; ++t) any = link_into(pnb[t], pnd[t], false) || any;
            // every chosen neighbour had a full list of closer entries: p would have no incoming edge at all and could
            // never be found (seen on a 40-node index with M0 = 32) -- its nearest neighbour takes it regardless
            if (!any && pcnt > 0) (void)link_into(pnb[0], pnd[0], true);
        }
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

// Dynamic LDS above the 64 KB default needs the per-kernel attribute (gfx950 has 160 KB per CU); raised once
// per kernel to the ceiling this file ever asks for.
constexpr size_t HNSW_LDS_MAX = 160 * 1024;
template <typename K>
hipError_t allow_big_lds(K kernel, size_t lds)
{
    if (lds <= 64 * 1024) return hipSuccess;
    // keyed by the kernel's address (all metric instantiations share one function-pointer TYPE)
    static std::mutex mu;
    static const void* seen[64];
    static int n_seen = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i] == key) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HNSW_LDS_MAX);
    if (e == hipSuccess && n_seen < 64) seen[n_seen++] = key;
    return e;
}

// sorted-list entries per lane for a beam of width ef: the smallest of 1, 2, 4, 8 with 64 * slots >= ef
uint32_t hnsw_beam_slots(uint32_t ef) { return ef <= 64 ? 1u : (ef <= 128 ? 2u : (ef <= 256 ? 4u : 8u)); }

int grid_for(const HnswGraphView& g, uint32_t work)
{
    uint32_t waves = work < g.n_slots ? work : g.n_slots;
    uint32_t blocks = (waves + 3) / 4;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

hipError_t launch_hnsw_search(hipStream_t s, int metric, const HnswGraphView& g, const double* queries, uint32_t nq,
                              uint32_t ef, uint32_t entry, int max_level, uint32_t max_candidates, uint32_t refill,
                              uint32_t k_stride, unsigned long long* out_ids, double* out_scores, unsigned long long* out_n,
                              unsigned long long* stat_evals)
{
    if (nq == 0) return hipSuccess;
    if (ef == 0 || ef > (uint32_t)HNSW_MAX_EF || g.m0 > 64 || g.m > 64 || max_candidates > k_stride) return hipErrorInvalidValue;
    const size_t q_words = ((size_t)g.ld / 2 + g.dim + 1) & ~(size_t)1;
    const uint32_t slots = hnsw_beam_slots(ef);  // list entries per lane: 1, 2, 4 or 8
    const size_t lds = (size_t)4 * (q_words + 3 * (size_t)64 * slots) * sizeof(double);
    if (lds > HNSW_LDS_MAX) return hipErrorInvalidValue;
    const int grid = grid_for(g, nq);
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        auto launch = [&](auto kern) -> hipError_t {
            const hipError_t e = allow_big_lds(kern, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g, queries, nq, ef, entry, max_level, max_candidates,
                               refill, k_stride, out_ids, out_scores, out_n, stat_evals);
            return hipGetLastError();
        };
        switch (slots) {
        case 1: return launch(k_hnsw_search<MM, 1>);
        case 2: return launch(k_hnsw_search<MM, 2>);
        case 4: return launch(k_hnsw_search<MM, 4>);
        default: return launch(k_hnsw_search<MM, 8>);
        }
    });
}

hipError_t launch_hnsw_insert_search(hipStream_t s, int metric, const HnswGraphView& g, uint32_t first, uint32_t n,
                                     uint32_t ef_construction, uint32_t entry, int max_level, uint32_t flags)
{
    if (n == 0) return hipSuccess;
    if (ef_construction == 0 || ef_construction > (uint32_t)HNSW_MAX_EF || g.m0 > 64 || g.m > 64)
        return hipErrorInvalidValue;
    const size_t lds = (size_t)4 * 2 * g.ld * sizeof(float);
    if (lds > HNSW_LDS_MAX) return hipErrorInvalidValue;
    const int grid = grid_for(g, n);
    return dispatch_metric(metric, [&](auto M) -> hipError_t {
        constexpr int MM = decltype(M)::value;
        auto launch = [&](auto kern) -> hipError_t {
            const hipError_t e = allow_big_lds(kern, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g, first, n, ef_construction, entry, max_level, flags);
            return hipGetLastError();
        };
        switch (hnsw_beam_slots(ef_construction)) {
        case 1: return launch(k_hnsw_insert_search<MM, 1>);
        case 2: return launch(k_hnsw_insert_search<MM, 2>);
        case 4: return launch(k_hnsw_insert_search<MM, 4>);
        default: return launch(k_hnsw_insert_search<MM, 8>);
        }
    });
}

hipError_t launch_hnsw_insert_link(hipStream_t s, const HnswGraphView& g, uint32_t first, uint32_t n, int max_level)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (n + 3) / 4;
    const int grid = (int)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(k_hnsw_insert_link, dim3(grid), dim3(256), 0, s, g, first, n, max_level);
    return hipGetLastError();
}

}  // namespace vl
