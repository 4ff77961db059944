// hnsw.hpp -- device graph walk behind the reference's HNSWIndex (src/index/hnsw.rs).
//
// What the reference owns and this file reproduces exactly: the four `Metric::distance`
// callbacks -> u64 (src/index/hnsw.rs:113-174), evaluated on the GPU in the reference's f64
// operation order (device_common.hpp: Acc64 + hnsw_quantise) for every node a query returns
// and through vl_index_hnsw_distances; navigation inside the walks uses f32 distances.
// What the reference delegates to crate `hnsw 0.11.0` (source not vendored, Cargo.lock:1111-1123):
// the graph build and walk.  This is OUR OWN traversal (standard HNSW: greedy descent through the
// upper layers, beam search of width ef on layer 0; M = 16, M0 = 32 like src/index/hnsw.rs:95-109).
// Its parity with the crate is UNPINNED; it is judged by recall against exact search.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vl {

constexpr uint32_t HNSW_NONE = 0xFFFFFFFFu;
constexpr int HNSW_MAX_EF = 512;       // beam width ceiling (eight sorted-list entries per lane; 1 / 2 / 4 / 8 by the width asked for)
constexpr int HNSW_MAX_LEVEL = 15;

struct HnswGraphView {
    const double* master;  // [cap, dim] f64 rows (node index = storage position)
    const float* slab;     // [cap, ld] f32 copy of the rows (the flat scan's slab): navigation distances of the query walk
    const float* inv_norm; // [cap] f32 1/|row| (0 for zero rows)
    uint32_t ld;           // slab row stride in floats (dim rounded up to 4)
    uint32_t dim;
    uint32_t m, m0;        // max neighbours per node: upper layers / layer 0
    // layer 0
    uint32_t* nbr0;               // [cap, m0]
    unsigned long long* dist0;    // [cap, m0] u64 distance of each edge (for replace-farthest)
    uint32_t* cnt0;               // [cap]
    // upper layers: node -> first slot; slot s + (layer-1) holds that layer's list
    const uint8_t* level;         // [cap]
    const uint32_t* upper_off;    // [cap]
    uint32_t* nbrU;               // [slots, m]
    unsigned long long* distU;    // [slots, m]
    uint32_t* cntU;               // [slots]
    uint32_t* lock;               // [cap] link-phase spin locks
    uint32_t* indeg0;             // [cap] incoming layer-0 edges of each node: an edge is never evicted when it is the target's last
    // visited sets of the walks of ONE launch (a WalkScratch the launch borrowed; hnsw_index.cpp): per walk slot a
    // bitmap over the nodes and a log of the nodes it set; all zero between walks
    uint32_t* vis_bits;           // [n_slots, vis_words]
    uint32_t* vis_log;            // [n_slots, vis_log_cap]
    uint32_t vis_words;           // ceil(cap / 32)
    uint32_t vis_log_cap;
    uint32_t n_slots;             // walks in flight = waves the launch may use
    uint64_t cap;
    // result post-processing on the device (query walks)
    const unsigned long long* node_id;  // [cap] node -> caller's id
    const uint8_t* live;                // [cap] 0 = tombstoned (src/index/hnsw.rs:405-411)
};

// Query-time walk of nq queries (f64 [nq, dim]) with beam width ef, finished on the device the way
// HNSWIndex::search finishes it on the host (src/index/hnsw.rs:468-495): the beam in (Metric::distance, node)
// order, the closest max_candidates taken and the tombstoned nodes among them dropped (refill != 0: the freed slots are
// filled from the rest of the beam -- for callers that named their own ef), distances converted to scores
// (convert_distance_to_similarity), ordered by score (already is: the score never increases with the distance).
// out_ids / out_scores are [nq, k_stride], out_n is [nq]; stat_evals (optional) accumulates distance evaluations.
hipError_t launch_hnsw_search(hipStream_t s, int metric, const HnswGraphView& g, const double* queries, uint32_t nq,
                              uint32_t ef, uint32_t entry, int max_level, uint32_t max_candidates, uint32_t refill,
                              uint32_t k_stride, unsigned long long* out_ids, double* out_scores, unsigned long long* out_n,
                              unsigned long long* stat_evals);

// Build phase A: the rows [first, first+n) are new nodes; each walks the graph that holds the
// nodes < first (entry/max_level describe it) with beam width ef_construction and writes its own
// neighbour lists.  Phase B links them back into their neighbours' lists.
// flags: bit 0 = diversity heuristic for neighbour selection, bit 1 = back-fill pruned candidates, bit 2 = offer the
// earlier nodes of the same batch to the layer-0 beam (what a one-by-one insertion would have seen).
hipError_t launch_hnsw_insert_search(hipStream_t s, int metric, const HnswGraphView& g, uint32_t first, uint32_t n,
                                     uint32_t ef_construction, uint32_t entry, int max_level, uint32_t flags);
hipError_t launch_hnsw_insert_link(hipStream_t s, const HnswGraphView& g, uint32_t first, uint32_t n, int max_level);

}  // namespace vl
