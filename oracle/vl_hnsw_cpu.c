/*
 * CPU ORACLE -- test infrastructure, NOT product code (see vl_oracle.h).
 *
 * vl_hnsw_cpu.c -- the "CPU HNSW" of BASELINE config 4: a plain, single-threaded walk of an EXPORTED graph
 * (vl_index_hnsw_graph_export) with the reference's own distance callbacks, `Metric::distance` -> u64
 * (src/index/hnsw.rs:113-174, restated in vl_oracle.c: vlo_hnsw_distance).
 *
 * What it follows: the reference's search side only calls `hnsw.nearest(&q, ef, &mut searcher, &mut neighbors)`
 * (src/index/hnsw.rs:454-466) of crate hnsw 0.11.0, whose source is NOT in the reference tree (Cargo.lock:1111-1123):
 * PARITY UNPINNED.  This file restates the published HNSW search (Malkov & Yashunin, Alg. 5 + Alg. 2): greedy descent
 * with a beam of 1 through the upper layers, then a beam search of width ef on layer 0 -- candidates in ascending
 * u64 distance, ties first-seen-first (SURVEY 9.5 records the same recollection of the crate).  It is used
 *   - as a second recall reference for the GPU walk (same graph, reference distances, no f32 navigation), and
 *   - as the timed CPU traversal beside the GPU walk (tests/test_gpu_hnsw_1m.py, tools/hnsw_eval.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "vl_oracle.h"

typedef struct {
    uint64_t d;
    uint32_t node;
} cand_t;

/* sorted insert (ascending d; among equal d the earlier-seen entry stays in front), capacity cap; returns 1 if kept */
static int beam_insert(cand_t *beam, uint32_t *n, uint32_t cap, uint64_t d, uint32_t node)
{
    uint32_t lo = 0, hi = *n;
    while (lo < hi) { /* first position whose d is > d: partition_point(|x| x.d <= d) */
        uint32_t mid = (lo + hi) / 2;
        if (beam[mid].d <= d)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo >= cap)
        return 0;
    uint32_t last = *n < cap ? *n : cap - 1;
    memmove(beam + lo + 1, beam + lo, (size_t)(last - lo) * sizeof(cand_t));
    beam[lo].d = d;
    beam[lo].node = node;
    if (*n < cap)
        ++*n;
    return 1;
}

/* One layer: expand the closest unexpanded beam entry until every entry is expanded.  `flag` bit 31 marks expansion. */
static void search_layer(int metric, const double *rows, size_t dim, const double *q, const uint32_t *nbr,
                         const uint32_t *cnt, uint32_t stride, const uint32_t *slot_of /* NULL on layer 0 */, int layer,
                         const uint32_t *upper_off, cand_t *beam, uint8_t *expanded, uint32_t *n_beam, uint32_t ef,
                         uint32_t *stamp, uint32_t epoch, uint64_t *evals)
{
    (void)slot_of;
    for (;;) {
        uint32_t pick = UINT32_MAX;
        for (uint32_t i = 0; i < *n_beam; ++i)
            if (!expanded[i]) {
                pick = i;
                break;
            }
        if (pick == UINT32_MAX)
            return;
        expanded[pick] = 1;
        const uint32_t c = beam[pick].node;
        const uint32_t slot = layer == 0 ? c : upper_off[c] + (uint32_t)(layer - 1);
        const uint32_t *list = nbr + (size_t)slot * stride;
        const uint32_t deg = cnt[slot];
        for (uint32_t j = 0; j < deg; ++j) {
            const uint32_t e = list[j];
            if (stamp[e] == epoch)
                continue;
            stamp[e] = epoch;
            const uint64_t d = vlo_hnsw_distance(metric, q, rows + (size_t)e * dim, dim);
            ++*evals;
            if (*n_beam == ef && d >= beam[ef - 1].d)
                continue; /* not better than the current worst */
            /* keep `expanded` aligned with the beam while inserting */
            uint32_t lo = 0, hi = *n_beam;
            while (lo < hi) {
                uint32_t mid = (lo + hi) / 2;
                if (beam[mid].d <= d)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo >= ef)
                continue;
            uint32_t last = *n_beam < ef ? *n_beam : ef - 1;
            memmove(expanded + lo + 1, expanded + lo, (size_t)(last - lo));
            expanded[lo] = 0;
            uint32_t nb = *n_beam;
            beam_insert(beam, &nb, ef, d, e);
            *n_beam = nb;
        }
    }
}

/*
 * Walk for one query.  Graph arrays as vl_index_hnsw_graph_export writes them.  stamp: caller's [n_nodes] u32 scratch
 * (zeroed once), *epoch is advanced.  Writes up to k (node, u64 distance) pairs in ascending distance, returns how many.
 */
size_t vlo_hnsw_walk(int metric, const double *rows, size_t dim, size_t n_nodes, const uint8_t *level,
                     const uint32_t *upper_off, const uint32_t *cnt0, const uint32_t *nbr0, uint32_t m0,
                     const uint32_t *cntU, const uint32_t *nbrU, uint32_t m, uint32_t entry, int max_level,
                     const double *q, uint32_t ef, uint32_t k, uint32_t *stamp, uint32_t *epoch, uint32_t *out_nodes,
                     uint64_t *out_dist, uint64_t *evals)
{
    (void)level;
    if (n_nodes == 0 || ef == 0 || entry >= n_nodes)
        return 0;
    if (ef > 4096)
        ef = 4096;
    cand_t *beam = (cand_t *)malloc((size_t)(ef + 1) * sizeof(cand_t));
    uint8_t *expanded = (uint8_t *)malloc((size_t)ef + 1);
    if (!beam || !expanded) {
        free(beam);
        free(expanded);
        return 0;
    }
    uint64_t ev = 1;
    uint32_t n_beam = 1;
    beam[0].d = vlo_hnsw_distance(metric, q, rows + (size_t)entry * dim, dim);
    beam[0].node = entry;
    expanded[0] = 0;
    ++*epoch;
    stamp[entry] = *epoch;
    for (int layer = max_level; layer >= 1; --layer) {
        search_layer(metric, rows, dim, q, nbrU, cntU, m, NULL, layer, upper_off, beam, expanded, &n_beam, 1, stamp, *epoch, &ev);
        /* next layer: the survivor is the entry point, fresh visited set */
        ++*epoch;
        stamp[beam[0].node] = *epoch;
        expanded[0] = 0;
        n_beam = 1;
    }
    search_layer(metric, rows, dim, q, nbr0, cnt0, m0, NULL, 0, upper_off, beam, expanded, &n_beam, ef, stamp, *epoch, &ev);
    size_t out = n_beam < k ? n_beam : k;
    for (size_t i = 0; i < out; ++i) {
        out_nodes[i] = beam[i].node;
        out_dist[i] = beam[i].d;
    }
    if (evals)
        *evals += ev;
    free(beam);
    free(expanded);
    return out;
}
