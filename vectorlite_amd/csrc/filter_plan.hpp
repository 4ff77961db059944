// filter_plan.hpp -- the launch plan of the row-stationary MFMA filter (k_mfma_rows): how many blocks the sampling pass
// covers and on which grid, and where the pass-1 stages end.  Pure host arithmetic (no HIP): tests/native/filter_plan_test.cpp
// checks its invariants on the CPU; mfma_scan.hip fills the knobs from the environment and launches what this returns.
#pragma once

#include <stdint.h>

#include <algorithm>

namespace vl {

struct FilterKnobs {         // 0 = default
    uint32_t sample_div = 0;   // sample = blocks / sample_div            (VL_MFMA_SAMPLE_DIV)
    uint32_t sample_min_rows = 0;  // ... but at least this many rows     (VL_MFMA_SAMPLE_MIN)
    int stages = 0;            // pass-1 stages wanted, 1..4               (VL_MFMA_STAGES)
    uint32_t stage_end[3] = {0, 0, 0};  // stage ends in sixteenths       (VL_MFMA_STAGE1..3)
};

struct FilterPlan {
    uint32_t n_blocks;        // 32-row blocks of the index
    uint32_t sample_blocks;   // blocks [0, sample_blocks) are the sample
    uint32_t gx0;             // workgroups per query chunk of the sampling launch
    uint32_t gpw;             // groups each of them reports (its waves in equal shares): 1, 2, 4 or 8
    uint32_t groups;          // = gx0 * gpw <= max_groups: what k_thresholds selects the 64th largest maximum of
    int stages;               // pass-1 launches
    uint32_t st_end[5];       // stage s covers blocks [st_end[s], st_end[s + 1])
};

// n_rows rows; wg_cap = co-resident workgroups per query chunk (CUs / chunks); n_waves per workgroup; max_groups = MFMA_GROUPS
inline FilterPlan filter_plan(uint64_t n_rows, uint32_t wg_cap, uint32_t n_waves, uint32_t max_groups, const FilterKnobs& kn)
{
    FilterPlan p{};
    const uint32_t n_blocks = (uint32_t)((n_rows + 31) / 32);
    p.n_blocks = n_blocks;
    wg_cap = std::max<uint32_t>(wg_cap, 1u);
    // the sample: 1/32 of the blocks; 1/64 on long scans (>= 4 M rows), where the floor of 65536 rows is far away and the
    // looser thresholds only add a few hundred candidates per query to the first stage
    const uint32_t div = kn.sample_div ? kn.sample_div : (n_blocks >= 131072u ? 64u : 32u);
    // floor of the sample: 32768 rows (round 4; 65536 before).  Config 3's shard, one process, knob combinations interleaved
    // (tools/c3_knob_sweep.py, profiles/r04_c3_knob_sweep.jsonl): 1.624 ms per batch against 1.659 with 65536 -- the sampling
    // pass costs what a pass-1 scan of the same rows costs and collects nothing, and the looser thresholds only add ~100
    // candidates per query to the first stage; 16384 is slower again (1.639), as are three stages or a first stage of 1/16 or 3/16.
    const uint32_t min_rows = kn.sample_min_rows >= 2048u ? kn.sample_min_rows : 32768u;
    p.sample_blocks = std::max<uint32_t>(n_blocks / div, std::min<uint32_t>(n_blocks, min_rows / 32u));
    // sampling launch: the grid of pass 1 (every CU busy, one round), each workgroup reporting gpw groups so that
    // k_thresholds sees up to max_groups of them
    const uint32_t wgs_for_sample = (p.sample_blocks + n_waves - 1) / n_waves;
    uint32_t gx0 = std::max<uint32_t>(1u, std::min<uint32_t>({wgs_for_sample, wg_cap, max_groups}));
    // never fewer than 128 groups when the sample has the blocks for them (the 64th largest of 64 group maxima is the
    // smallest of them: a threshold so loose that every candidate buffer overflows); more workgroups than are co-resident
    // just queue up
    if (gx0 * n_waves < 128u) gx0 = std::max<uint32_t>(gx0, std::min<uint32_t>(128u / n_waves, wgs_for_sample));
    uint32_t gpw = 1;
    while (gpw < n_waves && gx0 * gpw * 2 <= max_groups) gpw *= 2;
    if (p.sample_blocks < gx0 * gpw) {  // tiny sample: one block per group at most
        gpw = n_waves;
        gx0 = std::max<uint32_t>(1u, wgs_for_sample);
    }
    p.gx0 = gx0;
    p.gpw = gpw;
    p.groups = gx0 * gpw;
    // Pass 1 in stages of growing size; between stages every query's threshold is tightened to its 64th best candidate so
    // far (k_refine_thresholds).  The candidate code is not free (ballots, ring writes: the wave leaves the MFMA stream for
    // hundreds of cycles), and how often a wave enters it is set by the threshold: with the sampled one about
    // 64 x (stage rows / sample rows) rows per query pass, after a refine 64 x (stage rows / rows scanned so far).  So the
    // first stage is short and the stages grow geometrically -- 1/16, 3/16, 7/16 of the blocks on long scans (>= 4 M rows),
    // where a refine launch (~50 us with its gaps) is noise; a shard of ~1 M rows (config 3) is fastest with two
    // (measured: 2.015 -> 1.953 ms at 1.25 M x 768, 1024 queries); short scans keep one.
    for (uint32_t& e : p.st_end) e = n_blocks;
    p.st_end[0] = 0;
    p.stages = 1;
    const int want = kn.stages ? kn.stages : (n_blocks >= 131072u ? 4 : 2);
    if (want >= 2 && n_blocks >= 128u * n_waves * wg_cap) {
        p.stages = std::min(want, 4);
        static const uint32_t dflt[5][3] = {{0, 0, 0}, {0, 0, 0}, {2, 0, 0}, {3, 7, 0}, {1, 3, 7}};  // sixteenths
        uint32_t prev = 0;
        for (int st = 1; st < p.stages; ++st) {
            uint32_t f = kn.stage_end[st - 1] ? kn.stage_end[st - 1] : dflt[p.stages][st - 1];
            f = std::min<uint32_t>(std::max<uint32_t>(f, prev), 16u);  // stage ends never go backwards
            p.st_end[st] = (uint32_t)((uint64_t)n_blocks * f / 16);
            prev = f;
        }
        p.st_end[p.stages] = n_blocks;
    }
    return p;
}

}  // namespace vl
