// CPU check of csrc/filter_plan.hpp: the launch plan of the MFMA batch filter over a sweep of index sizes, chunk counts and
// knob settings.  Invariants: the sample is a prefix of the blocks and holds min(n, 32768) rows at least; the sampling grid
// reports at most MFMA_GROUPS groups, every group owns a block, and never fewer than 128 groups (or one per block) --
// with 64 the threshold would be the smallest group maximum; the stages tile [0, n_blocks) in order; and the defaults
// are the measured ones (1 stage on short scans, 2 on shards of ~1 M rows, 4 from 4 M rows on).
#include <cstdio>
#include <cstdlib>

#include "../../vectorlite_amd/csrc/filter_plan.hpp"

#define CHECK(c)                                                                                              \
    do {                                                                                                      \
        if (!(c)) {                                                                                           \
            std::printf("FAILED %s  (n_rows %llu wg_cap %u stages %d)\n", #c, (unsigned long long)n, cap, kn.stages); \
            return 1;                                                                                         \
        }                                                                                                     \
    } while (0)

int main()
{
    const uint32_t NW = 8, MAXG = 256;
    const uint64_t sizes[] = {8192, 9000, 20000, 65536, 100000, 150000, 1000000, 1250000, 4000000, 4194304, 10000000, 200000000};
    const uint32_t caps[] = {256, 128, 85, 64, 51, 32, 23, 19, 16};
    long plans = 0;
    for (uint64_t n : sizes)
        for (uint32_t cap : caps)
            for (int want = 0; want <= 5; ++want)
                for (uint32_t smin : {0u, 16384u, 32768u}) {
                    vl::FilterKnobs kn;
                    kn.stages = want;
                    kn.sample_min_rows = smin;
                    const vl::FilterPlan p = vl::filter_plan(n, cap, NW, MAXG, kn);
                    const uint32_t nb = (uint32_t)((n + 31) / 32);
                    CHECK(p.n_blocks == nb);
                    CHECK(p.sample_blocks >= 1 && p.sample_blocks <= nb);
                    const uint64_t floor_rows = smin ? smin : 32768u;
                    CHECK((uint64_t)p.sample_blocks * 32 >= (n < floor_rows ? (n / 32) * 32 : floor_rows));
                    CHECK(p.groups == p.gx0 * p.gpw && p.groups <= MAXG && p.gpw >= 1 && p.gpw <= NW);
                    CHECK(p.groups <= p.sample_blocks || p.gpw == NW);          // a group per block at most, or per wave
                    CHECK(p.groups >= 128 || (uint64_t)p.gx0 * NW >= p.sample_blocks);  // >= 128 groups whenever the sample has them
                    CHECK(p.stages >= 1 && p.stages <= 4 && p.st_end[0] == 0 && p.st_end[p.stages] == nb);
                    for (int s = 0; s < p.stages; ++s) CHECK(p.st_end[s] <= p.st_end[s + 1]);
                    if (want == 0) {
                        if ((uint64_t)nb < 128ull * NW * cap) CHECK(p.stages == 1);
                        else CHECK(p.stages == (nb >= 131072u ? 4 : 2));
                    }
                    ++plans;
                }
    // the shapes the round measured
    {
        uint64_t n = 10000000; uint32_t cap = 16; vl::FilterKnobs kn;
        const vl::FilterPlan p = vl::filter_plan(n, cap, NW, MAXG, kn);   // config 5: 16 chunks of 128 queries
        CHECK(p.stages == 4 && p.st_end[1] == 312500u / 16 && p.st_end[2] == 312500u * 3 / 16 && p.st_end[3] == 312500u * 7 / 16);
        CHECK(p.sample_blocks == 312500u / 64 && p.groups == 128);
    }
    {
        uint64_t n = 1250000; uint32_t cap = 23; vl::FilterKnobs kn;
        const vl::FilterPlan p = vl::filter_plan(n, cap, NW, MAXG, kn);   // config 3's shard: 11 chunks of 96 queries
        CHECK(p.stages == 2 && p.st_end[1] == 39063u * 2 / 16 && p.sample_blocks == 39063u / 32 && p.groups == 184);
    }
    std::printf("filter_plan: %ld plans checked\n", plans);
    return 0;
}
