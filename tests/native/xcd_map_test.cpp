// CPU check of csrc/xcd_map.hpp: for every grid shape k_mfma_rows can be launched with, the workgroup -> (x, y) map is
// a bijection (every row-block lane x and query chunk y is worked on exactly once), and the workgroups that share a
// lane x sit on one XCD except for at most 7 lanes per grid (grids of at least 8 lanes; a lane never spans more than 2).
#include <cstdio>
#include <set>
#include <vector>

#include "../../vectorlite_amd/csrc/xcd_map.hpp"

int main()
{
    long grids = 0;
    for (uint32_t ny = 1; ny <= 17; ++ny) {
        for (uint32_t nx = 1; nx <= 300; ++nx) {
            const uint32_t total = nx * ny;
            std::vector<int> seen(total, 0);
            std::vector<std::set<uint32_t>> xcds_of_x(nx);
            for (uint32_t lin = 0; lin < total; ++lin) {
                uint32_t x = ~0u, y = ~0u;
                vl::xcd_pair(lin, nx, ny, x, y);
                if (x >= nx || y >= ny) {
                    std::printf("grid %u x %u: workgroup %u -> (%u, %u) out of range\n", nx, ny, lin, x, y);
                    return 1;
                }
                if (seen[x * ny + y]++) {
                    std::printf("grid %u x %u: pair (%u, %u) taken twice\n", nx, ny, x, y);
                    return 1;
                }
                xcds_of_x[x].insert(lin % vl::XCDS);
            }
            uint32_t straddling = 0;
            // (with fewer than 8 lanes an XCD holds less than one lane's chunks: nothing to keep together)
            for (uint32_t x = 0; nx >= vl::XCDS && x < nx; ++x) {
                if (xcds_of_x[x].size() > 2) {
                    std::printf("grid %u x %u: lane %u spread over %zu XCDs\n", nx, ny, x, xcds_of_x[x].size());
                    return 1;
                }
                straddling += xcds_of_x[x].size() == 2;
            }
            if (straddling > vl::XCDS - 1) {
                std::printf("grid %u x %u: %u lanes straddle two XCDs\n", nx, ny, straddling);
                return 1;
            }
            ++grids;
        }
    }
    std::printf("xcd_pair: %ld grids, every map a bijection, at most 7 straddling lanes each\n", grids);
    return 0;
}
