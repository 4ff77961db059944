// xcd_map.hpp -- which (row-block lane x, query chunk y) a workgroup of k_mfma_rows works on.
//
// MI355X dispatches the workgroups of a grid to its 8 XCDs round-robin by linear id, and each XCD has its own L2.  All
// query chunks y of one lane x stream the SAME row blocks, so they belong on ONE XCD: its L2 then fetches a block from HBM
// once and serves the other chunks.  With id = x + nx * y that only happens when 8 divides nx.  Here the workgroups of
// an XCD, in dispatch order, take a contiguous range of the (x, y) pairs sorted by x: whatever the grid, at most 7 lanes
// straddle two XCDs.  Plain C++ (host and device): tests/native/xcd_map_test.cpp checks it on the CPU.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define VL_XCD_HD __host__ __device__
#else
#define VL_XCD_HD
#endif

namespace vl {

constexpr uint32_t XCDS = 8;

// lin = blockIdx.x + nx * blockIdx.y of a grid (nx, ny); returns the pair through x, y (a bijection onto [0,nx) x [0,ny))
VL_XCD_HD inline void xcd_pair(uint32_t lin, uint32_t nx, uint32_t ny, uint32_t& x, uint32_t& y)
{
    const uint32_t total = nx * ny, xcd = lin % XCDS;
    uint32_t p = lin / XCDS;                                            // index among this XCD's workgroups
    for (uint32_t c = 0; c < xcd; ++c) p += (total - c + XCDS - 1) / XCDS;  // + the workgroups of the XCDs before it
    x = p / ny;
    y = p - x * ny;
}

}  // namespace vl
