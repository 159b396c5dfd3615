// scan_mark.h -- device helpers shared by the kernels that look at every frame byte (bright_cells_kernel in
// blob_filter.hip, bayer_gray_scan_kernel in bayer_gray.hip): the excess sum of a cell row and what a hot cell does.
// The bound behind them is derived in blob_filter.hip ("dark-tile early-out").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace mocap {

// base4 = the excess base c in every byte (c * 0x01010101), c8 = 8 * c
__device__ __forceinline__ uint32_t excess2_row(uint32_t lo, uint32_t hi, uint32_t base4, uint32_t c8)
{ // sum over the 8 bytes of |p - c| + p - c = 2 * max(0, p - c)   (|p - c| + |p - 0| = 2 max(0, p - c) + c per byte)
    uint32_t s = __builtin_amdgcn_sad_u8(lo, base4, 0u);
    s = __builtin_amdgcn_sad_u8(lo, 0u, s);
    s = __builtin_amdgcn_sad_u8(hi, base4, s);
    s = __builtin_amdgcn_sad_u8(hi, 0u, s);
    return s - c8;
}

// What a hot cell does: for every filter tile its reach (+ 4 pixels of blur and median) overlaps, that tile's box of
// reachable mask rows and columns is widened.  rc = reach of the cell (x0 | x1 << 16, y0 | y1 << 16; x0 > x1: none), rows: the
// image's tile boxes.
__device__ __forceinline__ void widen_tile_boxes(const BrightArgs& a, const uint2 rc, uint32_t* __restrict__ rows)
{
    const int x0 = (int)(rc.x & 0xffffu), x1 = (int)(rc.x >> 16), y0 = (int)(rc.y & 0xffffu), y1 = (int)(rc.y >> 16);
    if (x0 > x1) return;
    // a window with a set threshold bit is centred within 2 pixels of a pixel that reads a hot cell and spans 2
    // more; the median adds 2 again: exact pixels are needed, and mask bits can be set, within 4 of the reach
    const int xa = x0 - 4 > 0 ? x0 - 4 : 0, xb = x1 + 4 < a.W - 1 ? x1 + 4 : a.W - 1;
    const int ya = y0 - 4 > 0 ? y0 - 4 : 0, yb = y1 + 4 < a.H - 1 ? y1 + 4 : a.H - 1;
    // floor(v / d) = (v * ceil(2^23 / d)) >> 23 for v < 32768 and d >= 8: no integer division in this kernel
    const int ch0 = (int)(((uint32_t)ya * a.rows_magic) >> 23), ch1 = (int)(((uint32_t)yb * a.rows_magic) >> 23);
    const int st0 = (int)(((uint32_t)xa * 34953u) >> 23), st1 = (int)(((uint32_t)xb * 34953u) >> 23);
    for (int ch = ch0; ch <= ch1; ch++)
        for (int st = st0; st <= st1; st++) {
            const int t = ch * a.n_strips + st;
            // a box that already holds the rectangle needs no atomics (boxes only grow inside a launch, so a stale
            // value read here errs on the safe side): keeps a frame that is hot everywhere from serialising on them
            const uint4 cur = *(const uint4*)(rows + 4 * t);
            if (cur.x <= (uint32_t)ya && cur.y >= (uint32_t)yb && cur.z <= (uint32_t)xa && cur.w >= (uint32_t)xb) continue;
            atomicMin(&rows[4 * t], (uint32_t)ya);
            atomicMax(&rows[4 * t + 1], (uint32_t)yb);
            atomicMin(&rows[4 * t + 2], (uint32_t)xa);
            atomicMax(&rows[4 * t + 3], (uint32_t)xb);
        }
}

// A cell whose doubled excess sum `acc` exceeds its threshold widens the boxes of the tiles it can reach -- from inside the
// kernel that looks at the pixels (the fused Bayer pass).  reach / cflags: the tables of the image's undistort slot.
__device__ __forceinline__ void mark_hot_cell(const BrightArgs& a, const uint2* __restrict__ reach, const uint8_t* __restrict__ cflags,
                                              uint32_t* __restrict__ rows, int ci, uint32_t acc)
{
    if ((int)acc > a.hot_corner) { // rare: a few cells per marker
        // flag bits: the cell feeds windows cut by the image border in one axis (1) / in both (2): fewer taps, smaller bound
        const uint2 rc = reach[ci];
        const uint32_t fl = cflags[ci];
        if ((int)acc > ((fl & 2u) ? a.hot_corner : (fl & 1u) ? a.hot_edge : a.hot)) widen_tile_boxes(a, rc, rows);
    }
}

// The same decision in two steps (the streaming scan, bright_cells_kernel): the scan only writes how many of the three
// thresholds (hot_corner <= hot_edge <= hot) the cell's sum exceeds -- two bits per cell, no table lookups and no atomics
// behind its loads -- and mark_tiles_kernel turns the hot map into tile boxes afterwards.
__device__ __forceinline__ uint32_t hot_level(const BrightArgs& a, uint32_t acc)
{
    return (uint32_t)((int)acc > a.hot_corner) + (uint32_t)((int)acc > a.hot_edge) + (uint32_t)((int)acc > a.hot);
}
__device__ __forceinline__ bool level_is_hot(uint32_t level, uint32_t fl)
{ // level >= 1 <=> acc > hot_corner, >= 2 <=> acc > hot_edge, 3 <=> acc > hot
    return level >= ((fl & 2u) ? 1u : (fl & 1u) ? 2u : 3u);
}

} // namespace mocap
