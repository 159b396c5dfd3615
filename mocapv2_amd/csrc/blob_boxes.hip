// blob_boxes.hip -- the sparse half of the filter stage: undistort -> 5x5 in-bounds box sum -> threshold -> 5x5 majority,
// run only where the streaming scan (bright_cells_kernel, blob_filter.hip) could not prove the mask to be zero.
//
// Replaces, for the boxes of one batch of camera images,
//   cv.undistort (reference lib/ImageOperations.py:38) -> fast_cuda_blur (lib/CudaOperations.py:5-41)
//   -> cv.threshold (lib/ImageOperations.py:29) -> cv.medianBlur (lib/ImageOperations.py:30).
//
// Two kernels per batch, behind the scan:
//   settle_tiles_kernel  one thread per (tile, image): turns the box the scan left on the tile (reachable mask rows and
//                        columns) into work items of a bounded size in ONE global list -- or, a wide box, into an entry of the
//                        wide-tile list that the sliding row pipeline of blob_filter.hip works through --, clears what the
//                        previous batch left in the mask where this batch will not write, empties the next batch's boxes;
//   box_filter_kernel    a fixed grid of single-wave workgroups consumes the list.  One wave owns one item and keeps
//                        everything in LDS: the source pixels its undistortion reads (staged with coalesced row loads,
//                        zero border = cv::remap's BORDER_CONSTANT), the horizontal 5-sums of the undistorted patch, the
//                        packed window counts of the thresholded rows.  Lanes are dealt (row, 4-pixel quad) pairs
//                        compactly -- a 72-pixel-wide box keeps 60 of 64 lanes busy with three rows per instruction --
//                        and horizontal neighbours come from DPP wave shifts.  The undistort table is one 4-byte word per
//                        pixel (11-bit displacements, 5-bit fractions), read with one 16-byte load per quad.
// No workgroup barrier anywhere: a wave never waits for another one; items are taken from the list dynamically (one atomic
// per item on one of 8 per-XCD run heads), so waves with expensive items simply take fewer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "kernels.h"

namespace mocap {

namespace {

__device__ __forceinline__ uint32_t from_prev(uint32_t v)
{ // lane L receives lane L-1's value, lane 0 receives 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t from_next(uint32_t v)
{ // lane L receives lane L+1's value, lane 63 receives 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t sel, uint32_t acc) { return __builtin_amdgcn_udot4(a, sel, acc, false); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int taps5(int v, int n)
{ // number of in-image taps of a 5-wide window centred on v
    int lo = v - 2 < 0 ? 0 : v - 2, hi = v + 2 > n - 1 ? n - 1 : v + 2;
    return hi - lo + 1;
}
// floor(n / d) for 0 <= n < 4096, 1 <= d <= 4096 (float reciprocal, +0.5 keeps every quotient away from an integer)
__device__ __forceinline__ int small_div(int n, float rcp_d) { return (int)(((float)n + 0.5f) * rcp_d); }

} // namespace

// ---- set-up: per 8x8 output cell, the box of source pixels its undistortion reads ------------------------------------
__global__ void srcbox_kernel(const uint32_t* __restrict__ map4, ushort4* __restrict__ srcbox, int H, int W)
{
    const int ncx = (W + 7) >> 3, ncy = (H + 7) >> 3;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncx * ncy) return;
    const int cy = c / ncx, cx = c - cy * ncx;
    int x0 = 0x7fff, x1 = -0x8000, y0 = 0x7fff, y1 = -0x8000;
    for (int y = 8 * cy; y < 8 * cy + 8 && y < H; y++)
        for (int x = 8 * cx; x < 8 * cx + 8 && x < W; x++) {
            const uint32_t w = map4[(size_t)y * W + x];
            const int sx = x + ((int)(w << 21) >> 21), sy = y + ((int)(w << 10) >> 21);
            x0 = imin(x0, sx); x1 = imax(x1, sx + 1); y0 = imin(y0, sy); y1 = imax(y1, sy + 1);
        }
    // tap coordinates lie in [-2, W + 1] x [-2, H + 1]; stored + 2
    srcbox[c] = make_ushort4((unsigned short)(x0 + 2), (unsigned short)(x1 + 2), (unsigned short)(y0 + 2), (unsigned short)(y1 + 2));
}

void launch_srcbox(const uint32_t* map4, ushort4* srcbox, int H, int W, hipStream_t s)
{
    const int n = ((W + 7) >> 3) * ((H + 7) >> 3);
    hipLaunchKernelGGL(srcbox_kernel, dim3((n + 63) / 64), dim3(64), 0, s, map4, srcbox, H, W);
}

// ---- settle: boxes -> work items ----------------------------------------------------------------------------------------
// How a tile's output region (nb mask bytes wide, h rows) is cut into items: nx x ny parts (nx * ny <= BOX_MAX_PARTS) such that a
// part's patch -- (bytes * 2 + 2) quads x (rows + 8) -- fits BOX_HCAP quad-rows, minimising the wave instructions
// ("trips": rows per instruction = 64 / quads) plus a fixed cost per item.
__device__ __forceinline__ void choose_split(int nb, int h, int& nx, int& ny)
{
    int best = 0x7fffffff;
    nx = 1; ny = 1;
    for (int cx = 1; cx <= 4; cx++) {
        const int pb = (nb + cx - 1) / cx, Q = 2 * pb + 2;
        if (Q > 64) continue;
        const int rpw = 64 / Q, mr = BOX_HCAP / Q - 8;
        if (mr < 1) continue;
        const int cy = (h + mr - 1) / mr;
        if (cx * cy > BOX_MAX_PARTS) continue;
        const int ph = (h + cy - 1) / cy;
        const int cost = cx * cy * ((ph + 8 + rpw - 1) / rpw + 8);
        if (cost < best) { best = cost; nx = cx; ny = cy; }
    }
}

__global__ __launch_bounds__(256) void settle_tiles_kernel(BoxArgs a)
{
    const int lane = threadIdx.x & 63;
    if (a.zero8 && blockIdx.x == 0 && threadIdx.x < 8) a.zero8[threadIdx.x] = 0u; // the contour stage's walk counters
    const int tiles = a.n_chunks * a.n_strips;
    // the grid also covers the images beyond this batch whose boxes an earlier, larger batch may have left in the array the next
    // batch's scan will widen (n_clear): they are emptied, nothing else happens for them
    const int n_all = a.n_clear > a.n_images ? a.n_clear : a.n_images;
    const int steps_all = (n_all + a.cam_mod - 1) / a.cam_mod;
    const long long total = (long long)tiles * a.cam_mod * steps_all;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    // thread -> (tile, slot, time step), time fastest: the items of one tile of one camera lie together in the list, so
    // the waves that work at the same time share undistort-table lines (matters when every tile is filtered)
    int image = 0, chunk = 0, strip = 0;
    bool valid = i < total;
    if (valid) {
        const int t = (int)(i % steps_all);
        const long long rest = i / steps_all;
        const int slot = (int)(rest % a.cam_mod), tile = (int)(rest / a.cam_mod);
        image = t * a.cam_mod + slot;
        chunk = tile / a.n_strips; strip = tile - chunk * a.n_strips;
        valid = chunk * a.rows_per_chunk < a.H;
    }
    const size_t idx = ((size_t)image * a.n_chunks + chunk) * a.n_strips + strip;
    if (valid && !a.dense && image < n_all) { // the buffer the NEXT batch's scan will widen -- read by the previous batch's settle -- is emptied
        const uint4 nxt = *(const uint4*)(a.tile_rows_next + 4 * idx);
        if (nxt.x <= nxt.y) *(uint4*)(a.tile_rows_next + 4 * idx) = make_uint4(0xffffffffu, 0u, 0xffffffffu, 0u);
    }
    valid = valid && image < a.n_images;
    const int tile_r0 = chunk * a.rows_per_chunk, tile_r1 = imin(tile_r0 + a.rows_per_chunk, a.H) - 1;
    const int tile_x0 = 240 * strip, tile_x1 = imin(tile_x0 + 239, a.W - 1);
    int bx0 = 0, bx1 = a.W - 1, by0 = 0, by1 = a.H - 1; // the scan's box: where exact pixels are needed / bits can be set
    bool marked = valid;
    uint4 raw = make_uint4(0xffffffffu, 0u, 0xffffffffu, 0u);
    if (valid && !a.dense) {
        // the scan's boxes of this batch are only read here (neighbouring tiles look at each other's)
        raw = *(const uint4*)(a.tile_rows + 4 * idx);
        marked = raw.x <= raw.y;
        if (marked) { by0 = (int)raw.x; by1 = (int)raw.y; bx0 = (int)raw.z; bx1 = (int)raw.w; }
    }
    // output region of the tile: the box clipped to the tile, whole mask bytes
    int ox0 = imax(bx0, tile_x0) & ~7, ox1 = imin(imin(bx1, tile_x1) | 7, tile_x1);
    int oy0 = imax(by0, tile_r0), oy1 = imin(by1, tile_r1);
    if (marked && (ox0 > ox1 || oy0 > oy1)) marked = false;
    if (!marked) { ox0 = 1; ox1 = 0; oy0 = 1; oy1 = 0; }
    // how the tile is filtered: cut into items for the box kernel, or -- a wide box -- as the tile's rows through the
    // sliding row pipeline, which then writes the whole width of the tile
    int nx = 0, ny = 0, nb = 0, h = 0;
    bool wide = false;
    // The marked tiles of an aligned 2 x 2 block of tiles whose boxes together are small (one marker on a tile border,
    // nothing else nearby) are filtered ONCE, as the union of their boxes clipped to the block, by the item the first of them
    // emits -- instead of one clipped piece per tile, each with its own halo rows, staging and set-up.  Every tile of the
    // block takes the same decision from the same read-only data; each still records its own clipped region.
    int ix0 = ox0, ix1 = ox1, iy0 = oy0, iy1 = oy1; // region the emitted items cover
    int ux0 = bx0, ux1 = bx1, uy0 = by0, uy1 = by1; // box the items take their exact pixels from
    bool emits = marked;
    if (marked && !a.dense && a.cluster) {
        const int c0 = chunk & ~1, s0 = strip & ~1;
        int n_marked = 0, first = -1, X0 = 0x7fffffff, X1 = -1, Y0 = 0x7fffffff, Y1 = -1;
        for (int k = 0; k < 4; k++) {
            const int cy = c0 + (k >> 1), sx = s0 + (k & 1);
            if (cy >= a.n_chunks || sx >= a.n_strips || cy * a.rows_per_chunk >= a.H) continue;
            const uint4 o = *(const uint4*)(a.tile_rows + 4 * (((size_t)image * a.n_chunks + cy) * a.n_strips + sx));
            if (o.x > o.y) continue;
            if (first < 0) first = k;
            n_marked++;
            Y0 = imin(Y0, (int)o.x); Y1 = imax(Y1, (int)o.y); X0 = imin(X0, (int)o.z); X1 = imax(X1, (int)o.w);
        }
        if (n_marked >= 2) {
            // the union, clipped to the block and the image, in whole mask bytes
            const int bx_lo = 240 * s0, bx_hi = imin(240 * (s0 + 2) - 1, a.W - 1), by_lo = c0 * a.rows_per_chunk, by_hi = imin((c0 + 2) * a.rows_per_chunk, a.H) - 1;
            const int ex0 = imax(X0, bx_lo) & ~7, ex1 = imin(X1 | 7, bx_hi), ey0 = imax(Y0, by_lo), ey1 = imin(Y1, by_hi);
            if (((ex1 - ex0 + 8) >> 3) <= 13 && ey1 - ey0 + 1 <= 100) {
                emits = first == ((chunk - c0) * 2 + (strip - s0));
                ix0 = ex0; ix1 = ex1; iy0 = ey0; iy1 = ey1;
                ux0 = X0; ux1 = X1; uy0 = Y0; uy1 = Y1;
            }
        }
    }
    if (marked) {
        nb = (ix1 - ix0 + 8) >> 3; h = iy1 - iy0 + 1;
        choose_split(nb, h, nx, ny);
        const int slot_ = image % a.cam_mod;
        const int limit = ((a.remap_bits >> slot_) & 1ull) ? a.wide_quads_remap : a.wide_quads_identity;
        wide = emits && ix0 == ox0 && ix1 == ox1 && iy0 == oy0 && iy1 == oy1 && a.wide_tiles != nullptr && 2 * nb + 2 * nx >= limit;
        if (wide) { ox0 = tile_x0; ox1 = tile_x1; ix0 = ox0; ix1 = ox1; }
    }
    const uint32_t nout_x = (uint32_t)ox0 | ((uint32_t)ox1 << 16), nout_y = (uint32_t)oy0 | ((uint32_t)oy1 << 16);

    // What the previous batch wrote in this tile of the context's own mask and this batch will not overwrite is
    // cleared here (the mask keeps "zero outside the recorded regions" from batch to batch; a caller-owned mask was
    // cleared by the scan kernel instead, or is written whole when every tile is filtered).
    bool need_clear = false;
    uint32_t pout_x = 1u, pout_y = 1u; // empty
    if (valid) {
        uint4* cb = (uint4*)(a.cur_box + 4 * idx);
        if (!a.ext_mask) {
            const uint4 prev = *cb;
            pout_x = prev.x; pout_y = prev.y;
            const int px0 = (int)(prev.x & 0xffffu), px1 = (int)(prev.x >> 16), py0 = (int)(prev.y & 0xffffu), py1 = (int)(prev.y >> 16);
            if (px0 <= px1) need_clear = !(marked && ox0 <= px0 && px1 <= ox1 && oy0 <= py0 && py1 <= oy1);
            if (prev.x != nout_x || prev.y != nout_y || marked)
                *cb = make_uint4(nout_x, nout_y, (uint32_t)bx0 | ((uint32_t)bx1 << 16), (uint32_t)by0 | ((uint32_t)by1 << 16));
            if (a.cells[idx] != 0u) a.cells[idx] = 0u;
        } else {
            *cb = make_uint4(nout_x, nout_y, (uint32_t)bx0 | ((uint32_t)bx1 << 16), (uint32_t)by0 | ((uint32_t)by1 << 16));
            a.cells[idx] = 0u;
        }
    }
    for (uint64_t todo = __ballot(need_clear); todo; todo &= todo - 1) { // the whole wave clears one region: lane = row
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t cx = (uint32_t)__builtin_amdgcn_readlane((int)pout_x, src), cy = (uint32_t)__builtin_amdgcn_readlane((int)pout_y, src);
        const int cimage = __builtin_amdgcn_readlane(image, src);
        const int x0 = (int)(cx & 0xffffu), x1 = (int)(cx >> 16), y0 = (int)(cy & 0xffffu), y1 = (int)(cy >> 16);
        uint8_t* m = (uint8_t*)(a.mask + (size_t)cimage * a.H * a.words_per_row);
        const int b0 = x0 >> 3, b1 = x1 >> 3, rb = a.words_per_row * 4;
        for (int y = y0 + lane; y <= y1; y += 64)
            for (int b = b0; b <= b1; b++) m[(size_t)y * rb + b] = 0;
    }

    if (uint64_t wb = __ballot(wide)) { // the wide tiles of this wave: `wide_bands` slots each in their list (the row pipeline is a
        // serial walk down the rows, so a tile's latency, not its work, sets the duration of that short kernel: its rows are cut
        // into bands for several waves, 8 halo rows each)
        const int nbands = a.wide_bands;
        uint32_t base_w = 0;
        if (lane == 0) base_w = atomicAdd(a.n_items + 8, (uint32_t)(__popcll(wb) * nbands));
        base_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_w);
        const uint32_t ow = base_w + (uint32_t)(__popcll(wb & ((1ull << lane) - 1ull)) * nbands);
        if (wide) {
            const int hb = (oy1 - oy0 + nbands) / nbands; // rows per band
            for (int b = 0; b < nbands; b++) {
                const int y0 = oy0 + b * hb, y1 = imin(y0 + hb - 1, oy1);
                if (ow + b < a.cap_wide) // an empty band (y0 > y1) is skipped by the kernel
                    a.wide_tiles[ow + b] = make_uint4((uint32_t)image, (uint32_t)(chunk * a.n_strips + strip), (uint32_t)y0, (uint32_t)(y0 <= y1 ? y1 : y0 - 1));
            }
        }
    }
    // items
    const int cnt = (wide || !emits) ? 0 : nx * ny;
    int incl = cnt; // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    const int wave_total = __builtin_amdgcn_readlane(incl, 63);
    if (wave_total == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.n_items, (uint32_t)wave_total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    uint32_t o = base + (uint32_t)(incl - cnt);
    if (cnt) {
        const int pb = (nb + nx - 1) / nx, ph = (h + ny - 1) / ny;
        for (int jy = 0; jy < ny; jy++)
            for (int jx = 0; jx < nx; jx++, o++) {
                const int x0 = ix0 + 8 * pb * jx, x1 = imin(x0 + 8 * pb - 1, ix1);
                const int y0 = iy0 + ph * jy, y1 = imin(y0 + ph - 1, iy1);
                if (o >= a.cap_items) continue; // cannot happen: the list holds BOX_MAX_PARTS items per tile
                uint4 it;
                it.x = (uint32_t)image;
                it.y = (uint32_t)(chunk * a.n_strips + strip);
                it.z = (uint32_t)x0 | ((uint32_t)x1 << 16);
                it.w = (uint32_t)y0 | ((uint32_t)y1 << 16);
                if (x0 > x1 || y0 > y1) it.z = 1u; // an empty part (rounding of the split): the consumer skips it
                ((uint4*)a.items)[2 * o] = it;
                ((uint4*)a.items)[2 * o + 1] = make_uint4((uint32_t)ux0 | ((uint32_t)ux1 << 16), (uint32_t)uy0 | ((uint32_t)uy1 << 16), 0u, 0u);
            }
    }
}

// ---- the consumer ---------------------------------------------------------------------------------------------------------
// Geometry of one item (all wave-uniform):
//   output region   columns [ox0, ox1] (ox0 a multiple of 8), rows [oy0, oy1]
//   threshold rows  [ty0, ty1] = [oy0 - 2, oy1 + 2] clipped to the image (the median replicates the border rows)
//   patch           columns [px0, px0 + 4 Q) with px0 = ox0 - 4, Q = 2 * bytes + 2 quads; rows [hy0, hy1] = [ty0 - 2, ty1 + 2]
//   exact region    the patch pixels inside the scan's box and the image; every other patch pixel counts as 0, which is
//                   exact for the mask (blob_filter.hip, "dark-tile early-out")
// Lane -> (rsub, q): q = lane % Q the quad, rsub = lane / Q the row of the trip; trip k handles rows k * rpw + rsub.
// Global memory latency is taken out of the item loop: the next item's header and the box of source pixels it will read
// are fetched while the current item is filtered, all staging loads of an item are in flight together, and the table
// words of the next group of trips are requested before the current group is blended.
#ifndef MOCAP_BOX_GROUP      // (build-time knobs for A/B builds: scratch/build_variant.sh)
#define MOCAP_BOX_GROUP 3
#endif
#ifndef MOCAP_BOX_SU
#define MOCAP_BOX_SU 7
#endif
#ifndef MOCAP_BOX_WAVES
#define MOCAP_BOX_WAVES 4
#endif
constexpr int BOX_GROUP = MOCAP_BOX_GROUP; // trips whose table / frame loads are issued together

struct BoxGeom { // derived from an item header, wave-uniform
    int image, tile, slot, remap;
    int ox0, ox1, oy0, oy1, Q, px0, ty0, ty1, hy0, hy1, PR, ey0, ey1, eq0, eq1;
    bool valid, exact;
};
__device__ __forceinline__ BoxGeom box_geometry(int H, int W, int cam_mod, uint64_t remap_bits, uint4 h0, uint4 h1)
{
    BoxGeom g;
    g.image = uni((int)h0.x); g.tile = uni((int)h0.y);
    g.ox0 = uni((int)(h0.z & 0xffffu)); g.ox1 = uni((int)(h0.z >> 16)); g.oy0 = uni((int)(h0.w & 0xffffu)); g.oy1 = uni((int)(h0.w >> 16));
    const int bx0 = uni((int)(h1.x & 0xffffu)), bx1 = uni((int)(h1.x >> 16)), by0 = uni((int)(h1.y & 0xffffu)), by1 = uni((int)(h1.y >> 16));
    g.valid = g.ox0 <= g.ox1 && g.oy0 <= g.oy1;
    g.slot = g.image % cam_mod;
    g.remap = (int)((remap_bits >> g.slot) & 1ull);
    const int nb = (g.ox1 - g.ox0 + 8) >> 3;
    g.Q = 2 * nb + 2; g.px0 = g.ox0 - 4;
    g.ty0 = imax(g.oy0 - 2, 0); g.ty1 = imin(g.oy1 + 2, H - 1);
    g.hy0 = g.ty0 - 2; g.hy1 = g.ty1 + 2; g.PR = g.hy1 - g.hy0 + 1;
    g.ey0 = imax(imax(by0, g.hy0), 0); g.ey1 = imin(imin(by1, g.hy1), H - 1);
    g.eq0 = imax((imax(bx0, 0) - g.px0) >> 2, 0); g.eq1 = imin((imin(bx1, W - 1) - g.px0) >> 2, g.Q - 1);
    g.exact = g.valid && g.ey0 <= g.ey1 && g.eq0 <= g.eq1;
    return g;
}
// per-lane partial of the box of source pixels the exact region of `g` reads (union of the per-cell boxes); the caller
// reduces it over the wave when it needs it, so that the loads stay in flight meanwhile
struct SrcBounds { int xa, xb, ya, yb; };
__device__ __forceinline__ SrcBounds source_bounds_partial(const ushort4* __restrict__ srcbox, int H, int W, const BoxGeom& g, int lane)
{
    SrcBounds b{0x7fff, -0x8000, 0x7fff, -0x8000};
    if (!g.exact || !g.remap) return b;
    const int ncx8 = (W + 7) >> 3;
    const ushort4* __restrict__ sbx = srcbox + (size_t)g.slot * ncx8 * ((H + 7) >> 3);
    const int cx0 = imax(g.px0 + 4 * g.eq0, 0) >> 3, cx1 = imin(g.px0 + 4 * g.eq1 + 3, W - 1) >> 3, cy0 = g.ey0 >> 3, cy1 = g.ey1 >> 3;
    const int ncw = cx1 - cx0 + 1, ncells = ncw * (cy1 - cy0 + 1);
    const float rcpw = __builtin_amdgcn_rcpf((float)ncw);
    for (int c0 = 0; c0 < ncells; c0 += 64) {
        const int c = imin(c0 + lane, ncells - 1);
        const int cyi = small_div(c, rcpw), cxi = c - cyi * ncw;
        const ushort4 sb = sbx[(cy0 + cyi) * ncx8 + cx0 + cxi];
        b.xa = imin(b.xa, (int)sb.x - 2); b.xb = imax(b.xb, (int)sb.y - 2); b.ya = imin(b.ya, (int)sb.z - 2); b.yb = imax(b.yb, (int)sb.w - 2);
    }
    return b;
}

// Registers.  The kernel's 20 KB of LDS per wave allow two waves per SIMD, and left to itself the compiler takes every register
// two waves can have (242; 214 when told to aim at two).  What it does with them is keep loads in flight -- six trips' table words
// twice over, 28 staging dwords -- and a wave that size shuts the other batches' kernels out of its SIMD.
// Round 4: groups of 3 trips and 14 staging loads, compiled for four waves per SIMD: 128 registers, no spills, the same duration
// alone (0.49 ms for the benchmark batch's filters) and +3..7 % for the three-batch pipeline (profiles/history/r4_*.log).
constexpr size_t BOX_LDS_BYTES = 1024 + (size_t)(BOX_HCAP + 64) * 8 + BOX_SCAP;
__global__ __attribute__((amdgpu_waves_per_eu(MOCAP_BOX_WAVES, MOCAP_BOX_WAVES))) __launch_bounds__(64) void box_filter_kernel(BoxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t box_lds[];
    uint32_t* const lut = (uint32_t*)box_lds;
    uint2* const Hs = (uint2*)(box_lds + 1024);                         // + 64: where lanes without a row park their store
    uint8_t* const Sbuf = box_lds + 1024 + (size_t)(BOX_HCAP + 64) * 8; // staged source pixels; afterwards the window counts
    uint32_t* const Cs = (uint32_t*)Sbuf;
    static_assert((BOX_HCAP + 64) * 4 <= BOX_SCAP, "counts alias the source buffer");

    const int lane = threadIdx.x;
    // kernel arguments as plain scalars (a struct captured by the lambdas below would be kept in scratch memory)
    const uint8_t* __restrict__ const a_src = a.src;
    const size_t a_image_stride = a.image_stride;
    const int a_pitch = a.pitch, a_stage_bytes = a.stage_bytes, a_thr_mul = a.thr_mul, a_words_per_row = a.words_per_row;
    const int a_n_chunks = a.n_chunks, a_n_strips = a.n_strips, a_rows_per_chunk = a.rows_per_chunk;
    const uint32_t* __restrict__ const a_map4 = a.map4;
    uint32_t* __restrict__ const a_mask = a.mask;
    uint32_t* __restrict__ const a_cells = a.cells;
    const uint32_t a_cap_items = a.cap_items;
    const uint32_t* const a_n_items = a.n_items;
    uint64_t* const a_timing = a.timing;
    if (a.prio) __builtin_amdgcn_s_setprio(2); // A/B switch: these waves compute, the scan's waves of the next batch wait on HBM
    // lut[w]: byte k = number of set bits among bits k..k+4 of the 8-bit window w
#pragma unroll
    for (int e = 0; e < 4; e++) {
        uint32_t i = (uint32_t)(lane + 64 * e), v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)__popc((i >> k) & 0x1fu) << (8 * k);
        lut[i] = v;
    }
    uint32_t n_items = *a_n_items;
    n_items = n_items < a_cap_items ? n_items : a_cap_items;
    // The list is cut into 8 runs of consecutive items, one per XCD (blocks b and b + 8 share an XCD; consecutive items share
    // undistort-table lines, see settle).  A wave takes the next item of its run with one atomic add on the run's head word
    // -- requested an item ahead, so its latency hides behind the filtering -- and moves on to the next run when its own is
    // exhausted: items differ in cost, a fixed deal left the busiest wave with 1.4x the mean.
    uint32_t* const heads = a.n_items + 16; // 8 head words, 64 bytes apart
    int shard = blockIdx.x & 7, tries = 0;
    auto take = [&]() __attribute__((always_inline)) -> uint32_t { // index of the next item, or 0xffffffff when the list is exhausted
        for (;;) {
            const uint32_t lo = (uint32_t)(((uint64_t)n_items * (uint32_t)shard) >> 3), hi = (uint32_t)(((uint64_t)n_items * (uint32_t)(shard + 1)) >> 3);
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(heads + 16 * shard, 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
            if (i < hi - lo) return lo + i;
            if (++tries >= 8) return 0xffffffffu;
            shard = (shard + 1) & 7;
        }
    };
    const int H = a.H, W = a.W, Hm1 = H - 1;
    const int row_bytes = a_words_per_row * 4;
    const int cam_mod = a.cam_mod;
    const uint64_t remap_bits = a.remap_bits;
    const ushort4* __restrict__ srcbox = a.srcbox;
    const uint4* __restrict__ items = (const uint4*)a.items;
    // the same in two halves, so that the atomic's round trip hides behind an item's work: request now, look later
    auto take_request = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t i = 0;
        if (lane == 0 && tries < 8) i = atomicAdd(heads + 16 * shard, 1u);
        return i;
    };
    auto take_resolve = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t {
        if (tries >= 8) return 0xffffffffu;
        const uint32_t lo = (uint32_t)(((uint64_t)n_items * (uint32_t)shard) >> 3), hi = (uint32_t)(((uint64_t)n_items * (uint32_t)(shard + 1)) >> 3);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (i < hi - lo) return lo + i;
        if (++tries >= 8) return 0xffffffffu;
        shard = (shard + 1) & 7;
        return take(); // this run is exhausted: the next ones, synchronously (the tail of the kernel only)
    };
    uint32_t item = take();
    if (item == 0xffffffffu) return;
    uint32_t nitem = take();

    uint64_t tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = a.timing ? __builtin_readcyclecounter() : 0; // phase clock (debugging aid)
    auto tick = [&](int i) __attribute__((always_inline)) { if (a_timing) { const uint64_t now = __builtin_readcyclecounter(); tacc[i] += now - tlast; tlast = now; } };
    uint4 h0 = items[2 * item], h1 = items[2 * item + 1];
    BoxGeom g = box_geometry(H, W, cam_mod, remap_bits, h0, h1);
    SrcBounds sbp = source_bounds_partial(srcbox, H, W, g, lane);

    while (item != 0xffffffffu) {
        // the next item's header and the index of the one after it: requested now, looked at when this item's patch is done /
        // at the end of the iteration
        const uint32_t nxt = nitem != 0xffffffffu ? nitem : item;
        const uint4 n0 = items[2 * nxt], n1 = items[2 * nxt + 1];
        const uint32_t nn_raw = take_request();
        if (!g.valid) { // an empty part of a split
            g = box_geometry(H, W, cam_mod, remap_bits, n0, n1);
            sbp = source_bounds_partial(srcbox, H, W, g, lane);
            item = nitem;
            nitem = take_resolve(nn_raw);
            continue;
        }
        const int image = g.image, slot = g.slot;
        const int ox0 = g.ox0, oy0 = g.oy0, oy1 = g.oy1, Q = g.Q, px0 = g.px0, ty0 = g.ty0, ty1 = g.ty1, hy0 = g.hy0, PR = g.PR;
        const int ey0 = g.ey0, ey1 = g.ey1, eq0 = g.eq0, eq1 = g.eq1;
        const float rcpQ = __builtin_amdgcn_rcpf((float)Q);
        const int rpw = small_div(64, rcpQ);
        const int rsub = small_div(lane, rcpQ), q = lane - rsub * Q;
        const bool lane_on = rsub < rpw;
        const int x = px0 + 4 * q;
        uint32_t bytemask = 0, colmask = 0; // columns of this lane's quad inside the image
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((unsigned)(x + k) < (unsigned)W) { bytemask |= 0xffu << (8 * k); colmask |= 1u << k; }
        const bool q_exact = q >= eq0 && q <= eq1;
        const int qc = q < eq0 ? eq0 : (q > eq1 ? eq1 : q);
        const int xc = imax(px0 + 4 * qc, 0); // column of a quad of the exact region (loads of other lanes go there and are discarded)
        const uint8_t* __restrict__ img = a_src + (size_t)image * a_image_stride;
        const int ntrip_a = (PR + rpw - 1) / rpw;
        __syncthreads(); // (single wave) the previous item's LDS reads are done
        tick(0);

        // ---- pass A: undistorted patch -> horizontal 5-sums of every patch row, in LDS -----------------------------------
        // hsum of one patch row's quad B (4 bytes): packed 16-bit sums of the 5 columns around each pixel
        auto store_h = [&](uint32_t B, int r) __attribute__((always_inline)) {
            const uint32_t A = from_prev(B), C = from_next(B);
            const uint32_t sB = dot4(B, 0x01010101u, 0u);
            const uint32_t h0_ = dot4(A, 0x01010000u, dot4(B, 0x00010101u, 0u));
            const uint32_t h1_ = dot4(A, 0x01000000u, sB);
            const uint32_t h2_ = dot4(C, 0x00000001u, sB);
            const uint32_t h3_ = dot4(C, 0x00000101u, dot4(B, 0x01010100u, 0u));
            const int at = (lane_on && r < PR) ? __mul24(r, Q) + q : BOX_HCAP + lane;
            Hs[at] = make_uint2(h0_ | (h1_ << 16), h2_ | (h3_ << 16));
        };
        // patch rows outside the exact region are zeros: trips [0, ka0) and [ka1, ntrip_a) hold none of its rows
        const int ka0 = g.exact ? (ey0 - hy0) / rpw : ntrip_a, ka1 = g.exact ? (ey1 - hy0) / rpw + 1 : ntrip_a;
        for (int k = 0; k < ntrip_a; k++) {
            if (k == ka0) k = ka1;
            if (k >= ntrip_a) break;
            const int r = k * rpw + rsub;
            Hs[(lane_on && r < PR) ? __mul24(r, Q) + q : BOX_HCAP + lane] = make_uint2(0u, 0u);
        }
        const int ngroup_e = (ka1 - ka0 + BOX_GROUP - 1) / BOX_GROUP; // groups of trips of the exact region
        if (!g.exact) {
        } else if (!g.remap) {
            // identity map: the patch is the frame
            const int ax = xc > W - 4 ? W - 4 : xc; // W >= 4
            const uint32_t sh = (uint32_t)((xc - ax) * 8);
            auto load_rows = [&](uint32_t (&raw)[BOX_GROUP], int gi) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < BOX_GROUP; u++) {
                    const int y = hy0 + (ka0 + gi * BOX_GROUP + u) * rpw + rsub, yc = y < ey0 ? ey0 : (y > ey1 ? ey1 : y);
                    __builtin_memcpy(&raw[u], img + ((uint32_t)yc * (uint32_t)a_pitch + (uint32_t)ax), 4);
                }
            };
            auto use_rows = [&](const uint32_t (&raw)[BOX_GROUP], int gi) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < BOX_GROUP; u++) {
                    const int kt = ka0 + gi * BOX_GROUP + u, r = kt * rpw + rsub, y = hy0 + r;
                    const bool ok = q_exact && y >= ey0 && y <= ey1;
                    store_h(ok ? ((raw[u] >> sh) & bytemask) : 0u, kt < ka1 ? r : PR);
                }
            };
            uint32_t ra[BOX_GROUP], rb[BOX_GROUP];
            load_rows(ra, 0);
            for (int gi = 0; gi < ngroup_e; gi += 2) {
                load_rows(rb, gi + 1);
                use_rows(ra, gi);
                if (gi + 1 < ngroup_e) {
                    load_rows(ra, gi + 2);
                    use_rows(rb, gi + 1);
                }
            }
        } else {
            // 1. source pixels the exact region reads (their loads were issued during the previous item)
            int sxa = sbp.xa, sxb = sbp.xb, sya = sbp.ya, syb = sbp.yb;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sxa = imin(sxa, __shfl_xor(sxa, d)); sxb = imax(sxb, __shfl_xor(sxb, d));
                sya = imin(sya, __shfl_xor(sya, d)); syb = imax(syb, __shfl_xor(syb, d));
            }
            sxa = uni(sxa) & ~3; sxb = uni(sxb); sya = uni(sya); syb = uni(syb);
            const int SP = (sxb - sxa + 4) & ~3, SR = syb - sya + 1, dpr = SP >> 2;
            const bool staged = SP * SR <= a_stage_bytes && dpr <= 64;
            const uint32_t* __restrict__ map4 = a_map4 + (size_t)slot * H * W;
            auto load_table = [&](uint4 (&tw)[BOX_GROUP], int gi) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < BOX_GROUP; u++) {
                    const int y = hy0 + (ka0 + gi * BOX_GROUP + u) * rpw + rsub, yc = y < ey0 ? ey0 : (y > ey1 ? ey1 : y);
                    __builtin_memcpy(&tw[u], map4 + ((uint32_t)yc * (uint32_t)W + (uint32_t)xc), 16);
                }
            };
            uint4 ta[BOX_GROUP], tb[BOX_GROUP];
            if (staged) {
                // 2. stage them: coalesced dword loads, zeros outside the image (cv::remap's BORDER_CONSTANT).  Lane -> (row of
                // the round, dword of the row); all rounds' loads are in flight together.
                const float rcpd = __builtin_amdgcn_rcpf((float)dpr);
                const int rpi = small_div(64, rcpd), srs = small_div(lane, rcpd), sc = lane - srs * dpr;
                const bool s_on = srs < rpi;
                const int gx = sxa + 4 * sc, ax = gx < 0 ? 0 : (gx > W - 4 ? W - 4 : gx);
                const int sb = (gx - ax) * 8; // > 0 only at the right edge of the image
                uint32_t keep = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((unsigned)(gx + k) < (unsigned)W) keep |= 0xffu << (8 * k);
                if (sb >= 32 || sb < 0) keep = 0;
                const uint32_t shs = (uint32_t)(sb < 0 || sb > 31 ? 0 : sb);
                constexpr int SU = MOCAP_BOX_SU;
                const int nround = (SR + rpi - 1) / rpi;
                // the whole source rectangle inside the image (the usual case): no clamps, no masks
                const bool interior = sya >= 0 && syb <= Hm1 && sxa >= 0 && sxa + SP <= W;
                if (interior) {
                    const uint32_t step = (uint32_t)__mul24(rpi, a_pitch);
                    const int scc = sc < dpr ? sc : dpr - 1; // (lanes past the last dword of a row re-read it)
                    const uint32_t glast = (uint32_t)__mul24(syb, a_pitch) + (uint32_t)(sxa + 4 * scc); // same column, last row
                    uint32_t goff = (uint32_t)__mul24(sya + (srs < SR ? srs : SR - 1), a_pitch) + (uint32_t)(sxa + 4 * scc);
                    int li = __mul24(srs, dpr) + sc;
                    const int lstep = __mul24(rpi, dpr), lend = s_on ? __mul24(SR, dpr) : 0;
                    for (int r0 = 0; r0 < nround; r0 += 2 * SU) {
                        uint32_t v[2 * SU];
#pragma unroll
                        for (int u = 0; u < 2 * SU; u++) {
                            // rows past the rectangle re-read its last row: always inside the image
                            __builtin_memcpy(&v[u], img + goff, 4);
                            goff = goff + step < glast ? goff + step : glast;
                        }
                        if (r0 == 0) load_table(ta, 0);
#pragma unroll
                        for (int u = 0; u < 2 * SU; u++) {
                            if (li < lend) ((uint32_t*)Sbuf)[li] = v[u];
                            li += lstep;
                        }
                    }
                } else
                for (int r0 = 0; r0 < nround; r0 += 2 * SU) {
                    uint32_t v[2 * SU];
#pragma unroll
                    for (int u = 0; u < 2 * SU; u++) {
                        const int gy = sya + (r0 + u) * rpi + srs, gyc = gy < 0 ? 0 : (gy > Hm1 ? Hm1 : gy);
                        __builtin_memcpy(&v[u], img + ((uint32_t)gyc * (uint32_t)a_pitch + (uint32_t)ax), 4);
                    }
                    if (r0 == 0) load_table(ta, 0);
#pragma unroll
                    for (int u = 0; u < 2 * SU; u++) {
                        const int r = (r0 + u) * rpi + srs, gy = sya + r;
                        const uint32_t val = ((unsigned)gy < (unsigned)H) ? ((v[u] >> shs) & keep) : 0u;
                        if (s_on && r < SR) ((uint32_t*)Sbuf)[r * dpr + sc] = val;
                    }
                }
                __syncthreads();
                tick(1);
            } else {
                load_table(ta, 0);
            }
            // 3. table words (one 16-byte load per quad), taps from LDS, blend exactly as cv::remap's fixed point does
            auto blend_rows = [&](auto staged_c, const uint4 (&tw)[BOX_GROUP], int gi) __attribute__((always_inline)) {
                constexpr bool STAGED = decltype(staged_c)::value;
#pragma unroll
                for (int u = 0; u < BOX_GROUP; u++) {
                    const int kt = ka0 + gi * BOX_GROUP + u, r = kt * rpw + rsub, y = hy0 + r;
                    const int yc = y < ey0 ? ey0 : (y > ey1 ? ey1 : y);
                    // lanes outside the exact region blend a pixel of it (clamped row / quad) and drop the result: no branch,
                    // so that the trips of a group overlap their LDS reads
                    uint32_t okm = (q_exact && y >= ey0 && y <= ey1) ? bytemask : 0u;
                    asm volatile("" : "+v"(okm));
                    const uint32_t ww[4] = {tw[u].x, tw[u].y, tw[u].z, tw[u].w};
                    int rowbase = __mul24(yc - sya, SP) + (xc - sxa); // LDS offset of (xc, yc)
                    asm volatile("" : "+v"(rowbase)); // (keeps dy * SP a 24-bit multiply-add of its own)
                    uint32_t B = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t w = ww[k];
                        const int dx = (int)(w << 21) >> 21, dy = (int)(w << 10) >> 21;
                        const uint32_t fa = (w >> 22) & 31u, fb = w >> 27;
                        uint32_t p00, p01, p10, p11;
                        if (STAGED) {
                            const int A0 = __mul24(dy, SP) + (rowbase + k) + dx;
                            p00 = Sbuf[A0]; p01 = Sbuf[A0 + 1]; p10 = Sbuf[A0 + SP]; p11 = Sbuf[A0 + SP + 1];
                        } else { // source region too large for the LDS buffer (strong local distortion): taps from memory,
                                 // loaded from the nearest in-image position and zeroed by select (no branch around loads)
                            const int sx = xc + k + dx, sy = yc + dy;
                            const int sx0 = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx), sx1 = sx + 1 < 0 ? 0 : (sx + 1 > W - 1 ? W - 1 : sx + 1);
                            const int sy0 = sy < 0 ? 0 : (sy > Hm1 ? Hm1 : sy), sy1 = sy + 1 < 0 ? 0 : (sy + 1 > Hm1 ? Hm1 : sy + 1);
                            const uint32_t o0 = (uint32_t)sy0 * (uint32_t)a_pitch, o1 = (uint32_t)sy1 * (uint32_t)a_pitch;
                            const uint32_t t00 = img[o0 + (uint32_t)sx0], t01 = img[o0 + (uint32_t)sx1], t10 = img[o1 + (uint32_t)sx0], t11 = img[o1 + (uint32_t)sx1];
                            const bool c0 = sx0 == sx, c1 = sx1 == sx + 1, r0 = sy0 == sy, r1 = sy1 == sy + 1;
                            p00 = (c0 && r0) ? t00 : 0u; p01 = (c1 && r0) ? t01 : 0u; p10 = (c0 && r1) ? t10 : 0u; p11 = (c1 && r1) ? t11 : 0u;
                        }
                        const uint32_t wa = 32u - fa, wb = 32u - fb;
                        const uint32_t top = __umul24(p00, wa) + __umul24(p01, fa), bot = __umul24(p10, wa) + __umul24(p11, fa);
                        const uint32_t rr = (__umul24(top, wb) + __umul24(bot, fb) + 512u) >> 10; // == (sum of 32*w*p + 2^14) >> 15
                        B |= rr << (8 * k);
                    }
                    store_h(B & okm, kt < ka1 ? r : PR);
                }
            };
            auto remap_rows = [&](auto staged_c) __attribute__((always_inline)) {
                for (int gi = 0; gi < ngroup_e; gi += 2) {
                    load_table(tb, gi + 1);
                    blend_rows(staged_c, ta, gi);
                    if (gi + 1 < ngroup_e) {
                        load_table(ta, gi + 2);
                        blend_rows(staged_c, tb, gi + 1);
                    }
                }
            };
            if (staged) remap_rows(std::true_type{});
            else remap_rows(std::false_type{});
        }
        __syncthreads();
        tick(2);
        // the next item: its geometry, and the loads of its source box (in flight during passes B and C)
        const BoxGeom gn = box_geometry(H, W, cam_mod, remap_bits, n0, n1);
        const SrcBounds sbn = source_bounds_partial(srcbox, H, W, gn, lane);

        // ---- pass B: vertical 5-sums -> threshold -> horizontal window counts of the threshold bits --------------------
        const int c0t = taps5(x, W), c1t = taps5(x + 1, W), c2t = taps5(x + 2, W), c3t = taps5(x + 3, W);
        const uint32_t cx01 = (uint32_t)(c0t & 0xffff) | ((uint32_t)c1t << 16), cx23 = (uint32_t)(c2t & 0xffff) | ((uint32_t)c3t << 16);
        const bool left_edge = px0 < 0, right_edge = px0 + 4 * Q > W; // the patch sticks out of the image
        const int qe = (W - 1 - px0) >> 2, be = (W - 1 - px0) & 3;    // quad / bit of column W - 1
        const int ntrip_b = (ty1 - ty0 + 1 + rpw - 1) / rpw;
        const int hstep = Q;                                          // Hs entries per patch row
        auto pass_b = [&](auto edge_c) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_c)::value; // the patch sticks out of the image on the left or right
#pragma unroll 2
            for (int k = 0; k < ntrip_b; k++) {
                const int y = ty0 + k * rpw + rsub, yb = imin(y, ty1), rh = yb - hy0; // H row of the threshold row (>= 2)
                const uint2* hp = Hs + (__mul24(rh - 2, hstep) + q);
                const uint2 v0 = hp[0], v1 = hp[hstep], v2 = hp[2 * hstep], v3 = hp[3 * hstep], v4 = hp[4 * hstep];
                const uint32_t V01 = v0.x + v1.x + v2.x + v3.x + v4.x, V23 = v0.y + v1.y + v2.y + v3.y + v4.y;
                const uint32_t m = (uint32_t)__mul24(a_thr_mul, taps5(yb, H));
                const uint32_t T01 = __umul24(cx01, m), T23 = __umul24(cx23, m);
                const uint32_t d01 = (V01 | 0x80008000u) - T01, d23 = (V23 | 0x80008000u) - T23;
                const uint32_t t = (d01 >> 15) & 0x10001u, u = (d23 >> 15) & 0x10001u;
                const uint32_t wv = t | (u << 2);
                uint32_t nib = (wv | (wv >> 15)) & 0xfu;
                if (EDGE) {
                    // medianBlur replicates the border: columns outside the image take the edge column's bit
                    if (left_edge) { // quad 0 = columns -4..-1, quad 1 starts at column 0
                        const uint32_t e = from_next(nib) & 1u;
                        if (q == 0) nib = e ? 0xfu : 0u;
                    }
                    if (right_edge) {
                        const uint32_t ne = (uint32_t)__shfl((int)nib, lane - q + qe);
                        const uint32_t e = (ne >> be) & 1u, keep = (2u << be) - 1u;
                        if (q > qe) nib = e ? 0xfu : 0u;
                        else if (q == qe) nib = (nib & keep) | (e ? (0xfu & ~keep) : 0u);
                    }
                }
                const uint32_t nl = from_prev(nib), nr = from_next(nib);
                const uint32_t win = (nl >> 2) | (nib << 2) | ((nr & 3u) << 6);
                const uint32_t c = lut[win & 0xffu];
                const int at = (lane_on && y <= ty1) ? __mul24(y - ty0, Q) + q : BOX_HCAP + lane;
                Cs[at] = c;
            }
        };
        if (left_edge || right_edge) pass_b(std::true_type{});
        else pass_b(std::false_type{});
        __syncthreads();
        tick(3);

        // ---- pass C: majority (>= 13 of 25) of every output row, two quads -> one byte of the bit mask -----------------
        // the item's output region lies in at most 2 x 2 tiles: chunks chunk0 (+1), strips strip0 (+1)
        const int chunk0 = oy0 / a_rows_per_chunk, strip0 = ox0 / 240;
        const int tile_r0 = chunk0 * a_rows_per_chunk, tile_r1 = tile_r0 + a_rows_per_chunk; // first row of chunk0 / of chunk0 + 1
        uint8_t* __restrict__ mrow = (uint8_t*)(a_mask + (size_t)image * H * a_words_per_row);
        const int out_byte = (ox0 >> 3) + ((q - 1) >> 1);
        const bool stores = lane_on && (q & 1) && q <= Q - 3 && out_byte < ((W + 7) >> 3) && out_byte < row_bytes;
        uint32_t lacc = 0, lacc1 = 0; // bit g: rows 8g..8g+7 of chunk0 (lacc) / chunk0 + 1 (lacc1) hold set pixels in this lane's columns
        const int ntrip_c = (oy1 - oy0 + 1 + rpw - 1) / rpw;
        const bool inner = oy0 >= 2 && oy1 + 2 <= Hm1; // no row of the item's windows is replicated
        auto pass_c = [&](auto inner_c) __attribute__((always_inline)) {
            constexpr bool INNER = decltype(inner_c)::value; // no row of the item's windows is replicated
#pragma unroll 2
            for (int k = 0; k < ntrip_c; k++) {
                const int y = oy0 + k * rpw + rsub, yy = imin(y, oy1);
                uint32_t Cv = 0;
                if (INNER) {
                    const uint32_t* cp = Cs + (__mul24(yy - 2 - ty0, hstep) + q);
                    Cv = cp[0] + cp[hstep] + cp[2 * hstep] + cp[3 * hstep] + cp[4 * hstep];
                } else {
#pragma unroll
                    for (int d = -2; d <= 2; d++) {
                        const int yr = yy + d < 0 ? 0 : (yy + d > Hm1 ? Hm1 : yy + d); // BORDER_REPLICATE in y
                        Cv += Cs[__mul24(yr - ty0, Q) + q];
                    }
                }
                const uint32_t mm = ((Cv + 0x73737373u) >> 7) & 0x01010101u;
                const uint32_t t1 = mm | (mm >> 7);
                const uint32_t mn = (t1 | (t1 >> 14)) & colmask & 0xfu;
                const uint32_t odd = from_next(mn);
                const uint32_t byte = mn | ((odd & 0xfu) << 4);
                const bool st = stores && y <= oy1;
                if (st) mrow[(size_t)y * row_bytes + out_byte] = (uint8_t)byte;
                const uint32_t hit = (st && byte != 0u) ? 1u : 0u;
                if (yy < tile_r1) lacc |= hit << ((yy - tile_r0) >> 3);
                else lacc1 |= hit << ((yy - tile_r1) >> 3);
            }
        };
        if (inner) pass_c(std::true_type{});
        else pass_c(std::false_type{});
        {   // occupancy words of the tiles (read by the contour kernel): OR of every lane's row groups; bit 31 = filtered
            const int lstrip = (out_byte >= 30 * (strip0 + 1)) ? 1 : 0;   // this lane's bytes lie in strip0 or strip0 + 1
            const int chunk1 = oy1 / a_rows_per_chunk, strip1 = (g.ox1 >> 3) / 30;
            for (int cc = 0; cc <= chunk1 - chunk0; cc++)
                for (int ss = 0; ss <= strip1 - strip0; ss++) {
                    const uint32_t la = cc ? lacc1 : lacc;
                    uint32_t cellmask = 0;
                    for (int gg = 0; gg < 9; gg++)
                        if (__ballot(lstrip == ss && ((la >> gg) & 1u)) != 0ull) cellmask |= 1u << gg;
                    const size_t cidx = ((size_t)image * a_n_chunks + chunk0 + cc) * a_n_strips + strip0 + ss;
                    if (lane == 0) atomicOr(&a_cells[cidx], cellmask | 0x80000000u);
                }
        }
        g = gn; sbp = sbn; item = nitem;
        nitem = take_resolve(nn_raw);
        tick(4);
        tacc[5] += 1;
    }
    if (a_timing && lane == 0)
        for (int i = 0; i < 6; i++) a_timing[(size_t)blockIdx.x * 6 + i] = tacc[i];
}

void launch_settle_tiles(const BoxArgs& a, hipStream_t s)
{
    const int n_all = a.n_clear > a.n_images ? a.n_clear : a.n_images;
    const long long total = (long long)a.n_chunks * a.n_strips * a.cam_mod * ((n_all + a.cam_mod - 1) / a.cam_mod);
    hipLaunchKernelGGL(settle_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// resident workgroups of the box kernel per CU (the list is dealt to a grid of exactly the resident workgroups)
int box_filter_blocks_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, box_filter_kernel, 64, BOX_LDS_BYTES) != hipSuccess || n < 1) n = 8;
    return n;
}

void launch_box_filter(const BoxArgs& a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(box_filter_kernel, dim3(grid), dim3(64), BOX_LDS_BYTES, s, a);
}

} // namespace mocap
