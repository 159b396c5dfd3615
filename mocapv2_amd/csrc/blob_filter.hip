// blob_filter.hip -- the filter stage:  undistort -> 5x5 in-bounds box sum -> threshold -> 5x5 majority.
//
// Replaces, for one batch of camera images resident in HBM, the chain
//   cv.undistort (reference lib/ImageOperations.py:38) -> fast_cuda_blur (lib/CudaOperations.py:5-41)
//   -> cv.threshold (lib/ImageOperations.py:29) -> cv.medianBlur (lib/ImageOperations.py:30)
// and writes the filtered binary image as a bit mask (1 bit / pixel).
//
// Here:
//   bright_cells_kernel  streams every frame byte once (the algorithmic HBM traffic of the stage) and records, per
//                        filter tile, which mask rows and columns can possibly hold a set pixel -- an exact bound, see
//                        "dark-tile early-out" below.  The boxes it leaves are filtered by blob_boxes.hip;
//   filter_mask_kernel   the dense form of the filter (every tile, any image size, any lens model; one wave owns
//                        a strip of 256 source columns and slides down its rows, everything in registers: horizontal
//                        neighbours from DPP wave shifts, byte sums from v_dot4_u32_u8, vertical 5-row windows as
//                        running sums whose history sits in a per-wave LDS ring).  Off the hot path: tiny images, tables
//                        the compact format cannot hold, the single-image entry points;
//   the set-up kernels of the undistort tables and the single-image convenience kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "kernels.h"
#include "scan_mark.h"

namespace mocap {

__device__ __forceinline__ uint32_t lane_from_prev(uint32_t v)
{ // lane L receives lane L-1's value, lane 0 receives 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t lane_from_next(uint32_t v)
{ // lane L receives lane L+1's value, lane 63 receives 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t sel, uint32_t acc)
{
    return __builtin_amdgcn_udot4(a, sel, acc, false);
}

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p)
{ // possibly unaligned 4-byte load (the compiler emits one global_load_dword: unaligned access is enabled on amdhsa)
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t load_u16(const uint8_t* p)
{
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}

// One undistorted pixel (cv::remap, INTER_LINEAR, BORDER_CONSTANT 0) from the two set-up tables: `m` places the
// 2x2 tap window (already clamped into the image), `w` holds the four blend weights with border handling baked in
// (a tap outside the image weighs 0).  Taps with zero weight are not read.  General form, used off the hot path.
__device__ __forceinline__ uint32_t remap_px(const uint8_t* __restrict__ img, int pitch, uint32_t m, uint32_t w, int x, int y)
{
    int sx = x + (int)(int16_t)(m & 0xffffu), sy = y + ((int)m >> 16);
    uint32_t wx0 = w & 0xffu, wx1 = (w >> 8) & 0xffu, wy1 = (w >> 16) & 0xffu, wy0 = w >> 24;
    const uint8_t* r = img + (size_t)sy * pitch + sx;
    uint32_t p00 = (wx0 && wy0) ? r[0] : 0u, p01 = (wx1 && wy0) ? r[1] : 0u;
    uint32_t p10 = (wx0 && wy1) ? r[pitch] : 0u, p11 = (wx1 && wy1) ? r[pitch + 1] : 0u;
    uint32_t top = p00 * wx0 + p01 * wx1, bot = p10 * wx0 + p11 * wx1;
    return (top * wy0 + bot * wy1 + 512u) >> 10; // == (sum of 32*w*p + 2^14) >> 15
}

// per-lane column constants of a 4-pixel group starting at column xl
struct LaneCols {
    int addr_x;        // column actually loaded from: clamp(xl, 0, W-4)
    uint32_t shift;    // bits to shift the loaded dword right so that byte k is column xl+k
    uint32_t bytemask; // 0xff for every byte k with 0 <= xl+k < W
    bool interior;     // all four columns inside the image
};

__device__ __forceinline__ LaneCols lane_cols(int xl, int W)
{
    LaneCols c;
    int ax = xl < 0 ? 0 : (xl > W - 4 ? W - 4 : xl);
    if (ax < 0) ax = 0;
    c.addr_x = ax;
    int sh = (xl - ax) * 8;
    c.shift = sh < 0 ? 0u : (sh > 24 ? 24u : (uint32_t)sh);
    c.bytemask = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if ((unsigned)(xl + k) < (unsigned)W) c.bytemask |= 0xffu << (8 * k);
    c.interior = xl >= 0 && xl + 3 < W;
    return c;
}

// Raw fetch of the four source pixels of a lane (columns xl..xl+3 of row y).  For the plain path the dword is
// returned as loaded (row clamped into the image) and finish_src4 applies the column shift/mask and the row
// validity when the value is consumed, several iterations later, so the load stays in flight meanwhile.
template <bool REMAP, bool TINY>
__device__ __forceinline__ uint32_t fetch_src4(const FilterArgs& a, const uint8_t* __restrict__ img,
                                               const uint32_t* __restrict__ map, int y, int xl, const LaneCols& lc)
{
    const int yc = y < 0 ? 0 : (y > a.H - 1 ? a.H - 1 : y); // y is wave-uniform
    if (REMAP) {
        if ((unsigned)y >= (unsigned)a.H) return 0u;
        uint32_t out = 0;
        const uint32_t* mrow = map + (size_t)yc * a.W;
        const uint32_t* wrow = a.mapw + (map - a.map) + (size_t)yc * a.W;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int x = xl + k;
            if ((unsigned)x < (unsigned)a.W) out |= remap_px(img, a.pitch, mrow[x], wrow[x], x, yc) << (8 * k);
        }
        return out;
    } else {
        // uniform base + 32-bit offset keeps the scalar-base addressing form (and the global address space:
        // a pointer rebuilt from integers would turn these into flat loads, which vmcnt cannot count in order)
        if (!TINY) return load_u32(img + ((uint32_t)yc * (uint32_t)a.pitch + (uint32_t)lc.addr_x)); // needs W >= 4
        const uint8_t* p = img + (size_t)yc * a.pitch;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((unsigned)(xl + k) < (unsigned)a.W) v |= (uint32_t)p[xl + k] << (8 * k);
        return v;
    }
}

template <bool REMAP, bool TINY>
__device__ __forceinline__ uint32_t finish_src4(uint32_t raw, bool row_ok, const LaneCols& lc)
{
    if (REMAP) return raw;
    uint32_t v = TINY ? raw : ((raw >> lc.shift) & lc.bytemask);
    return row_ok ? v : 0u;
}

// ---- software-pipelined remap (three stages, each one source row apart in time) -----------------------------
//   A: issue the load of the row's four packed map words          (8 rows ahead of use)
//   B: decode them, issue the 2x2 tap loads and the weight load    (4 rows ahead of use)
//   C: blend the taps                                             (at use)
// so that neither memory latency is exposed.  Border handling lives in the tables (tap window clamped into the
// image, weights of outside taps zero), so the stages contain no image-edge logic at all.
struct MapSlot { uint4 m; };
struct TapSlot { uint32_t t0[4], t1[4], w[4]; };

__device__ __forceinline__ void remap_issue_map(MapSlot& ms, const uint32_t* __restrict__ map, int row, int H, int W,
                                                const LaneCols& lc)
{
    int rc = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
    __builtin_memcpy(&ms.m, map + ((uint32_t)rc * (uint32_t)W + (uint32_t)lc.addr_x), 16);
}

__device__ __forceinline__ void remap_issue_taps(TapSlot& ts, const MapSlot& ms, const uint8_t* __restrict__ img,
                                                 const uint32_t* __restrict__ mapw, int pitch, int H, int W, int row,
                                                 const int xq[4], const LaneCols& lc)
{
    // Rows outside the image contribute zeros; their loads are simply those of the nearest row (no branch around
    // loads: the compiler's in-flight counts stay exact) and next_row() discards the result.
    row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
    const uint32_t mm[4] = {ms.m.x, ms.m.y, ms.m.z, ms.m.w};
    uint4 wv4;
    __builtin_memcpy(&wv4, mapw + ((uint32_t)row * (uint32_t)W + (uint32_t)lc.addr_x), 16);
    ts.w[0] = wv4.x; ts.w[1] = wv4.y; ts.w[2] = wv4.z; ts.w[3] = wv4.w;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t m = mm[k];
        int sx = xq[k] + (int)(int16_t)(m & 0xffffu), sy = row + ((int)m >> 16); // inside the image by construction
        uint32_t off0 = __umul24((uint32_t)sy, (uint32_t)pitch) + (uint32_t)sx, off1 = off0 + (uint32_t)pitch;
        ts.t0[k] = load_u16(img + off0);
        ts.t1[k] = load_u16(img + off1);
    }
}

__device__ __forceinline__ uint32_t remap_combine(const TapSlot& ts, const LaneCols& lc)
{
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t w = ts.w[k];
        uint32_t top = dot4(ts.t0[k], w, 0u), bot = dot4(ts.t1[k], w, 0u); // tap bytes 2,3 are zero
        uint32_t r = __umul24(top, w >> 24) + 512u;
        r += __umul24(bot, (w >> 16) & 0xffu);
        out |= (r >> 10) << (8 * k);
    }
    return out & lc.bytemask; // columns outside the image do not exist
}

// ---- dark-tile early-out ---------------------------------------------------------------------------------------
// A thresholded pixel can only be 1 if its 5x5 box sum reaches thr_mul * taps.  Every undistorted pixel is at most
// (sum of weight * tap + 512) >> 10 with weights summing to <= 1024, so with the excess e(p) = max(0, p - 63) of a source
// pixel p (p <= 63 + e(p)):
//     box sum  <=  taps * 63.5  +  (total of weight * e over the source pixels feeding the window) / 1024
// and a source pixel's total weight over ALL output pixels is at most Wmax (measured on the table at set-up; 1024 for
// the identity).  The taps of one 5x5 window span at most 9 source pixels in x and y (checked at set-up), i.e. they
// lie inside some 2x2 block of 8x8-pixel cells of a fixed grid.  Hence: if no such block of the tile's source region
// has an excess sum E with Wmax * 2E >= 1024 * taps * (2 * thr_mul - 127) (2E <= allow, computed on the host with the
// smallest tap count), every threshold bit of the tile is 0, so is the majority, and the tile's mask rows are zero --
// without running the filter.  The test is made per cell: no cell of the region with 2E above hot = allow / 4.
// (A bound on the excess, not on the number of bright pixels: a background at 100 or the 3x3 halo a demosaiced hot
// pixel leaves costs what it weighs, not 192 per pixel.)  The base 63 of this text is a parameter c (BrightArgs::base,
// chosen on the host from the threshold): with e(p) = max(0, p - c) the bound reads box sum <= taps * (c + 0.5) + ...,
// i.e. Wmax * 2E < 1024 * taps * (2 * thr_mul - 2c - 1); |p - c| + |p - 0| = 2 e(p) + c per byte keeps it at two v_sad_u8
// per dword for any c.
// 16 aligned bytes of a frame for the streaming pass: a non-temporal load (global_load_dwordx4 ... nt).  The pass reads every
// pixel exactly once, so nothing is gained by keeping the lines in L2 / the Infinity Cache, and the streaming policy itself
// is faster: 6.37 GB per launch in 0.946 ms instead of 1.02 ms (6.7 against 6.2 TB/s; A/B on one box, profiles/README.md).
__device__ __forceinline__ uint4 load_once16(const uint8_t* p)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 q = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(q.x, q.y, q.z, q.w);
}

// One streaming pass over the frames -- the only time a dark tile's pixels are read.  A thread sums the excess over 63
// of two cells of the fixed 8x8-pixel grid (16 eight-byte loads in flight; consecutive lanes take consecutive cells of
// a cell row, so a wave's loads cover 512 contiguous bytes of each of 8 image rows; two v_sad_u8 per dword).  A cell
// whose doubled excess exceeds `hot` (4 * hot <= allow, so four dark cells can never exceed the 2x2-block bound above;
// `hot_edge` and `hot_corner`, from the bounds of the 15- and 9-tap windows, for the cells that feed windows cut by the image
// border in one axis or in both) marks every filter tile its reach touches (reach = the box of output pixels that read
// the cell, tabulated at set-up, + 4 pixels of blur and median) by widening the tile's range of reachable mask rows
// and columns (atomic min / max).  Tiles left unmarked, and rows outside the range, provably filter to zeros.
// WIDE (W, pitch, image stride and base multiples of 16): the two cells of a thread are neighbours in one cell row and
// come in with one 16-byte load per image row (8 loads of 16 B instead of 16 of 8 B per thread).
// FULL (H a multiple of 8, wide only): every cell has its 8 rows, no row clamping and no per-row validity test.
// One block of 256 threads of the pass: block `bx` of image `image` (the kernel below deals these to the workgroups).
template <bool WIDE, bool FULL, bool MAP>
__device__ __forceinline__ void bright_cells_block(const BrightArgs& a, const int bx, const int image)
{
    const int ncx = (a.W + 7) >> 3, ncy = (a.H + 7) >> 3, n = ncx * ncy;
    const uint8_t* __restrict__ img = a.src + (size_t)image * a.image_stride;
    uint2 v[2][8];
    int ci[2], cr[2];
    uint32_t sh[2];
    bool in_range[2];
    if (WIDE) {
        const int half = ncx >> 1, np = half * ncy; // cell pairs (ncx is even)
        int p = bx * 256 + threadIdx.x;
        in_range[0] = in_range[1] = p < np;
        p = p < np ? p : np - 1; // threads past the end recount the last pair (and mark the same tiles again)
        const int row = (int)__umulhi((uint32_t)p, a.ncx_magic), cxp = p - row * half; // ncx_magic: for ncx / 2 here
        cr[0] = cr[1] = row;
        ci[0] = row * ncx + 2 * cxp; ci[1] = ci[0] + 1;
        sh[0] = sh[1] = 0;
        if (FULL) {
            const uint32_t off0 = (uint32_t)(8 * row) * (uint32_t)a.pitch + 16u * (uint32_t)cxp;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint4 q = load_once16(img + (off0 + (uint32_t)(j * a.pitch))); // uniform base + 32-bit offset
                v[0][j] = make_uint2(q.x, q.y); v[1][j] = make_uint2(q.z, q.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int r = 8 * row + j;
                r = r < a.H ? r : a.H - 1;
                const uint4 q = load_once16(img + ((uint32_t)r * (uint32_t)a.pitch + 16u * (uint32_t)cxp));
                v[0][j] = make_uint2(q.x, q.y); v[1][j] = make_uint2(q.z, q.w);
            }
        }
    } else {
        const int i0 = bx * 512 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            int i = i0 + 256 * u;
            in_range[u] = i < n;
            i = i < n ? i : n - 1; // threads past the end recount the last cell (and mark the same tiles again)
            ci[u] = i;
            cr[u] = a.ncx_magic ? (int)__umulhi((uint32_t)i, a.ncx_magic) : i / ncx; // floor(i / ncx) without the division
            const int cx = i - cr[u] * ncx;
            const int c = 8 * cx, cc = c < a.W - 8 ? c : a.W - 8; // W >= 8 (checked on the host)
            sh[u] = (uint32_t)(8 * (c - cc));
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int r = 8 * cr[u] + j;
                r = r < a.H ? r : a.H - 1;
                __builtin_memcpy(&v[u][j], img + ((uint32_t)r * (uint32_t)a.pitch + (uint32_t)cc), 8);
            }
        }
    }
    const int slot = image % a.cam_mod;
    const uint2* __restrict__ reach = a.reach + (size_t)slot * n;
    const uint8_t* __restrict__ cflags = a.cflags + (size_t)slot * n;
    uint32_t* __restrict__ rows = a.tile_rows + (size_t)image * a.n_chunks * a.n_strips * 4;
    const uint32_t base4 = (uint32_t)a.base * 0x01010101u, c8 = 8u * (uint32_t)a.base;
    uint32_t level[2] = {0u, 0u};
#pragma unroll
    for (int u = 0; u < 2; u++) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (WIDE) { // cells are whole
                const uint32_t e2 = excess2_row(v[u][j].x, v[u][j].y, base4, c8);
                if (FULL || 8 * cr[u] + j < a.H) acc += e2;
            } else {
                uint64_t vv = (((uint64_t)v[u][j].y << 32) | v[u][j].x) >> sh[u]; // drops the bytes left of the cell at the right edge
                const uint32_t e2 = excess2_row((uint32_t)vv, (uint32_t)(vv >> 32), base4, c8);
                if (8 * cr[u] + j < a.H) acc += e2;
            }
        }
        if (MAP) level[u] = in_range[u] ? hot_level(a, acc) : 0u;
        else mark_hot_cell(a, reach, cflags, rows, ci[u], acc);
        if (a.probe && (image & 15) == 0) { // wave-uniform, rare: which base would leave fewer hot cells?
            const uint32_t alt4 = (uint32_t)a.base_alt * 0x01010101u, alt8 = 8u * (uint32_t)a.base_alt;
            uint32_t acc2 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint64_t vv = WIDE ? (((uint64_t)v[u][j].y << 32) | v[u][j].x) : ((((uint64_t)v[u][j].y << 32) | v[u][j].x) >> sh[u]);
                const uint32_t e2 = excess2_row((uint32_t)vv, (uint32_t)(vv >> 32), alt4, alt8);
                if (FULL || 8 * cr[u] + j < a.H) acc2 += e2;
            }
            const int n_cur = __popcll(__ballot((int)acc > a.hot)), n_alt = __popcll(__ballot((int)acc2 > a.hot_alt));
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&a.probe[PROBE_STRIDE * (bx & 127)], (uint32_t)n_cur);
                atomicAdd(&a.probe[PROBE_STRIDE * (bx & 127) + 1], (uint32_t)n_alt);
            }
        }
    }
    if (MAP) {
        // The hot map: two bits per cell in cell order, so the 128 cells of a wave are 8 whole words (WIDE: consecutive lanes hold
        // consecutive cell pairs) or two runs of 4 (lanes hold cells i and i + 256).  The lanes sharing a word OR their fields
        // together with DPP moves and one of them stores it if it is not zero (the map is all zeros between batches: mark_tiles_kernel
        // clears what it reads) -- no atomics, and a wave without a hot cell, the usual one, issues no memory operation behind its
        // frame loads at all (a store per wave, tried first, made every wave wait for it at its end: 0.99 against 0.95 ms).
        const int lane = threadIdx.x & 63;
        uint32_t* __restrict__ hm = a.hotmap + (size_t)image * a.hot_words;
        if (WIDE) {
            uint32_t w = (level[0] | (level[1] << 2)) << (4 * (lane & 7));
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1 /*quad_perm 1,0,3,2*/, 0xf, 0xf, true);
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E /*quad_perm 2,3,0,1*/, 0xf, 0xf, true);
            w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x141 /*row_half_mirror*/, 0xf, 0xf, true);
            if ((lane & 7) == 0 && w != 0u) hm[(bx * 256 + (int)threadIdx.x) >> 3] = w;
        } else {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                uint32_t w = level[u] << (2 * (lane & 15));
                w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xf, 0xf, true);
                w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xf, 0xf, true);
                w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x141, 0xf, 0xf, true);
                w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x140 /*row_mirror*/, 0xf, 0xf, true);
                if ((lane & 15) == 0 && w != 0u) hm[(bx * 512 + (int)threadIdx.x + 256 * u) >> 4] = w;
            }
        }
    }
}

// The hot map of the scan -> tile boxes.  One thread per map word (16 cells); a wave collects its hot cells in a list in LDS
// and works through it one cell per lane: reach and flags of the cell (the tables of the image's undistort slot), the
// threshold its flags ask for, and the boxes of the tiles it reaches -- what the scan used to do behind its own loads, where
// every hot wave then sat through two dependent round trips with no frame loads in flight (0.946 ms for the benchmark batch
// at 8 markers per frame, 1.133 at 32: most scan waves carry a hot cell then).
// The hot cells of a marker lie in the same few waves and reach the same one or two tiles, and a memory atomic costs what
// it costs whether it changes anything or not (the first version issued four per hot cell and tile: 4 M of them per batch
// at 8 markers, 0.29 ms; 0.89 ms at 32): the wave first merges its cells' rectangles per tile in a 32-entry table in LDS (tag
// = tile; a collision goes to memory directly) and then widens each tile it touched once.
__global__ __launch_bounds__(256) void mark_tiles_kernel(BrightArgs a)
{
    __shared__ uint16_t s_list[4][1024];
    __shared__ uint32_t s_tab[4][32][5]; // tile | first row | last row | first column | last column
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // a wave takes 256 consecutive words (4 per lane) of one image at a time; with a fixed grid (a few workgroups per CU) the
    // waves go on to further pieces: few workgroups to place beside another batch's scan, which holds every wave slot of the chip
    const int wpi = (a.hot_words + 255) >> 8; // pieces per image
    const int n_cells = ((a.W + 7) >> 3) * ((a.H + 7) >> 3);
    for (long long piece = (long long)blockIdx.x * 4 + wv; piece < (long long)wpi * a.n_images; piece += (long long)gridDim.x * 4) {
    const int image = (int)(piece / wpi), wave_word0 = (int)(piece - (long long)image * wpi) * 256;
    uint32_t* __restrict__ hm = a.hotmap + (size_t)image * a.hot_words;
    uint32_t wk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int wi = wave_word0 + 64 * k + lane;
        wk[k] = wi < a.hot_words ? hm[wi] : 0u;
    }
    if (__ballot((wk[0] | wk[1] | wk[2] | wk[3]) != 0u) == 0ull) continue;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (wk[k] != 0u) hm[wave_word0 + 64 * k + lane] = 0u; // the map is all zeros again for the next batch's scan
    uint32_t (*tab)[5] = s_tab[wv];
    if (lane < 32) { tab[lane][0] = 0xffffffffu; tab[lane][1] = 0xffffffffu; tab[lane][2] = 0u; tab[lane][3] = 0xffffffffu; tab[lane][4] = 0u; }
    const int slot = image % a.cam_mod;
    const uint2* __restrict__ reach = a.reach + (size_t)slot * n_cells;
    const uint8_t* __restrict__ cflags = a.cflags + (size_t)slot * n_cells;
    uint32_t* __restrict__ rows = a.tile_rows + (size_t)image * a.n_chunks * a.n_strips * 4;
    // widen one tile's box in memory (a box that already holds the rectangle needs no atomics)
    auto widen = [&](int t, uint32_t ya, uint32_t yb, uint32_t xa, uint32_t xb) __attribute__((always_inline)) {
        const uint4 cur = *(const uint4*)(rows + 4 * t);
        if (cur.x <= ya && cur.y >= yb && cur.z <= xa && cur.w >= xb) return;
        atomicMin(&rows[4 * t], ya);
        atomicMax(&rows[4 * t + 1], yb);
        atomicMin(&rows[4 * t + 2], xa);
        atomicMax(&rows[4 * t + 3], xb);
    };
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        const uint32_t w = k == 0 ? wk[0] : (k == 1 ? wk[1] : (k == 2 ? wk[2] : wk[3]));
        const uint32_t nz = (w | (w >> 1)) & 0x55555555u; // bit 2j: cell j of the word exceeds at least the lowest threshold
        const int cnt = __popc(nz);
        if (__ballot(cnt != 0) == 0ull) continue;
        const int word0 = wave_word0 + 64 * k;
        int incl = cnt; // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int at = incl - cnt;
        for (uint32_t m = nz; m; m &= m - 1) {
            const int j2 = __ffs((int)m) - 1; // = 2 j
            s_list[wv][at++] = (uint16_t)((lane << 6) | (j2 << 1) | ((w >> j2) & 3u)); // lane | cell of the word | level
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): list and table are written (one wave: no barrier needed)
        __builtin_amdgcn_wave_barrier();
        for (int e0 = 0; e0 < total; e0 += 64) {
            const int e = e0 + lane;
            if (e < total) {
                const uint32_t v = s_list[wv][e];
                const int ci = 16 * (word0 + (int)(v >> 6)) + (int)((v >> 2) & 15u);
                if (ci < n_cells) {
                    const uint2 rc = reach[ci];
                    const int x0 = (int)(rc.x & 0xffffu), x1 = (int)(rc.x >> 16), y0 = (int)(rc.y & 0xffffu), y1 = (int)(rc.y >> 16);
                    if (x0 <= x1 && level_is_hot(v & 3u, cflags[ci])) {
                        // (the rectangle and the tiles it overlaps: as widen_tile_boxes, scan_mark.h)
                        const int xa = x0 - 4 > 0 ? x0 - 4 : 0, xb = x1 + 4 < a.W - 1 ? x1 + 4 : a.W - 1;
                        const int ya = y0 - 4 > 0 ? y0 - 4 : 0, yb = y1 + 4 < a.H - 1 ? y1 + 4 : a.H - 1;
                        const int ch0 = (int)(((uint32_t)ya * a.rows_magic) >> 23), ch1 = (int)(((uint32_t)yb * a.rows_magic) >> 23);
                        const int st0 = (int)(((uint32_t)xa * 34953u) >> 23), st1 = (int)(((uint32_t)xb * 34953u) >> 23);
                        for (int ch = ch0; ch <= ch1; ch++)
                            for (int st = st0; st <= st1; st++) {
                                const int t = ch * a.n_strips + st, q = t & 31;
                                const uint32_t old = atomicCAS(&tab[q][0], 0xffffffffu, (uint32_t)t);
                                if (old == 0xffffffffu || old == (uint32_t)t) {
                                    atomicMin(&tab[q][1], (uint32_t)ya); atomicMax(&tab[q][2], (uint32_t)yb);
                                    atomicMin(&tab[q][3], (uint32_t)xa); atomicMax(&tab[q][4], (uint32_t)xb);
                                } else
                                    widen(t, (uint32_t)ya, (uint32_t)yb, (uint32_t)xa, (uint32_t)xb);
                            }
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier(); // (the list is rewritten by the next k)
    }
    if (lane < 32 && tab[lane][0] != 0xffffffffu) widen((int)tab[lane][0], tab[lane][1], tab[lane][2], tab[lane][3], tab[lane][4]);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier(); // (the table is initialised anew for the wave's next piece)
    }
}

// The grid is one-dimensional: workgroup b takes the blocks b, b + gridDim.x, ... of the batch's blocks_x * n_images blocks
// (block -> image = block / blocks_x).  Launched with as many workgroups as blocks it is the plain form (every workgroup one
// block); launched with a fixed number per CU it is a persistent pass that never holds more than that many wave slots and
// registers of a SIMD, whatever the batch size (BrightArgs::blocks_x, launch_bright_cells).
// MAP: the hot cells go into the hot map (BrightArgs::hotmap) instead of marking their tiles from here.
template <bool WIDE, bool FULL = false, bool MAP = true>
__global__ __launch_bounds__(256) void bright_cells_kernel(BrightArgs a)
{
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const uint32_t total = (uint32_t)a.blocks_x * (uint32_t)a.slice_images;
    if (a.zero_counters && blockIdx.x == 0) a.zero_counters[threadIdx.x] = 0u; // the counter block of the kernels behind this one
    if (a.block_ctr == nullptr) {
        for (uint32_t vb = blockIdx.x; vb < total; vb += gridDim.x) {
            const uint32_t image = vb / (uint32_t)a.blocks_x;
            bright_cells_block<WIDE, FULL, MAP>(a, (int)(vb - image * (uint32_t)a.blocks_x), a.image0 + (int)image);
        }
    } else {
        // persistent form: the workgroups take the blocks in the order of a shared counter, as the hardware dispatcher would hand
        // them out -- the blocks in flight stay one contiguous window of the frames (a fixed stride per workgroup lets them drift
        // apart: 1.20 ms against 0.95 at the same occupancy); the next index is fetched while this block's loads are in flight
        __shared__ uint32_t s_next[2];
        constexpr uint32_t CH = 16; // blocks per visit of the counter (one atomic per block would serialise on it: 2.4 ms)
        uint32_t v0 = blockIdx.x * CH;
        for (int it = 0; v0 < total; it++) {
            if (threadIdx.x == 0) s_next[it & 1] = gridDim.x * CH + atomicAdd(a.block_ctr, CH);
            for (uint32_t vb = v0; vb < v0 + CH && vb < total; vb++) {
                const uint32_t image = vb / (uint32_t)a.blocks_x;
                bright_cells_block<WIDE, FULL, MAP>(a, (int)(vb - image * (uint32_t)a.blocks_x), a.image0 + (int)image);
            }
            __syncthreads();
            v0 = s_next[it & 1];
        }
    }
    if (a.mask_words) {
        // caller-owned masks: clear them on the side (16 bytes per thread and round), the filter kernel then only
        // writes the tiles it filters.  The context's own mask needs no clearing (see the filter kernel).
        const size_t nthreads = (size_t)gridDim.x * 256;
        const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (a.mask_aligned16) {
            const size_t quads = a.mask_words >> 2;
            for (size_t q = g; q < quads; q += nthreads) ((uint4*)a.mask)[q] = make_uint4(0u, 0u, 0u, 0u);
            if (g < (a.mask_words & 3)) a.mask[(quads << 2) + g] = 0u;
        } else {
            for (size_t q = g; q < a.mask_words; q += nthreads) a.mask[q] = 0u;
        }
    }
}

// number of in-image taps of a 5-wide window centred on v
__device__ __forceinline__ int taps5(int v, int n)
{
    int lo = v - 2 < 0 ? 0 : v - 2, hi = v + 2 > n - 1 ? n - 1 : v + 2;
    return hi - lo + 1;
}

template <int J> struct IC { static constexpr int value = J; };


// workgroup -> (camera slot, group of 4 chunks, time step); wave w filters chunk 4 * group + w, strip after strip
struct TileId { int slot, cgroup, image; bool valid; };
__device__ __forceinline__ TileId decode_tile(const FilterArgs& a, int b)
{
    TileId t;
    const int groups = a.cam_mod * a.n_cgroups;
    const int grp = b / a.n_steps, tstep = b - grp * a.n_steps;
    t.slot = grp % a.cam_mod;
    t.cgroup = grp / a.cam_mod;
    t.image = tstep * a.cam_mod + t.slot;
    t.valid = grp < groups && t.image < a.n_images;
    return t;
}

// LIST: instead of every tile of every image, the waves work through a list of (image, tile, first row, last row)
// entries -- the tiles whose boxes settle_tiles_kernel found too wide for the box kernel to be the cheaper way.
#ifndef MOCAP_ROWS_WAVES      // (build-time knob for A/B builds: registers of the row pipeline, scratch/build_variant.sh)
#define MOCAP_ROWS_WAVES 1
#endif
template <bool REMAP, bool TINY, bool PIPE, bool LIST>
__global__ __attribute__((amdgpu_waves_per_eu(MOCAP_ROWS_WAVES))) __launch_bounds__(256) void filter_mask_kernel(FilterArgs a)
{
    __shared__ uint32_t lut[256];
    __shared__ uint2 hring[4][8][64];
    __shared__ uint32_t cring[4][8][64];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: keeps the row loop scalar

    int slot = 0, image = 0, chunk = 0;
    uint32_t n_list = 0;
    if (LIST) {
        n_list = *a.n_tiles;
        n_list = n_list < a.cap_tiles ? n_list : a.cap_tiles;
        if ((uint32_t)(blockIdx.x * 4 + wv) >= n_list) return;
    } else {
        const TileId tid_ = decode_tile(a, blockIdx.x);
        if (!tid_.valid) return;
        slot = tid_.slot; image = tid_.image;
        chunk = tid_.cgroup * 4 + wv;
        if (chunk * a.rows_per_chunk >= a.H) return;
    }
    // lut[w]: byte k = number of set bits among bits k..k+4 of the 8-bit window w.  Every wave writes the whole table
    // (identical values), so no workgroup barrier is needed.
#pragma unroll
    for (int e = 0; e < 4; e++) {
        uint32_t i = (uint32_t)(lane + 64 * e), v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)__popc((i >> k) & 0x1fu) << (8 * k);
        lut[i] = v;
    }
    const uint32_t it_first = LIST ? (uint32_t)(blockIdx.x * 4 + wv) : 0u, it_end = LIST ? n_list : (uint32_t)a.n_strips;
    const uint32_t it_step = LIST ? gridDim.x * 4u : 1u;
    for (uint32_t it = it_first; it < it_end; it += it_step) {
        int strip = (int)it, r0e = 0, r1e = 0x7fffffff;
        if (LIST) {
            const uint4 e = a.tiles[it];
            image = __builtin_amdgcn_readfirstlane((int)e.x);
            const int tile = __builtin_amdgcn_readfirstlane((int)e.y);
            r0e = __builtin_amdgcn_readfirstlane((int)e.z); r1e = __builtin_amdgcn_readfirstlane((int)e.w) + 1;
            chunk = tile / a.n_strips; strip = tile - chunk * a.n_strips;
            slot = image % a.cam_mod;
        }
        const int tile_r0 = chunk * a.rows_per_chunk;              // the tile's mask rows [tile_r0, tile_r1)
        const int tile_r1 = tile_r0 + a.rows_per_chunk < a.H ? tile_r0 + a.rows_per_chunk : a.H;
        const size_t cell_index = ((size_t)image * a.n_cgroups * 4 + chunk) * a.n_strips + strip;
        const int r0 = r0e > tile_r0 ? r0e : tile_r0, r1 = r1e < tile_r1 ? r1e : tile_r1; // the rows filtered
        if (LIST && r0 >= r1) continue; // an empty band
        const int xbase = strip * 240 - 8;

        const int Hm1 = a.H - 1;
        int kfirst = r0 - 2;
        kfirst = kfirst < 0 ? 0 : (kfirst > Hm1 ? Hm1 : kfirst);
        const int ks = r0 - 1 > 1 ? r0 - 1 : 1;                 // steady range: every iteration slides one source row
        const int ke = r1 + 1 < Hm1 ? r1 + 1 : Hm1;

    #pragma unroll
        for (int s = 0; s < 8; s++) {
            hring[wv][s][lane] = make_uint2(0u, 0u);
            cring[wv][s][lane] = 0u;
        }

        const uint8_t* __restrict__ img = a.src + (size_t)image * a.image_stride;
        const uint32_t* __restrict__ map = REMAP ? a.map + (size_t)slot * a.H * a.W : nullptr;
        const uint32_t* __restrict__ mapw = REMAP ? a.mapw + (size_t)slot * a.H * a.W : nullptr;
        uint8_t* __restrict__ mrow_base = (uint8_t*)(a.mask + (size_t)image * a.H * a.words_per_row);
        const int row_bytes = a.words_per_row * 4;

        const int xl = xbase + 4 * lane;

        // per-lane column constants
        uint32_t cx01, cx23, colmask = 0;
        {
            int c0 = taps5(xl, a.W), c1 = taps5(xl + 1, a.W), c2 = taps5(xl + 2, a.W), c3 = taps5(xl + 3, a.W);
            cx01 = (uint32_t)(c0 & 0xffff) | ((uint32_t)c1 << 16);
            cx23 = (uint32_t)(c2 & 0xffff) | ((uint32_t)c3 << 16);
    #pragma unroll
            for (int k = 0; k < 4; k++)
                if ((unsigned)(xl + k) < (unsigned)a.W) colmask |= 1u << k;
        }
        const LaneCols lc = lane_cols(xl, a.W);
        const bool left_edge = xbase < 0;
        const bool right_edge = xbase + 255 >= a.W;
        const int lane_r = (a.W - 1 - xbase) >> 2, bit_r = (a.W - 1 - xbase) & 3; // lane / bit of column W-1
        // byte of the output row written by this (even) lane
        const int out_byte = strip * 30 + ((lane - 2) >> 1);
        const bool stores = ((lane & 1) == 0) && lane >= 2 && lane <= 60 && out_byte < ((a.W + 7) >> 3) &&
                            out_byte < row_bytes;

        uint32_t V01 = 0, V23 = 0, Cv = 0;
        uint32_t lacc = 0; // per lane: bit g = this lane's columns have set pixels in output rows r0+8g .. r0+8g+7
        const bool out_lane = lane >= 2 && lane <= 61;
        // source-row queue, 8 deep.  q[3] holds the first row so that the five set-up slides consume q[3..7] and the
        // steady loop starts at q[0] / ring slot 0 with all indices static.
        uint32_t q[8];
        MapSlot mq[4];
        TapSlot tq[4];
        int xq[4];
    #pragma unroll
        // columns the lane's four table words belong to (lanes outside the image read the nearest in-image group:
        // their taps stay inside the image, their result is masked out)
        for (int k = 0; k < 4; k++) xq[k] = lc.addr_x + k;
        const int y0 = kfirst - 2;
        if (PIPE) {
            // slot of source row rho = (rho - (y0 + 5)) & 3, so that the steady loop starts at slot 0
            remap_issue_map(mq[3], map, y0, a.H, a.W, lc);
            remap_issue_map(mq[0], map, y0 + 1, a.H, a.W, lc);
            remap_issue_map(mq[1], map, y0 + 2, a.H, a.W, lc);
            remap_issue_map(mq[2], map, y0 + 3, a.H, a.W, lc);
            remap_issue_taps(tq[3], mq[3], img, mapw, a.pitch, a.H, a.W, y0, xq, lc);
            remap_issue_map(mq[3], map, y0 + 4, a.H, a.W, lc);
            remap_issue_taps(tq[0], mq[0], img, mapw, a.pitch, a.H, a.W, y0 + 1, xq, lc);
            remap_issue_map(mq[0], map, y0 + 5, a.H, a.W, lc);
            remap_issue_taps(tq[1], mq[1], img, mapw, a.pitch, a.H, a.W, y0 + 2, xq, lc);
            remap_issue_map(mq[1], map, y0 + 6, a.H, a.W, lc);
            remap_issue_taps(tq[2], mq[2], img, mapw, a.pitch, a.H, a.W, y0 + 3, xq, lc);
            remap_issue_map(mq[2], map, y0 + 7, a.H, a.W, lc);
        } else {
    #pragma unroll
            for (int j = 0; j < 8; j++) q[j] = fetch_src4<REMAP, TINY>(a, img, map, y0 + ((j + 5) & 7), xl, lc);
        }
        // next source row (row index `row`, queue slot J): its four pixels, and the refill of the pipeline behind it
        auto next_row = [&](auto Jc, int row) -> uint32_t {
            constexpr int J = decltype(Jc)::value;
            if (PIPE) {
                constexpr int S = J & 3;
                uint32_t B = remap_combine(tq[S], lc);
                if ((unsigned)row >= (unsigned)a.H) B = 0u; // wave-uniform select: rows outside the image are zero
                remap_issue_taps(tq[S], mq[S], img, mapw, a.pitch, a.H, a.W, row + 4, xq, lc);
                remap_issue_map(mq[S], map, row + 8, a.H, a.W, lc);
                return B;
            }
            uint32_t B = finish_src4<REMAP, TINY>(q[J], (unsigned)row < (unsigned)a.H, lc);
            // refill 8 rows ahead, unconditionally (rows past the chunk are clamped into the image and simply
            // unused: a branch here would make the compiler drain the whole queue at the join)
            q[J] = fetch_src4<REMAP, TINY>(a, img, map, row + 8, xl, lc);
            return B;
        };
        // horizontal 5-sums of one source row -> vertical running sums (history in the LDS ring)
        auto hsum_update = [&](uint32_t B, int s_new, int s_old) {
            uint32_t A = lane_from_prev(B), C = lane_from_next(B);
            uint32_t sB = dot4(B, 0x01010101u, 0u);
            uint32_t h0 = dot4(A, 0x01010000u, dot4(B, 0x00010101u, 0u));
            uint32_t h1 = dot4(A, 0x01000000u, sB);
            uint32_t h2 = dot4(C, 0x00000001u, sB);
            uint32_t h3 = dot4(C, 0x00000101u, dot4(B, 0x01010100u, 0u));
            uint32_t H01 = h0 | (h1 << 16), H23 = h2 | (h3 << 16);
            uint2 old = hring[wv][s_old][lane];
            hring[wv][s_new][lane] = make_uint2(H01, H23);
            V01 += H01 - old.x; // 16-bit fields never borrow: the window sum always contains the row removed
            V23 += H23 - old.y;
        };
        // threshold row kc from the running sums -> packed horizontal 5-window counts of the thresholded row
        auto thresh_counts = [&](int kc) -> uint32_t {
            uint32_t m = (uint32_t)(a.thr_mul * taps5(kc, a.H));
            uint32_t T01 = __umul24(cx01, m), T23 = __umul24(cx23, m);
            uint32_t d01 = (V01 | 0x80008000u) - T01, d23 = (V23 | 0x80008000u) - T23;
            uint32_t t = (d01 >> 15) & 0x10001u, u = (d23 >> 15) & 0x10001u;
            uint32_t w = t | (u << 2);
            uint32_t nib = (w | (w >> 15)) & 0xfu;
            // medianBlur replicates the border: columns outside the image take the edge column's bit
            if (left_edge) {
                uint32_t e = __builtin_amdgcn_readlane(nib, 2) & 1u;
                if (xl < 0) nib = e ? 0xfu : 0u;
            }
            if (right_edge) {
                uint32_t e = (__builtin_amdgcn_readlane(nib, lane_r) >> bit_r) & 1u;
                uint32_t keep = (2u << bit_r) - 1u;
                if (lane > lane_r) nib = e ? 0xfu : 0u;
                else if (lane == lane_r) nib = (nib & keep) | (e ? (0xfu & ~keep) : 0u);
            }
            uint32_t nl = lane_from_prev(nib), nr = lane_from_next(nib);
            uint32_t win = (nl >> 2) | (nib << 2) | ((nr & 3u) << 6);
            return lut[win];
        };
        auto push_counts = [&](uint32_t c, int s_new, int s_old) {
            uint32_t cold = cring[wv][s_old][lane];
            cring[wv][s_new][lane] = c;
            Cv += c - cold;
        };
        // majority (>= 13 of 25) of output row `row`, two lanes -> one byte of the bit mask
        auto emit = [&](int row, bool on) {
            uint32_t mm = ((Cv + 0x73737373u) >> 7) & 0x01010101u;
            uint32_t t1 = mm | (mm >> 7);
            uint32_t mn = (t1 | (t1 >> 14)) & colmask;
            lacc |= (mn != 0u ? 1u : 0u) << (((on ? row : r0) - tile_r0) >> 3); // rows not yet valid have mn from a partial window: harmless superset
            uint32_t odd = lane_from_next(mn);
            uint32_t byte = (mn & 0xfu) | ((odd & 0xfu) << 4);
            // direct byte store: all loads here are global-address-space loads, so the compiler keeps counted vmcnt
            // waits around this exec-masked store and the load pipeline stays full
            if (stores && on) mrow_base[(size_t)row * row_bytes + out_byte] = (uint8_t)byte;
        };

        // ---- set-up: source rows kfirst-2 .. kfirst+2 (ring slots 3..7), first threshold row, replicated top rows ----
        hsum_update(next_row(IC<3>{}, y0), 3, 6);
        hsum_update(next_row(IC<4>{}, y0 + 1), 4, 7);
        hsum_update(next_row(IC<5>{}, y0 + 2), 5, 0);
        hsum_update(next_row(IC<6>{}, y0 + 3), 6, 1);
        hsum_update(next_row(IC<7>{}, y0 + 4), 7, 2);
        uint32_t c_cur = thresh_counts(kfirst);
        // count-ring phase chosen so that the steady loop starts at slot 0: pushes so far = 1 (+2 at the image top)
        int cj = (r0 == 0) ? 5 : 7;
        push_counts(c_cur, cj & 7, (cj + 3) & 7);
        cj++;
        for (int kk = r0 - 1; kk < ks; ++kk) { // rows above the image replicate row 0 (only the top chunk gets here)
            push_counts(c_cur, cj & 7, (cj + 3) & 7);
            cj++;
            if (kk >= r0 + 2) emit(kk - 2, true);
        }

        // ---- steady state: one source row in, one threshold row, one output row per step; unrolled by 8 so that the
        // queue registers and both ring slots are compile-time constants ----
        auto step = [&](auto Jc, int k) {
            constexpr int J = decltype(Jc)::value;
            uint32_t B = next_row(Jc, k + 2);
            hsum_update(B, J, (J + 3) & 7);
            c_cur = thresh_counts(k);
            push_counts(c_cur, J, (J + 3) & 7);
            emit(k - 2, k >= r0 + 2);
        };
        int k = ks;
        for (; k + 7 <= ke; k += 8) { // hot loop: no guards, every index static
            step(IC<0>{}, k);
            step(IC<1>{}, k + 1);
            step(IC<2>{}, k + 2);
            step(IC<3>{}, k + 3);
            step(IC<4>{}, k + 4);
            step(IC<5>{}, k + 5);
            step(IC<6>{}, k + 6);
            step(IC<7>{}, k + 7);
        }
        if (k <= ke) step(IC<0>{}, k);
        if (k + 1 <= ke) step(IC<1>{}, k + 1);
        if (k + 2 <= ke) step(IC<2>{}, k + 2);
        if (k + 3 <= ke) step(IC<3>{}, k + 3);
        if (k + 4 <= ke) step(IC<4>{}, k + 4);
        if (k + 5 <= ke) step(IC<5>{}, k + 5);
        if (k + 6 <= ke) step(IC<6>{}, k + 6);
        // ---- rows below the image replicate the last row (only the bottom chunk gets here) ----
        {
            int n_steady = ke >= ks ? ke - ks + 1 : 0;
            cj = n_steady; // slot of the next push (the steady loop started at slot 0)
            int kb = ke + 1 > ks ? ke + 1 : ks;
            for (int kk = kb; kk <= r1 + 1; ++kk) {
                push_counts(c_cur, cj & 7, (cj + 3) & 7);
                cj++;
                if (kk >= r0 + 2) emit(kk - 2, true);
            }
        }
        {   // occupancy word of this (strip, chunk): OR of the output lanes' bits
            uint32_t cellmask = 0;
            const int groups = (tile_r1 - tile_r0 + 7) >> 3;
            for (int g = 0; g < groups; g++)
                if (__ballot(out_lane && ((lacc >> g) & 1u)) != 0ull) cellmask |= 1u << g;
            // bit 31 marks a filtered tile; bits 0..16 are the groups (list form: several bands of a tile add their bits)
            if (lane == 0) { if (LIST) atomicOr(&a.cells[cell_index], cellmask | 0x80000000u); else a.cells[cell_index] = cellmask | 0x80000000u; }
        }
    } // strips / list entries
}

// ---- the row pipeline on the compact table, its source pixels staged in LDS -------------------------------------------
// filter_mask_kernel<REMAP, PIPE> above reads 8 bytes of table per pixel and gathers every tap pair from memory (ten vector
// memory instructions per 256-pixel row, eight of them 2-byte gathers): 10.6 ms per 3072 1080p images as the dense path, and
// per row just as much for the wide tiles of the sparse path.  This form reads the box kernel's 4-byte table (one 16-byte load
// per lane and row) and takes the taps from LDS: the rows of the strip are worked through in bands of up to 8; the
// rectangle of source pixels a band reads is known from a per-(row, strip) table made at set-up (rowbox), it is staged with
// coalesced dword loads, zeros outside the image (cv::remap's BORDER_CONSTANT), while the previous band is being filtered.
// Everything behind the remapped row -- horizontal sums, running vertical sums, threshold, window counts, majority -- is the
// same arithmetic as above.  Requires W % 16 == 0 (16-byte staging units), H >= 2, every slot's table in the compact format.
constexpr int ROWS_LOADS = 9;                    // 16-byte staging loads per lane and band (always all of them: see stage_issue)
constexpr int ROWS_STAGE_U = 64 * ROWS_LOADS;    // 16-byte units of source pixels per wave and band (9 KB; 61 KB of LDS per workgroup in all)
constexpr int ROWS_STAGE_DW = 4 * ROWS_STAGE_U;

// per (row, strip): box of the tap coordinates the row's pixels of the strip (columns 240 strip - 8 .. + 255) read, + 2
__global__ void rowbox_kernel(const uint32_t* __restrict__ map4, ushort4* __restrict__ rowbox, int H, int W, int n_strips)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * n_strips) return;
    const int row = i / n_strips, strip = i - row * n_strips;
    const int xa = strip * 240 - 8 > 0 ? strip * 240 - 8 : 0, xb = strip * 240 + 247 < W - 1 ? strip * 240 + 247 : W - 1;
    int x0 = 0x7fff, x1 = -0x8000, y0 = 0x7fff, y1 = -0x8000;
    for (int x = xa; x <= xb; x++) {
        const uint32_t w = map4[(size_t)row * W + x];
        const int sx = x + ((int)(w << 21) >> 21), sy = row + ((int)(w << 10) >> 21);
        x0 = sx < x0 ? sx : x0; x1 = sx + 1 > x1 ? sx + 1 : x1; y0 = sy < y0 ? sy : y0; y1 = sy + 1 > y1 ? sy + 1 : y1;
    }
    rowbox[i] = make_ushort4((unsigned short)(x0 + 2), (unsigned short)(x1 + 2), (unsigned short)(y0 + 2), (unsigned short)(y1 + 2));
}
void launch_rowbox(const uint32_t* map4, ushort4* rowbox, int H, int W, int n_strips, hipStream_t s)
{
    const int n = H * n_strips;
    hipLaunchKernelGGL(rowbox_kernel, dim3((n + 63) / 64), dim3(64), 0, s, map4, rowbox, H, W, n_strips);
}

// source rectangle of a band in 16-byte units: origin (sxa a multiple of 16), pitch SP bytes = q units, SR rows, n units; its part
// inside the image: origin (ux0, iy0), iq units x iSR rows, first / last unit at li0 / llast of the rectangle
typedef uint32_t rows_u32x4 __attribute__((ext_vector_type(4))); // (a native vector: an array of HIP's uint4 filled by memcpy stays in scratch memory)
typedef rows_u32x4 rows_u32x4_any __attribute__((aligned(1)));   // ... at any address (unaligned access is enabled on amdhsa: one global_load_dwordx4)
struct BandRect { int sxa, sya, SP, SR, q, n; int ux0, iy0, iq, in, li0, llast; bool staged, interior; }; // source rectangle of a band: origin, pitch (bytes), rows, dwords per row, dwords

template <bool LIST>
__global__ __launch_bounds__(256) void filter_rows_staged_kernel(FilterArgs a)
{
    __shared__ uint32_t lut[256];
    __shared__ uint2 hring[4][8][64];
    __shared__ uint32_t cring[4][8][64];
    __shared__ rows_u32x4 sbuf[4][ROWS_STAGE_U];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: keeps the row loop scalar

    int slot = 0, image = 0, chunk = 0;
    uint32_t n_list = 0;
    if (LIST) {
        n_list = *a.n_tiles;
        n_list = n_list < a.cap_tiles ? n_list : a.cap_tiles;
        if ((uint32_t)(blockIdx.x * 4 + wv) >= n_list) return;
    } else {
        const TileId tid_ = decode_tile(a, blockIdx.x);
        if (!tid_.valid) return;
        slot = tid_.slot; image = tid_.image;
        chunk = tid_.cgroup * 4 + wv;
        if (chunk * a.rows_per_chunk >= a.H) return;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) { // lut[w]: byte k = number of set bits among bits k..k+4 of the 8-bit window w (every wave writes all of it)
        uint32_t i = (uint32_t)(lane + 64 * e), v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)__popc((i >> k) & 0x1fu) << (8 * k);
        lut[i] = v;
    }
    rows_u32x4* const Sb = sbuf[wv];
    const uint8_t* const Sbytes = (const uint8_t*)Sb;
    // kernel arguments as plain scalars (a struct captured by the lambdas below would be kept in scratch memory)
    const int H = a.H, W = a.W, Hm1 = a.H - 1, pitch = a.pitch, stage_units = a.stage_dw >> 2, n_strips = a.n_strips, thr_mul = a.thr_mul;
    const int rows_per_chunk = a.rows_per_chunk, n_cgroups = a.n_cgroups, cam_mod = a.cam_mod, words_per_row = a.words_per_row;
    const uint8_t* __restrict__ const a_src = a.src; const size_t a_image_stride = a.image_stride;
    const uint32_t* __restrict__ const a_map4 = a.map4; const ushort4* __restrict__ const a_rowbox = a.rowbox;
    uint32_t* __restrict__ const a_mask = a.mask; uint32_t* __restrict__ const a_cells = a.cells; const uint4* __restrict__ const a_tiles = a.tiles;
    const uint32_t it_first = LIST ? (uint32_t)(blockIdx.x * 4 + wv) : 0u, it_end = LIST ? n_list : (uint32_t)n_strips;
    const uint32_t it_step = LIST ? gridDim.x * 4u : 1u;
    for (uint32_t it = it_first; it < it_end; it += it_step) {
        int strip = (int)it, r0e = 0, r1e = 0x7fffffff;
        if (LIST) {
            const uint4 e = a_tiles[it];
            image = __builtin_amdgcn_readfirstlane((int)e.x);
            const int tile = __builtin_amdgcn_readfirstlane((int)e.y);
            r0e = __builtin_amdgcn_readfirstlane((int)e.z); r1e = __builtin_amdgcn_readfirstlane((int)e.w) + 1;
            chunk = tile / n_strips; strip = tile - chunk * n_strips;
            slot = image % cam_mod;
        }
        const int tile_r0 = chunk * rows_per_chunk;              // the tile's mask rows [tile_r0, tile_r1)
        const int tile_r1 = tile_r0 + rows_per_chunk < H ? tile_r0 + rows_per_chunk : H;
        const size_t cell_index = ((size_t)image * n_cgroups * 4 + chunk) * n_strips + strip;
        const int r0 = r0e > tile_r0 ? r0e : tile_r0, r1 = r1e < tile_r1 ? r1e : tile_r1; // the rows filtered
        if (LIST && r0 >= r1) continue; // an empty band
        const int xbase = strip * 240 - 8;

        int kfirst = r0 - 2;
        kfirst = kfirst < 0 ? 0 : (kfirst > Hm1 ? Hm1 : kfirst);
        const int ks = r0 - 1 > 1 ? r0 - 1 : 1;                 // steady range: every iteration slides one source row
        const int ke = r1 + 1 < Hm1 ? r1 + 1 : Hm1;

#pragma unroll
        for (int s = 0; s < 8; s++) {
            hring[wv][s][lane] = make_uint2(0u, 0u);
            cring[wv][s][lane] = 0u;
        }
        const uint8_t* __restrict__ img = a_src + (size_t)image * a_image_stride;
        const uint32_t* __restrict__ map4 = a_map4 + (size_t)slot * H * W;
        const ushort4* __restrict__ rbox = a_rowbox + (size_t)slot * H * n_strips + strip; // row r: rbox[r * n_strips]
        uint8_t* __restrict__ mrow_base = (uint8_t*)(a_mask + (size_t)image * H * words_per_row);
        const int row_bytes = words_per_row * 4;
        const int xl = xbase + 4 * lane;
        uint32_t cx01, cx23, colmask = 0;
        {
            int c0 = taps5(xl, W), c1 = taps5(xl + 1, W), c2 = taps5(xl + 2, W), c3 = taps5(xl + 3, W);
            cx01 = (uint32_t)(c0 & 0xffff) | ((uint32_t)c1 << 16);
            cx23 = (uint32_t)(c2 & 0xffff) | ((uint32_t)c3 << 16);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((unsigned)(xl + k) < (unsigned)W) colmask |= 1u << k;
        }
        const LaneCols lc = lane_cols(xl, W);
        const bool left_edge = xbase < 0;
        const bool right_edge = xbase + 255 >= W;
        const int lane_r = (W - 1 - xbase) >> 2, bit_r = (W - 1 - xbase) & 3; // lane / bit of column W-1
        const int out_byte = strip * 30 + ((lane - 2) >> 1);
        const bool stores = ((lane & 1) == 0) && lane >= 2 && lane <= 60 && out_byte < ((W + 7) >> 3) && out_byte < row_bytes;
        uint32_t V01 = 0, V23 = 0, Cv = 0;
        uint32_t lacc = 0;
        const bool out_lane = lane >= 2 && lane <= 61;
        const int y0 = kfirst - 2;

        // ---- bands: rectangle, staging, table words ----------------------------------------------------------------------
        // a lane's share of the rows' boxes (lanes 0..nr-1 hold one row each); reduced when the band is about to be staged
        auto rect_load = [&](int rb, int nr) __attribute__((always_inline)) -> ushort4 {
            int r = rb + (lane < nr ? lane : 0);
            r = r < 0 ? 0 : (r > Hm1 ? Hm1 : r);
            return rbox[(size_t)r * n_strips];
        };
        auto rect_reduce = [&](ushort4 p) __attribute__((always_inline)) -> BandRect {
            int xa = p.x, xb = p.y, ya = p.z, yb = p.w; // lanes beyond the band's rows hold a copy of its first row
#pragma unroll
            for (int d = 1; d <= 4; d <<= 1) {
                const int oxa = __shfl_xor(xa, d), oxb = __shfl_xor(xb, d), oya = __shfl_xor(ya, d), oyb = __shfl_xor(yb, d);
                xa = oxa < xa ? oxa : xa; xb = oxb > xb ? oxb : xb; ya = oya < ya ? oya : ya; yb = oyb > yb ? oyb : yb;
            }
            BandRect R;
            R.sxa = (__builtin_amdgcn_readfirstlane(xa) - 2) & ~15;
            const int sxb = __builtin_amdgcn_readfirstlane(xb) - 2;
            R.sya = __builtin_amdgcn_readfirstlane(ya) - 2;
            const int syb = __builtin_amdgcn_readfirstlane(yb) - 2;
            R.SP = (sxb - R.sxa + 16) & ~15; R.SR = syb - R.sya + 1; R.q = R.SP >> 4; R.n = R.SR * R.q;
            R.staged = R.n < stage_units; // (one unit is kept free: where the loads of a rectangle wholly outside the image end up)
            // the part inside the image: W % 16 == 0 and an origin that is a multiple of 16 put every unit entirely inside or outside
            const int ux1 = R.sxa + R.SP < W ? R.sxa + R.SP : W, iy1 = R.sya + R.SR < H ? R.sya + R.SR : H;
            R.ux0 = R.sxa > 0 ? R.sxa : 0; R.iy0 = R.sya > 0 ? R.sya : 0;
            R.iq = (ux1 - R.ux0) >> 4;
            const int iSR = iy1 - R.iy0;
            R.interior = R.iq == R.q && iSR == R.SR;
            if (R.iq <= 0 || iSR <= 0) { // nothing of it inside the image: one unit from somewhere valid, parked behind the rectangle
                R.iq = 1; R.in = 1; R.ux0 = 0; R.iy0 = 0; R.li0 = R.n; R.llast = R.n;
            } else {
                R.in = R.iq * iSR;
                R.li0 = (R.iy0 - R.sya) * R.q + ((R.ux0 - R.sxa) >> 4);
                R.llast = R.li0 + (iSR - 1) * R.q + R.iq - 1;
            }
            return R;
        };
        // The staging loads of a band: ROWS_LOADS 16-byte loads per lane, ALWAYS all of them and never inside a branch (the
        // counter that orders vector memory operations is counted at compile time: loads inside a branch make every later
        // wait a wait for all of them), all in flight while the previous band is filtered.  Lane -> units lane, lane + 64, ... of
        // the rectangle's part inside the image, row-major; the address and the place in LDS walk on by wave-uniform steps with a
        // carry into the next row (no division, no clamps); units past the end repeat the last one (same address, same place).
        struct StageWalk { int c; uint32_t goff; int li; };
        auto stage_walk = [&](const BandRect& R, StageWalk& w, int& rem, uint32_t& gstep, uint32_t& gcarry, int& lstep, int& lcarry, uint32_t& glast) __attribute__((always_inline)) {
            const float rcpd = __builtin_amdgcn_rcpf((float)R.iq);
            const int qr = (int)(64.5f * rcpd); // 64 = qr * iq + rem
            rem = 64 - qr * R.iq;
            const int r0_ = (int)(((float)lane + 0.5f) * rcpd);
            w.c = lane - r0_ * R.iq;
            w.goff = (uint32_t)(R.iy0 + r0_) * (uint32_t)pitch + (uint32_t)(R.ux0 + 16 * w.c);
            w.li = R.li0 + r0_ * R.q + w.c;
            gstep = (uint32_t)qr * (uint32_t)pitch + 16u * (uint32_t)rem; gcarry = (uint32_t)pitch - 16u * (uint32_t)R.iq;
            lstep = qr * R.q + rem; lcarry = R.q - R.iq;
            const int rl = (R.in - 1) / R.iq; // (scalar)
            glast = (uint32_t)(R.iy0 + rl) * (uint32_t)pitch + (uint32_t)(R.ux0 + 16 * (R.in - 1 - rl * R.iq));
        };
        auto stage_issue = [&](const BandRect& R, rows_u32x4 (&v)[ROWS_LOADS]) __attribute__((always_inline)) {
            StageWalk w; int rem, lstep, lcarry; uint32_t gstep, gcarry, glast;
            stage_walk(R, w, rem, gstep, gcarry, lstep, lcarry, glast);
#pragma unroll
            for (int u = 0; u < ROWS_LOADS; u++) {
                const uint32_t g = w.goff < glast ? w.goff : glast;
                v[u] = *(const rows_u32x4_any*)(img + g);
                w.c += rem; w.goff += gstep;
                if (w.c >= R.iq) { w.c -= R.iq; w.goff += gcarry; }
            }
        };
        auto stage_write = [&](const BandRect& R, const rows_u32x4 (&v)[ROWS_LOADS]) __attribute__((always_inline)) {
            if (!R.interior) { // units outside the image read 0 (cv::remap's BORDER_CONSTANT): clear, then the inside part on top
#pragma unroll
                for (int u = 0; u < ROWS_LOADS; u++) Sb[lane + 64 * u] = rows_u32x4{0u, 0u, 0u, 0u};
            }
            StageWalk w; int rem, lstep, lcarry; uint32_t gstep, gcarry, glast;
            stage_walk(R, w, rem, gstep, gcarry, lstep, lcarry, glast);
#pragma unroll
            for (int u = 0; u < ROWS_LOADS; u++) {
                Sb[w.li < R.llast ? w.li : R.llast] = v[u];
                w.c += rem; w.li += lstep;
                if (w.c >= R.iq) { w.c -= R.iq; w.li += lcarry; }
            }
        };
        auto table_issue = [&](uint4& tw, int row) __attribute__((always_inline)) {
            const int rc = row < 0 ? 0 : (row > Hm1 ? Hm1 : row);
            __builtin_memcpy(&tw, map4 + ((uint32_t)rc * (uint32_t)W + (uint32_t)lc.addr_x), 16);
        };
        // one remapped row of the strip: the lane's four pixels, blended exactly as cv::remap's fixed point does
        // (STAGED is a compile-time flag chosen once per band: a branch inside every row would cut the unrolled rows into separate
        // basic blocks, and the tap reads of one row could no longer be scheduled under the arithmetic of the previous one)
        auto blend_row = [&](auto staged_c, const uint4& tw, int row, const BandRect& R) __attribute__((always_inline)) -> uint32_t {
            constexpr bool STAGED = decltype(staged_c)::value;
            const int rc = row < 0 ? 0 : (row > Hm1 ? Hm1 : row);
            const uint32_t ww[4] = {tw.x, tw.y, tw.z, tw.w};
            uint32_t B = 0;
            if (STAGED) {
                const int rowbase = __mul24(rc - R.sya, R.SP) + (lc.addr_x - R.sxa);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w = ww[k];
                    const int dx = (int)(w << 21) >> 21, dy = (int)(w << 10) >> 21;
                    const uint32_t fa = (w >> 22) & 31u, fb = w >> 27;
                    const int A0 = __mul24(dy, R.SP) + (rowbase + k) + dx;
                    const uint32_t p00 = Sbytes[A0], p01 = Sbytes[A0 + 1], p10 = Sbytes[A0 + R.SP], p11 = Sbytes[A0 + R.SP + 1];
                    const uint32_t wa = 32u - fa, wb = 32u - fb;
                    const uint32_t top = __umul24(p00, wa) + __umul24(p01, fa), bot = __umul24(p10, wa) + __umul24(p11, fa);
                    B |= ((__umul24(top, wb) + __umul24(bot, fb) + 512u) >> 10) << (8 * k); // == (sum of 32*w*p + 2^14) >> 15
                }
            } else { // the band's source rectangle outgrew the buffer (strong local distortion): taps from memory
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w = ww[k];
                    const int dx = (int)(w << 21) >> 21, dy = (int)(w << 10) >> 21;
                    const uint32_t fa = (w >> 22) & 31u, fb = w >> 27;
                    const int sx = lc.addr_x + k + dx, sy = rc + dy;
                    const int sx0 = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx), sx1 = sx + 1 < 0 ? 0 : (sx + 1 > W - 1 ? W - 1 : sx + 1);
                    const int sy0 = sy < 0 ? 0 : (sy > Hm1 ? Hm1 : sy), sy1 = sy + 1 < 0 ? 0 : (sy + 1 > Hm1 ? Hm1 : sy + 1);
                    const uint32_t o0 = (uint32_t)sy0 * (uint32_t)pitch, o1 = (uint32_t)sy1 * (uint32_t)pitch;
                    const uint32_t t00 = img[o0 + (uint32_t)sx0], t01 = img[o0 + (uint32_t)sx1], t10 = img[o1 + (uint32_t)sx0], t11 = img[o1 + (uint32_t)sx1];
                    const bool c0 = sx0 == sx, c1 = sx1 == sx + 1, q0 = sy0 == sy, q1 = sy1 == sy + 1;
                    const uint32_t p00 = (c0 && q0) ? t00 : 0u, p01 = (c1 && q0) ? t01 : 0u, p10 = (c0 && q1) ? t10 : 0u, p11 = (c1 && q1) ? t11 : 0u;
                    const uint32_t wa = 32u - fa, wb = 32u - fb;
                    const uint32_t top = __umul24(p00, wa) + __umul24(p01, fa), bot = __umul24(p10, wa) + __umul24(p11, fa);
                    B |= ((__umul24(top, wb) + __umul24(bot, fb) + 512u) >> 10) << (8 * k);
                }
            }
            return (unsigned)row < (unsigned)H ? (B & lc.bytemask) : 0u; // rows and columns outside the image do not exist
        };
        // ---- the pipeline behind the remapped row: as in filter_mask_kernel ---------------------------------------------
        auto hsum_update = [&](uint32_t B, int s_new, int s_old) __attribute__((always_inline)) {
            uint32_t A = lane_from_prev(B), C = lane_from_next(B);
            uint32_t sB = dot4(B, 0x01010101u, 0u);
            uint32_t h0 = dot4(A, 0x01010000u, dot4(B, 0x00010101u, 0u));
            uint32_t h1 = dot4(A, 0x01000000u, sB);
            uint32_t h2 = dot4(C, 0x00000001u, sB);
            uint32_t h3 = dot4(C, 0x00000101u, dot4(B, 0x01010100u, 0u));
            uint32_t H01 = h0 | (h1 << 16), H23 = h2 | (h3 << 16);
            uint2 old = hring[wv][s_old][lane];
            hring[wv][s_new][lane] = make_uint2(H01, H23);
            V01 += H01 - old.x; // 16-bit fields never borrow: the window sum always contains the row removed
            V23 += H23 - old.y;
        };
        auto thresh_counts = [&](int kc) __attribute__((always_inline)) -> uint32_t {
            uint32_t m = (uint32_t)(thr_mul * taps5(kc, H));
            uint32_t T01 = __umul24(cx01, m), T23 = __umul24(cx23, m);
            uint32_t d01 = (V01 | 0x80008000u) - T01, d23 = (V23 | 0x80008000u) - T23;
            uint32_t t = (d01 >> 15) & 0x10001u, u = (d23 >> 15) & 0x10001u;
            uint32_t w = t | (u << 2);
            uint32_t nib = (w | (w >> 15)) & 0xfu;
            if (left_edge) { // medianBlur replicates the border: columns outside the image take the edge column's bit
                uint32_t e = __builtin_amdgcn_readlane(nib, 2) & 1u;
                if (xl < 0) nib = e ? 0xfu : 0u;
            }
            if (right_edge) {
                uint32_t e = (__builtin_amdgcn_readlane(nib, lane_r) >> bit_r) & 1u;
                uint32_t keep = (2u << bit_r) - 1u;
                if (lane > lane_r) nib = e ? 0xfu : 0u;
                else if (lane == lane_r) nib = (nib & keep) | (e ? (0xfu & ~keep) : 0u);
            }
            uint32_t nl = lane_from_prev(nib), nr = lane_from_next(nib);
            uint32_t win = (nl >> 2) | (nib << 2) | ((nr & 3u) << 6);
            return lut[win];
        };
        auto push_counts = [&](uint32_t c, int s_new, int s_old) __attribute__((always_inline)) {
            uint32_t cold = cring[wv][s_old][lane];
            cring[wv][s_new][lane] = c;
            Cv += c - cold;
        };
        auto emit = [&](int row, bool on) __attribute__((always_inline)) {
            uint32_t mm = ((Cv + 0x73737373u) >> 7) & 0x01010101u;
            uint32_t t1 = mm | (mm >> 7);
            uint32_t mn = (t1 | (t1 >> 14)) & colmask;
            lacc |= (mn != 0u ? 1u : 0u) << (((on ? row : r0) - tile_r0) >> 3);
            uint32_t odd = lane_from_next(mn);
            uint32_t byte = (mn & 0xfu) | ((odd & 0xfu) << 4);
            if (stores && on) mrow_base[(size_t)row * row_bytes + out_byte] = (uint8_t)byte;
        };

        // ---- band 0: the five rows y0 .. y0 + 4 of the set-up (ring slots 3 .. 7) ------------------------------------------
        uint4 tc[8], tn[8];  // table words of the band being filtered / of the next one
        rows_u32x4 sv[ROWS_LOADS]; // staging loads in flight
        BandRect Rc = rect_reduce(rect_load(y0, 5));
        stage_issue(Rc, sv);
#pragma unroll
        for (int j = 0; j < 5; j++) table_issue(tc[3 + j], y0 + j);
        // band 1 = the first rows of the steady loop: y0 + 5 = ks + 2 onwards
        int nr_next = ke - ks + 1 < 8 ? ke - ks + 1 : 8;
        ushort4 part = rect_load(ks + 2, nr_next > 0 ? nr_next : 1);
        stage_write(Rc, sv);
        // the next band's loads are issued before the current one is filtered, and land in LDS after it.  Every band does this,
        // the last one too (its successor's rows are clamped into the image and never used): no branch around a load
        BandRect Rn = Rc;
        auto prefetch = [&](int first_row, int nr) __attribute__((always_inline)) { // rows first_row .. first_row + nr - 1 -> tn / sv
            Rn = rect_reduce(part);
            stage_issue(Rn, sv);
#pragma unroll
            for (int j = 0; j < 8; j++) table_issue(tn[j], first_row + (j < nr ? j : nr - 1));
        };
        auto commit = [&]() __attribute__((always_inline)) { // the band just filtered has read its last tap: the next one moves in
            stage_write(Rn, sv);
#pragma unroll
            for (int j = 0; j < 8; j++) tc[j] = tn[j];
            Rc = Rn;
        };
        prefetch(ks + 2, nr_next > 0 ? nr_next : 1);
        {
            const int n2 = ke - (ks + 8) + 1 < 8 ? ke - (ks + 8) + 1 : 8;
            part = rect_load(ks + 10, n2 > 0 ? n2 : 1); // (the band after that one: its box loads have a whole band's time)
        }
        auto setup_rows = [&](auto staged_c) __attribute__((always_inline)) {
            hsum_update(blend_row(staged_c, tc[3], y0, Rc), 3, 6);
            hsum_update(blend_row(staged_c, tc[4], y0 + 1, Rc), 4, 7);
            hsum_update(blend_row(staged_c, tc[5], y0 + 2, Rc), 5, 0);
            hsum_update(blend_row(staged_c, tc[6], y0 + 3, Rc), 6, 1);
            hsum_update(blend_row(staged_c, tc[7], y0 + 4, Rc), 7, 2);
        };
        if (Rc.staged) setup_rows(std::true_type{}); else setup_rows(std::false_type{});
        uint32_t c_cur = thresh_counts(kfirst);
        int cj = (r0 == 0) ? 5 : 7; // count-ring phase chosen so that the steady loop starts at slot 0
        push_counts(c_cur, cj & 7, (cj + 3) & 7);
        cj++;
        for (int kk = r0 - 1; kk < ks; ++kk) { // rows above the image replicate row 0 (only the top chunk gets here)
            push_counts(c_cur, cj & 7, (cj + 3) & 7);
            cj++;
            if (kk >= r0 + 2) emit(kk - 2, true);
        }
        // ---- steady state: bands of 8 rows; one source row in, one threshold row, one output row per step -------------------
        auto step = [&](auto staged_c, auto Jc, int k) __attribute__((always_inline)) {
            constexpr int J = decltype(Jc)::value;
            const uint32_t B = blend_row(staged_c, tc[J], k + 2, Rc);
            hsum_update(B, J, (J + 3) & 7);
            c_cur = thresh_counts(k);
            push_counts(c_cur, J, (J + 3) & 7);
            emit(k - 2, k >= r0 + 2);
        };
        auto band_steps = [&](auto staged_c, int k, int nst) __attribute__((always_inline)) {
            if (nst == 8) { // one basic block: the rows' LDS reads and arithmetic interleave
                step(staged_c, IC<0>{}, k); step(staged_c, IC<1>{}, k + 1); step(staged_c, IC<2>{}, k + 2); step(staged_c, IC<3>{}, k + 3);
                step(staged_c, IC<4>{}, k + 4); step(staged_c, IC<5>{}, k + 5); step(staged_c, IC<6>{}, k + 6); step(staged_c, IC<7>{}, k + 7);
            } else {
                step(staged_c, IC<0>{}, k);
                if (nst > 1) step(staged_c, IC<1>{}, k + 1);
                if (nst > 2) step(staged_c, IC<2>{}, k + 2);
                if (nst > 3) step(staged_c, IC<3>{}, k + 3);
                if (nst > 4) step(staged_c, IC<4>{}, k + 4);
                if (nst > 5) step(staged_c, IC<5>{}, k + 5);
                if (nst > 6) step(staged_c, IC<6>{}, k + 6);
            }
        };
        for (int k = ks; k <= ke; k += 8) {
            commit(); // (the band of this iteration)
            const int nst = ke - k + 1 < 8 ? ke - k + 1 : 8;
            {
                const int n1 = ke - (k + 8) + 1 < 8 ? ke - (k + 8) + 1 : 8, n2 = ke - (k + 16) + 1 < 8 ? ke - (k + 16) + 1 : 8;
                prefetch(k + 10, n1 > 0 ? n1 : 1);
                part = rect_load(k + 18, n2 > 0 ? n2 : 1);
            }
            if (Rc.staged) band_steps(std::true_type{}, k, nst); else band_steps(std::false_type{}, k, nst);
        }
        // ---- rows below the image replicate the last row (only the bottom chunk gets here) ----
        {
            int n_steady = ke >= ks ? ke - ks + 1 : 0;
            cj = n_steady; // slot of the next push (the steady loop started at slot 0)
            int kb = ke + 1 > ks ? ke + 1 : ks;
            for (int kk = kb; kk <= r1 + 1; ++kk) {
                push_counts(c_cur, cj & 7, (cj + 3) & 7);
                cj++;
                if (kk >= r0 + 2) emit(kk - 2, true);
            }
        }
        {   // occupancy word of this (strip, chunk): OR of the output lanes' bits
            uint32_t cellmask = 0;
            const int groups = (tile_r1 - tile_r0 + 7) >> 3;
            for (int g = 0; g < groups; g++)
                if (__ballot(out_lane && ((lacc >> g) & 1u)) != 0ull) cellmask |= 1u << g;
            if (lane == 0) { if (LIST) atomicOr(&a_cells[cell_index], cellmask | 0x80000000u); else a_cells[cell_index] = cellmask | 0x80000000u; }
        }
    } // strips / list entries
}

// ---- map construction: cv::initUndistortRectifyMap as called by cv::undistort (stripe by stripe) -------------
// One thread per image row; the _x accumulation along the row is sequential exactly as in OpenCV.
__global__ void undistort_map_kernel(MapArgs m)
{
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m.H) return;
    int stripe0 = 4096 / (m.W > 1 ? m.W : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > m.H) stripe0 = m.H;
    int ys = (row / stripe0) * stripe0, i = row - ys;
    double A[9];
    for (int k = 0; k < 9; k++) A[k] = m.K[k];
    double fx = A[0], fy = A[4], u0 = A[2], v0 = A[5];
    A[5] = v0 - ys;
    double ir[9];
    {
        double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
                     A[2] * (A[3] * A[7] - A[4] * A[6]);
        double d = 1.0 / det;
        ir[0] = (A[4] * A[8] - A[5] * A[7]) * d;
        ir[1] = (A[2] * A[7] - A[1] * A[8]) * d;
        ir[2] = (A[1] * A[5] - A[2] * A[4]) * d;
        ir[3] = (A[5] * A[6] - A[3] * A[8]) * d;
        ir[4] = (A[0] * A[8] - A[2] * A[6]) * d;
        ir[5] = (A[2] * A[3] - A[0] * A[5]) * d;
        ir[6] = (A[3] * A[7] - A[4] * A[6]) * d;
        ir[7] = (A[1] * A[6] - A[0] * A[7]) * d;
        ir[8] = (A[0] * A[4] - A[1] * A[3]) * d;
    }
    double k1 = m.dist[0], k2 = m.dist[1], p1 = m.dist[2], p2 = m.dist[3], k3 = m.dist[4];
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    uint32_t* out = m.map + (size_t)row * m.W;
    uint32_t* outw = m.mapw + (size_t)row * m.W;
    uint32_t* out4 = m.map4 + (size_t)row * m.W;
    uint32_t flags = 0;
    for (int j = 0; j < m.W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        double w = 1. / _w, x = _x * w, y = _y * w;
        double x2 = x * x, y2 = y * y;
        double r2 = x2 + y2, _2xy = 2 * x * y;
        double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
        double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
        double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
        double u = fx * xd + u0;
        double v = fy * yd + v0;
        double ru = __builtin_rint(u * 32), rv = __builtin_rint(v * 32); // round half to even (cvRound)
        // saturate_cast<int>, then the (short) casts of the integer parts that cv::remap's fixed-point map applies
        ru = ru > 2147483647.0 ? 2147483647.0 : (ru < -2147483648.0 ? -2147483648.0 : ru);
        rv = rv > 2147483647.0 ? 2147483647.0 : (rv < -2147483648.0 ? -2147483648.0 : rv);
        int iu = (int)ru, iv = (int)rv;
        int sx = (int)(int16_t)(iu >> 5), sy = (int)(int16_t)(iv >> 5);
        uint32_t a = iu & 31, b = iv & 31;
        // 2x2 tap window clamped into the image; taps that fall outside read 0 (BORDER_CONSTANT): weight 0
        int sxc = sx < 0 ? 0 : (sx > m.W - 2 ? m.W - 2 : sx), syc = sy < 0 ? 0 : (sy > m.H - 2 ? m.H - 2 : sy);
        if (sxc < 0) sxc = 0;
        if (syc < 0) syc = 0;
        int ddx = sx - sxc, ddy = sy - syc;
        uint32_t wx0 = ddx == 0 ? 32u - a : (ddx == -1 ? a : 0u), wx1 = ddx == 0 ? a : (ddx == 1 ? 32u - a : 0u);
        uint32_t wy0 = ddy == 0 ? 32u - b : (ddy == -1 ? b : 0u), wy1 = ddy == 0 ? b : (ddy == 1 ? 32u - b : 0u);
        if (sxc + 1 > m.W - 1) wx1 = 0; // one-column / one-row images: the second tap does not exist
        if (syc + 1 > m.H - 1) wy1 = 0;
        int dx = sxc - j, dy = syc - row; // |.| < 32768 because both ends are inside the image
        uint32_t wq = wx0 | (wx1 << 8) | (wy1 << 16) | (wy0 << 24);
        if (iu != 32 * j || iv != 32 * row) flags |= 1u; // anything but the identity map
        out[j] = ((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16);
        outw[j] = wq;
        // compact table of the box kernel: unclamped tap origin, limited to [-2, W] x [-2, H] (from there on all four
        // taps lie outside the image and read 0 whatever the fractions are), as 11-bit displacements + 5-bit fractions
        const int sx2 = sx < -2 ? -2 : (sx > m.W ? m.W : sx), sy2 = sy < -2 ? -2 : (sy > m.H ? m.H : sy);
        const int dx4 = sx2 - j, dy4 = sy2 - row;
        if (dx4 < -1024 || dx4 > 1023 || dy4 < -1024 || dy4 > 1023) flags |= 2u;
        out4[j] = ((uint32_t)dx4 & 0x7ffu) | (((uint32_t)dy4 & 0x7ffu) << 11) | (a << 22) | (b << 27);
    }
    if (flags) atomicOr(m.flags, flags);
}

// ---- stand-alone stages (drop-in surface of lib/CudaOperations.py and lib/ImageOperations.py) ---------------

// fast_cuda_blur: floor(S/c) over the in-bounds taps of a ksize x ksize window
__global__ void box_blur_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                int spitch, int dpitch, int ksize)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    int k = ksize / 2;
    int y0 = y - k < 0 ? 0 : y - k, y1 = y + k > H - 1 ? H - 1 : y + k;
    int x0 = x - k < 0 ? 0 : x - k, x1 = x + k > W - 1 ? W - 1 : x + k;
    uint32_t s = 0;
    for (int yy = y0; yy <= y1; yy++)
        for (int xx = x0; xx <= x1; xx++) s += src[(size_t)yy * spitch + xx];
    dst[(size_t)y * dpitch + x] = (uint8_t)(s / (uint32_t)((y1 - y0 + 1) * (x1 - x0 + 1)));
}

// cv.undistort of one image through the packed map
__global__ void undistort_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                 int spitch, int dpitch, const uint32_t* __restrict__ map,
                                 const uint32_t* __restrict__ mapw)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    dst[(size_t)y * dpitch + x] = (uint8_t)remap_px(src, spitch, map[(size_t)y * W + x], mapw[(size_t)y * W + x], x, y);
}

// bit mask -> {0,255} image
__global__ void mask_expand_kernel(const uint32_t* __restrict__ mask, int words_per_row, uint8_t* __restrict__ dst,
                                   int H, int W, int dpitch)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    uint32_t w = mask[(size_t)y * words_per_row + (x >> 5)];
    dst[(size_t)y * dpitch + x] = ((w >> (x & 31)) & 1u) ? 255 : 0;
}

// image_filter_cpu order: exact 5x5 median (BORDER_REPLICATE) then threshold
__global__ void median5_threshold_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                         int spitch, int dpitch, int ithresh, int apply_threshold)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    uint8_t v[25];
    int n = 0;
    for (int dy = -2; dy <= 2; dy++) {
        int yy = y + dy;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        for (int dx = -2; dx <= 2; dx++) {
            int xx = x + dx;
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            v[n++] = src[(size_t)yy * spitch + xx];
        }
    }
    // median = the value with exactly 12 smaller-or-equal-ranked elements before it (rank by value, then index)
    int med = 0;
    for (int i = 0; i < 25; i++) {
        int rank = 0;
        for (int j = 0; j < 25; j++) rank += (v[j] < v[i]) || (v[j] == v[i] && j < i);
        if (rank == 12) med = v[i];
    }
    dst[(size_t)y * dpitch + x] = apply_threshold ? (med > ithresh ? 255 : 0) : (uint8_t)med;
}

// fast_cuda_demosaic (reference lib/CudaOperations.py:43-100)
__global__ void demosaic_kernel(const uint8_t* __restrict__ bayer, uint8_t* __restrict__ bgr, int H, int W,
                                int spitch)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    auto gp = [&](int xx, int yy) -> int {
        return ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? (int)bayer[(size_t)yy * spitch + xx] : 0;
    };
    int cross = gp(x - 1, y) + gp(x + 1, y) + gp(x, y - 1) + gp(x, y + 1);
    int diag = gp(x - 1, y - 1) + gp(x + 1, y - 1) + gp(x - 1, y + 1) + gp(x + 1, y + 1);
    int horiz = gp(x - 1, y) + gp(x + 1, y), vert = gp(x, y - 1) + gp(x, y + 1);
    int r, g, b, c = gp(x, y);
    if (!(y & 1) && !(x & 1)) { b = c; g = cross / 4; r = diag / 4; }
    else if (!(y & 1)) { g = c; b = horiz / 2; r = vert / 2; }
    else if (!(x & 1)) { g = c; r = horiz / 2; b = vert / 2; }
    else { r = c; g = cross / 4; b = diag / 4; }
    uint8_t* o = bgr + ((size_t)y * W + x) * 3;
    o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
}

// total blend weight every source pixel carries over all output pixels (scatter), for the dark-tile bound
__global__ void remap_weight_scatter_kernel(StatArgs a)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.W || y >= a.H) return;
    uint32_t m = a.map[(size_t)y * a.W + x], w = a.mapw[(size_t)y * a.W + x];
    int sx = x + (int)(int16_t)(m & 0xffffu), sy = y + ((int)m >> 16);
    uint32_t wx0 = w & 0xffu, wx1 = (w >> 8) & 0xffu, wy1 = (w >> 16) & 0xffu, wy0 = w >> 24;
    uint32_t* p = a.acc + (size_t)sy * a.W + sx;
    if (wx0 * wy0) atomicAdd(p, wx0 * wy0);
    if (wx1 * wy0) atomicAdd(p + 1, wx1 * wy0);
    if (wx0 * wy1) atomicAdd(p + a.W, wx0 * wy1);
    if (wx1 * wy1) atomicAdd(p + a.W + 1, wx1 * wy1);
    // reach of every 8x8 source cell: the bounding box of the output pixels that read it with a nonzero weight
    const int ncx = (a.W + 7) >> 3;
    auto touch = [&](int tx, int ty) {
        int* r = a.reach + 4 * ((ty >> 3) * ncx + (tx >> 3));
        atomicMin(r, x); atomicMax(r + 1, x); atomicMin(r + 2, y); atomicMax(r + 3, y);
    };
    if (wx0 * wy0) touch(sx, sy);
    if (wx1 * wy0) touch(sx + 1, sy);
    if (wx0 * wy1) touch(sx, sy + 1);
    if (wx1 * wy1) touch(sx + 1, sy + 1);
}
__global__ void remap_stats_kernel(StatArgs a)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.W || y >= a.H) return;
    atomicMax(&a.stats[0], a.acc[(size_t)y * a.W + x]);
    int x0 = 0x7fff, x1 = -1, y0 = 0x7fff, y1 = -1; // extent of the nonzero-weight taps of the 5x5 window around (x,y)
    // windows cut by the image border have fewer than 25 taps, hence a smaller bound: their source cells are marked
    const uint32_t cut = ((x < 2 || x >= a.W - 2) ? 1u : 0u) + ((y < 2 || y >= a.H - 2) ? 1u : 0u); // axes cut: 0, 1, 2
    const int ncx = (a.W + 7) >> 3;
    for (int dy = -2; dy <= 2; dy++)
        for (int dx = -2; dx <= 2; dx++) {
            int xx = x + dx, yy = y + dy;
            if ((unsigned)xx >= (unsigned)a.W || (unsigned)yy >= (unsigned)a.H) continue;
            uint32_t m = a.map[(size_t)yy * a.W + xx], w = a.mapw[(size_t)yy * a.W + xx];
            int sx = xx + (int)(int16_t)(m & 0xffffu), sy = yy + ((int)m >> 16);
            bool c0 = (w & 0xffu) != 0, c1 = ((w >> 8) & 0xffu) != 0, r1 = ((w >> 16) & 0xffu) != 0, r0 = (w >> 24) != 0;
            if ((c0 || c1) && (r0 || r1)) {
                int lo = c0 ? sx : sx + 1, hi = c1 ? sx + 1 : sx, lo2 = r0 ? sy : sy + 1, hi2 = r1 ? sy + 1 : sy;
                x0 = lo < x0 ? lo : x0; x1 = hi > x1 ? hi : x1; y0 = lo2 < y0 ? lo2 : y0; y1 = hi2 > y1 ? hi2 : y1;
                if (cut)
                    for (int cy = lo2 >> 3; cy <= hi2 >> 3; cy++)
                        for (int cx = lo >> 3; cx <= hi >> 3; cx++) atomicOr(&a.edge[cy * ncx + cx], cut);
            }
        }
    if (x1 >= x0) { atomicMax(&a.stats[1], (uint32_t)(x1 - x0 + 1)); atomicMax(&a.stats[2], (uint32_t)(y1 - y0 + 1)); }
}

// ---- launchers -------------------------------------------------------------------------------------------------
static inline dim3 grid2d(int W, int H) { return dim3((W + 63) / 64, (H + 3) / 4); }
void launch_remap_stats(const StatArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(remap_weight_scatter_kernel, grid2d(a.W, a.H), dim3(64, 4), 0, s, a);
    hipLaunchKernelGGL(remap_stats_kernel, grid2d(a.W, a.H), dim3(64, 4), 0, s, a);
}

int rows_stage_dwords() { return ROWS_STAGE_DW; }

void launch_filter_mask(const FilterArgs& a, bool remap, hipStream_t s)
{
    const int blocks = a.cam_mod * a.n_cgroups * a.n_steps;
    if (remap && a.staged) {
        hipLaunchKernelGGL(filter_rows_staged_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
        return;
    }
    if (remap && a.pipelined)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, true, false>), dim3(blocks), dim3(256), 0, s, a);
    else if (remap)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, false, false>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.W >= 4)
        hipLaunchKernelGGL((filter_mask_kernel<false, false, false, false>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((filter_mask_kernel<false, true, false, false>), dim3(blocks), dim3(256), 0, s, a);
}

// the same row pipeline over the list of wide tiles (a.tiles / a.n_tiles): a fixed grid, four entries per workgroup at a time
void launch_filter_tiles(const FilterArgs& a, bool remap, int blocks, hipStream_t s)
{
    if (remap && a.staged) {
        hipLaunchKernelGGL(filter_rows_staged_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
        return;
    }
    if (remap && a.pipelined)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, true, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (remap)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, false, true>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((filter_mask_kernel<false, false, false, true>), dim3(blocks), dim3(256), 0, s, a);
}
void launch_bright_cells(const BrightArgs& a_, hipStream_t s)
{
    BrightArgs a = a_;
    const int n = ((a.W + 7) >> 3) * ((a.H + 7) >> 3);
    a.blocks_x = a.wide ? (n / 2 + 255) / 256 : (n + 511) / 512;
    // The pass goes out as `slices` launches over consecutive runs of images (1 = one launch).  Between two slices the stream's
    // queue has a kernel boundary: while a slice drains, the short kernels of the other batches in flight get the registers and
    // wave slots that the pass, with its hundreds of thousands of ready workgroups, otherwise holds until its last block.
    const int slices = a.slices > 1 ? (a.slices < a.n_images ? a.slices : a.n_images) : 1;
    const int per = (a.n_images + slices - 1) / slices;
    const size_t mask_words = a.mask_words;
    for (int i0 = 0; i0 < a.n_images; i0 += per) {
        a.image0 = i0;
        a.slice_images = a.n_images - i0 < per ? a.n_images - i0 : per;
        a.mask_words = i0 == 0 ? mask_words : 0; // (the side job of clearing caller-owned masks goes with the first slice)
        const long long total = (long long)a.blocks_x * a.slice_images;
        long long grid = total;
        uint32_t* const ctr = a_.block_ctr;
        a.block_ctr = nullptr;
        if (a.max_blocks > 0 && a.max_blocks < grid) { // persistent form: a fixed number of workgroups
            grid = a.max_blocks;
            a.block_ctr = ctr ? ctr + (i0 / per) : nullptr; // one counter per slice (zeroed by the caller)
        }
        if (grid > 0x7fffffffLL) grid = 0x7fffffffLL;
        const dim3 g((unsigned)grid), b(256);
        if (a.hotmap) {
            if (a.wide && a.H % 8 == 0) hipLaunchKernelGGL((bright_cells_kernel<true, true, true>), g, b, 0, s, a);
            else if (a.wide) hipLaunchKernelGGL((bright_cells_kernel<true, false, true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((bright_cells_kernel<false, false, true>), g, b, 0, s, a);
        } else {
            if (a.wide && a.H % 8 == 0) hipLaunchKernelGGL((bright_cells_kernel<true, true, false>), g, b, 0, s, a);
            else if (a.wide) hipLaunchKernelGGL((bright_cells_kernel<true, false, false>), g, b, 0, s, a);
            else hipLaunchKernelGGL((bright_cells_kernel<false, false, false>), g, b, 0, s, a);
        }
    }
}
int hot_map_words(int H, int W, int wide)
{ // whole waves of the scan write the map: 32 words per block of 256 threads (128 cells per wave either way)
    const int n = ((W + 7) >> 3) * ((H + 7) >> 3);
    return 32 * (wide ? (n / 2 + 255) / 256 : (n + 511) / 512);
}
void launch_mark_tiles(const BrightArgs& a, hipStream_t s)
{
    long long grid = ((long long)((a.hot_words + 255) >> 8) * a.n_images + 3) / 4; // one piece per wave ...
    if (a.mark_grid > 0 && a.mark_grid < grid) grid = a.mark_grid;                  // ... or a fixed grid whose waves loop
    hipLaunchKernelGGL(mark_tiles_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
}
void launch_undistort_map(const MapArgs& m, hipStream_t s)
{
    hipLaunchKernelGGL(undistort_map_kernel, dim3((m.H + 63) / 64), dim3(64), 0, s, m);
}
void launch_box_blur(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, int ksize, hipStream_t s)
{
    hipLaunchKernelGGL(box_blur_kernel, grid2d(W, H), dim3(64, 4), 0, s, src, dst, H, W, sp, dp, ksize);
}
void launch_undistort(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, const uint32_t* map, const uint32_t* mapw,
                      hipStream_t s)
{
    hipLaunchKernelGGL(undistort_kernel, grid2d(W, H), dim3(64, 4), 0, s, src, dst, H, W, sp, dp, map, mapw);
}
void launch_mask_expand(const uint32_t* mask, int wpr, uint8_t* dst, int H, int W, int dp, hipStream_t s)
{
    hipLaunchKernelGGL(mask_expand_kernel, grid2d(W, H), dim3(64, 4), 0, s, mask, wpr, dst, H, W, dp);
}
void launch_median5(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, int ithresh, int apply, hipStream_t s)
{
    hipLaunchKernelGGL(median5_threshold_kernel, grid2d(W, H), dim3(64, 4), 0, s, src, dst, H, W, sp, dp, ithresh, apply);
}
void launch_demosaic(const uint8_t* bayer, uint8_t* bgr, int H, int W, int sp, hipStream_t s)
{
    hipLaunchKernelGGL(demosaic_kernel, grid2d(W, H), dim3(64, 4), 0, s, bayer, bgr, H, W, sp);
}

} // namespace mocap
