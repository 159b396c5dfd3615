// blob_filter.hip -- the filter stage:  undistort -> 5x5 in-bounds box sum -> threshold -> 5x5 majority.
//
// Replaces, for one batch of camera images resident in HBM, the chain
//   cv.undistort (reference lib/ImageOperations.py:38) -> fast_cuda_blur (lib/CudaOperations.py:5-41)
//   -> cv.threshold (lib/ImageOperations.py:29) -> cv.medianBlur (lib/ImageOperations.py:30)
// and writes the filtered binary image as a bit mask (1 bit / pixel).
//
// Two kernels per batch:
//   bright_cells_kernel  streams every frame byte once (the algorithmic HBM traffic of the stage) and records, per
//                        filter tile, which mask rows can possibly hold a set pixel -- an exact bound, see
//                        "dark-tile early-out" below;
//   filter_mask_kernel   runs the fused filter over those rows only.  One wave owns a strip of 256 source columns
//                        (4 px per lane, one dword load per lane and row = a 256-byte coalesced row segment) and
//                        slides down its rows.  Everything lives in registers; horizontal neighbours come from DPP
//                        wave shifts, byte sums from v_dot4_u32_u8, the vertical 5-row windows are running sums
//                        whose history sits in a per-wave LDS ring.
// Also here: the set-up kernels of the undistort tables and the single-image convenience kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
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

// ---- LDS-staged remap -------------------------------------------------------------------------------------------
// The per-pixel 2x2 taps of the gather variants cost one vector-memory instruction per tap row and pixel, and the
// vector memory pipeline, not HBM, bounds them.  Here each wave keeps a ring of the most recent source rows of its
// strip in LDS (RING_H rows x RING_W bytes, filled with coalesced 8-byte loads, one or two rows per step -- the same
// HBM traffic as the plain path) and takes the taps from LDS.  Per step:
//   1. rows requested 4 steps ago -> LDS        2. taps of the NEXT row: LDS -> registers
//   3. table loads 5 rows ahead                 4. request the source rows needed 5 rows ahead
//   5. blend THIS row from the registers filled one step ago
struct TabSlot { uint4 m, w; };                       // tap positions and weights of one row (4 pixels)
struct RowReq { uint2 va, vb; int qa, qb; };           // two source rows in flight (qa, qb wave-uniform)
struct LTaps { uint32_t a0[4], b0[4], a1[4], b1[4], ph[4], w[4]; }; // dword pairs of both tap rows, byte phase, weights

struct LdsRemap {
    uint8_t* ring;     // this wave's ring (LDS)
    int xs0;           // first source column held (multiple of 8)
    int loaded_hi;     // highest source row requested so far
    int nl;            // lanes that carry ring columns: ceil(width / 8)
    uint32_t col;      // per lane: byte offset of its 8-byte column group in the image row (clamped to W-8)
    uint32_t colsh;    // per lane: bits to shift the loaded 64-bit value right after the clamp
};

__device__ __forceinline__ void ldsr_request(RowReq& rq, LdsRemap& st, int need, const uint8_t* __restrict__ img, int pitch)
{ // branch-free: always two loads; when nothing new is needed they re-fetch the newest row (identical bytes)
    int qa = st.loaded_hi + 1 < need ? st.loaded_hi + 1 : need, qb = st.loaded_hi + 2 < need ? st.loaded_hi + 2 : need;
    st.loaded_hi = st.loaded_hi > qb ? st.loaded_hi : qb;
    rq.qa = qa; rq.qb = qb;
    __builtin_memcpy(&rq.va, img + ((uint32_t)qa * (uint32_t)pitch + st.col), 8);
    __builtin_memcpy(&rq.vb, img + ((uint32_t)qb * (uint32_t)pitch + st.col), 8);
}

__device__ __forceinline__ void ldsr_write(const RowReq& rq, const LdsRemap& st, int lane)
{
    uint64_t a = (((uint64_t)rq.va.y << 32) | rq.va.x) >> st.colsh, b = (((uint64_t)rq.vb.y << 32) | rq.vb.x) >> st.colsh;
    if (lane < st.nl) {
        *(uint2*)(st.ring + (rq.qa & (RING_H - 1)) * RING_W + 8 * lane) = make_uint2((uint32_t)a, (uint32_t)(a >> 32));
        *(uint2*)(st.ring + (rq.qb & (RING_H - 1)) * RING_W + 8 * lane) = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
    }
}

__device__ __forceinline__ void ldsr_issue_tables(TabSlot& ts, const uint32_t* __restrict__ map, const uint32_t* __restrict__ mapw,
                                                  int row, int H, int W, const LaneCols& lc)
{
    int rc = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
    uint32_t off = (uint32_t)rc * (uint32_t)W + (uint32_t)lc.addr_x;
    __builtin_memcpy(&ts.m, map + off, 16);
    __builtin_memcpy(&ts.w, mapw + off, 16);
}

__device__ __forceinline__ void ldsr_read_taps(LTaps& t, const TabSlot& ts, const LdsRemap& st, int row, int H, const int xq[4])
{
    row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
    const uint32_t mm[4] = {ts.m.x, ts.m.y, ts.m.z, ts.m.w};
    t.w[0] = ts.w.x; t.w[1] = ts.w.y; t.w[2] = ts.w.z; t.w[3] = ts.w.w;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t m = mm[k];
        int sx = xq[k] + (int)(int16_t)(m & 0xffffu), sy = row + ((int)m >> 16);
        uint32_t cx = (uint32_t)(sx - st.xs0);
        uint32_t A0 = (uint32_t)(sy & (RING_H - 1)) * RING_W + cx, A1 = (uint32_t)((sy + 1) & (RING_H - 1)) * RING_W + cx;
        const uint32_t* p0 = (const uint32_t*)(st.ring + (A0 & ~3u));
        const uint32_t* p1 = (const uint32_t*)(st.ring + (A1 & ~3u));
        t.a0[k] = p0[0]; t.b0[k] = p0[1];
        t.a1[k] = p1[0]; t.b1[k] = p1[1];
        t.ph[k] = A0 & 3u; // RING_W is a multiple of 4: both rows share the byte phase
    }
}

__device__ __forceinline__ uint32_t ldsr_combine(const LTaps& t, const LaneCols& lc)
{
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t w = t.w[k], wx = w & 0xffffu;
        uint32_t t0 = __builtin_amdgcn_alignbyte(t.b0[k], t.a0[k], t.ph[k]), t1 = __builtin_amdgcn_alignbyte(t.b1[k], t.a1[k], t.ph[k]);
        uint32_t top = dot4(t0, wx, 0u), bot = dot4(t1, wx, 0u);
        uint32_t r = __umul24(top, w >> 24) + 512u;
        r += __umul24(bot, (w >> 16) & 0xffu);
        out |= (r >> 10) << (8 * k);
    }
    return out & lc.bytemask;
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
// pixel leaves costs what it weighs, not 192 per pixel.)
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
template <bool WIDE, bool FULL = false>
__global__ __launch_bounds__(256) void bright_cells_kernel(BrightArgs a)
{
    const int ncx = (a.W + 7) >> 3, ncy = (a.H + 7) >> 3, n = ncx * ncy;
    const int image = blockIdx.y;
    const uint8_t* __restrict__ img = a.src + (size_t)image * a.image_stride;
    uint2 v[2][8];
    int ci[2], cr[2];
    uint32_t sh[2];
    if (WIDE) {
        const int half = ncx >> 1, np = half * ncy; // cell pairs (ncx is even)
        int p = blockIdx.x * 256 + threadIdx.x;
        p = p < np ? p : np - 1; // threads past the end recount the last pair (and mark the same tiles again)
        const int row = (int)__umulhi((uint32_t)p, a.ncx_magic), cxp = p - row * half; // ncx_magic: for ncx / 2 here
        cr[0] = cr[1] = row;
        ci[0] = row * ncx + 2 * cxp; ci[1] = ci[0] + 1;
        sh[0] = sh[1] = 0;
        if (FULL) {
            const uint32_t off0 = (uint32_t)(8 * row) * (uint32_t)a.pitch + 16u * (uint32_t)cxp;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint4 q = *(const uint4*)(img + (off0 + (uint32_t)(j * a.pitch))); // uniform base + 32-bit offset
                v[0][j] = make_uint2(q.x, q.y); v[1][j] = make_uint2(q.z, q.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int r = 8 * row + j;
                r = r < a.H ? r : a.H - 1;
                const uint4 q = *(const uint4*)(img + ((uint32_t)r * (uint32_t)a.pitch + 16u * (uint32_t)cxp));
                v[0][j] = make_uint2(q.x, q.y); v[1][j] = make_uint2(q.z, q.w);
            }
        }
    } else {
        const int i0 = blockIdx.x * 512 + threadIdx.x;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            int i = i0 + 256 * u;
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
#pragma unroll
    for (int u = 0; u < 2; u++) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (WIDE) { // cells are whole
                const uint32_t e2 = excess2_row(v[u][j].x, v[u][j].y);
                if (FULL || 8 * cr[u] + j < a.H) acc += e2;
            } else {
                uint64_t vv = (((uint64_t)v[u][j].y << 32) | v[u][j].x) >> sh[u]; // drops the bytes left of the cell at the right edge
                const uint32_t e2 = excess2_row((uint32_t)vv, (uint32_t)(vv >> 32));
                if (8 * cr[u] + j < a.H) acc += e2;
            }
        }
        mark_hot_cell(a, reach, cflags, rows, ci[u], acc);
    }
    if (a.mask_words) {
        // caller-owned masks: clear them on the side (16 bytes per thread and round), the filter kernel then only
        // writes the tiles it filters.  The context's own mask needs no clearing (see the filter kernel).
        const size_t nthreads = (size_t)gridDim.x * gridDim.y * 256;
        const size_t g = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
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


// workgroup -> (camera slot, group of CPB chunks, time step): its four waves share the tiles (CPB chunks x all strips; a
// half of a 1080p image; 4 and 16 measured slightly slower) of that group through a work list (below).  Only the marked tiles' boxes touch the undistort
// tables, far too little per camera to keep anything resident in an XCD's L2, so there is no XCD-aware placement here.
constexpr int CPB = 8; // chunks per workgroup (4 per wave for the dark test; the marked tiles are shared by all waves)
struct TileId { int slot, cgroup, image; bool valid; }; // cgroup: group of CPB chunks
__device__ __forceinline__ int block_groups(const FilterArgs& a) { return (a.n_cgroups * 4 + CPB - 1) / CPB; }
__device__ __forceinline__ TileId decode_tile(const FilterArgs& a, int b)
{
    TileId t;
    const int groups = a.cam_mod * block_groups(a);
    const int grp = b / a.n_steps, tstep = b - grp * a.n_steps;
    t.slot = grp % a.cam_mod;
    t.cgroup = grp / a.cam_mod;
    t.image = tstep * a.cam_mod + t.slot;
    t.valid = grp < groups && t.image < a.n_images;
    return t;
}

// Work list of a workgroup: the tiles (chunk of the group, strip) of its four chunks that need work -- with the
// early-out, those whose box is not empty, plus (filter kernel only) those that were filtered in the previous batch and
// must be cleared; occupancy words of the others are settled right here; without the early-out, all of them.  Each wave
// tests the strips of one chunk with one lane-parallel load (lane = strip); afterwards the four waves take tiles from the
// list one by one, so a chunk with several marked strips does not hold up one wave while the others idle.
constexpr int MAX_STRIPS = 144; // 32767 / 240 + 1 rounded up
struct WorkList { uint16_t tile[CPB * MAX_STRIPS]; int n, head; };
template <bool FILTER>
__device__ __forceinline__ void list_work(const FilterArgs& a, WorkList& wl, int image, int cgroup, int wv, int lane)
{
    if (threadIdx.x == 0) { wl.n = 0; wl.head = 0; }
    __syncthreads();
    for (int j = wv; j < CPB; j += 4) {
        const int chunk = cgroup * CPB + j;
        if (chunk * a.rows_per_chunk >= a.H) break;
        const size_t cell_row = ((size_t)image * a.n_cgroups * 4 + chunk) * a.n_strips;
        for (int sbase = 0; sbase < a.n_strips; sbase += 64) {
            const int strip = sbase + lane;
            bool work = false;
            if (strip < a.n_strips) {
                work = true;
                if (a.skip_allow >= 0) {
                    const size_t ci = cell_row + strip;
                    const uint32_t lo = a.tile_rows[4 * ci], hi = a.tile_rows[4 * ci + 1];
                    work = lo <= hi;
                    if (FILTER) {
                        const uint32_t old = a.ext_mask ? 0u : a.cells[ci];
                        work = work || (old >> 31);
                        if (!work && (a.ext_mask || old != 0u)) a.cells[ci] = 0u; // dark and clean: an empty occupancy word
                    }
                }
            }
            if (work) wl.tile[atomicAdd(&wl.n, 1)] = (uint16_t)((j << 8) | strip);
        }
    }
    __syncthreads();
}
// next tile of the list for this wave: (chunk of the group << 8 | strip), or -1 when the list is exhausted
__device__ __forceinline__ int next_work(WorkList& wl, int lane)
{
    int i = 0;
    if (lane == 0) i = atomicAdd(&wl.head, 1);
    i = __builtin_amdgcn_readfirstlane(i);
    return i < wl.n ? (int)wl.tile[i] : -1;
}

// Patch of one filter tile: the undistorted pixels the tile's row pipeline will read (mask rows [r0, r1) + 4 rows
// above and 3 below, the strip's 256 columns), row-major with a pitch of 256 bytes, first row = tile_r0 - 4.
constexpr int PATCH_PITCH = 256;
__device__ __forceinline__ size_t patch_offset(const FilterArgs& a, size_t cell_index)
{
    return cell_index * (size_t)(a.rows_per_chunk + 8) * PATCH_PITCH;
}

// The box of a marked tile inside its patch: rows [by0, by1] x column quads [qa, qb] (quad q = columns xbase + 4q .. +3),
// from the ranges bright_cells_kernel left (ylo, yhi, xlo, xhi), clipped to the rows [in0, in1] the pipeline reads and
// to the quads inside the image.  Empty if qa > qb or by0 > by1.
struct PatchBox { int by0, by1, qa, qb; };
__device__ __forceinline__ PatchBox patch_box(int ylo, int yhi, int xlo, int xhi, int in0, int in1, int xbase, int W)
{
    PatchBox b;
    b.by0 = ylo > in0 ? ylo : in0; b.by1 = yhi < in1 ? yhi : in1;
    b.qa = (xlo - xbase) >> 2; b.qb = (xhi - xbase) >> 2;
    const int qmin = xbase < 0 ? 2 : 0, qmax = (W - 4 - xbase) >> 2 < 63 ? (W - 4 - xbase) >> 2 : 63;
    b.qa = b.qa < qmin ? qmin : b.qa; b.qb = b.qb > qmax ? qmax : b.qb;
    return b;
}

// the four patch pixels of a lane (columns xbase + 4*lane ..) in image row y: loaded from the nearest box row (so the
// load is unconditional and always hits written memory); patch_valid says whether the value counts or is a zero
__device__ __forceinline__ uint32_t fetch_patch4(const uint8_t* __restrict__ patch, int y, int prow0, int lane, const PatchBox& b)
{
    const int yc = y < b.by0 ? b.by0 : (y > b.by1 ? b.by1 : y);
    const int lc = lane < b.qa ? b.qa : (lane > b.qb ? b.qb : lane);
    return *(const uint32_t*)(patch + (uint32_t)(yc - prow0) * PATCH_PITCH + 4u * (uint32_t)lc);
}
__device__ __forceinline__ bool patch_valid(int y, int lane, const PatchBox& b)
{
    return y >= b.by0 && y <= b.by1 && lane >= b.qa && lane <= b.qb;
}

// undistort_patches_kernel -- cv::remap only where it can matter.  For every tile that bright_cells_kernel marked, the
// box of pixels that hot cells can reach (rows and columns, including the 4 pixels of blur + median) is undistorted
// exactly; every other pixel the tile's pipeline reads counts as 0 (filter_mask_kernel's patch path).  That is exact
// for the mask: a 5x5 window with a set threshold bit lies inside the box (its centre is within 2 pixels of a pixel
// that touches a hot cell), and zeros elsewhere can only lower box sums that are provably below the threshold.
// The box is usually much narrower than the strip, so its pixels are dealt to the lanes compactly (several box rows
// per wave instruction); the filter kernel then runs its plain (no gather) path on the patch.
__global__ __launch_bounds__(256) void undistort_patches_kernel(FilterArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ WorkList wl;
    const TileId t = decode_tile(a, blockIdx.x);
    if (!t.valid) return;
    list_work<false>(a, wl, t.image, t.cgroup, wv, lane);
    // every marked tile is worked on by all four waves together: wave w takes trips w, w + 4, ... of its box
    for (int i = 0; i < wl.n; ++i) {
        const int e = (int)wl.tile[i];
        const int chunk = t.cgroup * CPB + (e >> 8), strip = e & 0xff;
        const int tile_r0 = chunk * a.rows_per_chunk;
        const int tile_r1 = tile_r0 + a.rows_per_chunk < a.H ? tile_r0 + a.rows_per_chunk : a.H;
        const size_t cell_index = ((size_t)t.image * a.n_cgroups * 4 + chunk) * a.n_strips + strip;
        const uint32_t ylo = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index]);
        const uint32_t yhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index + 1]);
        if (ylo > yhi) continue; // dark tile
        const int r0 = (int)ylo > tile_r0 ? (int)ylo : tile_r0, r1 = (int)yhi + 1 < tile_r1 ? (int)yhi + 1 : tile_r1;
        if (r0 >= r1) continue;
        const int xlo = __builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index + 2]);
        const int xhi = __builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index + 3]);
        const int Hm1 = a.H - 1;
        const int in0 = r0 - 4 > 0 ? r0 - 4 : 0, in1 = r1 + 3 < Hm1 ? r1 + 3 : Hm1;     // rows the pipeline reads
        const int xbase = strip * 240 - 8;
        // The ranges left by bright_cells_kernel already include the 4 pixels of blur + median: a window with a set
        // threshold bit is centred within 2 pixels of a pixel that reads a hot cell and extends 2 pixels further.
        const PatchBox box = patch_box((int)ylo, (int)yhi, xlo, xhi, in0, in1, xbase, a.W);
        const int by0 = box.by0, by1 = box.by1, qa = box.qa, qb = box.qb;
        uint8_t* __restrict__ patch = a.patch + patch_offset(a, cell_index);
        const int prow0 = tile_r0 - 4;
        if (qa > qb || by0 > by1) continue;
        // the box only (the filter kernel substitutes the zeros around it itself): nq quads per row, rpw rows per wave instruction
        const uint8_t* __restrict__ img = a.src + (size_t)t.image * a.image_stride;
        const uint32_t* __restrict__ map = a.map + (size_t)t.slot * a.H * a.W;
        const uint32_t* __restrict__ mapw = a.mapw + (size_t)t.slot * a.H * a.W;
        const int nq = qb - qa + 1, rpw = 64 / nq;
        const int rsub = lane / nq, q = qa + (lane - rsub * nq);
        const bool lane_on = rsub < rpw;
        const int x = xbase + 4 * q; // inside the image and a multiple of 4 by construction
        // U row groups per trip: all table loads first, then all tap loads, then the blends -- two memory round trips per
        // trip instead of per row group.  Rows past the box are clamped to its last row (computed again, stored again with
        // the same value): no branch around the loads.
        constexpr int U = 4;
        if (!lane_on) continue;
        for (int rb = by0 + wv * U * rpw; rb <= by1; rb += 4 * U * rpw) {
            int rows[U];
            uint4 m4[U], w4[U];
    #pragma unroll
            for (int u = 0; u < U; u++) {
                int row = rb + u * rpw + rsub;
                rows[u] = row > by1 ? by1 : row;
                __builtin_memcpy(&m4[u], map + ((uint32_t)rows[u] * (uint32_t)a.W + (uint32_t)x), 16);
                __builtin_memcpy(&w4[u], mapw + ((uint32_t)rows[u] * (uint32_t)a.W + (uint32_t)x), 16);
            }
            uint32_t t0[U][4], t1[U][4];
    #pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t mm[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w};
    #pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t m = mm[k];
                    const int sx = x + k + (int)(int16_t)(m & 0xffffu), sy = rows[u] + ((int)m >> 16); // inside the image by construction
                    const uint32_t off0 = __umul24((uint32_t)sy, (uint32_t)a.pitch) + (uint32_t)sx;
                    t0[u][k] = load_u16(img + off0);
                    t1[u][k] = load_u16(img + off0 + (uint32_t)a.pitch);
                }
            }
    #pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t ww[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
                uint32_t out = 0;
    #pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w = ww[k];
                    const uint32_t top = dot4(t0[u][k], w, 0u), bot = dot4(t1[u][k], w, 0u); // tap bytes 2,3 are zero
                    uint32_t r = __umul24(top, w >> 24) + 512u;
                    r += __umul24(bot, (w >> 16) & 0xffu);
                    out |= (r >> 10) << (8 * k);
                }
                *(uint32_t*)(patch + (size_t)(rows[u] - prow0) * PATCH_PITCH + 4 * q) = out;
            }
        }
    } // marked tiles
}

template <bool REMAP, bool TINY, bool PIPE, bool LDSR, bool PATCH = false>
__global__ __launch_bounds__(256) void filter_mask_kernel(FilterArgs a)
{
    __shared__ uint32_t lut[256];
    __shared__ uint2 hring[4][8][64];
    __shared__ uint32_t cring[4][8][64];
    __shared__ __attribute__((aligned(16))) uint8_t sring[LDSR ? 4 : 1][LDSR ? RING_H * RING_W : 16];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: keeps the row loop scalar

    const TileId tid_ = decode_tile(a, blockIdx.x);
    if (!tid_.valid) return;
    const int slot = tid_.slot, cgroup = tid_.cgroup, image = tid_.image;

    __shared__ WorkList wl;
    list_work<true>(a, wl, image, cgroup, wv, lane);
    for (int e = next_work(wl, lane); e >= 0; e = next_work(wl, lane)) {
        const int chunk = cgroup * CPB + (e >> 8), strip = e & 0xff;
        const int tile_r0 = chunk * a.rows_per_chunk;              // the tile's mask rows [tile_r0, tile_r1)
        const int tile_r1 = tile_r0 + a.rows_per_chunk < a.H ? tile_r0 + a.rows_per_chunk : a.H;
        const size_t cell_row = ((size_t)image * a.n_cgroups * 4 + chunk) * a.n_strips;
        int r0 = tile_r0, r1 = tile_r1;                            // the rows this wave filters
        int box_ylo = 0, box_yhi = a.H - 1, box_xlo = 0, box_xhi = a.W - 1; // patch path: the exact pixels' box
        const int xbase = strip * 240 - 8;
        const size_t cell_index = cell_row + strip;

        if (a.skip_allow >= 0) {
            // ---- dark-tile early-out (see the comment above excess2_row) ----
            // bright_cells_kernel has left, per tile, the range of mask rows that hot cells of its source region can reach
            // (empty = none: the tile is all zeros).  Only those rows are filtered.  The context's mask keeps the
            // invariant "a tile's mask bytes are zero unless its occupancy word has bit 31 set" from batch to batch, so
            // rows that are not filtered only have to be cleared if the tile was filtered last time.
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index]);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index + 1]);
            if (PATCH) {
                box_ylo = (int)lo; box_yhi = (int)hi;
                box_xlo = __builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index + 2]);
                box_xhi = __builtin_amdgcn_readfirstlane((int)a.tile_rows[4 * cell_index + 3]);
            }
            if (lo <= hi) {
                if (lane < 4) a.tile_rows[4 * cell_index + lane] = (lane & 1) ? 0u : 0xffffffffu; // ready for the next batch
                if (!LDSR) { // (the LDS-staged variant's ring schedule is verified for whole tiles only)
                    r0 = (int)lo > r0 ? (int)lo : r0;
                    r1 = (int)hi + 1 < r1 ? (int)hi + 1 : r1;
                }
            }
            const bool dark = lo > hi || r0 >= r1;
            const uint32_t old = a.ext_mask ? 0u : a.cells[cell_index]; // (a caller-owned mask was cleared by bright_cells_kernel)
            if (__builtin_amdgcn_readfirstlane((int)old) < 0) { // clear what will not be written below
                uint8_t* mrow = (uint8_t*)(a.mask + (size_t)image * a.H * a.words_per_row);
                const int rb = a.words_per_row * 4;
                const int half = lane >> 5, pair = lane & 31, byte0 = strip * 30 + 2 * pair, nb = (a.W + 7) >> 3;
                if (pair < 15 && byte0 < nb) {
                    for (int row = tile_r0 + half; row < tile_r1; row += 2) {
                        if (!dark && row >= r0 && row < r1) continue;
                        uint8_t* dst = mrow + (size_t)row * rb + byte0;
                        if (byte0 + 1 < nb) *(uint16_t*)dst = 0; // byte0 is even: aligned
                        else *dst = 0;
                    }
                }
            }
            if (dark) {
                if (lane == 0) a.cells[cell_index] = 0u;
                continue;
            }
        }

        const int Hm1 = a.H - 1;
        int kfirst = r0 - 2;
        kfirst = kfirst < 0 ? 0 : (kfirst > Hm1 ? Hm1 : kfirst);
        const int ks = r0 - 1 > 1 ? r0 - 1 : 1;                 // steady range: every iteration slides one source row
        const int ke = r1 + 1 < Hm1 ? r1 + 1 : Hm1;

        // lut[w]: byte k = number of set bits among bits k..k+4 of the 8-bit window w.  Every wave that filters writes the
        // whole table (identical values), so no workgroup barrier is needed and dark waves are gone before this point.
    #pragma unroll
        for (int e = 0; e < 4; e++) {
            uint32_t i = (uint32_t)(lane + 64 * e), v = 0;
    #pragma unroll
            for (int k = 0; k < 4; k++) v |= (uint32_t)__popc((i >> k) & 0x1fu) << (8 * k);
            lut[i] = v;
        }
    #pragma unroll
        for (int s = 0; s < 8; s++) {
            hring[wv][s][lane] = make_uint2(0u, 0u);
            cring[wv][s][lane] = 0u;
        }

        const uint8_t* __restrict__ img = a.src + (size_t)image * a.image_stride;
        const uint8_t* __restrict__ patch = PATCH ? a.patch + patch_offset(a, cell_index) : nullptr;
        const int prow0 = tile_r0 - 4;
        const PatchBox pbox = patch_box(box_ylo, box_yhi, box_xlo, box_xhi, r0 - 4 > 0 ? r0 - 4 : 0, r1 + 3 < a.H - 1 ? r1 + 3 : a.H - 1,
                                        xbase, a.W);
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
        TabSlot tabs[4];
        RowReq rq[4];
        LTaps tb[2];
        LdsRemap st;
        if (LDSR) {
            const uint2* spans = a.spans + ((size_t)slot * a.n_strips + strip) * a.H;
            auto span_row = [&](int r) { return r < 0 ? 0 : (r > Hm1 ? Hm1 : r); };
            // source columns this strip needs over the rows the chunk consumes: y0 .. ke+2 (clamped into the image)
            int xmin = 0x7fff, xmax = 0;
            const int ra = span_row(y0), rb = span_row((ke > kfirst ? ke : kfirst) + 2);
            for (int r = ra + lane; r <= rb; r += 64) {
                uint32_t xs = spans[r].y;
                int lo = (int)(xs & 0xffffu), hi = (int)(xs >> 16);
                xmin = lo < xmin ? lo : xmin; xmax = hi > xmax ? hi : xmax;
            }
    #pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                int o1 = __shfl_xor(xmin, d), o2 = __shfl_xor(xmax, d);
                xmin = o1 < xmin ? o1 : xmin; xmax = o2 > xmax ? o2 : xmax;
            }
            st.ring = &sring[wv][0];
            st.xs0 = __builtin_amdgcn_readfirstlane(xmin) & ~7;
            st.nl = (__builtin_amdgcn_readfirstlane(xmax) - st.xs0 + 8) >> 3;
            {
                int c = st.xs0 + 8 * lane, cc = c < a.W - 8 ? c : a.W - 8;
                st.col = (uint32_t)cc;
                int sh = 8 * (c - cc);
                st.colsh = (uint32_t)(sh > 63 ? 63 : sh);
            }
            // first tables, then the ring rows the first five rows of the pipeline need, synchronously
            ldsr_issue_tables(tabs[3], map, mapw, y0, a.H, a.W, lc);
            const int first = (int)(spans[span_row(y0)].x & 0xffffu);
            const int upto = (int)(spans[span_row(y0 + RING_LOOKAHEAD - 1)].x >> 16);
            for (int qrow = first; qrow <= upto; ++qrow) {
                uint2 v;
                __builtin_memcpy(&v, img + ((uint32_t)qrow * (uint32_t)a.pitch + st.col), 8);
                uint64_t vv = (((uint64_t)v.y << 32) | v.x) >> st.colsh;
                if (lane < st.nl) *(uint2*)(st.ring + (qrow & (RING_H - 1)) * RING_W + 8 * lane) = make_uint2((uint32_t)vv, (uint32_t)(vv >> 32));
            }
            st.loaded_hi = upto;
            // the four request slots start as harmless re-fetches of the newest row
            ldsr_request(rq[3], st, upto, img, a.pitch);
            ldsr_request(rq[0], st, upto, img, a.pitch);
            ldsr_request(rq[1], st, upto, img, a.pitch);
            ldsr_request(rq[2], st, upto, img, a.pitch);
            // taps of the first row; tables of the next four (row rho lives in slot (rho - y0 + 3) & 3)
            ldsr_read_taps(tb[1], tabs[3], st, y0, a.H, xq);
            ldsr_issue_tables(tabs[0], map, mapw, y0 + 1, a.H, a.W, lc);
            ldsr_issue_tables(tabs[1], map, mapw, y0 + 2, a.H, a.W, lc);
            ldsr_issue_tables(tabs[2], map, mapw, y0 + 3, a.H, a.W, lc);
            ldsr_issue_tables(tabs[3], map, mapw, y0 + 4, a.H, a.W, lc);
        } else if (PIPE) {
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
            for (int j = 0; j < 8; j++) {
                int row = y0 + ((j + 5) & 7);
                q[j] = PATCH ? fetch_patch4(patch, row, prow0, lane, pbox) : fetch_src4<REMAP, TINY>(a, img, map, row, xl, lc);
            }
        }
        // next source row (row index `row`, queue slot J): its four pixels, and the refill of the pipeline behind it
        auto next_row = [&](auto Jc, int row) -> uint32_t {
            constexpr int J = decltype(Jc)::value;
            if (LDSR) {
                const uint2* spans = a.spans + ((size_t)slot * a.n_strips + strip) * a.H;
                ldsr_write(rq[J & 3], st, lane);                                         // 1
                ldsr_read_taps(tb[(J + 1) & 1], tabs[(J + 1) & 3], st, row + 1, a.H, xq); // 2
                ldsr_issue_tables(tabs[(J + 1) & 3], map, mapw, row + 5, a.H, a.W, lc);   // 3
                int nr = row + RING_LOOKAHEAD;
                nr = nr < 0 ? 0 : (nr > Hm1 ? Hm1 : nr);
                ldsr_request(rq[J & 3], st, (int)(spans[nr].x >> 16), img, a.pitch);      // 4
                uint32_t B = ldsr_combine(tb[J & 1], lc);                                 // 5
                if ((unsigned)row >= (unsigned)a.H) B = 0u;
                return B;
            } else if (PIPE) {
                constexpr int S = J & 3;
                uint32_t B = remap_combine(tq[S], lc);
                if ((unsigned)row >= (unsigned)a.H) B = 0u; // wave-uniform select: rows outside the image are zero
                remap_issue_taps(tq[S], mq[S], img, mapw, a.pitch, a.H, a.W, row + 4, xq, lc);
                remap_issue_map(mq[S], map, row + 8, a.H, a.W, lc);
                return B;
            } else {
                uint32_t B = PATCH ? (patch_valid(row, lane, pbox) ? q[J] : 0u)
                                   : finish_src4<REMAP, TINY>(q[J], (unsigned)row < (unsigned)a.H, lc);
                // refill 8 rows ahead, unconditionally (rows past the chunk are clamped into the image and simply
                // unused: a branch here would make the compiler drain the whole queue at the join)
                q[J] = PATCH ? fetch_patch4(patch, row + 8, prow0, lane, pbox) : fetch_src4<REMAP, TINY>(a, img, map, row + 8, xl, lc);
                return B;
            }
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
            // bit 31 marks a tile that went through the full filter (the early-out writes 0); bits 0..16 are the groups
            if (lane == 0) a.cells[cell_index] = cellmask | 0x80000000u;
        }
    } // marked tiles
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

// per (strip, row): the source rows [smin, smax] and columns [xmin, xmax] read by the strip's pixels of that row
// (both taps included), from the tap-position table.  One thread per (strip, row); set-up time only.
__global__ void remap_spans_kernel(SpanArgs a)
{
    int y = blockIdx.x * blockDim.x + threadIdx.x, strip = blockIdx.y;
    if (y >= a.H) return;
    int xa = strip * 240 - 8, xb = xa + 256;
    xa = xa < 0 ? 0 : xa; xb = xb > a.W ? a.W : xb;
    int smin = 0x7fff, smax = 0, xmin = 0x7fff, xmax = 0;
    const uint32_t* row = a.map + (size_t)y * a.W;
    for (int x = xa; x < xb; x++) {
        uint32_t m = row[x];
        int sx = x + (int)(int16_t)(m & 0xffffu), sy = y + ((int)m >> 16);
        smin = sy < smin ? sy : smin; smax = sy + 1 > smax ? sy + 1 : smax;
        xmin = sx < xmin ? sx : xmin; xmax = sx + 1 > xmax ? sx + 1 : xmax;
    }
    a.spans[(size_t)strip * a.H + y] = make_uint2((uint32_t)smin | ((uint32_t)smax << 16), (uint32_t)xmin | ((uint32_t)xmax << 16));

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
void launch_remap_spans(const SpanArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(remap_spans_kernel, dim3((a.H + 63) / 64, a.n_strips), dim3(64), 0, s, a);
}

static int filter_blocks(const FilterArgs& a)
{
    const int groups = a.cam_mod * ((a.n_cgroups * 4 + CPB - 1) / CPB);
    return groups * a.n_steps;
}

void launch_undistort_patches(const FilterArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(undistort_patches_kernel, dim3(filter_blocks(a)), dim3(256), 0, s, a);
}

void launch_filter_mask(const FilterArgs& a, bool remap, hipStream_t s)
{
    const int blocks = filter_blocks(a);
    if (a.patch) // remapped cameras with the early-out: the plain pipeline on the undistorted patches
        hipLaunchKernelGGL((filter_mask_kernel<false, false, false, false, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (remap && a.remap_mode == 4)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, false, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (remap && a.remap_mode == 3)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, true, false>), dim3(blocks), dim3(256), 0, s, a);
    else if (remap)
        hipLaunchKernelGGL((filter_mask_kernel<true, false, false, false>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.W >= 4)
        hipLaunchKernelGGL((filter_mask_kernel<false, false, false, false>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((filter_mask_kernel<false, true, false, false>), dim3(blocks), dim3(256), 0, s, a);
}
void launch_bright_cells(const BrightArgs& a, hipStream_t s)
{
    const int n = ((a.W + 7) >> 3) * ((a.H + 7) >> 3);
    if (a.wide && a.H % 8 == 0) hipLaunchKernelGGL((bright_cells_kernel<true, true>), dim3((n / 2 + 255) / 256, a.n_images), dim3(256), 0, s, a);
    else if (a.wide) hipLaunchKernelGGL(bright_cells_kernel<true>, dim3((n / 2 + 255) / 256, a.n_images), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bright_cells_kernel<false>, dim3((n + 511) / 512, a.n_images), dim3(256), 0, s, a);
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
