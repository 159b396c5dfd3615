// blob_filter.hip -- fused  undistort -> 5x5 in-bounds box sum -> threshold -> 5x5 majority  kernel.
//
// Replaces, for one batch of camera images resident in HBM, the chain
//   cv.undistort (reference lib/ImageOperations.py:38) -> fast_cuda_blur (lib/CudaOperations.py:5-41)
//   -> cv.threshold (lib/ImageOperations.py:29) -> cv.medianBlur (lib/ImageOperations.py:30)
// and writes the filtered binary image as a bit mask (1 bit / pixel).
//
// gfx950 design: one wave owns a strip of 256 source columns (4 px per lane, one dword load per lane
// and row = a 256-byte coalesced row segment) and slides down the rows of its chunk.  Everything lives
// in registers; horizontal neighbours come from DPP wave shifts, byte sums from v_dot4_u32_u8, the
// vertical 5-row windows are running sums whose history sits in a per-wave LDS ring.  HBM traffic is
// the 1 B/px read (+6 % halo) and the 1/8 B/px mask write.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace mocap {

__device__ __forceinline__ uint32_t lane_from_prev(uint32_t v)
{ // lane L receives lane L-1's value, lane 0 receives 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lane_from_next(uint32_t v)
{ // lane L receives lane L+1's value, lane 63 receives 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t sel, uint32_t acc)
{
    return __builtin_amdgcn_udot4(a, sel, acc, false);
}

// one undistorted pixel (cv::remap, INTER_LINEAR, BORDER_CONSTANT 0) through the packed (dx,dy) map
__device__ __forceinline__ uint32_t remap_px(const uint8_t* __restrict__ img, int pitch, int H, int W,
                                             uint32_t m, int x, int y)
{
    int dx = (int)(int16_t)(m & 0xffffu), dy = (int)m >> 16;
    int iu = 32 * x + dx, iv = 32 * y + dy;
    int sx = iu >> 5, sy = iv >> 5, a = iu & 31, b = iv & 31;
    uint32_t p00 = 0, p01 = 0, p10 = 0, p11 = 0;
    bool x0ok = (unsigned)sx < (unsigned)W, x1ok = (unsigned)(sx + 1) < (unsigned)W;
    if ((unsigned)sy < (unsigned)H) {
        const uint8_t* r = img + (size_t)sy * pitch;
        if (x0ok) p00 = r[sx];
        if (x1ok) p01 = r[sx + 1];
    }
    if ((unsigned)(sy + 1) < (unsigned)H) {
        const uint8_t* r = img + (size_t)(sy + 1) * pitch;
        if (x0ok) p10 = r[sx];
        if (x1ok) p11 = r[sx + 1];
    }
    uint32_t top = p00 * (32 - a) + p01 * a, bot = p10 * (32 - a) + p11 * a;
    return ((top * (32 - b) + bot * b) * 32 + (1u << 14)) >> 15;
}

template <bool REMAP>
__device__ __forceinline__ uint32_t load_src4(const FilterArgs& a, const uint8_t* __restrict__ img,
                                              const uint32_t* __restrict__ map, int y, int xl)
{
    if (y < 0 || y >= a.H) return 0u; // wave-uniform
    if (REMAP) {
        uint32_t out = 0;
        const uint32_t* mrow = map + (size_t)y * a.W;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int x = xl + k;
            if ((unsigned)x < (unsigned)a.W) out |= remap_px(img, a.pitch, a.H, a.W, mrow[x], x, y) << (8 * k);
        }
        return out;
    } else {
        const uint8_t* p = img + (size_t)y * a.pitch + xl;
        if (xl >= 0 && xl + 3 < a.W) {
            if (a.aligned4) return *(const uint32_t*)p;
            return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        }
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((unsigned)(xl + k) < (unsigned)a.W) out |= (uint32_t)p[k] << (8 * k);
        return out;
    }
}

// number of in-image taps of a 5-wide window centred on v
__device__ __forceinline__ int taps5(int v, int n)
{
    int lo = v - 2 < 0 ? 0 : v - 2, hi = v + 2 > n - 1 ? n - 1 : v + 2;
    return hi - lo + 1;
}

template <bool REMAP>
__global__ __launch_bounds__(256) void filter_mask_kernel(FilterArgs a)
{
    __shared__ uint32_t lut[256];
    __shared__ uint2 hring[4][8][64];
    __shared__ uint32_t cring[4][8][64];

    // lut[w]: byte k = number of set bits among bits k..k+4 of the 8-bit window w
    {
        uint32_t i = threadIdx.x, v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) v |= (uint32_t)__popc((i >> k) & 0x1fu) << (8 * k);
        lut[i] = v;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < 8; s++) {
        hring[wv][s][lane] = make_uint2(0u, 0u);
        cring[wv][s][lane] = 0u;
    }
    __syncthreads();

    // block -> (tile, time step).  Blocks b and b+8 share an XCD (round-robin dispatch): all time steps of
    // one (camera, strip, chunk group) tile are dealt to the same XCD back to back, so the tile's undistort
    // map is fetched into that XCD's L2 once per batch instead of once per frame.
    const int tiles = a.cam_mod * a.n_strips * a.n_cgroups;
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    const int tile = (q / a.n_steps) * 8 + xcd, tstep = q % a.n_steps;
    if (tile >= tiles) return;
    const int slot = tile % a.cam_mod;
    const int strip = (tile / a.cam_mod) % a.n_strips;
    const int cgroup = tile / (a.cam_mod * a.n_strips);
    const int image = tstep * a.cam_mod + slot;
    if (image >= a.n_images) return;

    const int r0 = (cgroup * 4 + wv) * a.rows_per_chunk;
    if (r0 >= a.H) return;
    const int r1 = r0 + a.rows_per_chunk < a.H ? r0 + a.rows_per_chunk : a.H;

    const uint8_t* __restrict__ img = a.src + (size_t)image * a.image_stride;
    const uint32_t* __restrict__ map = REMAP ? a.map + (size_t)slot * a.H * a.W : nullptr;
    uint8_t* __restrict__ mrow_base = (uint8_t*)(a.mask + (size_t)image * a.H * a.words_per_row);
    const int row_bytes = a.words_per_row * 4;

    const int xbase = strip * 240 - 8;
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
    const bool left_edge = xbase < 0;
    const bool right_edge = xbase + 255 >= a.W;
    const int lane_r = (a.W - 1 - xbase) >> 2, bit_r = (a.W - 1 - xbase) & 3; // lane / bit of column W-1
    // byte of the output row written by this (even) lane
    const int out_byte = strip * 30 + ((lane - 2) >> 1);
    const bool stores = ((lane & 1) == 0) && lane >= 2 && lane <= 60 && out_byte < ((a.W + 7) >> 3) &&
                        out_byte < row_bytes;

    uint32_t V01 = 0, V23 = 0, Cv = 0;
    int hi = 0, cj = 0; // ring counters

    const int Hm1 = a.H - 1;
    int kfirst = r0 - 2;
    kfirst = kfirst < 0 ? 0 : (kfirst > Hm1 ? Hm1 : kfirst);

    // source-row prefetch queue (4 rows ahead)
    int ynext = kfirst - 2;
    uint32_t q0 = load_src4<REMAP>(a, img, map, ynext, xl);
    uint32_t q1 = load_src4<REMAP>(a, img, map, ynext + 1, xl);
    uint32_t q2 = load_src4<REMAP>(a, img, map, ynext + 2, xl);
    uint32_t q3 = load_src4<REMAP>(a, img, map, ynext + 3, xl);
    ynext += 4;

    auto slide = [&]() {
        uint32_t B = q0;
        q0 = q1; q1 = q2; q2 = q3;
        q3 = load_src4<REMAP>(a, img, map, ynext, xl);
        ynext++;
        uint32_t A = lane_from_prev(B), C = lane_from_next(B);
        uint32_t sB = dot4(B, 0x01010101u, 0u);
        uint32_t h0 = dot4(A, 0x01010000u, dot4(B, 0x00010101u, 0u));
        uint32_t h1 = dot4(A, 0x01000000u, sB);
        uint32_t h2 = dot4(C, 0x00000001u, sB);
        uint32_t h3 = dot4(C, 0x00000101u, dot4(B, 0x01010100u, 0u));
        uint32_t H01 = h0 | (h1 << 16), H23 = h2 | (h3 << 16);
        uint2 old = hring[wv][(hi + 3) & 7][lane];
        hring[wv][hi & 7][lane] = make_uint2(H01, H23);
        hi++;
        V01 += H01 - old.x; // 16-bit fields never borrow: the window sum always contains the row removed
        V23 += H23 - old.y;
    };

    // rows kfirst-2 .. kfirst+1
    slide(); slide(); slide(); slide();

    int kc_cur = kfirst - 1;
    uint32_t c_cur = 0;
    for (int k = r0 - 2; k <= r1 + 1; ++k) {
        int kc = k < 0 ? 0 : (k > Hm1 ? Hm1 : k);
        if (kc != kc_cur) { // wave-uniform
            kc_cur = kc;
            slide(); // V = sum of source rows kc-2 .. kc+2
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
            c_cur = lut[win];
        }
        uint32_t cold = cring[wv][(cj + 3) & 7][lane];
        cring[wv][cj & 7][lane] = c_cur;
        cj++;
        Cv += c_cur - cold;
        if (k >= r0 + 2) {
            // majority: count >= 13 in each byte
            uint32_t mm = ((Cv + 0x73737373u) >> 7) & 0x01010101u;
            uint32_t t1 = mm | (mm >> 7);
            uint32_t mn = (t1 | (t1 >> 14)) & colmask;
            uint32_t odd = lane_from_next(mn);
            uint32_t byte = (mn & 0xfu) | ((odd & 0xfu) << 4);
            if (stores) mrow_base[(size_t)(k - 2) * row_bytes + out_byte] = (uint8_t)byte;
        }
    }
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
        double ddx = ru - 32.0 * j, ddy = rv - 32.0 * row;
        if (!(ddx >= -32768.0 && ddx <= 32767.0 && ddy >= -32768.0 && ddy <= 32767.0)) {
            flags |= 2u; // displacement beyond +-1024 px: not representable in the packed map
            ddx = ddx < 0 ? -32768.0 : 32767.0;
            ddy = ddy < 0 ? -32768.0 : 32767.0;
        }
        int dx = (int)ddx, dy = (int)ddy;
        if (dx | dy) flags |= 1u;
        out[j] = ((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16);
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
                                 int spitch, int dpitch, const uint32_t* __restrict__ map)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    dst[(size_t)y * dpitch + x] = (uint8_t)remap_px(src, spitch, H, W, map[(size_t)y * W + x], x, y);
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

// ---- launchers -------------------------------------------------------------------------------------------------
void launch_filter_mask(const FilterArgs& a, bool remap, hipStream_t s)
{
    int tiles = a.cam_mod * a.n_strips * a.n_cgroups;
    int blocks = ((tiles + 7) / 8) * 8 * a.n_steps;
    if (remap)
        hipLaunchKernelGGL(filter_mask_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(filter_mask_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
}
void launch_undistort_map(const MapArgs& m, hipStream_t s)
{
    hipLaunchKernelGGL(undistort_map_kernel, dim3((m.H + 63) / 64), dim3(64), 0, s, m);
}
static inline dim3 grid2d(int W, int H) { return dim3((W + 63) / 64, (H + 3) / 4); }
void launch_box_blur(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, int ksize, hipStream_t s)
{
    hipLaunchKernelGGL(box_blur_kernel, grid2d(W, H), dim3(64, 4), 0, s, src, dst, H, W, sp, dp, ksize);
}
void launch_undistort(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, const uint32_t* map, hipStream_t s)
{
    hipLaunchKernelGGL(undistort_kernel, grid2d(W, H), dim3(64, 4), 0, s, src, dst, H, W, sp, dp, map);
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
