// bayer_gray.hip -- the pixel steps in front of the path: raw Bayer frame -> gray frame in one pass.
//
// Replaces the two OpenCV calls of the reference's camera loop (RealtimeTracking_FLIR.py:103-104)
//     image = cv2.cvtColor(raw, cv2.COLOR_BAYER_GR2BGR);  image = cv2.cvtColor(image, cv2.COLOR_BGR2GRAY)
// without the 3-byte BGR image in between: 1 B/px read + 1 B/px written, HBM bound.  Arithmetic as restated in
// oracle/blob_oracle.c (orc_bayer_gray_u8): bilinear demosaic with rounded means, first/last rows and columns repeat
// their inner neighbours, fixed-point luma with the 14-bit (default) or 15-bit coefficient set.
//
// Fast kernel (widths and pitches that are multiples of 16, or of 8): one lane = 16 (8) consecutive pixels (one aligned 16-
// (8-)byte load from each of the three rows around them), the pixels left and right of those come from the neighbouring
// lanes' registers (DPP wave shifts); the two outer lanes of a wave only feed their neighbours, so a wave writes 992 (496)
// pixels of one row.  The four
// waves of a workgroup take four consecutive rows: every frame byte is requested three times and comes from HBM once.
// The generic kernel (any width / pitch) does one pixel per lane.  bayer_gray_scan_kernel (below) is the fast kernel fused with
// the early-out's streaming scan: it walks 8 rows per wave and marks the filter tiles from the gray bytes it has in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "scan_mark.h"

namespace mocap {

__device__ __forceinline__ uint32_t luma(int b, int g, int r, const BayerArgs& a)
{
    return ((uint32_t)b * a.cb + (uint32_t)g * a.cg + (uint32_t)r * a.cr + (1u << (a.shift - 1))) >> a.shift;
}

// colours at a site from its 3x3 neighbourhood (centre c, l/r/u/d, the four diagonals)
__device__ __forceinline__ uint32_t site_gray(int c, int l, int r_, int u, int d, int ul, int ur, int dl, int dr, bool red_row,
                                              bool red_col, const BayerArgs& a)
{
    const int horiz = (l + r_ + 1) >> 1, vert = (u + d + 1) >> 1;
    const int cross = (l + r_ + u + d + 2) >> 2, diag = (ul + ur + dl + dr + 2) >> 2;
    int r, g, b;
    if (red_row == red_col) { // a red or a blue site
        g = cross;
        r = red_row ? c : diag;
        b = red_row ? diag : c;
    } else {                  // a green site: its row's colour left and right, the other one above and below
        g = c;
        r = red_row ? horiz : vert;
        b = red_row ? vert : horiz;
    }
    return luma(b, g, r, a);
}

// ---- the 16 (or 8) pixels-per-lane kernel ----------------------------------------------------------------------------
// Within a row, sites of equal column parity are of one kind, so the lane keeps its pixels (and their neighbours) as
// pairs of 16-bit fields: e = columns (0,2), (4,6), ..., o = columns (1,3), (5,7), ..., Le = the left neighbours of e, Ro = the
// right neighbours of o (v_perm_b32 each).  Rounded means are then plain 32-bit adds on two pixels at once, which
// kind of site the even columns hold is a wave-uniform branch (a wave is one row), and "which of the two interpolated
// colours is red" only swaps two luma coefficients.  Luma per pixel: three v_dot2_u32_u16 that pick the field as they multiply.
__device__ __forceinline__ uint32_t prm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ uint32_t wave_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t wave_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }

template <int ND> struct RowPairs { uint32_t e[ND], o[ND], Le[ND], Ro[ND]; };

// w[i] = columns 4i .. 4i+3 of the lane's ND * 4 pixels
template <int ND, bool NEED_LE, bool NEED_RO>
__device__ __forceinline__ RowPairs<ND> row_pairs(const uint32_t (&w)[ND])
{
    RowPairs<ND> r;
#pragma unroll
    for (int i = 0; i < ND; i++) {
        r.e[i] = prm(0, w[i], 0x0c020c00u);               // columns (4i, 4i+2)
        r.o[i] = prm(0, w[i], 0x0c030c01u);               // columns (4i+1, 4i+3)
    }
    if (NEED_LE) {
        const uint32_t pw = wave_prev(w[ND - 1]);         // the four columns left of the lane's
        r.Le[0] = prm(pw, w[0], 0x0c010c07u);             // columns (-1, 1)
#pragma unroll
        for (int i = 1; i < ND; i++) r.Le[i] = prm(w[i], w[i - 1], 0x0c050c03u); // columns (4i-1, 4i+1)
    }
    if (NEED_RO) {
        const uint32_t nw = wave_next(w[0]);              // the four columns right of the lane's
#pragma unroll
        for (int i = 0; i < ND - 1; i++) r.Ro[i] = prm(w[i + 1], w[i], 0x0c040c02u); // columns (4i+2, 4i+4)
        r.Ro[ND - 1] = prm(nw, w[ND - 1], 0x0c040c02u);
    }
    return r;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_shr(uint32_t v, int n)
{ // both 16-bit fields shifted right on their own (v_pk_lshrrev_b16): nothing leaks from the upper into the lower field
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, v) >> (unsigned short)n));
}
__device__ __forceinline__ uint32_t mean2(uint32_t a, uint32_t b) { return pk_shr(a + b + 0x00010001u, 1); }
__device__ __forceinline__ uint32_t mean4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return pk_shr(a + b + c + d + 0x00020002u, 2); }

// luma of the two pixels held in the 16-bit fields of (x, g, y): x = the row's own colour, y = the other one.
// v_dot2_u32_u16 against (c, 0) / (0, c) picks the field and multiplies in one instruction.
struct LumaCoef { uint32_t xa, ga, ya, xb, gb, yb, half; int shift; };
__device__ __forceinline__ uint32_t dot2(uint32_t v, uint32_t c, uint32_t acc)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, v), __builtin_bit_cast(u16x2, c), acc, false);
}
__device__ __forceinline__ void luma2(uint32_t x, uint32_t g, uint32_t y, const LumaCoef& k, uint32_t& out_a, uint32_t& out_b)
{
#ifdef BAYER_LUMA_MUL
    out_a = ((x & 0xffffu) * k.xa + ((g & 0xffffu) * k.ga + ((y & 0xffffu) * k.ya + k.half))) >> k.shift;
    out_b = ((x >> 16) * k.xa + ((g >> 16) * k.ga + ((y >> 16) * k.ya + k.half))) >> k.shift;
#else
    out_a = dot2(x, k.xa, dot2(g, k.ga, dot2(y, k.ya, k.half))) >> k.shift;
    out_b = dot2(x, k.xb, dot2(g, k.gb, dot2(y, k.yb, k.half))) >> k.shift;
#endif
}

// ND dwords (4 * ND pixels) per lane; the two outer lanes of a wave only feed their neighbours
template <int ND> struct LaneLoad;
template <> struct LaneLoad<2> { typedef uint2 type; };
template <> struct LaneLoad<4> { typedef uint4 type; };

template <int ND>
__global__ __launch_bounds__(256) void bayer_gray_kernel(BayerArgs a)
{
    constexpr int PX = 4 * ND, WAVE_PX = 62 * PX;
    typedef typename LaneLoad<ND>::type vec_t;
    const int lane = threadIdx.x;                       // blockDim = (64, 4): a wave works on one row
    const int y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y); // wave-uniform: row addresses stay scalar
    if (y >= a.H) return;
    const int x0 = blockIdx.x * WAVE_PX - PX + PX * lane;
    const int yc = y < 1 ? 1 : (y > a.H - 2 ? a.H - 2 : y);
    const uint8_t* __restrict__ src = a.src + (size_t)blockIdx.z * a.sstride + (size_t)(yc - 1) * a.spitch;
    const bool in = x0 >= 0 && x0 < a.W;                // W is a multiple of PX
    const uint32_t off = in ? (uint32_t)x0 : 0u;
    uint32_t wu[ND], wc[ND], wd[ND];
    {
        const vec_t vu = *(const vec_t*)(src + off), vc = *(const vec_t*)(src + a.spitch + off), vd = *(const vec_t*)(src + 2 * a.spitch + off);
        __builtin_memcpy(wu, &vu, sizeof(vu)); __builtin_memcpy(wc, &vc, sizeof(vc)); __builtin_memcpy(wd, &vd, sizeof(vd));
    }
    if (!in) {
#pragma unroll
        for (int i = 0; i < ND; i++) wu[i] = wc[i] = wd[i] = 0u;
    }

    const bool red_row = (yc & 1) == a.ry;
    const bool even_is_colour = (red_row ? a.rx : 1 - a.rx) == 0; // the row's red / blue sites sit on even columns
    // x = the row's own colour (red on a red row), y = the other one: only their luma coefficients differ
    const uint32_t cx = red_row ? a.cr : a.cb, cy = red_row ? a.cb : a.cr;
    const LumaCoef k{cx, a.cg, cy, cx << 16, a.cg << 16, cy << 16, 1u << (a.shift - 1), a.shift};

    uint32_t g[PX];
    const RowPairs<ND> C = row_pairs<ND, true, true>(wc);
    if (even_is_colour) {
        const RowPairs<ND> U = row_pairs<ND, true, false>(wu), D = row_pairs<ND, true, false>(wd);
#pragma unroll
        for (int i = 0; i < ND; i++) {
            // even columns: colour sites -- own value, green from the cross, the other colour from the diagonals
            luma2(C.e[i], mean4(C.Le[i], C.o[i], U.e[i], D.e[i]), mean4(U.Le[i], U.o[i], D.Le[i], D.o[i]), k, g[4 * i], g[4 * i + 2]);
            // odd columns: green sites -- the row's colour left and right, the other one above and below
            luma2(mean2(C.e[i], C.Ro[i]), C.o[i], mean2(U.o[i], D.o[i]), k, g[4 * i + 1], g[4 * i + 3]);
        }
    } else {
        const RowPairs<ND> U = row_pairs<ND, false, true>(wu), D = row_pairs<ND, false, true>(wd);
#pragma unroll
        for (int i = 0; i < ND; i++) {
            luma2(mean2(C.Le[i], C.o[i]), C.e[i], mean2(U.e[i], D.e[i]), k, g[4 * i], g[4 * i + 2]);
            luma2(C.o[i], mean4(C.e[i], C.Ro[i], U.o[i], D.o[i]), mean4(U.e[i], U.Ro[i], D.e[i], D.Ro[i]), k, g[4 * i + 1], g[4 * i + 3]);
        }
    }
    if (x0 == 0) g[0] = g[1];                           // column 0 repeats column 1
    if (x0 + PX == a.W) g[PX - 1] = g[PX - 2];          // column W-1 repeats column W-2
    if (in && lane >= 1 && lane <= 62) {
        uint32_t out[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) out[i] = g[4 * i] | (g[4 * i + 1] << 8) | (g[4 * i + 2] << 16) | (g[4 * i + 3] << 24);
        vec_t v;
        __builtin_memcpy(&v, out, sizeof(v));
        *(vec_t*)(a.dst + (size_t)blockIdx.z * a.dstride + (size_t)y * a.dpitch + x0) = v;
    }
}

// ---- Bayer -> gray fused with the early-out's streaming pass (widths that are multiples of 16, heights of 8) ----------
// One wave = one cell row (8 image rows) x 992 pixels.  A lane walks its 16 columns down the 8 rows with a sliding window
// of three rows of field pairs -- every source row is loaded and unpacked once instead of three times --, writes the gray
// rows and sums their excess per 8x8 cell on the way (the gray bytes are in registers: the separate scan of the gray
// frames, one more read of every byte, disappears); hot cells mark tiles exactly as bright_cells_kernel does.
// Rows 0 and H-1 repeat rows 1 and H-2: they are not computed, their neighbours are written and counted twice.
__device__ __forceinline__ LumaCoef luma_coef(const BayerArgs& a, bool red_row)
{
    const uint32_t cx = red_row ? a.cr : a.cb, cy = red_row ? a.cb : a.cr;
    return LumaCoef{cx, a.cg, cy, cx << 16, a.cg << 16, cy << 16, 1u << (a.shift - 1), a.shift};
}

__global__ __launch_bounds__(256) void bayer_gray_scan_kernel(BayerArgs a, BrightArgs b)
{
    constexpr int ND = 4, PX = 16, WAVE_PX = 62 * PX;
    const int lane = threadIdx.x;                       // blockDim = (64, 4): a wave works on one cell row
    const int cy = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    if (8 * cy >= a.H) return;                          // H is a multiple of 8
    const int image = blockIdx.z;
    const int x0 = blockIdx.x * WAVE_PX - PX + PX * lane;
    const bool in = x0 >= 0 && x0 < a.W;                // W is a multiple of 16
    const uint32_t off = in ? (uint32_t)x0 : 0u;
    const uint8_t* __restrict__ src = a.src + (size_t)image * a.sstride;
    uint8_t* __restrict__ dst = a.dst + (size_t)image * a.dstride;

    // per row parity: which kind of site the even columns hold, and the luma coefficient order (8 * cy is even)
    const bool red0 = a.ry == 0, red1 = a.ry == 1;
    const bool eic0 = (red0 ? a.rx : 1 - a.rx) == 0, eic1 = (red1 ? a.rx : 1 - a.rx) == 0;
    const LumaCoef k0 = luma_coef(a, red0), k1 = luma_coef(a, red1);

    auto load_row = [&](int r) {
        r = r < 0 ? 0 : (r > a.H - 1 ? a.H - 1 : r);    // rows -1 and H only stand in for windows that are never evaluated
        uint4 v = *(const uint4*)(src + (size_t)r * a.spitch + off);
        if (!in) v = make_uint4(0u, 0u, 0u, 0u);
        return v;
    };
    auto pairs_of = [&](uint4 v) {
        uint32_t w[ND];
        __builtin_memcpy(w, &v, sizeof(v));
        return row_pairs<ND, true, true>(w);
    };
    uint4 raw[10];
#pragma unroll
    for (int r = 0; r < 10; r++) raw[r] = load_row(8 * cy - 1 + r);

    RowPairs<ND> U = pairs_of(raw[0]), C = pairs_of(raw[1]);
    uint32_t acc0 = 0, acc1 = 0;
    const bool writer = in && lane >= 1 && lane <= 62;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int y = 8 * cy + j;
        const RowPairs<ND> D = pairs_of(raw[j + 2]);
        if (y != 0 && y != a.H - 1) {                   // wave-uniform
            const bool eic = (j & 1) ? eic1 : eic0;
            const LumaCoef& k = (j & 1) ? k1 : k0;
            uint32_t g[PX];
            if (eic) {
#pragma unroll
                for (int i = 0; i < ND; i++) {
                    luma2(C.e[i], mean4(C.Le[i], C.o[i], U.e[i], D.e[i]), mean4(U.Le[i], U.o[i], D.Le[i], D.o[i]), k, g[4 * i], g[4 * i + 2]);
                    luma2(mean2(C.e[i], C.Ro[i]), C.o[i], mean2(U.o[i], D.o[i]), k, g[4 * i + 1], g[4 * i + 3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < ND; i++) {
                    luma2(mean2(C.Le[i], C.o[i]), C.e[i], mean2(U.e[i], D.e[i]), k, g[4 * i], g[4 * i + 2]);
                    luma2(C.o[i], mean4(C.e[i], C.Ro[i], U.o[i], D.o[i]), mean4(U.e[i], U.Ro[i], D.e[i], D.Ro[i]), k, g[4 * i + 1], g[4 * i + 3]);
                }
            }
            if (x0 == 0) g[0] = g[1];                   // column 0 repeats column 1
            if (x0 + PX == a.W) g[PX - 1] = g[PX - 2];  // column W-1 repeats column W-2
            uint4 out;
            out.x = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
            out.y = g[4] | (g[5] << 8) | (g[6] << 16) | (g[7] << 24);
            out.z = g[8] | (g[9] << 8) | (g[10] << 16) | (g[11] << 24);
            out.w = g[12] | (g[13] << 8) | (g[14] << 16) | (g[15] << 24);
            const bool twice = y == 1 || y == a.H - 2;  // wave-uniform
            if (writer) {
                *(uint4*)(dst + (size_t)y * a.dpitch + x0) = out;
                if (y == 1) *(uint4*)(dst + x0) = out;
                if (y == a.H - 2) *(uint4*)(dst + (size_t)(a.H - 1) * a.dpitch + x0) = out;
            }
            const uint32_t base4 = (uint32_t)b.base * 0x01010101u, c8 = 8u * (uint32_t)b.base;
            const uint32_t e0 = excess2_row(out.x, out.y, base4, c8), e1 = excess2_row(out.z, out.w, base4, c8);
            acc0 += twice ? 2u * e0 : e0;
            acc1 += twice ? 2u * e1 : e1;
        }
        U = C; C = D;
    }
    if (writer) {
        const int ncx = a.W >> 3, n = ncx * (a.H >> 3);
        const int slot = image % b.cam_mod;
        const uint2* __restrict__ reach = b.reach + (size_t)slot * n;
        const uint8_t* __restrict__ cflags = b.cflags + (size_t)slot * n;
        uint32_t* __restrict__ rows = b.tile_rows + (size_t)image * b.n_chunks * b.n_strips * 4;
        const int ci = cy * ncx + (x0 >> 3);
        mark_hot_cell(b, reach, cflags, rows, ci, acc0);
        mark_hot_cell(b, reach, cflags, rows, ci + 1, acc1);
    }
}

__global__ __launch_bounds__(256) void bayer_gray_any_kernel(BayerArgs a)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= a.W || y >= a.H) return;
    const int yc = y < 1 ? 1 : (y > a.H - 2 ? a.H - 2 : y), xc = x < 1 ? 1 : (x > a.W - 2 ? a.W - 2 : x);
    const uint8_t* __restrict__ p = a.src + (size_t)blockIdx.z * a.sstride + (size_t)yc * a.spitch + xc;
    const long sp = a.spitch;
    a.dst[(size_t)blockIdx.z * a.dstride + (size_t)y * a.dpitch + x] =
        (uint8_t)site_gray(p[0], p[-1], p[1], p[-sp], p[sp], p[-sp - 1], p[-sp + 1], p[sp - 1], p[sp + 1], (yc & 1) == a.ry,
                           (xc & 1) == a.rx, a);
}

void launch_bayer_gray(const BayerArgs& a, hipStream_t s)
{
    auto aligned = [&](int n) {
        return a.W % n == 0 && a.spitch % n == 0 && a.dpitch % n == 0 && a.sstride % n == 0 && a.dstride % n == 0 &&
               (uintptr_t)a.src % n == 0 && (uintptr_t)a.dst % n == 0;
    };
    if (aligned(16))
        hipLaunchKernelGGL(bayer_gray_kernel<4>, dim3((a.W + 62 * 16 - 1) / (62 * 16), (a.H + 3) / 4, a.n_images), dim3(64, 4), 0, s, a);
    else if (aligned(8))
        hipLaunchKernelGGL(bayer_gray_kernel<2>, dim3((a.W + 62 * 8 - 1) / (62 * 8), (a.H + 3) / 4, a.n_images), dim3(64, 4), 0, s, a);
    else
        hipLaunchKernelGGL(bayer_gray_any_kernel, dim3((a.W + 63) / 64, (a.H + 3) / 4, a.n_images), dim3(64, 4), 0, s, a);
}

bool bayer_scan_fusable(const BayerArgs& a)
{
    return a.W % 16 == 0 && a.H % 8 == 0 && a.H >= 8 && a.spitch % 16 == 0 && a.dpitch % 16 == 0 && a.sstride % 16 == 0 &&
           a.dstride % 16 == 0 && (uintptr_t)a.src % 16 == 0 && (uintptr_t)a.dst % 16 == 0;
}

void launch_bayer_gray_scan(const BayerArgs& a, const BrightArgs& b, hipStream_t s)
{
    hipLaunchKernelGGL(bayer_gray_scan_kernel, dim3((a.W + 62 * 16 - 1) / (62 * 16), (a.H / 8 + 3) / 4, a.n_images), dim3(64, 4), 0, s, a, b);
}

} // namespace mocap
