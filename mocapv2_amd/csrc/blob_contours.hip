// blob_contours.hip -- borders, polygon moments and ordered centroids from the filtered bit mask.
//
// Replaces cv.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) + cv.contourArea + cv.arcLength + cv.moments
// and the filter/centroid loops of reference lib/ImageOperations.py:41-65, for a batch of masks.
//
// The serial raster scan of Suzuki-Abe is replaced by its fixed point: every border is followed exactly
// once, from its raster-first pixel, so
//   * an outer border starts at a foreground pixel whose W, NW, N, NE neighbours are background and that is
//     the raster-minimum of the border it lies on;
//   * a hole border starts at the foreground pixel left of a background pixel whose W and N neighbours are
//     foreground and that is the raster-minimum of the left-side cracks of the border.
// Candidates are found with word-parallel bit tests on the mask; each candidate is followed by ONE LANE with
// literally the reference border-following step (same neighbour order, same CHAIN_APPROX_SIMPLE vertex rule),
// and is dropped as soon as it meets an earlier pixel of its own border.  The Green's-theorem sums are exact
// integers (int64), the perimeter is a sum of float32 square roots held exactly in a double.
// Tree order (parent = enclosing border, siblings in reverse discovery order, pre-order walk) is rebuilt
// from "which border owns the crack left of my start pixel", found by following that border once.
// One workgroup per image; the mask is 1/8 B per pixel and is read through L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace mocap {

namespace {

constexpr int MAXC = 2048;  // candidates per image
constexpr int MAXR = 384;   // borders per image
constexpr int MAXK = 256;   // kept contours per image
constexpr int MAXD = 8;     // nesting depth of a kept contour

struct Mask {
    const uint32_t* w;
    int wpr, H, W;
    int RS; // raster stride W+1, so the virtual background column right of the image has its own index
    __device__ __forceinline__ uint32_t word(int y, int k) const
    {
        return ((unsigned)y < (unsigned)H && (unsigned)k < (unsigned)wpr) ? w[(size_t)y * wpr + k] : 0u;
    }
    // bits x-1, x, x+1 of row y in bits 0..2
    __device__ __forceinline__ uint32_t three(int y, int x) const
    {
        if ((unsigned)y >= (unsigned)H) return 0u;
        int k = x >> 5, b = x & 31;
        uint64_t cur = word(y, k);
        uint64_t v = cur << 1;
        if (b == 0) v |= (word(y, k - 1) >> 31);
        if (b == 31) v |= ((uint64_t)(word(y, k + 1) & 1u) << 33);
        return (uint32_t)(v >> b) & 7u;
    }
    // occupancy of the 8 neighbours of (x,y), bit s = direction code s (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE)
    __device__ __forceinline__ uint32_t nbr8(int x, int y) const
    {
        uint32_t up = three(y - 1, x), mid = three(y, x), dn = three(y + 1, x);
        return ((mid >> 2) & 1u) | (((up >> 2) & 1u) << 1) | (((up >> 1) & 1u) << 2) | ((up & 1u) << 3) |
               ((mid & 1u) << 4) | ((dn & 1u) << 5) | (((dn >> 1) & 1u) << 6) | (((dn >> 2) & 1u) << 7);
    }
};

__device__ const int DXc[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__device__ const int DYc[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Trace {
    int64_t a00, a10, a01;
    double per;
    int npts, steps;
    int min_fg;   // raster-minimum border pixel
    int min_ebg;  // raster-minimum background pixel right of a border pixel whose East side was examined
    int status;   // 0 ok, 1 aborted (not the raster-first start), 2 step limit
    // polygon state
    int fx, fy, px, py;
    __device__ __forceinline__ void edge(int xp, int yp, int xi, int yi)
    {
        int64_t d = (int64_t)xp * yi - (int64_t)xi * yp;
        a00 += d;
        a10 += d * (xp + xi);
        a01 += d * (yp + yi);
        float dx = (float)xi - (float)xp, dy = (float)yi - (float)yp;
        per += (double)__fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    }
    __device__ __forceinline__ void vertex(int x, int y)
    {
        if (npts == 0) { fx = x; fy = y; }
        else edge(px, py, x, y);
        px = x; py = y;
        npts++;
    }
};

// Follow the border through pixel (sx,sy) whose neighbour in direction `first` (4 = W for an outer start,
// 0 = E for a hole start) is background.  Aborts when a border pixel with raster index < abort_fg or an
// East-side background pixel with raster index < abort_ebg is met.
__device__ void follow(const Mask& M, int sx, int sy, int first, int abort_fg, int abort_ebg, int max_steps, Trace& T)
{
    T.a00 = T.a10 = T.a01 = 0;
    T.per = 0.0;
    T.npts = T.steps = 0;
    T.status = 0;
    T.min_fg = sy * M.RS + sx;
    T.min_ebg = 0x7fffffff;
    uint32_t n = M.nbr8(sx, sy);
    int s = first, s_end = first;
    do {
        s = (s - 1) & 7;
    } while (!((n >> s) & 1u) && s != s_end);
    if (s == s_end) { // isolated pixel
        T.vertex(sx, sy);
        T.min_ebg = sy * M.RS + sx + 1;
        T.edge(T.px, T.py, T.fx, T.fy);
        return;
    }
    const int i1x = sx + DXc[s], i1y = sy + DYc[s];
    int x = sx, y = sy;
    int prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        // first occupied neighbour counter-clockwise from s_end+1
        uint32_t rot = ((n | (n << 8)) >> (s_end + 1)) & 0xffu;
        int u = s_end + 1 + (__ffs((int)rot) - 1);
        s = u & 7;
        int r = y * M.RS + x;
        if ((unsigned)(s - 1) < (unsigned)s_end) { // the East neighbour was examined and is background
            if (r + 1 < T.min_ebg) T.min_ebg = r + 1;
            if (r + 1 < abort_ebg) { T.status = 1; return; }
        }
        if (r < T.min_fg) T.min_fg = r;
        if (r < abort_fg) { T.status = 1; return; }
        if (s != prev_s) {
            T.vertex(x, y);
            prev_s = s;
        }
        int nx = x + DXc[s], ny = y + DYc[s];
        T.steps++;
        if (nx == sx && ny == sy && x == i1x && y == i1y) break;
        if (T.steps > max_steps) { T.status = 2; return; }
        x = nx; y = ny;
        s = (s + 4) & 7;
        n = M.nbr8(x, y);
    }
    T.edge(T.px, T.py, T.fx, T.fy); // close the polygon
}

// reference lib/ImageOperations.py:43-65 for one contour
__device__ void select_contour(ContourRec& r, double min_area, double min_circ)
{
    r.kept = 0; r.cx = r.cy = 0;
    double area = r.area, perimeter = r.perimeter;
    if (perimeter != 0.0) {
        double pi4 = 4 * 3.141592653589793;
        double circ = pi4 * area / (perimeter * perimeter);
        if (circ > min_circ && area > min_area) {
            double a00 = (double)r.a00, a10 = (double)r.a10, a01 = (double)r.a01;
            if (fabs(a00) > 1.1920928955078125e-07) {
                double h = 0.5, s = 0.16666666666666666666666666666667;
                if (a00 < 0) { h = -h; s = -s; }
                double m00 = a00 * h, m10 = a10 * s, m01 = a01 * s;
                if (m00 != 0) {
                    r.kept = 1;
                    r.cx = (int)(m10 / m00);
                    r.cy = (int)(m01 / m00);
                }
            }
        }
    }
}

} // namespace

__global__ __launch_bounds__(256) void contours_kernel(ContourArgs a)
{
    __shared__ uint32_t cand[MAXC];
    __shared__ ContourRec recs[MAXR];
    __shared__ int16_t kept_idx[MAXK];
    __shared__ int32_t kept_path[MAXK][MAXD];
    __shared__ int8_t kept_depth[MAXK];
    __shared__ int ncand, nrec, nkept, err;

    const int image = blockIdx.x;
    const int tid = threadIdx.x;
    Mask M{a.mask + (size_t)image * a.H * a.words_per_row, a.words_per_row, a.H, a.W, a.W + 1};
    if (tid == 0) { ncand = 0; nrec = 0; nkept = 0; err = 0; }
    __syncthreads();

    // ---- phase A: candidate starts ------------------------------------------------------------------------
    const int total_words = a.H * a.words_per_row;
    for (int i = tid; i < total_words; i += 256) {
        int y = i / a.words_per_row, k = i - y * a.words_per_row;
        uint32_t w = M.w[i];
        uint32_t lc = k > 0 ? (M.w[i - 1] >> 31) : 0u;
        if ((w | lc) == 0u) continue;
        uint32_t n = M.word(y - 1, k);
        uint32_t nlc = M.word(y - 1, k - 1) >> 31, nrc = M.word(y - 1, k + 1) & 1u;
        uint32_t Wn = (w << 1) | lc, NW = (n << 1) | nlc, NE = (n >> 1) | (nrc << 31);
        uint32_t outer = w & ~Wn & ~n & ~NW & ~NE;
        uint32_t hole = ~w & Wn & n;
        // pixels beyond the image width are background and can never be hole pixels
        int valid = a.W - 32 * k;
        if (valid < 32) hole &= (1u << valid) - 1u;
        while (outer) {
            int b = __ffs((int)outer) - 1;
            outer &= outer - 1;
            int slot = atomicAdd(&ncand, 1);
            if (slot < MAXC) cand[slot] = (uint32_t)(32 * k + b) | ((uint32_t)y << 16);
        }
        while (hole) {
            int b = __ffs((int)hole) - 1;
            hole &= hole - 1;
            int slot = atomicAdd(&ncand, 1);
            if (slot < MAXC) cand[slot] = (uint32_t)(32 * k + b) | ((uint32_t)y << 16) | 0x8000u;
        }
    }
    __syncthreads();
    if (ncand > MAXC) {
        if (tid == 0) { a.out_count[(size_t)image * a.count_stride] = BLOB_ERR_CANDIDATES; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }

    // ---- phase B: follow every candidate, keep the raster-first ones ---------------------------------------
    for (int c = tid; c < ncand; c += 256) {
        uint32_t v = cand[c];
        int is_hole = (v >> 15) & 1, x = v & 0x7fff, y = v >> 16;
        int key = y * M.RS + x;
        Trace T;
        if (!is_hole) follow(M, x, y, 4, key, -1, a.max_steps, T);
        else follow(M, x - 1, y, 0, -1, key, a.max_steps, T);
        if (T.status == 2) atomicMax(&err, 1);
        if (T.status != 0) continue;
        int slot = atomicAdd(&nrec, 1);
        if (slot >= MAXR) continue;
        ContourRec& r = recs[slot];
        r.key = key; r.is_hole = is_hole;
        r.sx = x - is_hole; r.sy = y;
        r.npts = T.npts; r.steps = T.steps;
        r.a00 = T.a00; r.a10 = T.a10; r.a01 = T.a01;
        r.area = fabs((double)T.a00 * 0.5);
        r.perimeter = T.npts > 1 ? T.per : 0.0;
        r.link = -1; r.parent = -1; r.order = -1;
        select_contour(r, a.min_area, a.min_circ);
    }
    __syncthreads();
    if (nrec > MAXR || err) {
        if (tid == 0) { a.out_count[(size_t)image * a.count_stride] = err ? BLOB_ERR_STEPS : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }

    // ---- phase C1: link = the border that owns the crack met when scanning left from the start ---------------
    //   outer border: nearest foreground pixel left of the start on the same row -> its East crack
    //   hole border : left end of the foreground run holding the start pixel     -> its West crack
    for (int c = tid; c < nrec; c += 256) {
        ContourRec& r = recs[c];
        int y = r.sy, qx = -1;
        if (!r.is_hole) {
            int x = r.sx - 1; // background
            for (int k = x >> 5; k >= 0 && x >= 0; k--) {
                uint32_t w = M.word(y, k);
                if (k == (x >> 5)) w &= (x & 31) == 31 ? 0xffffffffu : ((2u << (x & 31)) - 1u);
                if (w) { qx = 32 * k + 31 - __clz((int)w); break; }
            }
        } else {
            int x = r.sx; // foreground; find the nearest background pixel to the left
            qx = 0;
            for (int k = x >> 5; k >= 0; k--) {
                uint32_t w = ~M.word(y, k);
                if (k == (x >> 5)) w &= (x & 31) == 31 ? 0xffffffffu : ((2u << (x & 31)) - 1u);
                if (w) { qx = 32 * k + 31 - __clz((int)w) + 1; break; }
            }
        }
        if (qx < 0) { r.link = -1; continue; } // nothing to the left: the frame
        Trace T;
        follow(M, qx, y, r.is_hole ? 4 : 0, -1, -1, a.max_steps, T);
        if (T.status) { atomicMax(&err, 1); continue; }
        int ltype = T.a00 > 0 ? 1 : 0;            // hole borders run the other way round
        int lkey = ltype ? T.min_ebg : T.min_fg;
        int found = -2;
        for (int j = 0; j < nrec; j++)
            if (recs[j].key == lkey && recs[j].is_hole == ltype) { found = j; break; }
        if (found == -2) atomicMax(&err, 2);
        r.link = found;
    }
    __syncthreads();
    if (err) {
        if (tid == 0) { a.out_count[(size_t)image * a.count_stride] = err == 1 ? BLOB_ERR_STEPS : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }

    // ---- phase C2: parents (Suzuki's table: same kind -> the link's parent, else the link itself) -----------
    for (int c = tid; c < nrec; c += 256) {
        int me = recs[c].is_hole, j = recs[c].link, guard = 0;
        while (j >= 0 && recs[j].is_hole == me && guard++ < MAXR) j = recs[j].link;
        recs[c].parent = j;
    }
    __syncthreads();
    for (int c = tid; c < nrec; c += 256) {
        if (!recs[c].kept) continue;
        int slot = atomicAdd(&nkept, 1);
        if (slot >= MAXK) continue;
        kept_idx[slot] = (int16_t)c;
        int chain[MAXD], d = 0, j = c;
        while (j >= 0 && d < MAXD) { chain[d++] = recs[j].key; j = recs[j].parent; }
        if (j >= 0) { atomicMax(&err, 3); d = MAXD; }
        kept_depth[slot] = (int8_t)d;
        for (int i = 0; i < d; i++) kept_path[slot][i] = chain[d - 1 - i]; // root first
    }
    __syncthreads();
    if (err || nkept > MAXK) {
        if (tid == 0) { a.out_count[(size_t)image * a.count_stride] = err ? BLOB_ERR_DEPTH : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }

    // ---- phase C3: position in the pre-order walk with siblings in reverse discovery order -------------------
    for (int c = tid; c < nkept; c += 256) {
        int rank = 0, da = kept_depth[c];
        for (int o = 0; o < nkept; o++) {
            if (o == c) continue;
            int db = kept_depth[o], l = 0;
            while (l < da && l < db && kept_path[c][l] == kept_path[o][l]) l++;
            bool other_first;
            if (l == db) other_first = true;        // the other one is my ancestor
            else if (l == da) other_first = false;  // I am its ancestor
            else other_first = kept_path[o][l] > kept_path[c][l]; // later discovery comes first
            rank += other_first;
        }
        ContourRec& r = recs[kept_idx[c]];
        r.order = rank;
        if (rank < a.max_blobs) {
            int32_t* o = a.out_xy + (size_t)image * a.xy_stride + (size_t)rank * 2;
            o[0] = r.cx; o[1] = r.cy;
        }
    }
    if (tid == 0) a.out_count[(size_t)image * a.count_stride] = nkept;
    if (a.dbg) {
        for (int c = tid; c < nrec && c < a.dbg_cap; c += 256) a.dbg[(size_t)image * a.dbg_cap + c] = recs[c];
        if (tid == 0) a.dbg_count[image] = nrec;
    }
}

void launch_contours(const ContourArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(contours_kernel, dim3(a.n_images), dim3(256), 0, s, a);
}

} // namespace mocap
