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
// Candidates are found with word-parallel bit tests on the mask, guided by the occupancy words the filter kernels leave
// per tile and by the tiles' boxes (settle_tiles_kernel); a group of 8 lanes holds consecutive words of a mask row.  Every
// candidate becomes one entry of a batch-wide walk list and is followed by a PAIR OF LANES of contour_follow_kernel: one lane
// forwards, one backwards from the same start, until they meet (see there); a lane keeps the three 64-column mask rows around
// its current pixel in registers (a vertical move takes one new row from a 64 x 64 window staged in LDS), the walker state and
// the integer Green's-theorem sums are per-lane registers.  The step itself is literally the reference border-following step
// (same neighbour order, same CHAIN_APPROX_SIMPLE vertex rule); a candidate is dropped as soon as one of its lanes meets an
// earlier pixel of its own border.  The polygon sums are exact integers (int64), the perimeter is a sum of correctly rounded
// float32 square roots held exactly in a double.
// Tree order (parent = enclosing border, siblings in reverse discovery order, pre-order walk) is rebuilt from
// "which border owns the crack left of my start pixel": from the bounding boxes when that is unambiguous, else by
// following that border once -- as one more entry of a (second) walk list.
// Five launches per batch: contours_kernel<1> = the candidates kernel (one workgroup of 4 waves per image, or a fixed grid looping) -> contour_follow_kernel (persistent
// waves, every walk of the batch whatever image it belongs to) -> contours_kernel<2> (tree and output, one workgroup per image;
// images with an ambiguous link wait) -> contour_follow_kernel (the link walks) -> contours_kernel<2> (the images that waited);
// the hand-over is the per-image global workspace (L2) and the two batch-wide walk lists.  contours_kernel<0> is the same work
// as one kernel per image with the lone-lane walker `follow` below (contours_split = 0, and whenever contour_timing is on).
// The mask is 1/8 B per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace mocap {

namespace {

constexpr int MAXC = 1024;  // candidates per image (after the run-level filters)
constexpr int MAXR = 384;   // borders per image
constexpr int MAXK = 256;   // kept contours per image
constexpr int MAXD = 8;     // nesting depth of a kept contour
constexpr int MAXCELL = 4096; // occupancy cells (strip x 8 rows) scanned per image
constexpr int MAXA = 64;    // links per image whose owner has to be found by a walk that are handed to the packed second pass

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// lane L receives lane L - 1's / L + 1's value (0 at the ends of the wave and from lanes that are switched off)
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true); }

struct Mask {
    const uint32_t* w;
    int wpr, H, W;
    int RS; // raster stride W+1, so the virtual background column right of the image has its own index
    __device__ __forceinline__ uint32_t word(int y, int k) const
    {
        return ((unsigned)y < (unsigned)H && (unsigned)k < (unsigned)wpr) ? w[(size_t)y * wpr + k] : 0u;
    }
};

// columns x0 .. x0+63 of row y as a 64-bit word (bit c = column x0 + c), zero outside the image; per lane and
// branch-free: the three words are loaded from clamped in-image positions and zeroed by select
__device__ __forceinline__ uint64_t row64(const Mask& M, int y, int x0)
{
    const int k0 = x0 >> 5; // arithmetic shift = floor for negative x0
    const uint32_t sh = (uint32_t)x0 & 31u;
    const int yc = y < 0 ? 0 : (y > M.H - 1 ? M.H - 1 : y);
    const uint32_t* __restrict__ rowp = M.w + (uint32_t)yc * (uint32_t)M.wpr;
    const bool yin = (unsigned)y < (unsigned)M.H;
    uint32_t w[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int k = k0 + i, kc = k < 0 ? 0 : (k > M.wpr - 1 ? M.wpr - 1 : k);
        const uint32_t keep = (yin && (unsigned)k < (unsigned)M.wpr) ? 0xffffffffu : 0u;
        w[i] = rowp[kc] & keep; // an AND, not a select: the load stays unconditional (no branch around it)
    }
    const uint32_t lo = __builtin_amdgcn_alignbit(w[1], w[0], sh), hi = __builtin_amdgcn_alignbit(w[2], w[1], sh);
    return ((uint64_t)hi << 32) | lo;
}

// step of direction code s (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE), from packed 2-bit tables (value + 1) so that the
// scalar walker needs no memory access per step
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901au >> (2 * s)) & 3u) - 1; }  // 1,1,0,-1,-1,-1,0,1
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xa901u >> (2 * s)) & 3u) - 1; }  // 0,-1,-1,-1,0,1,1,1

struct Trace {
    int64_t a00, a10, a01; // Green's-theorem sums over the border polygon (exact)
    double per;            // cv.arcLength of the CHAIN_APPROX_SIMPLE polygon (float32 sqrt per segment, exact sum)
    int npts, steps;       // SIMPLE vertex count, border steps
    int min_fg;            // raster-minimum border pixel
    int min_ebg;           // raster-minimum background pixel right of a border pixel whose East side was examined
    int status;            // 0 ok, 1 aborted (not the raster-first start), 2 step limit
    int bx0, by0, bx1, by1; // bounding box of the border pixels
};

// float32 length of a straight run of k unit steps in direction code s, as cv.arcLength computes it
__device__ __forceinline__ double run_length(int s, int k)
{
    float d = (float)k;
    float q = (s & 1) ? __fadd_rn(__fmul_rn(d, d), __fmul_rn(d, d)) : __fmul_rn(d, d);
    // cv.arcLength takes a correctly rounded float32 square root (sqrtss); the device's float32 root (v_sqrt_f32) is good to 1 ulp
    // only, which showed on diagonal runs of a few lengths (tests/test_gpu_blob.py::test_contours_match_oracle_large_masks).  The
    // FP64 root of the (exact, integer-valued) float32 q, rounded to float32, IS the correctly rounded float32 root: sqrt(q) is never
    // within 2^-26 relative of a float32 midpoint for an integer q, and the FP64 root errs by 2^-53.
    return (double)(float)sqrt((double)q);
}

// Border following by one LANE.  Follows the border through pixel (sx,sy) whose neighbour in direction `first`
// (4 = W for an outer start, 0 = E for a hole start) is background.  Aborts when a border pixel with raster index
// < abort_fg or an East-side background pixel with raster index < abort_ebg is met.  Every lane of a wave follows
// its own border: the walker state, the three 64-column mask rows around the current pixel and all sums live in
// the lane's registers; a vertical move loads one new row (three mask words through L1/L2), a move near the edge of
// the 64-column window re-centres it.
//
// The polygon sums are accumulated per border step: splitting a straight polygon edge at the pixels it passes
// through leaves a00, a10, a01 unchanged (they are exact line integrals), so no vertex list is needed.  The
// perimeter needs the CHAIN_APPROX_SIMPLE segments: axis-parallel runs add their integer length, a diagonal run of
// k steps adds the float32 sqrt(2k^2) -- every term is a float32 >= 1 and the total stays far below 2^29, so the
// double sum is exact in any order.
// `win` (optional): 64 rows of the mask from row sy - 1 down, columns sx - 31 .. sx + 32, staged in LDS by the caller;
// rows are taken from there while the walk stays inside that window's columns.
// WS: distance (in 64-bit words) between consecutive rows of `win` (1: a window of its own; 64: row-major over the 64 windows
// of a wave, so that the lanes' reads fall into different LDS banks).
template <int WS = 1>
__device__ __forceinline__ void follow(const Mask& M, int sx, int sy, int first, int abort_fg, int abort_ebg, int max_steps,
                                       Trace& T, const double* diag_len, const uint64_t* win = nullptr)
{
    int64_t a00 = 0, a10 = 0, a01 = 0;
    int npts = 0, steps = 0;
    int min_fg = sy * M.RS + sx, min_ebg = 0x7fffffff;
    int bx0 = sx, bx1 = sx, by0 = sy, by1 = sy;
    T.status = 0;
    T.per = 0.0;

    int x0 = sx - 31; // window columns x0 .. x0+63
    int x = sx, y = sy;
    const int wy0 = sy - 1;
    bool staged = win != nullptr; // the LDS window still matches x0
    auto fetch = [&](int yy) -> uint64_t {
        const unsigned r = (unsigned)(yy - wy0);
        if (staged && r < 64u) return win[r * WS];
        return row64(M, yy, x0);
    };
    uint64_t rU = fetch(y - 1), rM = fetch(y), rD = fetch(y + 1);

    // occupancy of the 8 neighbours of (x,y), bit s = direction code s (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE)
    auto nbr8 = [&]() -> uint32_t {
        const int c = x - x0 - 1; // column x-1 at bit 0
        const uint32_t up = (uint32_t)(rU >> c) & 7u, mid = (uint32_t)(rM >> c) & 7u, dn = (uint32_t)(rD >> c) & 7u;
        const uint32_t up_rev = (0x73516240u >> (4u * up)) & 7u; // bit order NE, N, NW = columns x+1, x, x-1
        return (mid >> 2) | (up_rev << 1) | ((mid & 1u) << 4) | (dn << 5);
    };

    uint32_t n = nbr8();
    int s = first, s_end = first;
    do {
        s = (s - 1) & 7;
    } while (!((n >> s) & 1u) && s != s_end);
    if (s == s_end) { // isolated pixel: one vertex, zero area, zero perimeter
        T.a00 = T.a10 = T.a01 = 0;
        T.npts = 1; T.steps = 0;
        T.min_fg = min_fg; T.min_ebg = sy * M.RS + sx + 1;
        T.bx0 = T.bx1 = sx; T.by0 = T.by1 = sy;
        return;
    }
    const int i1x = sx + dir_dx(s), i1y = sy + dir_dy(s);
    int prev_s = s ^ 4;       // direction of the step that will close the border (arrives at the start)
    int run = 0;              // steps taken in direction prev_s since the last vertex
    int first_len = 0;        // length of the run leaving the start when the start is not a vertex (merged at the end)
    int axis = 0;             // total length of the axis-parallel segments
    double diag = 0.0;        // total length of the diagonal segments
    double pend = 0.0;        // table value fetched in the previous step, added one step later (hides the LDS latency)
    const int abort_lt = abort_fg > abort_ebg ? abort_fg : abort_ebg; // exactly one of the two is armed (the other is -1)
    const bool abort_on_fg = abort_fg >= 0;
    int status = 0;
    // The loop body is written with selects: a lane-divergent branch costs two EXEC updates and their wait states,
    // a select costs one instruction.  Branches remain only for leaving the loop, for re-centring the window and
    // for the two row sources.
    for (;;) {
        s_end = s;
        // first occupied neighbour counter-clockwise from s_end+1
        const uint32_t rot = ((n | (n << 8)) >> (s_end + 1)) & 0xffu;
        s = (s_end + __ffs((int)rot)) & 7;
        const int r = y * M.RS + x;
        const bool east_bg = (unsigned)(s - 1) < (unsigned)s_end; // the East neighbour was examined and is background
        const int re = east_bg ? r + 1 : 0x7fffffff;
        min_ebg = re < min_ebg ? re : min_ebg;
        min_fg = r < min_fg ? r : min_fg;
        // (x,y) is a CHAIN_APPROX_SIMPLE vertex when the direction changes: close the run that ends here
        const bool vertex = s != prev_s;
        const bool open_start = vertex && npts == 0 && steps > 0; // the start was not a vertex: its run is closed at the end
        first_len = open_start ? run : first_len;
        const int k = (vertex && !open_start) ? run : 0;
        const bool odd = (prev_s & 1) != 0;
        axis += odd ? 0 : k;
        diag += pend;
        const int kd = odd ? k : 0;                                  // diag_len[0] = 0
        pend = diag_len[kd < 63 ? kd : 63];
        if (kd > 63) pend = run_length(1, kd);                        // (a diagonal run longer than the table: rare)
        npts += vertex ? 1 : 0;
        prev_s = s;
        run = vertex ? 1 : run + 1;
        const int dx = dir_dx(s), dy = dir_dy(s);
        const int nx = x + dx, ny = y + dy;
        const int cross = x * dy - dx * y; // x*ny - nx*y
        a00 += cross;
        a10 += (int64_t)cross * (2 * x + dx);
        a01 += (int64_t)cross * (2 * y + dy);
        steps++;
        const bool aborted = (abort_on_fg ? r : re) < abort_lt;
        const bool closed = nx == sx && ny == sy && x == i1x && y == i1y;
        if (aborted || closed || steps > max_steps) {
            status = aborted ? 1 : (closed ? 0 : 2);
            break;
        }
        bx0 = nx < bx0 ? nx : bx0; bx1 = nx > bx1 ? nx : bx1;
        by0 = ny < by0 ? ny : by0; by1 = ny > by1 ? ny : by1;
        // move, keeping the three cached rows around the current pixel
        const int lx = nx - x0;
        if (lx < 1 || lx > 62) { // left the window: re-centre it on the new pixel (rare), or return to the staged one
            const int wl = nx - (sx - 31); // column of the new pixel in the staged window
            staged = win != nullptr && wl >= 1 && wl <= 62;
            x0 = staged ? sx - 31 : nx - 31;
            rU = fetch(ny - 1); rM = fetch(ny); rD = fetch(ny + 1);
        } else {
            const uint64_t nw = fetch(ny + dy); // (dy = 0: the middle row again, unused)
            const uint64_t oU = rU, oM = rM, oD = rD;
            rU = dy > 0 ? oM : (dy < 0 ? nw : oU);
            rM = dy > 0 ? oD : (dy < 0 ? oU : oM);
            rD = dy > 0 ? nw : (dy < 0 ? oM : oD);
        }
        x = nx; y = ny;
        s = (s + 4) & 7;
        n = nbr8();
    }
    T.status = status;
    if (status) return;
    // the run that arrives at the start, merged with the run that left it when the start is not a vertex
    diag += pend;
    {
        const int k = run + first_len;
        if (prev_s & 1) diag += k < 64 ? diag_len[k] : run_length(1, k);
        else axis += k;
    }
    T.a00 = a00; T.a10 = a10; T.a01 = a01;
    T.npts = npts; T.steps = steps;
    T.min_fg = min_fg; T.min_ebg = min_ebg;
    T.bx0 = bx0; T.bx1 = bx1; T.by0 = by0; T.by1 = by1;
    T.per = (double)axis + diag;
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

constexpr int NTHREADS = 256, NWAVES = NTHREADS / 64;
constexpr int NWIN = 16; // candidates per image whose mask window is staged in LDS

// per-image workspace in global memory (L2-resident): the full border records and the ancestor paths of the kept ones
struct ContourWork {
    ContourRec recs[MAXR];
    int32_t kept_path[MAXK][MAXD];
    // hand-over between the kernels of the split form (candidates -> follow -> tree [-> follow the ambiguous links -> tree])
    int32_t st_ncand;        // candidates of the image, or -1: the candidates kernel reported an error for it
    int32_t st_nrec;         // borders recorded by the follow kernel (atomic)
    int32_t st_err;          // follow kernel: 1 = a walk ran into the step limit
    int32_t st_pending;      // tree kernel, first pass: links left to the second follow pass (0 = the image is finished)
    int32_t rkey[MAXR];
    int16_t rsx[MAXR], rsy[MAXR];
    int16_t rbox[MAXR][4];
    uint8_t rhole[MAXR], rkept[MAXR];
    int16_t rlink[MAXR];     // first tree pass -> second: the links found so far (-2 = waits for its walk)
    int32_t link_key[MAXR];  // second follow pass: discovery key of the border that owns border c's link crack
    uint8_t link_type[MAXR]; //   and its kind (1 = hole border)
};

// One entry of the batch-wide walk lists (64 bits): x | y << 15 | kind << 30 | image << 32 | border << 52.
//   kind 0 / 1: a candidate start of an outer / a hole border at scan position (x, y) (contour_candidates_kernel);
//   kind 2 / 3: the link of border `border` of the image: follow the border through pixel (x, y) whose West (2) / East (3)
//               neighbour is background, to learn which border it is (tree kernel, first pass).
__device__ __forceinline__ uint64_t walk_entry(int image, int x, int y, int kind, int border = 0)
{
    return (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 15) | ((uint64_t)(uint32_t)kind << 30) | ((uint64_t)(uint32_t)image << 32) |
           ((uint64_t)(uint32_t)border << 52);
}
constexpr int MAX_SPLIT_IMAGES = 1 << 20; // image field of a walk entry

// MODE 0: the whole job for one image (candidates, walks, tree).  MODE 1 / MODE 2: the first and the last part of the split
// form -- candidates only (handed to contour_follow_kernel through the workspace and the batch-wide walk list) / tree only
// (from the records that kernel left).
template <int MODE>
__device__ __forceinline__ void contours_body(const ContourArgs& a, const int image)
{
    __shared__ uint32_t cand[MAXC];
    // phase A: cell_list (uint16 [MAXCELL]); afterwards: rbox (int16 [MAXR][4]), kept_idx (int16 [MAXK]), kept_depth (int8 [MAXK])
    __shared__ __attribute__((aligned(8))) uint8_t scratch[MAXCELL * 2];
    static_assert(MAXR * 8 + MAXK * 2 + MAXK <= MAXCELL * 2, "scratch overlay");
    __shared__ int32_t rkey[MAXR];                 // per border: discovery key, start pixel, kind, links
    __shared__ int16_t rsx[MAXR], rsy[MAXR], rlink[MAXR], rparent[MAXR];
    __shared__ uint8_t rhole[MAXR], rkept[MAXR];
    __shared__ uint8_t rres[MAXR];                 // 1 = the border's parent follows from the bounding boxes alone (phase C0)
    __shared__ int ncand, nrec, nkept, err, ncell, dbg_steps;
    __shared__ double diag_len[64]; // float32 length of a diagonal run of k steps, as a double
    __shared__ uint64_t win[NWIN][64]; // mask windows of the first NWIN candidates (see follow)
    uint16_t* const cell_list = (uint16_t*)scratch;
    uint16_t* const cell_rng = (uint16_t*)win;   // phase A only: first word | words << 12 of each listed cell's column range
    static_assert(sizeof(uint64_t) * NWIN * 64 >= MAXCELL * 2, "cell ranges overlay the windows");
    int16_t (*const rbox)[4] = (int16_t (*)[4])scratch;          // bounding box of each border: x0, y0, x1, y1
    int16_t* const kept_idx = (int16_t*)(scratch + MAXR * 8);
    int8_t* const kept_depth = (int8_t*)(scratch + MAXR * 8 + MAXK * 2);

    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = uni(tid >> 6);
    Mask M{a.mask + (size_t)image * a.H * a.words_per_row, a.words_per_row, a.H, a.W, a.W + 1};
    ContourWork& work = ((ContourWork*)a.work)[image];
    int32_t* const out_count = a.out_count + (size_t)image * a.count_stride;
    if (tid == 0) { ncand = 0; nrec = 0; nkept = 0; err = 0; ncell = 0; dbg_steps = 0; }
    if (tid < 64) diag_len[tid] = run_length(1, tid);
    // optional phase clock (MOCAP_CONTOUR_TIMING=1, a debugging aid): 100 MHz ticks at the phase boundaries
    uint64_t* const tick = a.timing ? a.timing + (size_t)image * 8 : nullptr;
    auto stamp = [&](int i) { if (tick && tid == 0) tick[i] = wall_clock64(); };
    stamp(0);
    __syncthreads();
    __shared__ uint64_t amb[MAXA]; // MODE 2, first pass: the links left to the second follow pass
    __shared__ int n_amb;
    const bool second_pass = MODE == 2 && a.tree_pass == 2;
    if constexpr (MODE == 2) {
        // the records of the follow kernel: counts, then the small per-border fields into LDS
        if (work.st_ncand < 0) return; // the candidates kernel has reported this image's error
        if (second_pass && work.st_pending <= 0) return; // finished by the first pass
        if (tid == 0) { nrec = work.st_nrec; err = work.st_err; n_amb = 0; }
        __syncthreads();
        if (nrec > MAXR || err) {
            if (tid == 0) { *out_count = err ? BLOB_ERR_STEPS : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
            return;
        }
        for (int c = tid; c < nrec; c += NTHREADS) {
            rkey[c] = work.rkey[c]; rsx[c] = work.rsx[c]; rsy[c] = work.rsy[c];
            rhole[c] = work.rhole[c]; rkept[c] = work.rkept[c];
            rlink[c] = second_pass ? work.rlink[c] : (int16_t)-1; rparent[c] = -1;
            rbox[c][0] = work.rbox[c][0]; rbox[c][1] = work.rbox[c][1]; rbox[c][2] = work.rbox[c][2]; rbox[c][3] = work.rbox[c][3];
        }
        __syncthreads();
    }

    // ---- phase A: candidate starts -------------------------------------------------------------------------------
    if constexpr (MODE != 2) {
    // A border can only start where the mask has set pixels.  The filter kernel leaves an occupancy word per
    // (strip, chunk): bit g = rows 8g..8g+7 of the chunk contain set pixels in that 240-column strip.  Occupied
    // cells, plus their right and lower neighbours (a hole can start in an empty cell whose W / N neighbour pixel
    // lies in the occupied one), are scanned row by row with word-parallel bit tests; without the occupancy words
    // (mask supplied by the caller) every cell is scanned.
    {
        const int R = a.rows_per_chunk, NS = a.n_strips, NCH = a.n_chunks;
        const int gpc = (R + 7) >> 3;                       // 8-row groups per chunk
        const uint32_t* cells = a.cells ? a.cells + (size_t)image * NCH * NS : nullptr;
        const uint32_t* boxes = (a.cells && a.boxes) ? a.boxes + (size_t)image * NCH * NS * 4 : nullptr;
        const bool ranged = cells != nullptr; // cell_rng holds the words to examine (the whole strip without boxes)
        const int n_cells = NCH * NS * gpc;
        if (cells) {
            // one task = one (chunk, strip) occupancy word: its own groups, and the groups its right and lower
            // neighbours must scan because of it
            for (int t = tid; t < NCH * NS; t += NTHREADS) {
                const int ch = t / NS, st = t - ch * NS;
                if (ch * R >= a.H) continue;
                const uint32_t own = cells[t] & 0x7fffffffu;
                const uint32_t left = st > 0 ? cells[t - 1] & 0x7fffffffu : 0u;
                const uint32_t up = ch > 0 ? cells[t - NS] & 0x7fffffffu : 0u;
                const uint32_t upbit = (up >> (gpc - 1)) & 1u;
                uint32_t scan = own | left | (own << 1) | upbit;
                scan &= (1u << gpc) - 1u;
                if (!scan) continue;
                // columns that can hold set pixels: the tile's output region and the scan's box (settle, BoxArgs::cur_box);
                // a hole start lies at most one column right of them
                const int xa = 240 * st, xb = xa + 240 < a.W ? xa + 240 : a.W;
                int ox0 = xa, ox1 = xb - 1, ux0 = xa, ux1 = xb - 1;
                if (boxes) {
                    const uint4 ob = *(const uint4*)(boxes + 4 * (size_t)t);
                    const int r0 = (int)(ob.x & 0xffffu), r1 = (int)(ob.x >> 16), b0 = (int)(ob.z & 0xffffu) & ~7, b1 = (int)(ob.z >> 16) | 7;
                    if (r0 <= r1) { ox0 = r0 > b0 ? r0 : b0; ox1 = (r1 < b1 ? r1 : b1) + 1; }
                    if (upbit) {
                        const uint4 ub = *(const uint4*)(boxes + 4 * (size_t)(t - NS));
                        const int u0 = (int)(ub.x & 0xffffu), u1 = (int)(ub.x >> 16), c0 = (int)(ub.z & 0xffffu) & ~7, c1 = (int)(ub.z >> 16) | 7;
                        if (u0 <= u1) { ux0 = u0 > c0 ? u0 : c0; ux1 = (u1 < c1 ? u1 : c1) + 1; }
                    }
                    ox0 = ox0 < xa ? xa : ox0; ox1 = ox1 > xb - 1 ? xb - 1 : ox1;
                    ux0 = ux0 < xa ? xa : ux0; ux1 = ux1 > xb - 1 ? xb - 1 : ux1;
                    if (ox0 > ox1) { ox0 = xa; ox1 = xb - 1; }
                    if (ux0 > ux1) { ux0 = xa; ux1 = xb - 1; }
                }
                const uint32_t ownish = own | (own << 1);
                while (scan) {
                    const int g = __ffs((int)scan) - 1;
                    scan &= scan - 1;
                    if (ch * R + 8 * g >= a.H || 8 * g >= R) continue;
                    const int slot = atomicAdd(&ncell, 1);
                    if (slot >= MAXCELL) continue;
                    cell_list[slot] = (uint16_t)((ch * NS + st) * gpc + g);
                    int c0 = 0x7fffffff, c1 = -1;
                    if ((ownish >> g) & 1u) { c0 = ox0; c1 = ox1; }
                    if ((left >> g) & 1u) { c0 = c0 < xa ? c0 : xa; c1 = c1 > xa ? c1 : xa; }
                    if (g == 0 && upbit) { c0 = c0 < ux0 ? c0 : ux0; c1 = c1 > ux1 ? c1 : ux1; }
                    const int k0 = c0 >> 5, k1 = c1 >> 5;
                    cell_rng[slot] = (uint16_t)(k0 | ((k1 - k0 + 1) << 12)); // k0 < 4096 (checked on the host), at most 10 words
                }
            }
            __syncthreads();
            if (ncell > MAXCELL && tid == 0) atomicMax(&err, 4);
        }
        const int ncl = cells ? (ncell < MAXCELL ? ncell : MAXCELL) : n_cells;
        // One task = one cell, taken by a group of 8 lanes: lane i of the group holds word kf - 1 + i of a row (kf = first
        // word of the cell's column range, see above) and of the row above it, so a wave's loads cover 8 rows x 32 contiguous
        // bytes; lanes 1..6 test their word with the neighbours' words from lanes i - 1 / i + 1 (DPP), ranges longer than 6
        // words take further passes.  The 16 loads of a cell's 8 rows are in flight together.
        const int sub = tid & 7, grp8 = tid >> 3;
        for (int ci0 = 0; ci0 < ncl; ci0 += NTHREADS / 8) {
            const int ci = ci0 + grp8;
            const bool cv = ci < ncl;
            const int cell = cv ? (cells ? (int)cell_list[ci] : ci) : 0;
            const int g = cell % gpc, st = (cell / gpc) % NS, ch = cell / (gpc * NS);
            const int y0 = ch * R + 8 * g;
            const int yend = (ch + 1) * R < a.H ? (ch + 1) * R : a.H;
            const int xa = 240 * st, xb = xa + 240 < a.W ? xa + 240 : a.W; // the strip's columns [xa, xb)
            const int ka = xa >> 5, kb = (xb - 1) >> 5;
            int kf = ka, cnt = kb - ka + 1;
            if (ranged && cv) { const uint32_t rg = cell_rng[ci]; kf = (int)(rg & 0xfffu); cnt = (int)(rg >> 12); }
            if (!cv) cnt = 0;
            for (int p0 = 0; p0 < cnt; p0 += 6) {
                const int k = kf - 1 + p0 + sub;
                const bool tests = sub >= 1 && sub <= 6 && k < kf + cnt && k >= ka && k <= kb;
                // the cell's 8 rows and the row above the first: 9 loads in flight together (the row above row j is row j - 1)
                uint32_t rw[9];
                rw[0] = cv ? M.word(y0 - 1, k) : 0u;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int y = y0 + j;
                    const bool rowv = y < yend && 8 * g + j < R;
                    rw[1 + j] = rowv ? M.word(y, k) : 0u;
                }
                // the neighbours' words, exchanged while every lane of the group is active (before any lane-divergent code)
                uint32_t pr[9], nx[9];
#pragma unroll
                for (int j = 0; j < 9; j++) { pr[j] = lane_prev(rw[j]); nx[j] = lane_next(rw[j]); }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int y = y0 + j;
                    const uint32_t w = rw[1 + j], n = rw[j];
                    const uint32_t prev_w = pr[1 + j], prev_n = pr[j], next_w = nx[1 + j], next_n = nx[j];
                    if (!tests || !(y < yend && 8 * g + j < R)) continue;
                    const uint32_t Wn = (w << 1) | (prev_w >> 31);
                    // Necessary conditions, evaluated on the 64 columns starting at this word (this word + the next):
                    // a raster-first foreground pixel starts a run none of whose pixels touches (8-connectivity) the
                    // row above; a raster-first hole pixel starts a background run none of whose pixels has
                    // background directly above (4-connectivity).  "Touches" are spread leftwards along the run for 12
                    // columns; beyond that the candidate is merely kept -- the follow step decides.
                    const uint64_t w64 = (uint64_t)w | ((uint64_t)next_w << 32), n64 = (uint64_t)n | ((uint64_t)next_n << 32);
                    const uint64_t above64 = n64 | (n64 << 1) | (uint64_t)(prev_n >> 31) | (n64 >> 1); // NE of column 63 unknown: treated as clear
                    uint32_t outer = w & ~Wn & ~(uint32_t)above64;
                    if (outer) {
                        uint64_t touch = w64 & above64;
#pragma unroll
                        for (int i = 0; i < 12; i++) touch |= (touch >> 1) & w64;
                        outer &= ~(uint32_t)touch;
                    }
                    uint32_t hole = ~w & Wn & n;
                    if (hole) {
                        uint64_t bg64 = ~w64, touch = bg64 & ~n64;
#pragma unroll
                        for (int i = 0; i < 12; i++) touch |= (touch >> 1) & bg64;
                        hole &= ~(uint32_t)touch;
                    }
                    // keep only this strip's columns (and, for holes, columns inside the image)
                    const int lo = xa - 32 * k, hi = xb - 32 * k; // bit range [lo, hi)
                    uint32_t m = 0xffffffffu;
                    if (lo > 0) m &= ~((1u << lo) - 1u);
                    if (hi < 32) m &= (1u << hi) - 1u;
                    outer &= m; hole &= m;
                    while (outer) {
                        const int b = __ffs((int)outer) - 1;
                        outer &= outer - 1;
                        const int slot = atomicAdd(&ncand, 1);
                        if (slot < MAXC) cand[slot] = (uint32_t)(32 * k + b) | ((uint32_t)y << 16);
                    }
                    while (hole) {
                        const int b = __ffs((int)hole) - 1;
                        hole &= hole - 1;
                        const int slot = atomicAdd(&ncand, 1);
                        if (slot < MAXC) cand[slot] = (uint32_t)(32 * k + b) | ((uint32_t)y << 16) | 0x8000u;
                    }
                }
            }
        }
    }
    __syncthreads();
    stamp(1);
    if (ncand > MAXC || err) {
        if (tid == 0) { *out_count = BLOB_ERR_CANDIDATES; if (a.dbg_count) a.dbg_count[image] = 0; if (MODE == 1) work.st_ncand = -1; }
        return;
    }
    }
    if constexpr (MODE == 1) {
        // hand the candidates over: one self-contained entry each into the batch's walk list
        __shared__ uint32_t wbase;
        const int nc1 = ncand;
        if (tid == 0) {
            work.st_ncand = nc1; work.st_nrec = 0; work.st_err = 0; work.st_pending = 0;
            wbase = nc1 ? atomicAdd(&a.walk_count[0], (uint32_t)nc1) : 0u;
        }
        __syncthreads();
        for (int c = tid; c < nc1; c += NTHREADS) {
            const uint32_t v = cand[c];
            a.walk_list[wbase + (uint32_t)c] = walk_entry(image, (int)(v & 0x7fffu), (int)(v >> 16), (int)((v >> 15) & 1u));
        }
        return;
    }

    // ---- phase B: one lane follows one candidate; the raster-first ones become records ---------------------------
    if constexpr (MODE == 0) {
    const int nc = ncand;
    // The walks read the mask rows below each start one at a time.  For the first NWIN candidates (all of them, in a
    // typical frame) the 64 rows from the start downwards are staged in LDS first, one row per lane: one round of
    // parallel loads instead of a dependent L2 round trip per vertical move.
    for (int c = wv; c < nc && c < NWIN; c += NWAVES) {
        const uint32_t v = cand[c];
        const int sx = (int)(v & 0x7fff) - (int)((v >> 15) & 1u), sy = (int)(v >> 16);
        win[c][lane] = row64(M, sy - 1 + lane, sx - 31);
    }
    __syncthreads();
    for (int c = tid; c < nc; c += NTHREADS) {
        const uint32_t v = cand[c];
        const int is_hole = (v >> 15) & 1, x = v & 0x7fff, y = v >> 16;
        const int key = y * M.RS + x;
        Trace T;
        const uint64_t* w = c < NWIN ? win[c] : nullptr;
        if (!is_hole) follow(M, x, y, 4, key, -1, a.max_steps, T, diag_len, w);
        else follow(M, x - 1, y, 0, -1, key, a.max_steps, T, diag_len, w);
        if (T.status == 2) atomicMax(&err, 1);
        if (tick) atomicMax(&dbg_steps, T.status == 0 ? T.steps : 0);
        if (T.status != 0) continue;
        const int slot = atomicAdd(&nrec, 1);
        if (slot >= MAXR) continue;
        ContourRec r;
        r.key = key; r.is_hole = is_hole;
        r.sx = x - is_hole; r.sy = y;
        r.npts = T.npts; r.steps = T.steps;
        r.a00 = T.a00; r.a10 = T.a10; r.a01 = T.a01;
        r.area = fabs((double)T.a00 * 0.5);
        r.perimeter = T.npts > 1 ? T.per : 0.0;
        r.link = -1; r.parent = -1; r.order = -1;
        select_contour(r, a.min_area, a.min_circ);
        work.recs[slot] = r;
        rkey[slot] = key; rsx[slot] = (int16_t)r.sx; rsy[slot] = (int16_t)y;
        rhole[slot] = (uint8_t)is_hole; rkept[slot] = (uint8_t)r.kept;
        rlink[slot] = -1; rparent[slot] = -1;
        rbox[slot][0] = (int16_t)T.bx0; rbox[slot][1] = (int16_t)T.by0;
        rbox[slot][2] = (int16_t)T.bx1; rbox[slot][3] = (int16_t)T.by1;
    }
    __syncthreads();
    stamp(2);
    if (tick && tid == 0) { tick[5] = (uint64_t)ncand; tick[6] = (uint64_t)dbg_steps; tick[7] = (uint64_t)nrec; }
    if (nrec > MAXR || err) {
        if (tid == 0) { *out_count = err ? BLOB_ERR_STEPS : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }
    }

    // ---- phase C1: link = the border that owns the crack met when scanning left from the start ------------------
    //   outer border: nearest foreground pixel left of the start on the same row -> its East crack
    //   hole border : left end of the foreground run holding the start pixel     -> its West crack
    // One wave per border (wave-uniform control flow).
    const int nr = nrec;
    // ---- phase C0: the parents the bounding boxes alone decide (no mask access, no link) ----------------------------
    //   An outer border's parent is the hole border of the hole its component lies in, or the frame; lying in a hole puts every
    //   pixel of the border, its start included, inside that hole border's box: no hole border's box around the start -> the frame.
    //   A hole border's parent is the outer border of its own component, whose box contains every pixel of the component, the
    //   hole's border pixels included: exactly one outer border's box around the hole border's box -> that one.
    //   Everything else (nested rings, boxes that overlap) goes through the link below.  A frame of separate, hole-free markers --
    //   the usual IR frame -- needs nothing more than this.
    for (int c = tid; c < nr; c += NTHREADS) {
        const int me = rhole[c];
        int hits = 0, found = -1;
        if (!me) {
            const int x = rsx[c], y = rsy[c];
            for (int j = 0; j < nr; j++)
                hits += rhole[j] && rbox[j][0] <= x && x <= rbox[j][2] && rbox[j][1] <= y && y <= rbox[j][3];
        } else {
            for (int j = 0; j < nr; j++)
                if (!rhole[j] && rbox[j][0] <= rbox[c][0] && rbox[c][2] <= rbox[j][2] && rbox[j][1] <= rbox[c][1] && rbox[c][3] <= rbox[j][3]) { hits++; found = j; }
        }
        const bool res = me ? hits == 1 : hits == 0;
        rres[c] = res ? 1 : 0;
        if (res) rparent[c] = (int16_t)found;
    }
    __syncthreads();
    if (second_pass) {
        // the links the first pass left open: the second follow pass has identified the border each of their cracks belongs to
        for (int c = tid; c < nr; c += NTHREADS) {
            if (rlink[c] != -2) continue;
            const int lkey = work.link_key[c], ltype = work.link_type[c];
            int found = -2;
            for (int j = 0; j < nr; j++)
                if (rkey[j] == lkey && rhole[j] == ltype) { found = j; break; }
            if (found == -2) atomicMax(&err, 2);
            rlink[c] = (int16_t)found;
        }
    }
    for (int c = wv; c < nr && !second_pass; c += NWAVES) {
        if (rres[c]) continue; // (wave-uniform)
        const int r_is_hole = rhole[c], r_sx = rsx[c], y = rsy[c];
        // hole: nearest background pixel at/left of the start; outer: nearest foreground pixel left of it.
        // The words of the row up to that column are examined 64 at a time, one per lane, right to left.
        const int xs = r_is_hole ? r_sx : r_sx - 1;
        int qx = r_is_hole ? 0 : -1;
        for (int kbase = xs >> 5; kbase >= 0 && xs >= 0; kbase -= 64) {
            int k = kbase - lane;
            uint32_t w = k >= 0 ? M.word(y, k) : 0u;
            if (r_is_hole) w = k >= 0 ? ~w : 0u;
            if (k == (xs >> 5)) w &= (2u << (xs & 31)) - 1u; // only columns <= xs
            uint64_t bal = __ballot(w != 0u);
            if (bal) {
                int src = __ffsll((long long)bal) - 1; // lowest lane = rightmost word
                uint32_t ww = (uint32_t)__builtin_amdgcn_readlane((int)w, src);
                int px = 32 * (kbase - src) + 31 - __clz((int)ww);
                qx = r_is_hole ? px + 1 : px;
                break;
            }
        }
        if (qx < 0) continue; // nothing to the left: the frame (link stays -1)
        // The crack's owner passes through pixel (qx, y), so its bounding box contains it, and it is never this
        // border itself (its pixels are all raster-later than its start).  If exactly one other border's box
        // contains the pixel, that border is the owner; only otherwise is the owner found by following it.
        int found = -2;
        {
            int hits = 0, which = -1;
            for (int jb = 0; jb < nr; jb += 64) {
                int j = jb + lane;
                bool in = j < nr && j != c && rbox[j][0] <= qx && qx <= rbox[j][2] && rbox[j][1] <= y && y <= rbox[j][3];
                uint64_t bal = __ballot(in);
                hits += __popcll(bal);
                if (bal && which < 0) which = jb + __ffsll((long long)bal) - 1;
            }
            if (hits == 1) found = which;
        }
        if (MODE == 2 && found == -2 && a.defer_links) {
            // split form: the walk joins the batch's second packed follow pass (one lane there, not a whole wave here)
            int slot = 0;
            if (lane == 0) slot = atomicAdd(&n_amb, 1);
            slot = uni(slot);
            if (slot < MAXA) {
                if (lane == 0) { amb[slot] = walk_entry(image, qx, y, r_is_hole ? 2 : 3, c); rlink[c] = -2; }
                continue;
            }
        }
        if (found == -2) {
            Trace T; // every lane walks the same border (uniform arguments): rare path
            follow(M, qx, y, r_is_hole ? 4 : 0, -1, -1, a.max_steps, T, diag_len);
            if (T.status) { if (lane == 0) atomicMax(&err, 1); continue; }
            int ltype = T.a00 > 0 ? 1 : 0; // hole borders run the other way round
            int lkey = ltype ? T.min_ebg : T.min_fg;
            for (int jb = 0; jb < nr; jb += 64) {
                int j = jb + lane;
                bool hit = j < nr && rkey[j] == lkey && rhole[j] == ltype;
                uint64_t bal = __ballot(hit);
                if (bal) { found = jb + __ffsll((long long)bal) - 1; break; }
            }
        }
        if (lane == 0) {
            if (found == -2) atomicMax(&err, 2);
            rlink[c] = (int16_t)found;
        }
    }
    __syncthreads();
    stamp(3);
    if (err) {
        if (tid == 0) { *out_count = err == 1 ? BLOB_ERR_STEPS : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }
    if constexpr (MODE == 2) {
        if (!second_pass && n_amb > 0) { // hand the open links to the second follow pass; this image is finished by the second tree pass
            __shared__ uint32_t lbase;
            const int na = n_amb < MAXA ? n_amb : MAXA;
            if (tid == 0) {
                lbase = atomicAdd(&a.walk_count[2], (uint32_t)na); work.st_pending = na;
                a.wait_list[atomicAdd(&a.walk_count[4], 1u)] = (uint32_t)image; // the second tree pass runs over this list
            }
            __syncthreads();
            for (int i = tid; i < na; i += NTHREADS) a.link_list[lbase + (uint32_t)i] = amb[i];
            for (int c = tid; c < nr; c += NTHREADS) work.rlink[c] = rlink[c];
            return;
        }
    }

    // ---- phase C2: parents (Suzuki's table: same kind -> the link's parent, else the link itself) -----------
    for (int c = tid; c < nr; c += NTHREADS) {
        if (rres[c]) continue;
        int me = rhole[c], j = rlink[c], guard = 0;
        while (j >= 0 && rhole[j] == me && guard++ < MAXR) {
            if (rres[j]) { j = rparent[j]; break; } // a border of my kind whose parent the boxes gave: its parent is mine
            j = rlink[j];
        }
        rparent[c] = (int16_t)j;
    }
    __syncthreads();
    for (int c = tid; c < nr; c += NTHREADS) {
        if (!rkept[c]) continue;
        int slot = atomicAdd(&nkept, 1);
        if (slot >= MAXK) continue;
        kept_idx[slot] = (int16_t)c;
        int chain[MAXD], d = 0, j = c;
        while (j >= 0 && d < MAXD) { chain[d++] = rkey[j]; j = rparent[j]; }
        if (j >= 0) { atomicMax(&err, 3); d = MAXD; }
        kept_depth[slot] = (int8_t)d;
        for (int i = 0; i < d; i++) work.kept_path[slot][i] = chain[d - 1 - i]; // root first
    }
    __syncthreads();
    if (err || nkept > MAXK) {
        if (tid == 0) { *out_count = err ? BLOB_ERR_DEPTH : BLOB_ERR_CONTOURS; if (a.dbg_count) a.dbg_count[image] = 0; }
        return;
    }

    // ---- phase C3: position in the pre-order walk with siblings in reverse discovery order -------------------
    for (int c = tid; c < nkept; c += NTHREADS) {
        int rank = 0, da = kept_depth[c];
        for (int o = 0; o < nkept; o++) {
            if (o == c) continue;
            int db = kept_depth[o], l = 0;
            while (l < da && l < db && work.kept_path[c][l] == work.kept_path[o][l]) l++;
            bool other_first;
            if (l == db) other_first = true;        // the other one is my ancestor
            else if (l == da) other_first = false;  // I am its ancestor
            else other_first = work.kept_path[o][l] > work.kept_path[c][l]; // later discovery comes first
            rank += other_first;
        }
        ContourRec& r = work.recs[kept_idx[c]];
        r.order = rank;
        if (rank < a.max_blobs) {
            int32_t* o = a.out_xy + (size_t)image * a.xy_stride + (size_t)rank * 2;
            o[0] = r.cx; o[1] = r.cy;
        }
    }
    if (tid == 0) *out_count = nkept;
    stamp(4);
    if (a.dbg) {
        __syncthreads(); // the records' order fields are written by other threads just above
        for (int c = tid; c < nr && c < a.dbg_cap; c += NTHREADS) {
            ContourRec r = work.recs[c];
            r.link = rlink[c]; r.parent = rparent[c];
            a.dbg[(size_t)image * a.dbg_cap + c] = r;
        }
        if (tid == 0) a.dbg_count[image] = nr;
    }
}

// One workgroup per image.  The second tree pass runs over the list of the images that wait for link walks (mostly none: its
// workgroups read one counter and leave).  (The kernels as loops over the images, fewer workgroups than images: 19 more registers for
// the candidates kernel, 42 for the tree kernel, 8 us slower each, nothing gained in the pipeline -- profiles/README.md.)
// LOOP: a fixed grid (a few workgroups per CU) works through the images -- workgroup b takes b, b + grid, ... -- instead of one
// workgroup per image.  Alone that is a few microseconds slower (more registers), but beside another batch's streaming scan, which
// holds every wave slot of the chip, each of 3072 four-wave workgroups has to wait for a place of its own: the tree kernel, 12 us
// alone, took 0.34-0.60 ms there, the candidates kernel 0.27-0.52 instead of 0.08 (profiles/history/r4_timeline_depth3.txt).
template <int MODE, bool LOOP>
__global__ __launch_bounds__(NTHREADS) void contours_kernel(ContourArgs a)
{
    if (MODE == 2 && a.tree_pass == 2) { // the images that waited for link walks (mostly none): a small grid over their list
        const uint32_t n = a.walk_count[4];
        for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
            contours_body<MODE>(a, uni((int)a.wait_list[e]));
            __syncthreads(); // (the next image reuses the workgroup's LDS)
        }
        return;
    }
    if (!LOOP) {
        contours_body<MODE>(a, blockIdx.x);
        return;
    }
    for (int image = blockIdx.x; image < a.n_images; image += gridDim.x) {
        contours_body<MODE>(a, image);
        __syncthreads();
    }
}

// The walks of the whole batch, whatever image they belong to, TWO LANES PER BORDER: lane 2i follows the border forwards from
// its start, lane 2i + 1 backwards from the same start (border following is reversible: the backward walk is the forward rule
// on the vertically mirrored neighbourhood), and the pair stops where the two meet -- half the steps of the longest border, which
// is what the kernel's duration comes down to (a merged pair of markers has a border of ~900 steps, ~0.5 us each).  Each lane
// accounts for the forward steps it covers (Green sums, vertices, runs, raster minima, box); the sums add, the two runs that
// straddle the seams (at the start pixel and at the meeting point) are joined before their float32 lengths are taken, exactly
// as the reference measures the whole polygon.
// The waves are PERSISTENT and refill their lane pairs: every FOLLOW_K steps a wave looks at its lanes; pairs whose walk has
// ended store their record (one round of atomics and stores for all of them), walks that have reached the rim of their mask
// window pause until it is staged anew around them, and when enough pairs are idle the wave takes that many new entries from
// the batch-wide list (one atomic on its head).  The windows (64 rows x 64 columns, in LDS, row-major over the lanes:
// bank-conflict-free for lanes at different rows) are staged by the whole wave, lane = row, eight windows per round; a pair
// shares one window until one of its lanes leaves it.  No global memory access happens inside the step loop.
// The step is `follow` above cut into resumable, direction-symmetric steps: same neighbour search, same vertex rule, same sums.
#ifndef FOLLOW_K_STEPS
#define FOLLOW_K_STEPS 16
#endif
constexpr int FOLLOW_K = FOLLOW_K_STEPS; // steps between two looks at the lanes
constexpr int FOLLOW_REFILL = 8;  // idle pairs that make a refill worth its three dependent memory round trips
#ifndef STAGE_N
#define STAGE_N 8                 // windows staged per round (3 loads each in flight together)
#endif

__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true); }
__device__ __forceinline__ int64_t pair_swap64(int64_t v)
{
    const uint32_t lo = (uint32_t)pair_swap((int)(uint32_t)v), hi = (uint32_t)pair_swap((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double pair_swap_f64(double v) { return __longlong_as_double(pair_swap64(__double_as_longlong(v))); }

struct Walk { // one lane's half of a walk
    int r, r_ahead;      // raster key y * (W + 1) + x of the current pixel; backward lane: of the pixel ahead of it (the one it came from)
    int64_t a00, a10, a01;
    double diag, pend;
    int sx, sy, x, y, known; // start pixel of the border; current pixel; the direction this lane knows there: the way back
                             //   (forward lane) / the way forward (backward lane)
    int run, head, axis, npts, steps, min_fg, min_ebg, bx0, bx1, by0, by1;
    int wx0, wy0, wslot; // the lane's window: columns wx0 .. wx0 + 63, rows wy0 .. wy0 + 63 of its image's mask, in LDS column wslot
    int abort_lt, key, s0;
    uint32_t n, meta;    // n: neighbourhood (mirrored for the backward lane); meta: kind | border << 2 | image << 11
    bool abort_on_fg, has_head;
    int status;          // 0 closed, 1 aborted, 2 step limit
};

__global__ __launch_bounds__(64) void contour_follow_kernel(ContourArgs a)
{
    __shared__ uint64_t win[64][64]; // [row of the window][window]
    __shared__ double diag_len[64];
    const int lane = threadIdx.x;
    const bool isB = (lane & 1) != 0;
    diag_len[lane] = run_length(1, lane);
    const uint64_t* const list = a.follow_list ? a.link_list : a.walk_list;
    const uint32_t total = a.walk_count[2 * a.follow_list];
    uint32_t* const head_ctr = &a.walk_count[2 * a.follow_list + 1];
    ContourWork* const works = (ContourWork*)a.work;
    const uint32_t image_words = (uint32_t)a.H * (uint32_t)a.words_per_row;
    const int RS = a.W + 1;
    Walk w;
    w.status = 0; w.meta = 0; w.n = 0; w.known = 0; w.x = 0; w.y = 0; w.wx0 = 0; w.wy0 = 0; w.wslot = lane; w.steps = 0; w.s0 = 0;
    // A lane is idle, or holds half a walk that is running, or paused (its pixel lies on the rim of its window: it waits for the
    // next look at the lanes), or finished (the pair has met, or given up; the record is not stored yet).
    bool active = false, paused = false, finished = false;
    bool drained = false;                  // (wave-uniform) the list has no more entries
    bool first_fill = true;                // (wave-uniform) the wave has not taken entries yet
    // optional phase clock (follow_timing = 1, a debugging aid): per wave, 100 MHz ticks in store / refill / walk, wave steps, lane steps, walks, refills
    uint64_t tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool clk = a.follow_dbg != nullptr && a.follow_list == a.follow_dbg_list;
    uint64_t t_prev = clk ? wall_clock64() : 0;
    auto lap = [&](int i) { if (clk) { const uint64_t t = wall_clock64(); tk[i] += t - t_prev; t_prev = t; } };
    __syncthreads();
    // occupancy of the 8 neighbours of (x,y), bit s = direction code s (0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE) -- for the backward
    // lane of the vertically mirrored image (rows swapped: NE <-> SE, N <-> S, NW <-> SW).  Straight from the lane's window: the
    // three columns x-1 .. x+1 of a row lie in the 16 bits at byte (x - 1 - wx0) >> 3 of the window row (one ds_read_u16 each; the
    // pixel is inside the rim, see the pause rule -- a paused lane reads some bytes of the array and does not use them).
    const uint8_t* const win_bytes = (const uint8_t*)&win[0][0];
    const int up_dy = isB ? 1 : -1; // the row that plays "up"
    auto nbr8 = [&](int x, int y, int up_dy) -> uint32_t {
        const int c = x - w.wx0 - 1; // column x-1 at bit 0
        const uint32_t col = (uint32_t)w.wslot * 8u + (((uint32_t)c >> 3) & 7u), sh = (uint32_t)c & 7u;
        const int ly = y - w.wy0;
        uint16_t vu, vm, vd;
        __builtin_memcpy(&vu, win_bytes + ((((uint32_t)(ly + up_dy) & 63u) << 9) + col), 2);
        __builtin_memcpy(&vm, win_bytes + ((((uint32_t)ly & 63u) << 9) + col), 2);
        __builtin_memcpy(&vd, win_bytes + ((((uint32_t)(ly - up_dy) & 63u) << 9) + col), 2);
        const uint32_t up = ((uint32_t)vu >> sh) & 7u, mid = ((uint32_t)vm >> sh) & 7u, dn = ((uint32_t)vd >> sh) & 7u;
        const uint32_t up_rev = (0x73516240u >> (4u * up)) & 7u; // bit order NE, N, NW = columns x+1, x, x-1
        return (mid >> 2) | (up_rev << 1) | ((mid & 1u) << 4) | (dn << 5);
    };
    auto on_rim = [&]() { return (unsigned)(w.x - w.wx0 - 1) > 61u || (unsigned)(w.y - w.wy0 - 1) > 61u; };
    auto dir_off = [&](int d) { return __mul24(dir_dy(d), RS) + dir_dx(d); }; // what a step in direction d adds to the raster key
    auto close_run = [&](int len, int parity) { // one CHAIN_APPROX_SIMPLE segment of `len` steps: its cv.arcLength term
        if (parity) w.diag += len < 64 ? diag_len[len] : run_length(1, len);
        else w.axis += len;
    };
    for (;;) {
        // ---- the pairs whose walk has ended: the forward lane gathers the backward lane's half and stores the record ----
        if (__ballot(finished)) {
            // (both lanes of a pair are finished together; the exchange runs for the whole wave, idle lanes carry zeros)
            const int64_t o_a00 = pair_swap64(w.a00), o_a10 = pair_swap64(w.a10), o_a01 = pair_swap64(w.a01);
            const double o_diag = pair_swap_f64(w.diag + w.pend);
            const int o_axis = pair_swap(w.axis), o_npts = pair_swap(w.npts), o_steps = pair_swap(w.steps), o_run = pair_swap(w.run);
            const int o_head = pair_swap(w.head), o_has_head = pair_swap(w.has_head ? 1 : 0);
            const int o_min_fg = pair_swap(w.min_fg), o_min_ebg = pair_swap(w.min_ebg);
            const int o_bx0 = pair_swap(w.bx0), o_bx1 = pair_swap(w.bx1), o_by0 = pair_swap(w.by0), o_by1 = pair_swap(w.by1);
            const int o_status = pair_swap(w.status);
            if (finished && !isB) {
                const int kind = (int)(w.meta & 3u), image = (int)(w.meta >> 11), border = (int)((w.meta >> 2) & 511u);
                ContourWork& work = works[image];
                const bool single = w.npts == 1 && w.steps == 0 && o_steps == 0 && w.status == 0 && w.s0 < 0; // an isolated pixel
                int status = w.status > o_status ? w.status : o_status;
                if (!single && status == 0) {
                    // join the halves: sums add; the run through the start pixel = the two heads (or, when a half has no vertex at
                    // all, that half's whole run as well), the run through the meeting point = the two tails
                    w.a00 += o_a00; w.a10 += o_a10; w.a01 += o_a01;
                    w.diag += w.pend; w.pend = 0.0; w.diag += o_diag; w.axis += o_axis;
                    w.npts += o_npts; w.steps += o_steps;
                    w.min_fg = o_min_fg < w.min_fg ? o_min_fg : w.min_fg; w.min_ebg = o_min_ebg < w.min_ebg ? o_min_ebg : w.min_ebg;
                    w.bx0 = o_bx0 < w.bx0 ? o_bx0 : w.bx0; w.bx1 = o_bx1 > w.bx1 ? o_bx1 : w.bx1;
                    w.by0 = o_by0 < w.by0 ? o_by0 : w.by0; w.by1 = o_by1 > w.by1 ? o_by1 : w.by1;
                    const int par0 = w.s0 & 1, par_tail = w.known & 1; // direction parity of the run through the start / of the forward tail
                    if (w.has_head && o_has_head) {
                        close_run(w.head + o_head, par0);
                        close_run(w.run + o_run, par_tail);
                    } else
                        close_run((w.has_head ? w.head : 0) + (o_has_head ? o_head : 0) + w.run + o_run, par0);
                }
                if (status == 2) atomicMax(&work.st_err, 1);
                if (kind >= 2) { // a link walk: which border is this?
                    if (status == 0) {
                        const int ltype = w.a00 > 0 ? 1 : 0; // hole borders run the other way round
                        work.link_key[border] = ltype ? w.min_ebg : w.min_fg;
                        work.link_type[border] = (uint8_t)ltype;
                    } else atomicMax(&work.st_err, 1);
                } else if (status == 0) {
                    const int slot = atomicAdd(&work.st_nrec, 1);
                    if (slot < MAXR) {
                        ContourRec r;
                        r.key = w.key; r.is_hole = kind;
                        r.sx = w.sx; r.sy = w.sy;
                        r.npts = w.npts; r.steps = w.steps;
                        r.a00 = w.a00; r.a10 = w.a10; r.a01 = w.a01;
                        r.area = fabs((double)w.a00 * 0.5);
                        r.perimeter = w.npts > 1 ? (double)w.axis + w.diag : 0.0;
                        r.link = -1; r.parent = -1; r.order = -1;
                        select_contour(r, a.min_area, a.min_circ);
                        work.recs[slot] = r;
                        work.rkey[slot] = w.key; work.rsx[slot] = (int16_t)w.sx; work.rsy[slot] = (int16_t)w.sy;
                        work.rhole[slot] = (uint8_t)kind; work.rkept[slot] = (uint8_t)r.kept;
                        work.rbox[slot][0] = (int16_t)w.bx0; work.rbox[slot][1] = (int16_t)w.by0;
                        work.rbox[slot][2] = (int16_t)w.bx1; work.rbox[slot][3] = (int16_t)w.by1;
                    }
                }
            }
            finished = false;
        }
        lap(0);
        // ---- refill: new walks for the idle pairs; new windows for them and for the paused lanes ----
        const uint64_t busy = __ballot(active);
        const uint64_t busy_pairs = (busy | (busy >> 1)) & 0x5555555555555555ull; // bit 2i: pair i holds a walk
        const int n_idle = 32 - __popcll(busy_pairs);
        bool take = false;
        int kind = 0, ex = 0;
        if (!drained && (n_idle >= FOLLOW_REFILL || busy == 0)) {
            // the first 32 entries of a wave are those at 32 * its number (no atomic, no round trip); the counter hands out the rest
            uint32_t base = blockIdx.x * 32u;
            if (!first_fill) {
                if (lane == 0) base = atomicAdd(head_ctr, (uint32_t)n_idle);
                base = (uint32_t)uni((int)base) + gridDim.x * 32u;
            }
            first_fill = false;
            const int n_new = base < total ? (int)(total - base < (uint32_t)n_idle ? total - base : (uint32_t)n_idle) : 0;
            drained = n_new < n_idle;
            if (clk) { tk[5] += (uint64_t)n_new; tk[6]++; }
            const uint64_t idle_pairs = ~busy_pairs & 0x5555555555555555ull;
            const int rank = __popcll(idle_pairs & ((1ull << (lane & ~1)) - 1ull)); // this pair's number among the idle ones
            take = ((idle_pairs >> (lane & ~1)) & 1ull) != 0 && rank < n_new;
            uint64_t e = 0;
            if (take) e = list[base + (uint32_t)rank];
            ex = (int)(e & 0x7fffu); kind = (int)((e >> 30) & 3u);
            if (take) {
                const int ey = (int)((e >> 15) & 0x7fffu);
                const int sx = ex - (kind == 1 ? 1 : 0); // a hole candidate's border pixel lies left of its scan position
                w.meta = (uint32_t)kind | ((uint32_t)((e >> 52) & 511u) << 2) | ((uint32_t)((e >> 32) & 0xfffffu) << 11);
                w.sx = sx; w.sy = ey; w.x = sx; w.y = ey;
                // an outer border starts at its topmost row: its window reaches down from there; a hole border has pixels one row
                // higher; a link walk starts anywhere on its border
                w.wx0 = sx - 31; w.wy0 = kind == 0 ? ey - 1 : (kind == 1 ? ey - 2 : ey - 31);
                w.wslot = lane & ~1; // the pair shares the forward lane's window until one of the two leaves it
            }
        }
        if (paused) { w.wx0 = w.x - 31; w.wy0 = w.y - 31; } // anew around the current pixel ...
        {   // ... in the one of the pair's two windows that the partner does not use (both paused: each takes its own)
            const int partner_slot = pair_swap(w.wslot), partner_paused = pair_swap(paused ? 1 : 0);
            if (paused) w.wslot = partner_paused ? lane : (partner_slot == (lane & ~1) ? (lane | 1) : (lane & ~1));
        }
        const uint64_t stage = __ballot((take && !isB) || paused);
        if (__ballot(take || paused)) {
            // lane = row of the window, eight windows per round so that their 24 loads are in flight together
            const int image = (int)(w.meta >> 11);
            for (uint64_t todo = stage; todo;) {
                int cs[STAGE_N];
                uint64_t rows[STAGE_N];
#pragma unroll
                for (int j = 0; j < STAGE_N; j++) {
                    cs[j] = todo ? __ffsll((long long)todo) - 1 : -1;
                    todo &= todo - 1; // (0 & anything = 0)
                }
#pragma unroll
                for (int j = 0; j < STAGE_N; j++) {
                    const int c = cs[j] < 0 ? cs[0] : cs[j]; // a short last round repeats its first window (not stored)
                    const int img_c = __builtin_amdgcn_readlane(image, c), x0_c = __builtin_amdgcn_readlane(w.wx0, c), y0_c = __builtin_amdgcn_readlane(w.wy0, c);
                    const Mask Mc{a.mask + (size_t)img_c * image_words, a.words_per_row, a.H, a.W, RS};
                    rows[j] = row64(Mc, y0_c + lane, x0_c);
                }
#pragma unroll
                for (int j = 0; j < STAGE_N; j++)
                    if (cs[j] >= 0) win[lane][__builtin_amdgcn_readlane(w.wslot, cs[j])] = rows[j];
            }
            __syncthreads(); // (one wave) the windows are in LDS before any lane reads its own
            if (take) { // both lanes of the pair: the start's first neighbour, clockwise from the one known to be background
                const uint32_t n0 = nbr8(w.x, w.y, -1);
                const int sx = w.sx, ey = w.sy;
                w.key = ey * RS + ex;
                const int first = (kind == 0 || kind == 2) ? 4 : 0; // W (outer start) / E
                w.abort_on_fg = kind == 0;
                w.abort_lt = kind <= 1 ? w.key : -1;  // link walks run all the way round
                w.a00 = w.a10 = w.a01 = 0; w.npts = 0; w.steps = 0; w.axis = 0; w.diag = 0.0; w.pend = 0.0;
                w.run = 0; w.head = 0; w.has_head = false;
                w.min_fg = 0x7fffffff; w.min_ebg = 0x7fffffff;
                w.bx0 = 0x7fffffff; w.bx1 = -1; w.by0 = 0x7fffffff; w.by1 = -1;
                int s = first;
                do {
                    s = (s - 1) & 7;
                } while (!((n0 >> s) & 1u) && s != first);
                w.status = 0;
                if (s == first) { // isolated pixel: one vertex, zero area, zero perimeter
                    w.s0 = -1;
                    if (!isB) { w.npts = 1; w.min_fg = ey * RS + sx; w.min_ebg = ey * RS + sx + 1; w.bx0 = w.bx1 = sx; w.by0 = w.by1 = ey; }
                    finished = true;
                } else {
                    w.s0 = s;
                    w.r = ey * RS + sx; w.r_ahead = w.r;
                    if (isB) { w.x = sx + dir_dx(s); w.y = ey + dir_dy(s); w.r += dir_off(s); w.known = s ^ 4; } // one step back along the border: the way forward from there
                    else w.known = s;
                    active = true;
                    paused = on_rim();
                }
            } else if (paused) paused = false; // (its window now lies around its pixel)
            if (active && !paused) w.n = nbr8(w.x, w.y, up_dy);
        }
        lap(1);
        if (__ballot(active || finished) == 0) break; // nothing in flight (and nothing left in the list, or the refill would have run)
        // ---- FOLLOW_K steps of every running lane ----
        for (int k = 0; k < FOLLOW_K; k++) {
            const bool go = active && !paused;
            // the step, tentatively: search the next border pixel (counter-clockwise from known + 1; the backward lane does the same on
            // its mirrored neighbourhood, i.e. clockwise from known - 1)
            const int kn = isB ? (8 - w.known) & 7 : w.known;
            const uint32_t rot = ((w.n | (w.n << 8)) >> (kn + 1)) & 0xffu;
            const int su = (kn + __ffs((int)rot)) & 7;
            const int srch = isB ? (8 - su) & 7 : su;
            const int nx = w.x + dir_dx(srch), ny = w.y + dir_dy(srch);
            // where the two lanes stand on the border's cycle of (pixel, way back) states: the forward lane at its own state, the
            // backward lane just behind the state (pixel ahead of it, way back to it) -- pixels as raster keys
            const int r_next = w.r + dir_off(srch);
            const int st_r = isB ? w.r_ahead : w.r, st_d = isB ? w.known ^ 4 : w.known;
            const int nw_r = isB ? w.r : r_next, nw_d = isB ? srch : srch ^ 4; // ... and after this step
            const int fl = (go ? 1 : 0) | (active ? 2 : 0) | (w.steps > 0 ? 4 : 0) | (w.status << 3);
            const int o_st_r = pair_swap(st_r), o_nw_r = pair_swap(nw_r), o_misc = pair_swap(st_d | (nw_d << 3) | (fl << 6));
            const int o_st_d = o_misc & 7, o_nw_d = (o_misc >> 3) & 7, o_fl = o_misc >> 6;
            const bool o_go = (o_fl & 1) != 0;
            const bool met = active && (o_fl & 2) && st_r == o_st_r && st_d == o_st_d && ((fl | o_fl) & 4);
            // the forward lane's step completes the cycle: the backward lane must not take the same step from the other side
            const bool fwd_closes = (isB ? o_nw_r : nw_r) == (isB ? st_r : o_st_r) && (isB ? o_nw_d : nw_d) == (isB ? st_d : o_st_d) && (o_go || !isB);
            const bool partner_gave_up = active && (o_fl >> 3) != 0;
            if (met || partner_gave_up || (isB && active && fwd_closes)) {
                // met: the halves cover the whole border; the last case: the forward lane's step of this round completes the cycle, the
                // backward lane must not take the same step from the other side
                active = false; finished = true; paused = false;
            } else if (go) {
                // this lane accounts for one forward step of the border: from its pixel, in direction s, having arrived from s_end
                const int s = isB ? w.known : srch, s_end = isB ? srch : w.known;
                const int r = w.r;
                const bool east_bg = (unsigned)(s - 1) < (unsigned)s_end; // the East neighbour was examined and is background
                const int re = east_bg ? r + 1 : 0x7fffffff;
                w.min_ebg = re < w.min_ebg ? re : w.min_ebg;
                w.min_fg = r < w.min_fg ? r : w.min_fg;
                w.bx0 = w.x < w.bx0 ? w.x : w.bx0; w.bx1 = w.x > w.bx1 ? w.x : w.bx1;
                w.by0 = w.y < w.by0 ? w.y : w.by0; w.by1 = w.y > w.by1 ? w.y : w.by1;
                // (x,y) is a CHAIN_APPROX_SIMPLE vertex when the direction changes there.  Forward lane: the run that ENDS here closes,
                // then this step opens / extends the next one; backward lane: this step extends the run that STARTS here, then it closes.
                const bool vertex = s != (s_end ^ 4);
                w.run += isB ? 1 : 0;
                const int len = vertex ? w.run : 0;
                const bool first_vertex = vertex && !w.has_head; // the lane's first run is joined with the partner's at the end
                w.head = first_vertex ? len : w.head;
                w.has_head = w.has_head || vertex;
                const int kk = first_vertex ? 0 : len;
                const bool odd = ((isB ? s : s_end) & 1) != 0;
                w.axis += odd ? 0 : kk;
                w.diag += w.pend;
                const int kd = odd ? kk : 0;                                  // diag_len[0] = 0
                w.pend = diag_len[kd < 63 ? kd : 63];
                if (kd > 63) w.pend = run_length(1, kd);                      // (a diagonal run longer than the table: rare)
                w.npts += vertex ? 1 : 0;
                w.run = (vertex ? 0 : w.run) + (isB ? 0 : 1);
                const int dx = dir_dx(s), dy = dir_dy(s);
                const int cross = __mul24(w.x, dy) - __mul24(dx, w.y); // x*ny - nx*y (coordinates below 2^15)
                w.a00 += cross;
                w.a10 += (int64_t)cross * (2 * w.x + dx);
                w.a01 += (int64_t)cross * (2 * w.y + dy);
                w.steps++;
                const bool aborted = (w.abort_on_fg ? r : re) < w.abort_lt, limit = w.steps > a.max_steps;
                // the walk ends here when it has left its border's claim (aborted), ran too long, or -- forward lane -- this step completes
                // the cycle (the backward lane sees the same condition and stops as well; known = the way back from the meeting pixel: the
                // direction of the forward tail)
                const bool stop = aborted || limit || (!isB && fwd_closes);
                w.status = aborted ? 1 : (limit ? 2 : w.status);
                // move (a lane that stops moves too: nothing reads its position afterwards).  The three rows around the new pixel and its
                // left / right neighbour columns must lie inside the window: a lane that steps onto the window's rim pauses until its
                // window is staged anew (no global load in this loop).
                w.x = nx; w.y = ny; w.known = srch ^ 4;
                w.r_ahead = w.r; w.r = r_next;
                paused = !stop && on_rim();
                w.n = nbr8(nx, ny, up_dy); // (garbage while paused: read again from the new window)
                active = !stop; finished = stop;
            }
            if (clk) { tk[3]++; tk[4] += (uint64_t)__popcll(__ballot(active && !paused)); }
            if (__ballot(active && !paused) == 0) break;
        }
        lap(2);
    }
    if (clk && lane == 0)
        for (int i = 0; i < 8; i++) a.follow_dbg[(size_t)blockIdx.x * 8 + i] = tk[i];
}

size_t contour_work_bytes() { return sizeof(ContourWork); }
size_t contour_walk_bytes() { return sizeof(uint64_t) * MAXC; }  // candidates of one image in the batch's walk list
size_t contour_link_bytes() { return sizeof(uint64_t) * MAXA; }  // links of one image handed to the second follow pass

void launch_contours(const ContourArgs& a_, hipStream_t s)
{
    if (a_.walk_list && !a_.timing && a_.n_images < MAX_SPLIT_IMAGES) {
        // candidates per image -> every walk of the batch -> tree per image; the (few) links whose owner only a walk can tell go
        // through a second, equally packed, follow pass, and the second tree pass finishes the images that waited for them
        ContourArgs a = a_;
        if (!a.counters_zeroed) (void)hipMemsetAsync(a.walk_count, 0, 8 * sizeof(uint32_t), s);
        const bool loop = a.image_grid > 0 && a.image_grid < a.n_images;
        const int pass2_grid = a.n_images < 64 ? a.n_images : 64;
        if (loop) hipLaunchKernelGGL((contours_kernel<1, true>), dim3(a.image_grid), dim3(NTHREADS), 0, s, a);
        else hipLaunchKernelGGL((contours_kernel<1, false>), dim3(a.n_images), dim3(NTHREADS), 0, s, a);
        a.follow_list = 0;
        hipLaunchKernelGGL(contour_follow_kernel, dim3(a.follow_grid), dim3(64), 0, s, a);
        a.tree_pass = 1;
        if (loop) hipLaunchKernelGGL((contours_kernel<2, true>), dim3(a.image_grid), dim3(NTHREADS), 0, s, a);
        else hipLaunchKernelGGL((contours_kernel<2, false>), dim3(a.n_images), dim3(NTHREADS), 0, s, a);
        if (!a.defer_links) return; // every link was settled in the first tree pass (walked in place where the boxes did not decide)
        a.follow_list = 1;
        hipLaunchKernelGGL(contour_follow_kernel, dim3(a.follow_grid2), dim3(64), 0, s, a);
        a.tree_pass = 2;
        hipLaunchKernelGGL((contours_kernel<2, false>), dim3(pass2_grid), dim3(NTHREADS), 0, s, a);
        return;
    }
    hipLaunchKernelGGL((contours_kernel<0, false>), dim3(a_.n_images), dim3(NTHREADS), 0, s, a_);
}

} // namespace mocap
