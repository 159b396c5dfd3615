// geom.hip -- epipolar correspondence, batched N-view DLT triangulation and reprojection error (FP64).
//
// Replaces, per time step, reference lib/Helpers.py:178-280 (find_point_correspondance_and_object_points)
// and its callees triangulate_point(s) (:43-99) and calculate_reprojection_error(s) (:102-143), including
// the two OpenCV primitives they use (cv.computeCorrespondEpilines :207, cv.projectPoints :133).
// The library is built with -ffp-contract=off: every product and sum below is a separately rounded FP64
// (or, where the reference rounds to float32, FP32) operation, as in the NumPy/OpenCV path.
//
// Work decomposition: one workgroup per time step.  (root, camera) pairs are scored one per lane; the
// candidate groups of all roots of the time step (the reference's cartesian expansion) are flattened into one
// index space and triangulated one group per lane; per-root error means use NumPy's pairwise summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "kernels.h"

namespace mocap {

namespace {

constexpr int MAXM = 16; // candidate matches kept per (root, camera)

__device__ __forceinline__ double np_block_sum(const double* a, int n)
{ // numpy pairwise_sum for n <= 128
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// numpy's pairwise_sum for any n, without recursion and without a stack: the leaves (runs of <= 128 values) are visited
// in order -- the leaf holding offset o is found by walking down from the root, which also yields the path (bit d set =
// right child at depth d) --, and a finished subtree's sum is parked IN the array, at the subtree's first offset (those
// values are consumed by then).  A node's left child starts at the node's own offset, so "left + right" of a node whose
// right child has just been finished reads a[offset of the node].  The array is overwritten.  Same additions in the same
// order as numpy's recursion: n2 = (n / 2) rounded down to a multiple of 8, left = [0, n2), right = [n2, n).
__device__ double np_pairwise_sum(double* a, int n)
{
    if (n <= 128) return np_block_sum(a, n);
    int o = 0;
    double ret = 0.;
    while (o < n) {
        int off = 0, len = n, depth = 0;
        uint32_t path = 0;
        while (len > 128) {
            int n2 = len / 2; n2 -= n2 % 8;
            if (o < off + n2) len = n2;
            else { off += n2; len -= n2; path |= 1u << depth; }
            depth++;
        }
        ret = np_block_sum(a + off, len);
        o = off + len;
        // every ancestor this leaf completes as the last leaf of its right subtree: deepest first
        while (depth > 0 && ((path >> (depth - 1)) & 1u)) {
            depth--;
            int poff = 0, plen = n; // the ancestor at `depth`: walk down the same path
            for (int d = 0; d < depth; d++) {
                int n2 = plen / 2; n2 -= n2 % 8;
                if ((path >> d) & 1u) { poff += n2; plen -= n2; } else plen = n2;
            }
            ret = a[poff] + ret;
        }
        if (depth > 0) { // a left child is complete: park its sum where its parent will look for it
            int poff = 0, plen = n;
            for (int d = 0; d < depth - 1; d++) {
                int n2 = plen / 2; n2 -= n2 % 8;
                if ((path >> d) & 1u) { poff += n2; plen -= n2; } else plen = n2;
            }
            a[poff] = ret;
        }
    }
    return ret;
}

// numpy's pairwise_sum of n <= 128 values that arrive two at a time, in order, without keeping them (np_block_sum above on
// the fly; n even).  The values are squares (>= +0), so starting the eight partial sums at +0 instead of a[0..7] adds nothing.
struct NpPairStream {
    double r0, r1, r2, r3, r4, r5, r6, r7, res;
    int nfull, i;
    __device__ __forceinline__ void begin(int n)
    {
        nfull = n < 8 ? 0 : n - (n % 8);
        i = 0;
        r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = 0.; res = 0.;
    }
    __device__ __forceinline__ void push2(double a, double b)
    {
        if (i < nfull) {
            // value i goes to partial sum i % 8.  The eight sums are rotated by two after every pair instead of being selected by
            // index (a register cannot be indexed; the compiler would move them to scratch memory): the next pair again lands in
            // r0 / r1, and after nfull values -- a multiple of 8 -- every sum is back in its own place
            r0 += a; r1 += b;
            const double t0 = r0, t1 = r1;
            r0 = r2; r1 = r3; r2 = r4; r3 = r5; r4 = r6; r5 = r7; r6 = t0; r7 = t1;
            i += 2;
            if (i == nfull) res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        } else {
            res += a; res += b;
            i += 2;
        }
    }
};

// cv.computeCorrespondEpilines for one float32 point
__device__ __forceinline__ void epiline(const double* F, float xf, float yf, float line[3])
{
    double x = xf, y = yf;
    double a = F[0] * x + F[1] * y + F[2];
    double b = F[3] * x + F[4] * y + F[5];
    double c = F[6] * x + F[7] * y + F[8];
    double nu = a * a + b * b;
    nu = nu ? 1. / sqrt(nu) : 1.;
    a *= nu; b *= nu; c *= nu;
    line[0] = (float)a; line[1] = (float)b; line[2] = (float)c;
}

__device__ __forceinline__ double epi_distance(const float line[3], double x, double y)
{ // reference lib/Helpers.py:217
    double a = line[0], b = line[1], c = line[2];
    return fabs(a * x + b * y + c) / sqrt(a * a + b * b);
}

// eigenvector of the smallest eigenvalue of a symmetric 4x4 (cyclic Jacobi); same rotations as the oracle
__device__ __forceinline__ void smallest_eigvec4(double B[4][4], double v[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            diag += B[p][p] * B[p][p];
#pragma unroll
            for (int q = p + 1; q < 4; q++) off += B[p][q] * B[p][q];
        }
        if (off == 0.0 || off <= 1e-40 * diag) break;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double apq = B[p][q];
                if (apq == 0.0) continue;
                double theta = (B[q][q] - B[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    double bkp = B[k][p], bkq = B[k][q];
                    B[k][p] = c * bkp - s * bkq;
                    B[k][q] = s * bkp + c * bkq;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    double bpk = B[p][k], bqk = B[q][k];
                    B[p][k] = c * bpk - s * bqk;
                    B[q][k] = s * bpk + c * bqk;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    double best = B[0][0];
#pragma unroll
    for (int k = 1; k < 4; k++)
        if (B[k][k] < best) { best = B[k][k]; m = k; }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = m == 0 ? V[k][0] : (m == 1 ? V[k][1] : (m == 2 ? V[k][2] : V[k][3]));
}

struct DltAcc {
    double B[4][4];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) B[j][k] = 0.;
    }
    // rows  y*P[2]-P[1]  and  P[0]-x*P[2]  of the DLT system, P = K @ [R|t]  (reference lib/Helpers.py:58-73)
    __device__ __forceinline__ void add(const double* K, const double* R, const double* t, double x, double y)
    {
        double P[12];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) s += K[3 * r + k] * (c < 3 ? R[3 * k + c] : t[k]);
                P[4 * r + c] = s;
            }
        double r0[4], r1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r0[k] = y * P[8 + k] - P[4 + k];
            r1[k] = P[k] - x * P[8 + k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) B[j][k] += r0[j] * r0[k] + r1[j] * r1[k];
    }
    // the same two rows from a projection matrix formed beforehand (the identical sums, formed once per camera)
    __device__ __forceinline__ void add_rows(const double* P, double x, double y)
    {
        double r0[4], r1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r0[k] = y * P[8 + k] - P[4 + k];
            r1[k] = P[k] - x * P[8 + k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) B[j][k] += r0[j] * r0[k] + r1[j] * r1[k];
    }
    __device__ __forceinline__ void solve(double X[3])
    {
        double v[4];
        smallest_eigvec4(B, v);
        X[0] = v[0] / v[3]; X[1] = v[1] / v[3]; X[2] = v[2] / v[3];
    }
};

// cv.projectPoints for one float32 object point; squared pixel errors against (px,py)
__device__ __forceinline__ void reproj_sq(const double* K, const double* d, const double* R, const double* t, const float Xf[3],
                          double px, double py, double& ex, double& ey)
{
    double X = Xf[0], Y = Xf[1], Z = Xf[2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6;
    double xd = x * cdist + d[2] * a1 + d[3] * a2;
    double yd = y * cdist + d[2] * a3 + d[3] * a1;
    float u = (float)(xd * K[0] + K[2]), v = (float)(yd * K[4] + K[5]);
    double dx = px - (double)u, dy = py - (double)v;
    ex = dx * dx; ey = dy * dy;
}

template <typename PT>
__device__ __forceinline__ void load_pt(const void* base, size_t idx, double& x, double& y)
{
    const PT* p = (const PT*)base + 2 * idx;
    x = (double)p[0]; y = (double)p[1];
}

} // namespace

// LDS plan of correspond_kernel (dynamic shared memory), the same on host and device.  What the kernel cannot do without
// comes first (camera block, counts, candidate lists); the staged points and the per-group errors take what `budget` leaves
// (both have a fallback: points from global memory, errors in the context's scratch array).
struct CorrLds {
    int cam, F, pts, errs, rerr, ints, nm, midx, total; // byte offsets
    int stage_pts, err_cap;
};
__host__ __device__ inline CorrLds corr_lds_plan(int P, int C, int budget)
{
    CorrLds L;
    int o = 0;
    L.cam = o; o += C * 38 * 8;                 // per camera: projection matrix P = K [R|t] (12), K (9), dist (5), R (9), t (3)
    L.F = o; o += (C > 1 ? C - 1 : 0) * 9 * 8;
    L.rerr = o; o += P * 8;
    L.ints = o; o += (3 * P + 2 + 32) * 4;      // G[P], goff[P + 1], slot[P], counts[32]
    L.nm = o; o += P * C;
    L.midx = o; o += P * C * MAXM;
    o = (o + 15) & ~15;
    L.stage_pts = C * P <= 2048 && o + C * P * 16 + 128 * 8 <= budget; // all image points of the time step in LDS (32 KB at most)
    L.pts = o; o += L.stage_pts ? C * P * 16 : 0;
    int cap = (budget - o) / 8;                 // per-group errors in LDS when the time step has no more groups than this
    L.err_cap = cap > 1024 ? 1024 : (cap < 0 ? 0 : cap);
    L.errs = o; o += L.err_cap * 8;
    L.total = (o + 15) & ~15;
    return L;
}

// One workgroup per time step.  Everything the step needs -- the cameras' image points, the camera table (with the
// projection matrices K [R|t] formed once instead of once per group and camera), the fundamental matrices -- comes in with
// ONE round of global loads into LDS; all scoring, triangulation and ranking then runs from LDS and registers, and the
// results leave with one round of stores.  (The first version loaded points and matrices where it used them: ~100 dependent
// global round trips per time step, 47 us alone but 552 us beside the other batches' streaming scans, whose queued loads every
// one of those round trips waits behind.)
template <typename PT>
__global__ __launch_bounds__(256) void correspond_kernel(CorrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int C = a.C, P = a.P, t = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const CorrLds L = corr_lds_plan(P, C, a.lds_budget);
    double (*cam)[38] = (double (*)[38])(smem + L.cam);
    double (*Fm)[9] = (double (*)[9])(smem + L.F);
    double* spts = (double*)(smem + L.pts);
    double* serr = (double*)(smem + L.errs);
    double* rerr = (double*)(smem + L.rerr);
    int* G = (int*)(smem + L.ints);
    int* goff = G + P;
    int* slot = goff + P + 1;
    int* scnt = slot + P;
    uint8_t* nm = smem + L.nm;
    uint8_t* midx = smem + L.midx;
    __shared__ int s_err, s_nout;

    const CameraTable* cams = a.cams;
    // element offset (in points) of camera c's list; strides are given in scalars, a point is 2 scalars
    auto pbase = [&](int c) { return ((size_t)t * a.pt_st + (size_t)c * a.pt_sc) / 2; };
    // ---- the one round of loads: counts, points (every slot up to P: the address does not depend on the count), cameras ----
    if (tid == 0) { s_err = 0; s_nout = 0; }
    if (tid < C) scnt[tid] = a.counts[(size_t)t * a.cnt_st + (size_t)tid * a.cnt_sc];
    if (L.stage_pts)
        for (int w = tid; w < C * P; w += nth) {
            const int c = w / P, p = w - c * P;
            double x, y;
            load_pt<PT>(a.pts, pbase(c) + p, x, y);
            spts[2 * w] = x; spts[2 * w + 1] = y;
        }
    for (int w = tid; w < C * 26; w += nth) {
        const int c = w / 26, k = w - c * 26;
        cam[c][12 + k] = k < 9 ? cams->K[c][k] : (k < 14 ? cams->dist[c][k - 9] : (k < 23 ? cams->R[c][k - 14] : cams->t[c][k - 23]));
    }
    for (int w = tid; w < (C - 1) * 9; w += nth) Fm[w / 9][w % 9] = cams->F[w / 9][w % 9];
    __syncthreads();
    for (int w = tid; w < C * 12; w += nth) { // P = K @ [R|t], the sums in the order of reference lib/Helpers.py:58-62 (as DltAcc::add forms them)
        const int c = w / 12, r = (w % 12) / 4, cc = w % 4;
        const double* K = &cam[c][12]; const double* R = &cam[c][26]; const double* tt = &cam[c][35];
        double sum = 0;
        for (int k = 0; k < 3; k++) sum += K[3 * r + k] * (cc < 3 ? R[3 * k + cc] : tt[k]);
        cam[c][4 * r + cc] = sum;
    }
    // A camera whose count does not fit the P points read here, or whose blob stage reported a capacity error (negative
    // count), fails the time step: the reference has no such limits (lib/Helpers.py:191,203-245), so a shortened list
    // must never pass for a result.  s_err: 1 = groups, 2 = more than P points, 3 = negative count.
    if (tid < C) {
        const int n = scnt[tid];
        if (n < 0) atomicMax(&s_err, 3);
        else if (n > P) atomicMax(&s_err, 2);
    }
    __syncthreads();
    if (s_err) {
        if (tid == 0) a.n_roots[t] = s_err == 3 ? CORR_ERR_BLOB : CORR_ERR_TRUNCATED;
        return;
    }
    auto getp = [&](int c, int p, double& x, double& y) {
        if (L.stage_pts) { x = spts[2 * (c * P + p)]; y = spts[2 * (c * P + p) + 1]; }
        else load_pt<PT>(a.pts, pbase(c) + p, x, y);
    };
    const int n0 = scnt[0];

    // ---- phase 1: per (root, camera) candidate lists, sorted by distance to the epipolar line -----------
    for (int w = tid; w < n0 * (C - 1); w += nth) {
        int j = w / (C - 1), i = 1 + w % (C - 1);
        double rx, ry;
        getp(0, j, rx, ry);
        float line[3];
        epiline(Fm[i - 1], (float)rx, (float)ry, line);
        // the candidates of (root j, camera i), kept sorted by distance (stable) in their LDS row.  No per-lane arrays: the
        // distance of an entry that has to be compared again is computed again from its point (a handful of operations, the
        // same value), so nothing here lives in scratch memory
        uint8_t* const mrow = midx + ((size_t)j * C + i) * MAXM;
        int k = 0;
        const int ni = scnt[i];
        bool over = false;
        for (int p = 0; p < ni; p++) {
            double x, y;
            getp(i, p, x, y);
            const double d = epi_distance(line, x, y);
            if (d < a.cutoff) {
                if (k == MAXM) { over = true; break; }
                int q = k++; // stable insertion by distance
                while (q > 0) {
                    const int pq = mrow[q - 1];
                    double xq, yq;
                    getp(i, pq, xq, yq);
                    if (!(epi_distance(line, xq, yq) > d)) break;
                    mrow[q] = (uint8_t)pq;
                    q--;
                }
                mrow[q] = (uint8_t)p;
            }
        }
        if (over) atomicMax(&s_err, 1);
        nm[j * C + i] = (uint8_t)k;
    }
    __syncthreads();
    // ---- group counts and offsets ---------------------------------------------------------------------------
    for (int j = tid; j < n0; j += nth) {
        long g = C >= 2 ? 1 : 0;
        for (int i = 1; i < C; i++) {
            g *= nm[j * C + i];
            if (g > a.max_groups) { atomicMax(&s_err, 1); g = 0; break; }
        }
        G[j] = (int)g;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, no = 0;
        for (int j = 0; j < n0; j++) {
            goff[j] = acc; acc += G[j];
            slot[j] = G[j] > 0 ? no++ : -1;
        }
        goff[n0] = acc;
        s_nout = no;
        if (acc > a.step_budget) s_err = 1; // the time step's groups do not fit its share of the error scratch
    }
    __syncthreads();
    if (s_err) {
        if (tid == 0) a.n_roots[t] = CORR_ERR_GROUPS;
        return;
    }
    // ---- phase 2: one candidate group per lane: DLT + reprojection error --------------------------------
    const int total = goff[n0];
    const bool err_in_lds = total <= L.err_cap;
    double* const errs = err_in_lds ? serr : a.scratch + (size_t)t * a.step_budget;
    for (int w = tid; w < total; w += nth) {
        int lo = 0, hi = n0 - 1; // root owning flat group index w
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (goff[mid] <= w) lo = mid; else hi = mid - 1;
        }
        while (G[lo] == 0) lo++; // skip dead roots sharing the offset
        const int j = lo, g = w - goff[j];
        // pass 1 over the cameras: the group's points (camera 1 is the fastest-varying digit, reference lib/Helpers.py:239-245)
        // into the DLT system.  The points are looked up again in pass 2 instead of being kept in per-lane arrays.
        double x0, y0;
        getp(0, j, x0, y0);
        DltAcc acc;
        acc.clear();
        acc.add_rows(cam[0], x0, y0);
        const uint8_t* const nmj = nm + j * C;
        const uint8_t* const mj = midx + (size_t)j * C * MAXM;
        int rem = g;
        for (int i = 1; i < C; i++) {
            const int n_i = nmj[i], dgt = rem % n_i;
            rem /= n_i;
            double x, y;
            getp(i, mj[i * MAXM + dgt], x, y);
            acc.add_rows(cam[i], x, y);
        }
        double X[3];
        acc.solve(X);
        const float Xf[3] = {(float)X[0], (float)X[1], (float)X[2]};
        // pass 2: squared reprojection errors, summed in numpy's order as they come
        const bool first = g == 0;
        const size_t ro = (size_t)t * P + (first ? slot[j] : 0);
        NpPairStream sum;
        sum.begin(2 * C);
        rem = g;
        for (int i = 0; i < C; i++) {
            double x = x0, y = y0;
            if (i > 0) {
                const int n_i = nmj[i], dgt = rem % n_i;
                rem /= n_i;
                getp(i, mj[i * MAXM + dgt], x, y);
            }
            double ex, ey;
            reproj_sq(&cam[i][12], &cam[i][21], &cam[i][26], &cam[i][35], Xf, x, y, ex, ey);
            sum.push2(ex, ey);
            if (first) { a.root_grp[(ro * C + i) * 2] = x; a.root_grp[(ro * C + i) * 2 + 1] = y; }
        }
        errs[goff[j] + g] = sum.res / (double)(2 * C);
        if (first) {
            a.root_xyz[ro * 3] = X[0]; a.root_xyz[ro * 3 + 1] = X[1]; a.root_xyz[ro * 3 + 2] = X[2];
            a.root_idx[ro] = j;
        }
    }
    __syncthreads(); // the errors are read back by other lanes of this workgroup
    __threadfence_block();
    // ---- phase 3: per-root mean over its groups (NumPy pairwise order), then argsort ---------------------
    for (int j = tid; j < n0; j += nth) {
        if (G[j] == 0) continue;
        const double m = np_pairwise_sum(errs + goff[j], G[j]) / (double)G[j];
        rerr[slot[j]] = m;
        a.root_err[(size_t)t * P + slot[j]] = m;
    }
    __syncthreads();
    const int nout = s_nout;
    for (int o = tid; o < nout; o += nth) {
        double eo = rerr[o];
        int rank = 0;
        for (int q = 0; q < nout; q++) {
            double eq = rerr[q];
            rank += (eq < eo) || (eq == eo && q < o);
        }
        a.order[(size_t)t * P + rank] = o;
    }
    if (tid == 0) a.n_roots[t] = nout;
}

// triangulate_point over a batch (reference lib/Helpers.py:43-84): invalid entries dropped, fewer than two
// left -> not triangulated; with compact_k the intrinsics are taken by position after the drop (:59-61)
__global__ void triangulate_kernel(TriArgs a)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const CameraTable* cams = a.cams;
    DltAcc acc;
    acc.clear();
    int m = 0;
    for (int c = 0; c < a.C; c++) {
        if (!a.valid[(size_t)n * a.C + c]) continue;
        int ki = a.compact_k ? m : c;
        acc.add(cams->K[ki], cams->R[c], cams->t[c], a.pts[((size_t)n * a.C + c) * 2], a.pts[((size_t)n * a.C + c) * 2 + 1]);
        m++;
    }
    if (m <= 1) { a.ok[n] = 0; return; }
    double X[3];
    acc.solve(X);
    a.xyz[3 * n] = X[0]; a.xyz[3 * n + 1] = X[1]; a.xyz[3 * n + 2] = X[2];
    a.ok[n] = 1;
}

// calculate_reprojection_error over a batch (reference lib/Helpers.py:113-143)
__global__ void reproject_kernel(ReprojArgs a)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const CameraTable* cams = a.cams;
    float Xf[3] = {(float)a.xyz[3 * n], (float)a.xyz[3 * n + 1], (float)a.xyz[3 * n + 2]};
    int nv = 0;
    for (int c = 0; c < a.C; c++) nv += a.valid[(size_t)n * a.C + c] != 0;
    if (nv <= 1) { a.ok[n] = 0; return; }
    NpPairStream sum; // the errors of the valid cameras, summed in numpy's order as they come (no per-lane array)
    sum.begin(2 * nv);
    int m = 0;
    for (int c = 0; c < a.C; c++) {
        if (!a.valid[(size_t)n * a.C + c]) continue;
        int ki = a.compact_k ? m : c;
        double ex, ey;
        reproj_sq(cams->K[ki], cams->dist[ki], cams->R[c], cams->t[c], Xf, a.pts[((size_t)n * a.C + c) * 2],
                  a.pts[((size_t)n * a.C + c) * 2 + 1], ex, ey);
        sum.push2(ex, ey);
        m++;
    }
    a.mse[n] = sum.res / (double)(2 * nv);
    a.ok[n] = 1;
}

// The distance matrix of reference lib/Helpers.py:205-220 for one camera pair: line of root r = cv.computeCorrespondEpilines
// of the float32 root point under F (:207), distance of candidate p to it by the expression of :217.  One lane per (root,
// candidate) pair; the same `epiline` / `epi_distance` the correspondence kernel scores with.
template <typename PT>
__global__ __launch_bounds__(256) void epipolar_scores_kernel(EpiArgs a)
{
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (long)a.n_roots * a.n_cand) return;
    const int r = (int)(w / a.n_cand), p = (int)(w - (long)r * a.n_cand);
    double rx, ry, x, y;
    load_pt<PT>(a.roots, (size_t)r, rx, ry);
    load_pt<PT>(a.cand, (size_t)p, x, y);
    float line[3];
    epiline(a.cams->F[a.f_index], (float)rx, (float)ry, line);
    a.dist[w] = epi_distance(line, x, y);
    if (a.lines && p == 0) { a.lines[3 * r] = line[0]; a.lines[3 * r + 1] = line[1]; a.lines[3 * r + 2] = line[2]; }
}

// scipy Rotation.from_rotvec(v).as_matrix() (reference lib/Helpers.py:152): rotation vector -> unit quaternion -> matrix
__device__ __forceinline__ void rotvec_to_matrix(const double v[3], double Rm[9])
{
    const double angle = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double scale;
    if (angle <= 1e-3) {
        const double a2 = angle * angle;
        scale = 0.5 - a2 / 48 + a2 * a2 / 3840;
    } else
        scale = sin(angle / 2) / angle;
    const double x = scale * v[0], y = scale * v[1], z = scale * v[2], w = cos(angle / 2);
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    Rm[0] = x2 - y2 - z2 + w2; Rm[1] = 2 * (xy - zw);        Rm[2] = 2 * (xz + yw);
    Rm[3] = 2 * (xy + zw);        Rm[4] = -x2 + y2 - z2 + w2; Rm[5] = 2 * (yz - xw);
    Rm[6] = 2 * (xz - yw);        Rm[7] = 2 * (yz + xw);        Rm[8] = -x2 - y2 + z2 + w2;
}

// bundle_adjustment.residual_function (reference lib/Helpers.py:161-167) for B parameter vectors, one workgroup each:
// params_to_camera_poses (:145-156: camera 0 = identity / zero, cameras 1.. from rotvec + t sextuples) into LDS, then
// triangulate_points (groups holding a None are skipped, :93), then calculate_reprojection_errors, which pairs groups and
// object points POSITIONALLY (:104) and drops pairs seen by fewer than two cameras (:127-128), then the float32 cast (:165).
// Both compactions are block-wide prefix counts, so the residual vector has the reference's length and order.
__global__ __launch_bounds__(256) void ba_residuals_kernel(BaArgs a)
{
    __shared__ double sR[32][9], sT[32][3];
    __shared__ int s_wave[4], s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, N = a.N;
    const CameraTable* cams = a.cams;
    if (tid < C) {
        if (tid == 0) {
            for (int k = 0; k < 9; k++) sR[0][k] = (k % 4 == 0) ? 1.0 : 0.0;
            sT[0][0] = sT[0][1] = sT[0][2] = 0.0;
        } else {
            const double* p = a.params + (size_t)b * 6 * (C - 1) + 6 * (tid - 1);
            const double v[3] = {p[0], p[1], p[2]};
            double Rm[9];
            rotvec_to_matrix(v, Rm);
            for (int k = 0; k < 9; k++) sR[tid][k] = Rm[k];
            sT[tid][0] = p[3]; sT[tid][1] = p[4]; sT[tid][2] = p[5];
        }
    }
    if (tid == 0) s_base = 0;
    __syncthreads();
    double* const obj = a.obj + (size_t)b * N * 3;
    // block-wide exclusive prefix of a flag over the 256 threads of one round; every thread calls it
    auto place = [&](bool flag, int& total) -> int {
        const uint64_t bal = __ballot(flag);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        return off + __popcll(bal & ((1ull << lane) - 1ull));
    };
    for (int n0 = 0; n0 < N; n0 += 256) { // triangulate_points: the groups without a None, in order
        const int n = n0 + tid;
        bool full = n < N;
        if (full)
            for (int c = 0; c < C; c++) full = full && a.valid[(size_t)n * C + c] != 0;
        int total;
        const int o = place(full, total);
        if (full) {
            DltAcc acc;
            acc.clear();
            for (int c = 0; c < C; c++)
                acc.add(cams->K[c], sR[c], sT[c], a.pts[((size_t)n * C + c) * 2], a.pts[((size_t)n * C + c) * 2 + 1]);
            double X[3];
            acc.solve(X);
            obj[3 * o] = X[0]; obj[3 * o + 1] = X[1]; obj[3 * o + 2] = X[2];
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    const int nobj = s_base;
    __threadfence_block(); // the object points are read back by other threads of this workgroup
    __syncthreads();
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int n0 = 0; n0 < nobj; n0 += 256) { // zip(groups, object points): group n with object point n
        const int n = n0 + tid;
        bool has = n < nobj;
        double mse = 0.0;
        if (has) {
            const float Xf[3] = {(float)obj[3 * n], (float)obj[3 * n + 1], (float)obj[3 * n + 2]};
            int nv = 0;
            for (int c = 0; c < C; c++) nv += a.valid[(size_t)n * C + c] != 0;
            has = nv > 1;
            if (has) {
                NpPairStream sum;
                sum.begin(2 * nv);
                int m = 0;
                for (int c = 0; c < C; c++) {
                    if (!a.valid[(size_t)n * C + c]) continue;
                    double ex, ey;
                    reproj_sq(cams->K[m], cams->dist[m], sR[c], sT[c], Xf, a.pts[((size_t)n * C + c) * 2], a.pts[((size_t)n * C + c) * 2 + 1],
                              ex, ey); // intrinsics by position after the None entries are dropped (:121-123,137-138)
                    sum.push2(ex, ey);
                    m++;
                }
                mse = sum.res / (double)(2 * nv);
            }
        }
        int total;
        const int o = place(has, total);
        if (has) a.res[(size_t)b * N + o] = (float)mse;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) a.counts[b] = s_base;
}

// Dynamic LDS a workgroup of correspond_kernel may use: 64 KiB without asking; the first call asks the runtime for gfx950's
// whole 160 KiB per workgroup (large P x C: the candidate lists alone are P * C * 16 bytes) and remembers the answer.
static int correspond_lds_limit()
{
    static std::mutex mu;
    static int limit[64] = {0}; // per device (the attribute belongs to the function on the current device); 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 64 * 1024 - 64;
    std::lock_guard<std::mutex> lk(mu);
    if (limit[dev] == 0) {
        const int want = 160 * 1024 - 64; // minus the kernel's static words
        const bool ok = hipFuncSetAttribute((const void*)correspond_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                        hipFuncSetAttribute((const void*)correspond_kernel<int32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        limit[dev] = ok ? want : 64 * 1024 - 64;
    }
    return limit[dev];
}
// the budget the plan is made for: 64 KiB when that holds everything (four workgroups per CU stay possible), else the limit
int correspond_lds_budget(int P, int C)
{
    const int small = 64 * 1024 - 64;
    const CorrLds L = corr_lds_plan(P, C, small);
    if (L.total <= small && L.stage_pts == (C * P <= 2048) && L.err_cap == 1024) return small;
    return correspond_lds_limit();
}
size_t correspond_smem_bytes(int P, int C) { return (size_t)corr_lds_plan(P, C, correspond_lds_budget(P, C)).total; }
bool correspond_fits(int P, int C) { return correspond_smem_bytes(P, C) <= (size_t)correspond_lds_budget(P, C); }

void launch_correspond(const CorrArgs& a_, hipStream_t s)
{
    CorrArgs a = a_;
    a.lds_budget = correspond_lds_budget(a.P, a.C);
    size_t sm = (size_t)corr_lds_plan(a.P, a.C, a.lds_budget).total;
    // threads per time step (MOCAP_CORR_THREADS: A/B switch; one wave per step measured 0.064 against 0.051 ms alone and the
    // same in the three-batch pipeline)
    const int threads = (a.threads == 64 || a.threads == 128) ? a.threads : 256;
    if (a.pts_f64)
        hipLaunchKernelGGL(correspond_kernel<double>, dim3(a.T), dim3(threads), sm, s, a);
    else
        hipLaunchKernelGGL(correspond_kernel<int32_t>, dim3(a.T), dim3(threads), sm, s, a);
}
void launch_epipolar_scores(const EpiArgs& a, hipStream_t s)
{
    const long n = (long)a.n_roots * a.n_cand;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (a.pts_f64) hipLaunchKernelGGL(epipolar_scores_kernel<double>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(epipolar_scores_kernel<int32_t>, dim3(blocks), dim3(256), 0, s, a);
}
void launch_ba_residuals(const BaArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(ba_residuals_kernel, dim3(a.B), dim3(256), 0, s, a);
}
void launch_triangulate(const TriArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(triangulate_kernel, dim3((a.N + 63) / 64), dim3(64), 0, s, a);
}
void launch_reproject(const ReprojArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(reproject_kernel, dim3((a.N + 63) / 64), dim3(64), 0, s, a);
}

} // namespace mocap
