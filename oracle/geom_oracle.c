/*
 * geom_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, FP64, no FMA contraction) of the correspondence + triangulation half
 * of MocapV2's per-frame hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (mocapv2_amd/) never does.
 *
 * Pinned by the npz files under tests/golden/, which were produced by the reference's own lib/Helpers.py run in
 * the build container (oracle/gen_golden.py; the two cv2 primitives it calls were supplied as
 * closed-form stand-ins, see that script) and by the reference's bundled known-answer data
 * jsons/image_points.json + after_ba_extrinsics.json -> after_ba_objects.json.
 *
 *   reference lib/Helpers.py:43-84    triangulate_point                 -> orc_triangulate
 *   reference lib/Helpers.py:113-143  calculate_reprojection_error      -> orc_reproj_mse
 *   reference lib/Helpers.py:178-280  find_point_correspondance_and_object_points -> orc_correspond
 *   reference lib/Helpers.py:145-167  params_to_camera_poses + residual_function  -> orc_ba_residuals
 *   cv.computeCorrespondEpilines (Helpers.py:207) -> orc_epiline
 *   cv.projectPoints            (Helpers.py:133) -> project_point
 *   scipy.linalg.svd of the 4x4 A^T A (Helpers.py:76-78) -> cyclic Jacobi eigen-solve (same null vector)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

/* numpy's pairwise summation (np.add.reduce on a contiguous float64 vector) */
static double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

ORC_API double orc_np_mean(const double *a, long n) { return np_pairwise_sum(a, n) / (double)n; }

/* cv.computeCorrespondEpilines(points float32, whichImage=1, F): l = F [x y 1]^T, scaled so that
 * a^2+b^2 = 1, rounded to float32. */
ORC_API void orc_epiline(const double F[9], double px, double py, float line[3])
{
    float xf = (float)px, yf = (float)py;
    double x = xf, y = yf;
    double a = F[0] * x + F[1] * y + F[2];
    double b = F[3] * x + F[4] * y + F[5];
    double c = F[6] * x + F[7] * y + F[8];
    double nu = a * a + b * b;
    nu = nu ? 1. / sqrt(nu) : 1.;
    a *= nu;
    b *= nu;
    c *= nu;
    line[0] = (float)a;
    line[1] = (float)b;
    line[2] = (float)c;
}

/* distance of point (x,y) to the float32 line, the expression of reference lib/Helpers.py:217 */
ORC_API double orc_epi_distance(const float line[3], double x, double y)
{
    double a = line[0], b = line[1], c = line[2];
    return fabs(a * x + b * y + c) / sqrt(a * a + b * b);
}

/* symmetric 4x4 eigen-decomposition by cyclic Jacobi rotations; returns the eigenvector of the
 * smallest eigenvalue in v[4] */
static void smallest_eigvec4(double B[4][4], double v[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int p = 0; p < 4; p++) {
            diag += B[p][p] * B[p][p];
            for (int q = p + 1; q < 4; q++) off += B[p][q] * B[p][q];
        }
        if (off == 0.0 || off <= 1e-40 * diag) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double apq = B[p][q];
                if (apq == 0.0) continue;
                double theta = (B[q][q] - B[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) { /* columns p,q of B */
                    double bkp = B[k][p], bkq = B[k][q];
                    B[k][p] = c * bkp - s * bkq;
                    B[k][q] = s * bkp + c * bkq;
                }
                for (int k = 0; k < 4; k++) { /* rows p,q of B */
                    double bpk = B[p][k], bqk = B[q][k];
                    B[p][k] = c * bpk - s * bqk;
                    B[q][k] = s * bpk + c * bqk;
                }
                for (int k = 0; k < 4; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int k = 1; k < 4; k++)
        if (B[k][k] < B[m][m]) m = k;
    for (int k = 0; k < 4; k++) v[k] = V[k][m];
}

/* P = K @ [R|t] */
static void projection_matrix(const double K[9], const double R[9], const double t[3], double P[12])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[3 * r + k] * (c < 3 ? R[3 * k + c] : t[k]);
            P[4 * r + c] = s;
        }
}

/* DLT over n observations.  pose_idx[i] selects [R|t], k_idx[i] selects the intrinsics
 * (the reference indexes intrinsics by position after dropping None entries, Helpers.py:59-61). */
ORC_API int orc_triangulate(const double *pts, const int32_t *pose_idx, const int32_t *k_idx, int n,
                            const double *K, const double *R, const double *t, double X[3])
{
    if (n <= 1) return 1; /* reference returns [None, None, None] */
    double B[4][4];
    memset(B, 0, sizeof(B));
    for (int i = 0; i < n; i++) {
        double P[12];
        projection_matrix(K + 9 * k_idx[i], R + 9 * pose_idx[i], t + 3 * pose_idx[i], P);
        double x = pts[2 * i], y = pts[2 * i + 1];
        double r0[4], r1[4];
        for (int k = 0; k < 4; k++) {
            r0[k] = y * P[8 + k] - P[4 + k];
            r1[k] = P[k] - x * P[8 + k];
        }
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) B[j][k] += r0[j] * r0[k] + r1[j] * r1[k];
    }
    double v[4];
    smallest_eigvec4(B, v);
    X[0] = v[0] / v[3];
    X[1] = v[1] / v[3];
    X[2] = v[2] / v[3];
    return 0;
}

/* cv.projectPoints for one float32 object point, R given as a matrix, 5 distortion coefficients;
 * result rounded to float32 */
static void project_point(const float Xf[3], const double R[9], const double t[3], const double K[9],
                          const double d[5], float out[2])
{
    double X = Xf[0], Y = Xf[1], Z = Xf[2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6;
    double xd = x * cdist + d[2] * a1 + d[3] * a2;
    double yd = y * cdist + d[2] * a3 + d[3] * a1;
    out[0] = (float)(xd * K[0] + K[2]);
    out[1] = (float)(yd * K[4] + K[5]);
}

/* mean squared pixel error over n observations (reference lib/Helpers.py:113-143) */
ORC_API int orc_reproj_mse(const double *pts, const int32_t *pose_idx, const int32_t *k_idx, int n,
                           const double X[3], const double *K, const double *dist, const double *R,
                           const double *t, double *mse)
{
    if (n <= 1) return 1; /* reference returns None */
    double e[64];
    if (2 * n > 64) return -1;
    float Xf[3] = {(float)X[0], (float)X[1], (float)X[2]};
    for (int i = 0; i < n; i++) {
        float pr[2];
        project_point(Xf, R + 9 * pose_idx[i], t + 3 * pose_idx[i], K + 9 * k_idx[i], dist + 5 * k_idx[i], pr);
        double dx = pts[2 * i] - (double)pr[0], dy = pts[2 * i + 1] - (double)pr[1];
        e[2 * i] = dx * dx;
        e[2 * i + 1] = dy * dy;
    }
    *mse = np_pairwise_sum(e, 2 * n) / (double)(2 * n);
    return 0;
}

typedef struct {
    double d;
    int idx;
} match_t;

static void sort_matches(match_t *m, int n)
{ /* stable insertion sort by distance (numpy argsort on < 17 elements is an insertion sort) */
    for (int i = 1; i < n; i++) {
        match_t k = m[i];
        int j = i - 1;
        while (j >= 0 && m[j].d > k.d) {
            m[j + 1] = m[j];
            j--;
        }
        m[j + 1] = k;
    }
}

/*
 * One time step of find_point_correspondance_and_object_points (reference lib/Helpers.py:178-280).
 * Only the camera-0 roots can reach the outputs (roots created from cameras i>0 start with a None
 * entry and are skipped at Helpers.py:93,265), so only those are evaluated.
 *
 *   pts      [C][max_pts][2]  image points per camera (sentinel already removed), counts in npts[C]
 *   F        [(C-1)][9]       F[i-1] maps a camera-0 pixel to a line in camera i
 *   out_xyz  [n0][3], out_err [n0], out_grp [n0][C][2] : per surviving root (in root order)
 *   out_root [n0]             camera-0 index of each surviving root
 *   order    [n0]             argsort of out_err (stable)
 * returns the number of surviving roots, or <0 (-2: more than max_groups groups for one root).
 */
ORC_API int orc_correspond(int C, int max_pts, const double *pts, const int32_t *npts, const double *K,
                           const double *dist, const double *R, const double *t, const double *F,
                           double cutoff, long max_groups, double *out_xyz, double *out_err,
                           double *out_grp, int32_t *out_root, int32_t *order)
{
    if (C < 1 || C > 32) return -1;
    int n0 = npts[0];
    int nout = 0;
    match_t *m = (match_t *)malloc(sizeof(match_t) * (size_t)(C > 1 ? C : 1) * (max_pts > 0 ? max_pts : 1));
    double *errs = 0;
    long errs_cap = 0;
    int rc = 0;
    int32_t ident[32];
    for (int i = 0; i < 32; i++) ident[i] = i;
    if (!m) return -1;
    for (int j = 0; j < n0; j++) {
        const double *root = pts + 2 * j;
        int nm[32];
        long G = 1;
        int dead = 0;
        for (int i = 1; i < C; i++) {
            float line[3];
            orc_epiline(F + 9 * (i - 1), root[0], root[1], line);
            match_t *mi = m + (size_t)i * max_pts;
            int k = 0;
            const double *pi = pts + (size_t)i * max_pts * 2;
            for (int p = 0; p < npts[i]; p++) {
                double d = orc_epi_distance(line, pi[2 * p], pi[2 * p + 1]);
                if (d < cutoff) {
                    mi[k].d = d;
                    mi[k].idx = p;
                    k++;
                }
            }
            sort_matches(mi, k);
            nm[i] = k;
            if (k == 0) { dead = 1; break; }
            if (G > max_groups / k) { rc = -2; goto done; }
            G *= k;
        }
        if (dead) continue;
        if (C < 2) continue; /* a single camera never triangulates (Helpers.py:55-56) */
        if (G > errs_cap) {
            free(errs);
            errs = (double *)malloc(sizeof(double) * G);
            errs_cap = G;
            if (!errs) { rc = -1; goto done; }
        }
        for (long g = 0; g < G; g++) {
            double gp[64];
            long rem = g;
            gp[0] = root[0];
            gp[1] = root[1];
            for (int i = 1; i < C; i++) { /* camera 1 is the fastest-varying digit */
                int dgt = (int)(rem % nm[i]);
                rem /= nm[i];
                int p = m[(size_t)i * max_pts + dgt].idx;
                gp[2 * i] = pts[((size_t)i * max_pts + p) * 2];
                gp[2 * i + 1] = pts[((size_t)i * max_pts + p) * 2 + 1];
            }
            double X[3], e;
            orc_triangulate(gp, ident, ident, C, K, R, t, X);
            orc_reproj_mse(gp, ident, ident, C, X, K, dist, R, t, &e);
            errs[g] = e;
            if (g == 0) {
                memcpy(out_xyz + 3 * nout, X, sizeof(X));
                memcpy(out_grp + (size_t)nout * C * 2, gp, sizeof(double) * 2 * C);
            }
        }
        out_err[nout] = np_pairwise_sum(errs, G) / (double)G;
        out_root[nout] = j;
        nout++;
    }
    /* np.argsort(errors): stable order used (ties between float64 means do not occur in practice) */
    for (int i = 0; i < nout; i++) order[i] = i;
    for (int i = 1; i < nout; i++) {
        int k = order[i], j = i - 1;
        while (j >= 0 && out_err[order[j]] > out_err[k]) {
            order[j + 1] = order[j];
            j--;
        }
        order[j + 1] = k;
    }
    rc = nout;
done:
    free(m);
    free(errs);
    return rc;
}

/* scipy Rotation.from_rotvec(v).as_matrix() */
ORC_API void orc_rotvec_to_matrix(const double v[3], double Rm[9])
{
    double angle = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double scale;
    if (angle <= 1e-3) {
        double a2 = angle * angle;
        scale = 0.5 - a2 / 48 + a2 * a2 / 3840;
    } else
        scale = sin(angle / 2) / angle;
    double x = scale * v[0], y = scale * v[1], z = scale * v[2], w = cos(angle / 2);
    double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    Rm[0] = x2 - y2 - z2 + w2;
    Rm[3] = 2 * (xy + zw);
    Rm[6] = 2 * (xz - yw);
    Rm[1] = 2 * (xy - zw);
    Rm[4] = -x2 + y2 - z2 + w2;
    Rm[7] = 2 * (yz + xw);
    Rm[2] = 2 * (xz + yw);
    Rm[5] = 2 * (yz - xw);
    Rm[8] = -x2 - y2 + z2 + w2;
}

/* bundle_adjustment.residual_function (reference lib/Helpers.py:161-167): camera 0 at the origin,
 * cameras 1.. from (rotvec, t) parameter sextuples; per-point reprojection MSE as float32.
 * pts [N][C][2]; valid [N][C] = 0 where the observation is [None, None].  Groups holding a None are
 * not triangulated (Helpers.py:93) and the error loop pairs groups with object points positionally
 * (Helpers.py:104), both as in the reference. */
ORC_API int orc_ba_residuals(const double *params, int C, const double *pts, const uint8_t *valid, int N,
                             const double *K, const double *dist, float *out)
{
    if (C < 2 || C > 32) return -1;
    double R[32 * 9], t[32 * 3];
    memset(R, 0, sizeof(R));
    memset(t, 0, sizeof(t));
    R[0] = R[4] = R[8] = 1;
    for (int i = 1; i < C; i++) {
        orc_rotvec_to_matrix(params + 6 * (i - 1), R + 9 * i);
        memcpy(t + 3 * i, params + 6 * (i - 1) + 3, sizeof(double) * 3);
    }
    int32_t ident[32];
    for (int i = 0; i < 32; i++) ident[i] = i;
    double *obj = (double *)malloc(sizeof(double) * 3 * (N > 0 ? N : 1));
    if (!obj) return -1;
    int nobj = 0;
    for (int n = 0; n < N; n++) {
        int full = 1;
        for (int c = 0; c < C; c++) full &= valid[(size_t)n * C + c] != 0;
        if (!full) continue;
        orc_triangulate(pts + (size_t)n * C * 2, ident, ident, C, K, R, t, obj + 3 * nobj);
        nobj++;
    }
    int k = 0;
    for (int n = 0; n < N && n < nobj; n++) {
        double gp[64];
        int32_t pose_idx[32], k_idx[32];
        int m = 0;
        for (int c = 0; c < C; c++)
            if (valid[(size_t)n * C + c]) {
                gp[2 * m] = pts[((size_t)n * C + c) * 2];
                gp[2 * m + 1] = pts[((size_t)n * C + c) * 2 + 1];
                pose_idx[m] = c;
                k_idx[m] = m;
                m++;
            }
        double e;
        if (orc_reproj_mse(gp, pose_idx, k_idx, m, obj + 3 * n, K, dist, R, t, &e) == 0) out[k++] = (float)e;
    }
    free(obj);
    return k;
}
