/*
 * blob_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, no FMA contraction) of the blob -> centroid half of
 * MocapV2's per-frame hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (mocapv2_amd/) never does.
 *
 * PARITY UNPINNED for this half: every arithmetic step of the reference lives in OpenCV
 * (opencv-python, version unpinned -- reference README.md:65) and numba, neither of which is
 * installed here, and the reference ships no (image -> centroid) known-answer pair.  The
 * structure follows the reference files cited below; the arithmetic follows OpenCV's published
 * algorithms as restated in SURVEY.md section 8a / Appendix A.
 *
 *   reference lib/ImageOperations.py:33-78   _find_dot            -> orc_find_dot
 *   reference lib/ImageOperations.py:23-31   image_filter_gpu     -> orc_filter (order 0)
 *   reference lib/ImageOperations.py:15-21   image_filter_cpu     -> orc_filter (order 1)
 *   reference lib/CudaOperations.py:5-41     blur_kernel/fast_cuda_blur -> orc_box_blur_u8
 *   reference lib/CudaOperations.py:43-100   demosaic_kernel      -> orc_demosaic_u8
 *   reference RealtimeTracking_FLIR.py:103-104  cvtColor BayerGR2BGR + BGR2GRAY -> orc_bayer_gray_u8
 *   cv.undistort / threshold / medianBlur / findContours / contourArea / arcLength / moments
 *                                            -> orc_undistort_u8, orc_threshold_u8,
 *                                               orc_median5_u8, orc_find_contours
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * A2  cv.undistort(img, K, dist)  (reference lib/ImageOperations.py:38)
 * newCameraMatrix = K, R = I, 5 distortion coefficients (k1 k2 p1 p2 k3), INTER_LINEAR,
 * BORDER_CONSTANT(0).  The map is built in row stripes exactly like cv::undistort does
 * (stripe = min(max(1, 4096/cols), rows) rows, principal point shifted by the stripe origin,
 * _x accumulated with += ir[0] along the row), quantised to 1/32 px with round-half-even,
 * and the 8-bit bilinear remap uses the 15-bit fixed-point weights 32*(32-a|a)*(32-b|b).
 * ------------------------------------------------------------------------------------------ */

/* inverse of a 3x3 by cofactors, the form OpenCV uses for n == 3 */
static void inv3(const double m[9], double o[9])
{
    double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                 m[2] * (m[3] * m[7] - m[4] * m[6]);
    double d = 1.0 / det;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * d;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * d;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * d;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * d;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * d;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * d;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * d;
}

static inline int round_half_even_sat(double v)
{
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int)(-2147483647 - 1);
    return (int)nearbyint(v); /* default rounding mode = to nearest even, like cvRound */
}

/* Quantised undistort map: iu/iv = round(32 * source coordinate) for every destination pixel. */
ORC_API void orc_undistort_map(int H, int W, const double K[9], const double dist[5], int32_t *iu,
                               int32_t *iv)
{
    double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    int stripe0 = 4096 / (W > 1 ? W : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > H) stripe0 = H;
    for (int ys = 0; ys < H; ys += stripe0) {
        int sh = stripe0 < H - ys ? stripe0 : H - ys;
        double Ar[9], ir[9];
        memcpy(Ar, K, sizeof(Ar));
        Ar[5] = v0 - ys;
        inv3(Ar, ir);
        for (int i = 0; i < sh; i++) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            int32_t *ru = iu + (size_t)(ys + i) * W, *rv = iv + (size_t)(ys + i) * W;
            for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                double w = 1. / _w, x = _x * w, y = _y * w;
                double x2 = x * x, y2 = y * y;
                double r2 = x2 + y2, _2xy = 2 * x * y;
                double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
                double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
                double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
                double u = fx * xd + u0;
                double v = fy * yd + v0;
                ru[j] = round_half_even_sat(u * 32);
                rv[j] = round_half_even_sat(v * 32);
            }
        }
    }
}

static inline int tap(const uint8_t *src, int H, int W, int y, int x)
{
    return ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) ? src[(size_t)y * W + x] : 0;
}

/* bilinear remap of an 8-bit image through a quantised map (cv::remap, CV_16SC2 + CV_16UC1 form) */
ORC_API void orc_remap_u8(const uint8_t *src, uint8_t *dst, int H, int W, const int32_t *iu,
                          const int32_t *iv)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t o = (size_t)y * W + x;
            int sx = (int16_t)(iu[o] >> 5), sy = (int16_t)(iv[o] >> 5);
            int a = iu[o] & 31, b = iv[o] & 31;
            int w00 = 32 * (32 - a) * (32 - b), w01 = 32 * a * (32 - b);
            int w10 = 32 * (32 - a) * b, w11 = 32 * a * b;
            int acc = w00 * tap(src, H, W, sy, sx) + w01 * tap(src, H, W, sy, sx + 1) +
                      w10 * tap(src, H, W, sy + 1, sx) + w11 * tap(src, H, W, sy + 1, sx + 1);
            dst[o] = (uint8_t)((acc + (1 << 14)) >> 15);
        }
}

ORC_API int orc_undistort_u8(const uint8_t *src, uint8_t *dst, int H, int W, const double K[9],
                             const double dist[5])
{
    int32_t *iu = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * W * 2);
    if (!iu) return -1;
    int32_t *iv = iu + (size_t)H * W;
    orc_undistort_map(H, W, K, dist, iu, iv);
    orc_remap_u8(src, dst, H, W, iu, iv);
    free(iu);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A3  fast_cuda_blur (reference lib/CudaOperations.py:5-41): mean over the in-bounds taps of
 * a k x k window, float64 accumulator, result stored to float32, then .astype(uint8).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_box_blur_u8(const uint8_t *src, uint8_t *dst, int H, int W, int ksize)
{
    int k = ksize / 2;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            double sum = 0.0;
            int count = 0;
            for (int di = -k; di <= k; di++)
                for (int dj = -k; dj <= k; dj++) {
                    int ni = i + di, nj = j + dj;
                    if (ni >= 0 && ni < H && nj >= 0 && nj < W) {
                        sum += src[(size_t)ni * W + nj];
                        count++;
                    }
                }
            float f = (float)(sum / count);
            dst[(size_t)i * W + j] = (uint8_t)f; /* truncation, value is in [0,255] */
        }
}

/* A4  cv.threshold(img, thresh, maxval, THRESH_BINARY) on uint8: threshold floored, maxval rounded */
ORC_API void orc_threshold_u8(const uint8_t *src, uint8_t *dst, size_t n, double thresh, double maxval)
{
    int it = (int)floor(thresh);
    int im = (int)nearbyint(maxval);
    if (im < 0) im = 0;
    if (im > 255) im = 255;
    for (size_t i = 0; i < n; i++) dst[i] = src[i] > it ? (uint8_t)im : 0;
}

/* A5  cv.medianBlur(img, 5): exact 5x5 median, BORDER_REPLICATE */
ORC_API void orc_median5_u8(const uint8_t *src, uint8_t *dst, int H, int W)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int hist[256];
            memset(hist, 0, sizeof(hist));
            for (int dy = -2; dy <= 2; dy++) {
                int yy = y + dy;
                yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                for (int dx = -2; dx <= 2; dx++) {
                    int xx = x + dx;
                    xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                    hist[src[(size_t)yy * W + xx]]++;
                }
            }
            int c = 0, v = 0;
            for (v = 0; v < 256; v++) {
                c += hist[v];
                if (c >= 13) break;
            }
            dst[(size_t)y * W + x] = (uint8_t)v;
        }
}

/* Separable forms used for the timed CPU baseline; tests check them against the literal forms above.
 * floor(S/c) equals the float path of fast_cuda_blur for uint8 input. */
ORC_API int orc_box_blur_u8_fast(const uint8_t *src, uint8_t *dst, int H, int W, int ksize)
{
    int k = ksize / 2;
    uint16_t *hs = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)H * W);
    if (!hs) return -1;
    for (int y = 0; y < H; y++) {
        const uint8_t *r = src + (size_t)y * W;
        uint16_t *h = hs + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            int lo = x - k < 0 ? 0 : x - k, hi = x + k >= W ? W - 1 : x + k, s = 0;
            for (int j = lo; j <= hi; j++) s += r[j];
            h[x] = (uint16_t)s;
        }
    }
    for (int y = 0; y < H; y++) {
        int lo = y - k < 0 ? 0 : y - k, hi = y + k >= H ? H - 1 : y + k;
        int cy = hi - lo + 1;
        for (int x = 0; x < W; x++) {
            int xl = x - k < 0 ? 0 : x - k, xh = x + k >= W ? W - 1 : x + k;
            int c = cy * (xh - xl + 1), s = 0;
            for (int i = lo; i <= hi; i++) s += hs[(size_t)i * W + x];
            dst[(size_t)y * W + x] = (uint8_t)(s / c);
        }
    }
    free(hs);
    return 0;
}

/* 5x5 median of a {0,v} image with replicated borders = majority vote (>= 13 of 25 taps) */
ORC_API int orc_majority5_u8(const uint8_t *src, uint8_t *dst, int H, int W)
{
    uint8_t *hs = (uint8_t *)malloc((size_t)H * W);
    if (!hs) return -1;
    uint8_t v = 0;
    for (size_t i = 0; i < (size_t)H * W; i++)
        if (src[i]) { v = src[i]; break; }
    for (int y = 0; y < H; y++) {
        const uint8_t *r = src + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            int s = 0;
            for (int d = -2; d <= 2; d++) {
                int xx = x + d;
                xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                s += r[xx] != 0;
            }
            hs[(size_t)y * W + x] = (uint8_t)s;
        }
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int s = 0;
            for (int d = -2; d <= 2; d++) {
                int yy = y + d;
                yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                s += hs[(size_t)yy * W + x];
            }
            dst[(size_t)y * W + x] = s >= 13 ? v : 0;
        }
    free(hs);
    return 0;
}

/* image_filter_gpu (order 0: blur -> threshold -> median) / image_filter_cpu (order 1: median -> threshold);
 * order 2 = order 0 computed with the separable forms */
ORC_API int orc_filter(const uint8_t *src, uint8_t *dst, int H, int W, int order, int ksize,
                       double thresh)
{
    size_t n = (size_t)H * W;
    uint8_t *tmp = (uint8_t *)malloc(n);
    if (!tmp) return -1;
    if (order == 2) {
        int rc = orc_box_blur_u8_fast(src, dst, H, W, ksize);
        orc_threshold_u8(dst, tmp, n, thresh, 255.0);
        if (!rc) rc = orc_majority5_u8(tmp, dst, H, W);
        free(tmp);
        return rc;
    }
    if (order == 0) {
        orc_box_blur_u8(src, dst, H, W, ksize);
        orc_threshold_u8(dst, tmp, n, thresh, 255.0);
        orc_median5_u8(tmp, dst, H, W);
    } else {
        orc_median5_u8(src, tmp, H, W);
        orc_threshold_u8(tmp, dst, n, thresh, 255.0);
    }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * A6  cv.findContours(grey, RETR_TREE, CHAIN_APPROX_SIMPLE): Suzuki-Abe border following on
 * grey != 0 with a virtual 1-px zero frame, 8-connected foreground, full-width labels.
 * Output order = pre-order walk of the tree in which every new border is inserted at the head
 * of its parent's child list.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int is_hole;
    int parent;      /* contour index, 0 = frame */
    int first_child; /* head of child list */
    int next_sibling;
    int ox, oy;      /* origin (image coordinates, unpadded) */
    int pt_begin, pt_count; /* SIMPLE vertices in pts[] */
    int steps;       /* number of border steps (CHAIN_APPROX_NONE length) */
} orc_cinfo;

typedef struct {
    orc_cinfo *c;
    int n, cap;
    int *pts; /* x,y pairs */
    int npts, pcap;
} orc_cstore;

static int cstore_new(orc_cstore *s)
{
    if (s->n == s->cap) {
        int nc = s->cap ? s->cap * 2 : 256;
        orc_cinfo *p = (orc_cinfo *)realloc(s->c, sizeof(orc_cinfo) * nc);
        if (!p) return -1;
        s->c = p;
        s->cap = nc;
    }
    memset(&s->c[s->n], 0, sizeof(orc_cinfo));
    return s->n++;
}

static int cstore_pt(orc_cstore *s, int x, int y)
{
    if (s->npts == s->pcap) {
        int nc = s->pcap ? s->pcap * 2 : 4096;
        int *p = (int *)realloc(s->pts, sizeof(int) * 2 * nc);
        if (!p) return -1;
        s->pts = p;
        s->pcap = nc;
    }
    s->pts[2 * s->npts] = x;
    s->pts[2 * s->npts + 1] = y;
    s->npts++;
    return 0;
}

static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* follow one border starting at padded position (px,py); label with nbd */
static int trace_border(int32_t *f, int step, int px, int py, int is_hole, int nbd, orc_cstore *st,
                        orc_cinfo *ci)
{
    int deltas[16];
    for (int s = 0; s < 8; s++) deltas[s] = deltas[s + 8] = DY[s] * step + DX[s];
    int32_t *i0 = f + (size_t)py * step + px, *i1, *i3, *i4 = 0;
    int x = px - 1, y = py - 1; /* reported coordinates are unpadded */
    int s_end, s, prev_s;
    ci->pt_begin = st->npts;
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) { /* isolated pixel */
        *i0 = -nbd;
        if (cstore_pt(st, x, y)) return -1;
        ci->steps = 0;
    } else {
        i3 = i0;
        prev_s = s ^ 4;
        for (;;) {
            s_end = s;
            while (s < 15) {
                i4 = i3 + deltas[++s];
                if (*i4 != 0) break;
            }
            s &= 7;
            if ((unsigned)(s - 1) < (unsigned)s_end)
                *i3 = -nbd; /* the East neighbour was examined and is zero */
            else if (*i3 == 1)
                *i3 = nbd;
            if (s != prev_s) {
                if (cstore_pt(st, x, y)) return -1;
                prev_s = s;
            }
            x += DX[s];
            y += DY[s];
            ci->steps++;
            if (i4 == i0 && i3 == i1) break;
            i3 = i4;
            s = (s + 4) & 7;
        }
    }
    ci->pt_count = st->npts - ci->pt_begin;
    return 0;
}

/* per-contour measurements (A7/A8) */
typedef struct {
    int32_t is_hole, parent_order; /* parent's position in the output order, -1 = frame */
    int32_t ox, oy, npts, steps;
    int64_t a00, a10, a01; /* Green's-theorem integer sums over the SIMPLE polygon */
    double area, perimeter;
    int32_t kept, cx, cy;  /* passes the _find_dot filter; truncated centroid */
    int32_t pad;
} orc_contour;

static void measure(const int *p, int n, orc_contour *o)
{
    int64_t a00 = 0, a10 = 0, a01 = 0;
    double per = 0.0, a = 0.0;
    if (n > 0) {
        int xp = p[2 * (n - 1)], yp = p[2 * (n - 1) + 1];
        for (int i = 0; i < n; i++) {
            int xi = p[2 * i], yi = p[2 * i + 1];
            int64_t d = (int64_t)xp * yi - (int64_t)xi * yp;
            a00 += d;
            a10 += d * (xp + xi);
            a01 += d * (yp + yi);
            /* contourArea accumulates the same cross products in double (exact integers) */
            a += (double)(float)xp * (float)yi - (double)(float)xi * (float)yp;
            if (n > 1) {
                float dx = (float)xi - (float)xp, dy = (float)yi - (float)yp;
                per += sqrtf(dx * dx + dy * dy);
            }
            xp = xi;
            yp = yi;
        }
    }
    o->a00 = a00;
    o->a10 = a10;
    o->a01 = a01;
    o->area = fabs(a * 0.5);
    o->perimeter = per;
}

/* Finds all borders of (img != 0).  Returns the number of contours (in OpenCV output order) or <0.
 * out[] receives up to max_out measurements; pts_out (optional) receives the concatenated SIMPLE
 * vertices (x,y) of the contours in output order, up to max_pts points, pt_off[i] = first vertex. */
ORC_API int orc_find_contours(const uint8_t *img, int H, int W, orc_contour *out, int max_out,
                              int32_t *pts_out, int32_t *pt_off, int max_pts)
{
    int step = W + 2;
    int32_t *f = (int32_t *)calloc((size_t)(H + 2) * step, sizeof(int32_t));
    if (!f) return -1;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) f[(size_t)(y + 1) * step + x + 1] = img[(size_t)y * W + x] != 0;
    orc_cstore st;
    memset(&st, 0, sizeof(st));
    int rc = 0;
    if (cstore_new(&st) < 0) { rc = -1; goto done; } /* index 0 = frame (a hole) */
    st.c[0].is_hole = 1;
    st.c[0].parent = -1;
    int nbd = 1;
    for (int y = 1; y <= H; y++) {
        int32_t *row = f + (size_t)y * step;
        int lnbd_x = 0;
        int prev = 0;
        for (int x = 1; x <= W; x++) {
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            int start = 1;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1)
                    start = 0;
                else {
                    if (prev & -2) lnbd_x = x - 1;
                    is_hole = 1;
                }
            }
            if (start) {
                int par;
                if (lnbd_x <= 0)
                    par = 0;
                else {
                    int lval = abs(row[lnbd_x]);
                    /* labels are nbd = contour index + 1 */
                    par = lval - 1;
                    if (st.c[par].is_hole == is_hole) par = st.c[par].parent;
                    if (par < 0) par = 0;
                }
                int ox = x - is_hole;
                lnbd_x = ox;
                int id = cstore_new(&st);
                if (id < 0) { rc = -1; goto done; }
                nbd = id + 1;
                orc_cinfo *ci = &st.c[id];
                ci->is_hole = is_hole;
                ci->parent = par;
                ci->ox = ox - 1;
                ci->oy = y - 1;
                if (trace_border(f, step, ox, y, is_hole, nbd, &st, ci) < 0) { rc = -1; goto done; }
                ci = &st.c[id];
                ci->next_sibling = st.c[par].first_child; /* insert at head */
                st.c[par].first_child = id;
                p = row[x]; /* may have been relabelled by the trace */
            }
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    (void)nbd;
    /* pre-order walk from the frame's first child */
    {
        int total = st.n - 1;
        int *order = (int *)malloc(sizeof(int) * (total + 1));
        int *pos = (int *)malloc(sizeof(int) * (st.n));
        if (!order || !pos) { free(order); free(pos); rc = -1; goto done; }
        int n = 0;
        int cur = st.c[0].first_child;
        while (cur) {
            pos[cur] = n;
            order[n++] = cur;
            if (st.c[cur].first_child)
                cur = st.c[cur].first_child;
            else {
                while (cur && !st.c[cur].next_sibling) cur = st.c[cur].parent > 0 ? st.c[cur].parent : 0;
                if (cur) cur = st.c[cur].next_sibling;
            }
        }
        int np = 0;
        for (int i = 0; i < n; i++) {
            orc_cinfo *ci = &st.c[order[i]];
            if (i < max_out) {
                orc_contour *o = &out[i];
                memset(o, 0, sizeof(*o));
                o->is_hole = ci->is_hole;
                o->parent_order = ci->parent > 0 ? pos[ci->parent] : -1;
                o->ox = ci->ox;
                o->oy = ci->oy;
                o->npts = ci->pt_count;
                o->steps = ci->steps;
                measure(st.pts + 2 * ci->pt_begin, ci->pt_count, o);
            }
            if (pts_out && pt_off) {
                if (i < max_out) pt_off[i] = np;
                for (int k = 0; k < ci->pt_count && np < max_pts; k++, np++) {
                    pts_out[2 * np] = st.pts[2 * (ci->pt_begin + k)];
                    pts_out[2 * np + 1] = st.pts[2 * (ci->pt_begin + k) + 1];
                }
            }
        }
        if (pts_out && pt_off && n < max_out) pt_off[n] = np;
        rc = n;
        free(order);
        free(pos);
    }
done:
    free(f);
    free(st.c);
    free(st.pts);
    return rc;
}

/* A7 + A8: the filter and centroid of reference lib/ImageOperations.py:43-65 applied to one contour */
ORC_API void orc_contour_select(orc_contour *o, double min_area, double min_circ)
{
    o->kept = 0;
    o->cx = o->cy = 0;
    double area = o->area, perimeter = o->perimeter;
    if (perimeter != 0.0) {
        double pi4 = 4 * 3.141592653589793;
        double circularity = pi4 * area / (perimeter * perimeter);
        if (circularity > min_circ && area > min_area) {
            /* cv.moments: m00 = a00/2, m10 = a10/6, m01 = a01/6, signs flipped when a00 < 0 */
            double a00 = (double)o->a00, a10 = (double)o->a10, a01 = (double)o->a01;
            if (fabs(a00) > FLT_EPSILON) {
                double db1_2 = 0.5, db1_6 = 0.16666666666666666666666666666667;
                if (a00 < 0) { db1_2 = -db1_2; db1_6 = -db1_6; }
                double m00 = a00 * db1_2, m10 = a10 * db1_6, m01 = a01 * db1_6;
                if (m00 != 0) {
                    o->kept = 1;
                    o->cx = (int32_t)(m10 / m00); /* Python int(): truncation toward zero */
                    o->cy = (int32_t)(m01 / m00);
                }
            }
        }
    }
}

typedef struct {
    int32_t ksize;       /* 5 */
    int32_t filter_order; /* 0 = image_filter_gpu order, 1 = image_filter_cpu order */
    double thresh;       /* 255*0.85 */
    double min_area;     /* 500 */
    double min_circ;     /* 0.5 */
    int32_t undistort;   /* 1 = apply cv.undistort first */
    int32_t pad;
} orc_blob_params;

/* _find_dot (reference lib/ImageOperations.py:33-78) minus the drawing: returns the number of
 * image points written to xy (pairs), in the reference's order.  mask_out (optional) receives the
 * filtered binary image. */
ORC_API int orc_find_dot(const uint8_t *img, int H, int W, const double K[9], const double dist[5],
                         const orc_blob_params *prm, int32_t *xy, int max_pts, uint8_t *mask_out)
{
    size_t n = (size_t)H * W;
    uint8_t *und = (uint8_t *)malloc(n), *flt = (uint8_t *)malloc(n);
    int rc = -1;
    orc_contour *cs = 0;
    if (!und || !flt) goto done;
    if (prm->undistort) {
        if (orc_undistort_u8(img, und, H, W, K, dist)) goto done;
    } else
        memcpy(und, img, n);
    if (orc_filter(und, flt, H, W, prm->filter_order, prm->ksize, prm->thresh)) goto done;
    if (mask_out) memcpy(mask_out, flt, n);
    int cap = 1 << 16;
    cs = (orc_contour *)malloc(sizeof(orc_contour) * cap);
    if (!cs) goto done;
    int nc = orc_find_contours(flt, H, W, cs, cap, 0, 0, 0);
    if (nc < 0) goto done;
    if (nc > cap) nc = cap;
    int k = 0;
    for (int i = 0; i < nc; i++) {
        orc_contour_select(&cs[i], prm->min_area, prm->min_circ);
        if (cs[i].kept) {
            if (k < max_pts) {
                xy[2 * k] = cs[i].cx;
                xy[2 * k + 1] = cs[i].cy;
            }
            k++;
        }
    }
    rc = k;
done:
    free(und);
    free(flt);
    free(cs);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * fast_cuda_demosaic (reference lib/CudaOperations.py:43-100): bilinear Bayer demosaic, pattern
 * (even y, even x)=B, (even,odd)=G, (odd,even)=G, (odd,odd)=R; out-of-range taps read 0 while
 * the divisors stay 4/2; output channel order B,G,R.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_demosaic_u8(const uint8_t *bayer, uint8_t *bgr, int H, int W)
{
#define GP(xx, yy) (((xx) >= 0 && (xx) < W && (yy) >= 0 && (yy) < H) ? (int)bayer[(size_t)(yy) * W + (xx)] : 0)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int r, g, b;
            if ((y % 2 == 0) && (x % 2 == 0)) {
                b = GP(x, y);
                g = (GP(x - 1, y) + GP(x + 1, y) + GP(x, y - 1) + GP(x, y + 1)) / 4;
                r = (GP(x - 1, y - 1) + GP(x + 1, y - 1) + GP(x - 1, y + 1) + GP(x + 1, y + 1)) / 4;
            } else if ((y % 2 == 0) && (x % 2 == 1)) {
                g = GP(x, y);
                b = (GP(x - 1, y) + GP(x + 1, y)) / 2;
                r = (GP(x, y - 1) + GP(x, y + 1)) / 2;
            } else if ((y % 2 == 1) && (x % 2 == 0)) {
                g = GP(x, y);
                r = (GP(x - 1, y) + GP(x + 1, y)) / 2;
                b = (GP(x, y - 1) + GP(x, y + 1)) / 2;
            } else {
                r = GP(x, y);
                g = (GP(x - 1, y) + GP(x + 1, y) + GP(x, y - 1) + GP(x, y + 1)) / 4;
                b = (GP(x - 1, y - 1) + GP(x + 1, y - 1) + GP(x - 1, y + 1) + GP(x + 1, y + 1)) / 4;
            }
            uint8_t *o = bgr + ((size_t)y * W + x) * 3;
            o[0] = (uint8_t)b;
            o[1] = (uint8_t)g;
            o[2] = (uint8_t)r;
        }
#undef GP
}

/* ------------------------------------------------------------------------------------------
 * Pre-path pixel steps of the tracker loop (reference RealtimeTracking_FLIR.py:103-104):
 *     gray = cv2.cvtColor(cv2.cvtColor(raw, cv2.COLOR_BAYER_GR2BGR), cv2.COLOR_BGR2GRAY)
 * PARITY UNPINNED: both calls live in OpenCV (version unpinned, absent here); this restates the published
 * behaviour of its portable code paths:
 *  - bilinear Bayer demosaic: a missing colour is the rounded mean of the nearest samples of that colour --
 *    (a + b + 1) >> 1 for the two horizontal or vertical neighbours of a green site, (a + b + c + d + 2) >> 2 for the
 *    cross / diagonal neighbours of a red or blue site; only rows 1..H-2 and columns 1..W-2 are interpolated, column
 *    0 / W-1 of every row then repeats column 1 / W-2, and rows 0 / H-1 repeat rows 1 / H-2;
 *  - pattern names give the colours at (row 1, col 1) and (row 1, col 2): GR = G B / R G rows starting with G B;
 *    pattern: 0 = BG, 1 = GB, 2 = RG, 3 = GR (cv2.COLOR_BayerBG2BGR + pattern);
 *  - BGR2GRAY, 8 bit: (B*1868 + G*9617 + R*4899 + 2^13) >> 14  (shift 14; OpenCV's R2Y/G2Y/B2Y), or the 15-bit
 *    coefficient set (3735, 19235, 9798, + 2^14, >> 15) its source names as the alternative (shift 15).
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_bayer_gray_u8(const uint8_t *bayer, uint8_t *gray, int H, int W, int pattern, int shift)
{
    if (H < 3 || W < 3 || pattern < 0 || pattern > 3 || (shift != 14 && shift != 15)) return -1;
    /* the red sites: (ry, rx) parities; blue sits at the opposite parities */
    const int ry = (pattern == 0) ? 0 : (pattern == 1) ? 0 : 1;
    const int rx = (pattern == 0) ? 0 : (pattern == 1) ? 1 : (pattern == 2) ? 1 : 0;
    const int cb = shift == 14 ? 1868 : 3735, cg = shift == 14 ? 9617 : 19235, cr = shift == 14 ? 4899 : 9798;
#define BP(yy, xx) ((int)bayer[(size_t)(yy) * W + (xx)])
    for (int y = 0; y < H; y++) {
        const int yc = y < 1 ? 1 : (y > H - 2 ? H - 2 : y);
        for (int x = 0; x < W; x++) {
            const int xc = x < 1 ? 1 : (x > W - 2 ? W - 2 : x);
            const int c = BP(yc, xc);
            const int horiz = (BP(yc, xc - 1) + BP(yc, xc + 1) + 1) >> 1;
            const int vert = (BP(yc - 1, xc) + BP(yc + 1, xc) + 1) >> 1;
            const int cross = (BP(yc, xc - 1) + BP(yc, xc + 1) + BP(yc - 1, xc) + BP(yc + 1, xc) + 2) >> 2;
            const int diag = (BP(yc - 1, xc - 1) + BP(yc - 1, xc + 1) + BP(yc + 1, xc - 1) + BP(yc + 1, xc + 1) + 2) >> 2;
            int r, g, b;
            const int on_red_row = (yc & 1) == ry, on_red_col = (xc & 1) == rx;
            if (on_red_row && on_red_col) { r = c; g = cross; b = diag; }
            else if (!on_red_row && !on_red_col) { b = c; g = cross; r = diag; }
            else if (on_red_row) { g = c; r = horiz; b = vert; }   /* green between reds */
            else { g = c; b = horiz; r = vert; }                    /* green between blues */
            gray[(size_t)y * W + x] = (uint8_t)((b * cb + g * cg + r * cr + (1 << (shift - 1))) >> shift);
        }
    }
#undef BP
    return 0;
}
