/*
 * lane_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the arithmetic that the
 * reference's lane scripts obtain from the third-party `cv2` module
 * (opencv-python, unpinned by the reference, absent from /root/reference and
 * from this image).  Contract restated: OpenCV >= 4.5.2 generic (non-IPP,
 * no-FMA) code path of modules/imgproc {color_rgb.simd.hpp, smooth.simd.hpp,
 * canny.cpp, drawing.cpp, hough.cpp, color_hsv.simd.hpp} and modules/core
 * {arithm.cpp, rng}.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or sample
 * inputs for this path and no cv2 can be imported here, so this file is
 * pinned only by hand-derivable known-answer tests (tests/golden/) and by an
 * independent numpy/scipy restatement (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the shipped HIP path never does.
 *
 * Reference call sites that fix the parameters (file:line under
 * /root/reference):
 *   cvtColor      Simple Lane Detection with OpenCV_2(images).py:69 (RGB2GRAY),
 *                 F1_vo_Lane_Tracking.py:186 (BGR2GRAY)
 *   GaussianBlur  F1_vo_Lane_Tracking.py:97   ((5,5), sigma 0)
 *   Canny         ..._2(images).py:72 (100,200), F1_vo_Lane_Tracking.py:98 (50,150)
 *   fillPoly      ..._2(images).py:14-20, F1_vo_Lane_Tracking.py:105-111,158-175
 *   rectangle     F1_vo_Lane_Tracking.py:179-180 (filled)
 *   bitwise_and   ..._2(images).py:23, F1_vo_Lane_Tracking.py:102 (mask=), :112
 *   BGR2HSV, inRange, bitwise_or, bitwise_and(mask=)  ..._6(mp4).py:65-88
 *   HoughLinesP   ..._2(images).py:84-92 (6, pi/60, 160, 40, 25),
 *                 ..._5(mp4).py:76-84    (6, pi/60, 130, 40, 150)
 *
 * Build: see oracle/Makefile (-ffp-contract=off is required: the Hough rho
 * projection must round each product and the sum separately).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

#define LO_OK       0
#define LO_EINVAL  -1
#define LO_ENOMEM  -2
#define LO_ECAP    -3   /* caller's output buffer too small */
#define LO_ERANGE  -4   /* an index OpenCV would not bounds-check left its array */

#define LO_COLOR_BGR2GRAY 6
#define LO_COLOR_RGB2GRAY 7

/* ------------------------------------------------------------------ */
/* helpers                                                            */
/* ------------------------------------------------------------------ */

/* cv::borderInterpolate for BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba). */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else       p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

static int clampi(int p, int len) { return p < 0 ? 0 : (p >= len ? len - 1 : p); }

/* cvRound(float): round to nearest, ties to even (SSE cvtss2si semantics). */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------ */
/* cvtColor -> gray, 8U.  color_rgb.simd.hpp RGB2Gray<uchar>:          */
/* 15-bit weights R 9798, G 19235, B 3735, rounding +2^14, >>15.       */
/* ------------------------------------------------------------------ */
int lo_cvt_gray(const uint8_t *src, int w, int h, long sstride, int code,
                uint8_t *dst, long dstride)
{
    int k0, k1 = 19235, k2;
    if (!src || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    if (code == LO_COLOR_BGR2GRAY)      { k0 = 3735; k2 = 9798; }
    else if (code == LO_COLOR_RGB2GRAY) { k0 = 9798; k2 = 3735; }
    else return LO_EINVAL;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (long)y * sstride;
        uint8_t *d = dst + (long)y * dstride;
        for (int x = 0; x < w; x++, s += 3)
            d[x] = (uint8_t)((s[0] * k0 + s[1] * k1 + s[2] * k2 + (1 << 14)) >> 15);
    }
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* GaussianBlur(src,(5,5),0), 8U.  sigma<=0 & ksize 5 -> fixed table    */
/* [1 4 6 4 1]/16; 8U runs in ufixedpoint16 (8 fractional bits):        */
/* hline: (p-2 + p+2 + 4(p-1 + p+1) + 6 p0) << 4   (fits 16 bits)       */
/* vline: (sum of the same taps over 5 hline rows + 2^11) >> 12         */
/* Border BORDER_REFLECT_101 on both axes.                              */
/* ------------------------------------------------------------------ */
int lo_gaussian5(const uint8_t *src, int w, int h, long sstride,
                 uint8_t *dst, long dstride)
{
    if (!src || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    uint16_t *hb = (uint16_t *)malloc((size_t)w * h * sizeof(uint16_t));
    if (!hb) return LO_ENOMEM;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (long)y * sstride;
        for (int x = 0; x < w; x++) {
            int a = s[reflect101(x - 2, w)], b = s[reflect101(x - 1, w)], c = s[x];
            int d = s[reflect101(x + 1, w)], e = s[reflect101(x + 2, w)];
            hb[(size_t)y * w + x] = (uint16_t)((a + e + 4 * (b + d) + 6 * c) << 4);
        }
    }
    for (int y = 0; y < h; y++) {
        const uint16_t *r0 = hb + (size_t)reflect101(y - 2, h) * w;
        const uint16_t *r1 = hb + (size_t)reflect101(y - 1, h) * w;
        const uint16_t *r2 = hb + (size_t)y * w;
        const uint16_t *r3 = hb + (size_t)reflect101(y + 1, h) * w;
        const uint16_t *r4 = hb + (size_t)reflect101(y + 2, h) * w;
        uint8_t *d = dst + (long)y * dstride;
        for (int x = 0; x < w; x++) {
            uint32_t v = (uint32_t)r0[x] + r4[x] + 4u * ((uint32_t)r1[x] + r3[x]) + 6u * r2[x];
            d[x] = (uint8_t)((v + (1u << 11)) >> 12);
        }
    }
    free(hb);
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* Sobel 3x3, CV_16S, BORDER_REPLICATE (as Canny calls it).            */
/* ------------------------------------------------------------------ */
int lo_sobel3(const uint8_t *src, int w, int h, long sstride,
              int16_t *dx, int16_t *dy)
{
    if (!src || !dx || !dy || w <= 0 || h <= 0) return LO_EINVAL;
    for (int y = 0; y < h; y++) {
        const uint8_t *rm = src + (long)clampi(y - 1, h) * sstride;
        const uint8_t *r0 = src + (long)y * sstride;
        const uint8_t *rp = src + (long)clampi(y + 1, h) * sstride;
        for (int x = 0; x < w; x++) {
            int xm = clampi(x - 1, w), xp = clampi(x + 1, w);
            int gx = (rm[xp] + 2 * r0[xp] + rp[xp]) - (rm[xm] + 2 * r0[xm] + rp[xm]);
            int gy = (rp[xm] + 2 * rp[x] + rp[xp]) - (rm[xm] + 2 * rm[x] + rm[xp]);
            dx[(size_t)y * w + x] = (int16_t)gx;
            dy[(size_t)y * w + x] = (int16_t)gy;
        }
    }
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* Canny(img, t1, t2), apertureSize 3, L2gradient false (canny.cpp).   */
/* cls_out (optional, w*h): the pre-hysteresis map, OpenCV's encoding:  */
/*   0 = weak candidate, 1 = not an edge, 2 = strong.                   */
/* ------------------------------------------------------------------ */
int lo_canny(const uint8_t *src, int w, int h, long sstride,
             double t1, double t2, uint8_t *dst, long dstride, uint8_t *cls_out)
{
    if (!src || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    if (t1 > t2) { double t = t1; t1 = t2; t2 = t; }
    const int low = (int)floor(t1), high = (int)floor(t2);
    const int TG22 = 13573; /* (int)(0.4142135623730950488016887242097*(1<<15) + 0.5) */
    const size_t n = (size_t)w * h;
    const int mw = w + 2; /* magnitude plane with a zero ring */
    int16_t *dx = (int16_t *)malloc(n * sizeof(int16_t));
    int16_t *dy = (int16_t *)malloc(n * sizeof(int16_t));
    int32_t *mag = (int32_t *)calloc((size_t)mw * (h + 2), sizeof(int32_t));
    uint8_t *map = (uint8_t *)malloc(n);
    int32_t *stack = (int32_t *)malloc(n * sizeof(int32_t));
    int rc = LO_OK;
    if (!dx || !dy || !mag || !map || !stack) { rc = LO_ENOMEM; goto done; }
    lo_sobel3(src, w, h, sstride, dx, dy);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            mag[(size_t)(y + 1) * mw + x + 1] = abs((int)dx[i]) + abs((int)dy[i]);
        }
    size_t sp = 0;
    for (int y = 0; y < h; y++) {
        const int32_t *mp = mag + (size_t)y * mw + 1;       /* row y-1 */
        const int32_t *ma = mag + (size_t)(y + 1) * mw + 1; /* row y   */
        const int32_t *mn = mag + (size_t)(y + 2) * mw + 1; /* row y+1 */
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            int m = ma[x];
            uint8_t c = 1;
            if (m > low) {
                int xs = dx[i], ys = dy[i];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * TG22;
                int keep = 0;
                if (ay < tg22x) {
                    keep = (m > ma[x - 1] && m >= ma[x + 1]);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > mp[x] && m >= mn[x]);
                    } else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > mp[x - s] && m > mn[x + s]);
                    }
                }
                if (keep) {
                    if (m > high) { c = 2; stack[sp++] = (int32_t)i; }
                    else c = 0;
                }
            }
            map[i] = c;
        }
    }
    if (cls_out) memcpy(cls_out, map, n);
    while (sp) {
        int32_t i = stack[--sp];
        int y = i / w, x = i % w;
        for (int yy = y - 1; yy <= y + 1; yy++) {
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int xx = x - 1; xx <= x + 1; xx++) {
                if ((unsigned)xx >= (unsigned)w) continue;
                size_t j = (size_t)yy * w + xx;
                if (map[j] == 0) { map[j] = 2; stack[sp++] = (int32_t)j; }
            }
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[(long)y * dstride + x] = (uint8_t)-(map[(size_t)y * w + x] >> 1);
done:
    free(dx); free(dy); free(mag); free(map); free(stack);
    return rc;
}

/* ------------------------------------------------------------------ */
/* bitwise_and(a, b[, mask]) on 8UC1 (arithm.cpp): dst = a & b where    */
/* mask != 0 (or everywhere without a mask); a freshly allocated dst    */
/* is zero where mask == 0.                                            */
/* ------------------------------------------------------------------ */
int lo_and(const uint8_t *a, long astride, const uint8_t *b, long bstride,
           const uint8_t *mask, long mstride, uint8_t *dst, long dstride, int w, int h)
{
    if (!a || !b || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t v = a[(long)y * astride + x] & b[(long)y * bstride + x];
            if (mask && !mask[(long)y * mstride + x]) v = 0;
            dst[(long)y * dstride + x] = v;
        }
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* drawing.cpp: clipLine, LineIterator/Line (8-connected),              */
/* CollectPolyEdges + FillEdgeCollection (fillPoly, LINE_8, shift 0).   */
/* ------------------------------------------------------------------ */
typedef struct { int64_t x, y; } pt64;

static int clip_line(int64_t width, int64_t height, pt64 *p1, pt64 *p2)
{
    int c1, c2;
    int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    int64_t x1 = p1->x, y1 = p1->y, x2 = p2->x, y2 = p2->y;
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (int64_t)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (int64_t)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (int64_t)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (int64_t)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
        p1->x = x1; p1->y = y1; p2->x = x2; p2->y = y2;
    }
    return (c1 | c2) == 0;
}

/* Line(img, pt1, pt2, color, 8): LineIterator(connectivity 8, leftToRight). */
static void draw_line8(uint8_t *img, int w, int h, long stride, pt64 a, pt64 b, uint8_t color)
{
    if ((uint64_t)a.x >= (uint64_t)w || (uint64_t)b.x >= (uint64_t)w ||
        (uint64_t)a.y >= (uint64_t)h || (uint64_t)b.y >= (uint64_t)h) {
        if (!clip_line(w, h, &a, &b)) return;
    }
    int delta_x = 1, delta_y = 1;
    int dx = (int)(b.x - a.x), dy = (int)(b.y - a.y);
    if (dx < 0) { dx = -dx; dy = -dy; a = b; }   /* leftToRight */
    if (dy < 0) { dy = -dy; delta_y = -1; }
    int vert = dy > dx;
    if (vert) { int t = dx; dx = dy; dy = t; t = delta_x; delta_x = delta_y; delta_y = t; }
    /* after the optional swap: delta_x steps the major axis, delta_y the minor */
    int err = dx - (dy + dy);
    const int plus_delta = dx + dx, minus_delta = -(dy + dy);
    int count = dx + 1;
    int px = (int)a.x, py = (int)a.y;
    for (int i = 0; i < count; i++) {
        img[(long)py * stride + px] = color;
        int neg = err < 0;
        err += minus_delta + (neg ? plus_delta : 0);
        if (!vert) { px += delta_x; if (neg) py += delta_y; }
        else       { py += delta_x; if (neg) px += delta_y; }
    }
}

typedef struct poly_edge {
    int y0, y1;
    int64_t x, dx;
    struct poly_edge *next;
} poly_edge;

#define XY_SHIFT 16
#define XY_ONE   (1 << XY_SHIFT)

static int cmp_edges(const void *pa, const void *pb)
{
    const poly_edge *a = (const poly_edge *)pa, *b = (const poly_edge *)pb;
    if (a->y0 != b->y0) return a->y0 < b->y0 ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->dx != b->dx) return a->dx < b->dx ? -1 : 1;
    return 0;
}

static void hline(uint8_t *row, int x1, int x2, uint8_t color)
{
    if (x2 >= x1) memset(row + x1, color, (size_t)(x2 - x1 + 1));
}

/* pts: concatenated (x,y) int32 pairs of all contours; npts[i] vertices each. */
int lo_fill_poly(uint8_t *img, int w, int h, long stride,
                 const int32_t *pts, const int *npts, int ncontours, int color)
{
    if (!img || !pts || !npts || ncontours < 0 || w <= 0 || h <= 0) return LO_EINVAL;
    const uint8_t col = (uint8_t)(color < 0 ? 0 : (color > 255 ? 255 : color));
    int total = 0;
    for (int i = 0; i < ncontours; i++) { if (npts[i] < 0) return LO_EINVAL; total += npts[i]; }
    poly_edge *edges = (poly_edge *)malloc(((size_t)total + 1) * sizeof(poly_edge));
    if (!edges) return LO_ENOMEM;
    int ne = 0;

    /* CollectPolyEdges, once per contour */
    const int32_t *v = pts;
    for (int c = 0; c < ncontours; v += 2 * npts[c], c++) {
        int count = npts[c];
        if (count == 0) continue;
        pt64 pt0 = { (int64_t)v[2 * (count - 1)] << XY_SHIFT, v[2 * (count - 1) + 1] }, pt1;
        for (int i = 0; i < count; i++, pt0 = pt1) {
            pt64 t0, t1, pt0c, pt1c;
            pt1.x = (int64_t)v[2 * i] << XY_SHIFT;
            pt1.y = v[2 * i + 1];
            pt0c = pt0; pt1c = pt1;
            t0.y = pt0.y; t1.y = pt1.y;
            t0.x = (pt0.x + (XY_ONE >> 1)) >> XY_SHIFT;
            t1.x = (pt1.x + (XY_ONE >> 1)) >> XY_SHIFT;
            draw_line8(img, w, h, stride, t0, t1, col);
            /* use the clipped end points for a more accurate edge */
            if ((uint64_t)t0.x >= (uint64_t)w || (uint64_t)t1.x >= (uint64_t)w ||
                (uint64_t)t0.y >= (uint64_t)h || (uint64_t)t1.y >= (uint64_t)h) {
                clip_line(w, h, &t0, &t1);
                if (t0.y != t1.y) {
                    pt0c.y = t0.y; pt1c.y = t1.y;
                    pt0c.x = t0.x << XY_SHIFT;
                    pt1c.x = t1.x << XY_SHIFT;
                }
            } else {
                pt0c.x += XY_ONE >> 1;
                pt1c.x += XY_ONE >> 1;
            }
            if (pt0.y == pt1.y) continue;
            poly_edge e;
            e.dx = (pt1c.x - pt0c.x) / (pt1c.y - pt0c.y);
            if (pt0.y < pt1.y) {
                e.y0 = (int)pt0.y; e.y1 = (int)pt1.y;
                e.x = pt0c.x + (pt0.y - pt0c.y) * e.dx;
            } else {
                e.y0 = (int)pt1.y; e.y1 = (int)pt0.y;
                e.x = pt1c.x + (pt1.y - pt1c.y) * e.dx;
            }
            e.next = 0;
            edges[ne++] = e;
        }
    }

    /* FillEdgeCollection (line_type < LINE_AA => delta = 0) */
    total = ne;
    if (total < 2) { free(edges); return LO_OK; }
    int y_max = INT_MIN, y_min = INT_MAX;
    int64_t x_max = -1, x_min = INT64_MAX;
    for (int i = 0; i < total; i++) {
        poly_edge *e1 = &edges[i];
        int64_t x1 = e1->x + (int64_t)(e1->y1 - e1->y0) * e1->dx;
        if (e1->y0 < y_min) y_min = e1->y0;
        if (e1->y1 > y_max) y_max = e1->y1;
        if (e1->x < x_min) x_min = e1->x;
        if (e1->x > x_max) x_max = e1->x;
        if (x1 < x_min) x_min = x1;
        if (x1 > x_max) x_max = x1;
    }
    if (y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((int64_t)w << XY_SHIFT)) {
        free(edges); return LO_OK;
    }
    qsort(edges, (size_t)total, sizeof(poly_edge), cmp_edges);
    poly_edge tmp;
    tmp.y0 = INT_MAX; tmp.next = 0;
    edges[total] = tmp; /* sentinel */
    int i = 0;
    poly_edge *e = &edges[0];
    if (y_max > h) y_max = h;
    for (int y = e->y0; y < y_max; y++) {
        poly_edge *last, *prelast, *keep_prelast;
        int sort_flag = 0, draw = 0, clipline = y < 0;
        prelast = &tmp;
        last = tmp.next;
        while (last || e->y0 == y) {
            if (last && last->y1 == y) {
                prelast->next = last->next;
                last = last->next;
                continue;
            }
            keep_prelast = prelast;
            if (last && (e->y0 > y || last->x < e->x)) {
                prelast = last;
                last = last->next;
            } else if (i < total) {
                prelast->next = e;
                e->next = last;
                prelast = e;
                e = &edges[++i];
            } else
                break;
            if (draw) {
                if (!clipline) {
                    uint8_t *row = img + (long)y * stride;
                    int x1, x2;
                    if (keep_prelast->x > prelast->x) {
                        x1 = (int)(prelast->x >> XY_SHIFT);
                        x2 = (int)(keep_prelast->x >> XY_SHIFT);
                    } else {
                        x1 = (int)(keep_prelast->x >> XY_SHIFT);
                        x2 = (int)(prelast->x >> XY_SHIFT);
                    }
                    if (x1 < w && x2 >= 0) {
                        if (x1 < 0) x1 = 0;
                        if (x2 >= w) x2 = w - 1;
                        hline(row, x1, x2, col);
                    }
                }
                keep_prelast->x += keep_prelast->dx;
                prelast->x += prelast->dx;
            }
            draw ^= 1;
        }
        /* bubble-sort the active list by x */
        keep_prelast = 0;
        do {
            prelast = &tmp;
            last = tmp.next;
            while (last != keep_prelast && last && last->next != 0) {
                poly_edge *te = last->next;
                if (last->x > te->x) {
                    prelast->next = te;
                    last->next = te->next;
                    te->next = last;
                    prelast = te;
                    sort_flag = 1;
                } else {
                    prelast = last;
                    last = te;
                }
            }
            keep_prelast = prelast;
        } while (sort_flag && keep_prelast != tmp.next && keep_prelast != &tmp);
    }
    free(edges);
    return LO_OK;
}

/* rectangle(img, pt1, pt2, color, thickness=-1): FillConvexPoly of the
 * axis-aligned 4-gon = the clipped rectangle with INCLUSIVE corners. */
int lo_rectangle_filled(uint8_t *img, int w, int h, long stride,
                        int x1, int y1, int x2, int y2, int color)
{
    if (!img || w <= 0 || h <= 0) return LO_EINVAL;
    const uint8_t col = (uint8_t)(color < 0 ? 0 : (color > 255 ? 255 : color));
    if (x1 > x2) { int t = x1; x1 = x2; x2 = t; }
    if (y1 > y2) { int t = y1; y1 = y2; y2 = t; }
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    if (x2 > w - 1) x2 = w - 1;
    if (y2 > h - 1) y2 = h - 1;
    for (int y = y1; y <= y2; y++) hline(img + (long)y * stride, x1, x2, col);
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* drawing.cpp: line / polylines with a thickness (ThickLine, Line2,    */
/* FillConvexPoly, Circle) on 8-bit images of cn channels, LINE_8,      */
/* shift 0.  Reference call sites: Simple Lane Detection with           */
/* OpenCV_2(images).py:27-51 (cv2.line, thickness 5 as called at :146), */
/* ..._5(mp4).py:21-52 and F1_vo_Lane_Tracking.py:137-140               */
/* (cv2.polylines, thickness 8 / 5).                                    */
/* Restated from memory of OpenCV 4.5+, NOT checked against a cv2       */
/* (parity unpinned).  By this code a thick line is the convex quad     */
/* p +- dp with |dp| = thickness / 2 (+ 1/2 for odd thicknesses), filled */
/* AND outlined, plus a filled circle of radius (thickness + 1) / 2     */
/* (truncated) at the ends that have a cap: a horizontal line of        */
/* thickness 5 covers 7 rows, thickness 8 covers 9 rows.                */
/* ------------------------------------------------------------------ */
static void put_px(uint8_t *img, int w, int h, long stride, int cn, int64_t x, int64_t y, const uint8_t *col)
{
    if (x >= 0 && x < w && y >= 0 && y < h) memcpy(img + y * stride + x * cn, col, (size_t)cn);
}

/* ICV_HLINE: x1 .. x2 inclusive, already inside the image */
static void hline_cn(uint8_t *img, long stride, int cn, int y, int x1, int x2, const uint8_t *col)
{
    for (int x = x1; x <= x2; x++) memcpy(img + (long)y * stride + (long)x * cn, col, (size_t)cn);
}

/* Line(img, pt1, pt2, color, 8) for any channel count (the thin case of ThickLine) */
static void draw_line8_cn(uint8_t *img, int w, int h, long stride, int cn, pt64 a, pt64 b, const uint8_t *col)
{
    if ((uint64_t)a.x >= (uint64_t)w || (uint64_t)b.x >= (uint64_t)w ||
        (uint64_t)a.y >= (uint64_t)h || (uint64_t)b.y >= (uint64_t)h) {
        if (!clip_line(w, h, &a, &b)) return;
    }
    int delta_x = 1, delta_y = 1;
    int dx = (int)(b.x - a.x), dy = (int)(b.y - a.y);
    if (dx < 0) { dx = -dx; dy = -dy; a = b; }
    if (dy < 0) { dy = -dy; delta_y = -1; }
    int vert = dy > dx;
    if (vert) { int t = dx; dx = dy; dy = t; t = delta_x; delta_x = delta_y; delta_y = t; }
    int err = dx - (dy + dy);
    const int plus_delta = dx + dx, minus_delta = -(dy + dy);
    int count = dx + 1;
    int px = (int)a.x, py = (int)a.y;
    for (int i = 0; i < count; i++) {
        memcpy(img + (long)py * stride + (long)px * cn, col, (size_t)cn);
        int neg = err < 0;
        err += minus_delta + (neg ? plus_delta : 0);
        if (!vert) { px += delta_x; if (neg) py += delta_y; }
        else       { py += delta_x; if (neg) px += delta_y; }
    }
}

/* Line2: a line between 16.16 fixed-point end points, one pixel per step of the major axis */
static void line2(uint8_t *img, int w, int h, long stride, int cn, pt64 pt1, pt64 pt2, const uint8_t *col)
{
    if (!clip_line((int64_t)w << XY_SHIFT, (int64_t)h << XY_SHIFT, &pt1, &pt2)) return;
    int64_t dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
    const int64_t j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
    const int64_t i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
    int64_t x_step, y_step;
    int ecount;
    if (ax > ay) {
        dy = (dy ^ j) - j;
        if (j) { pt64 t = pt1; pt1 = pt2; pt2 = t; }
        x_step = XY_ONE;
        y_step = dy * (1 << XY_SHIFT) / (ax | 1);
        ecount = (int)((pt2.x - pt1.x) >> XY_SHIFT);
    } else {
        dx = (dx ^ i) - i;
        if (i) { pt64 t = pt1; pt1 = pt2; pt2 = t; }
        x_step = dx * (1 << XY_SHIFT) / (ay | 1);
        y_step = XY_ONE;
        ecount = (int)((pt2.y - pt1.y) >> XY_SHIFT);
    }
    pt1.x += XY_ONE >> 1;
    pt1.y += XY_ONE >> 1;
    put_px(img, w, h, stride, cn, (pt2.x + (XY_ONE >> 1)) >> XY_SHIFT, (pt2.y + (XY_ONE >> 1)) >> XY_SHIFT, col);
    if (ax > ay) {
        pt1.x >>= XY_SHIFT;
        while (ecount >= 0) {
            put_px(img, w, h, stride, cn, pt1.x, pt1.y >> XY_SHIFT, col);
            pt1.x++;
            pt1.y += y_step;
            ecount--;
        }
    } else {
        pt1.y >>= XY_SHIFT;
        while (ecount >= 0) {
            put_px(img, w, h, stride, cn, pt1.x >> XY_SHIFT, pt1.y, col);
            pt1.x += x_step;
            pt1.y++;
            ecount--;
        }
    }
    (void)x_step; (void)y_step;
}

/* FillConvexPoly(img, v, npts, color, LINE_8, shift = XY_SHIFT): outline by Line2, then two edge chains walked
 * down from the top vertex, one span per row */
static void fill_convex_poly16(uint8_t *img, int w, int h, long stride, int cn, const pt64 *v, int npts, const uint8_t *col)
{
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    const int shift = XY_SHIFT;
    const int64_t delta = 1 << shift >> 1;
    int i, y, imin = 0;
    int edges = npts;
    int64_t xmin, xmax, ymin, ymax;
    const int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
    pt64 p0 = v[npts - 1];
    xmin = xmax = v[0].x;
    ymin = ymax = v[0].y;
    for (i = 0; i < npts; i++) {
        pt64 p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        if (p.y > ymax) ymax = p.y;
        if (p.x > xmax) xmax = p.x;
        if (p.x < xmin) xmin = p.x;
        line2(img, w, h, stride, cn, p0, p, col);
        p0 = p;
    }
    xmin = (xmin + delta) >> shift;
    xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift;
    ymax = (ymax + delta) >> shift;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    if (ymax > h - 1) ymax = h - 1;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y = (int)ymin;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)((v[idx].y + delta) >> shift);
                    if (ty > y) {
                        const int64_t xs = v[idx0].x, xe = v[idx].x;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < w) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= w) xx2 = w - 1;
                hline_cn(img, stride, cn, y, xx1, xx2, col);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

/* Circle(img, center, radius, color, fill = 1): the midpoint circle, every step four horizontal spans */
static void circle_filled(uint8_t *img, int w, int h, long stride, int cn, int cx, int cy, int radius, const uint8_t *col)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        int mask;
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        if (x11 < w && x12 >= 0 && y21 < h && y22 >= 0) {
            if (x11 < 0) x11 = 0;
            if (x12 > w - 1) x12 = w - 1;
            if ((unsigned)y11 < (unsigned)h) hline_cn(img, stride, cn, y11, x11, x12, col);
            if ((unsigned)y12 < (unsigned)h) hline_cn(img, stride, cn, y12, x11, x12, col);
            if (x21 < w && x22 >= 0) {
                if (x21 < 0) x21 = 0;
                if (x22 > w - 1) x22 = w - 1;
                if ((unsigned)y21 < (unsigned)h) hline_cn(img, stride, cn, y21, x21, x22, col);
                if ((unsigned)y22 < (unsigned)h) hline_cn(img, stride, cn, y22, x21, x22, col);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* ThickLine(img, p0, p1, color, thickness, LINE_8, flags, shift = 0): flags bit 0 / 1 = round cap at p0 / p1 */
static void thick_line(uint8_t *img, int w, int h, long stride, int cn, pt64 p0, pt64 p1, const uint8_t *col,
                       int thickness, int flags)
{
    p0.x <<= XY_SHIFT; p0.y <<= XY_SHIFT; p1.x <<= XY_SHIFT; p1.y <<= XY_SHIFT;
    if (thickness <= 1) {
        pt64 a = {(p0.x + (XY_ONE >> 1)) >> XY_SHIFT, (p0.y + (XY_ONE >> 1)) >> XY_SHIFT};
        pt64 b = {(p1.x + (XY_ONE >> 1)) >> XY_SHIFT, (p1.y + (XY_ONE >> 1)) >> XY_SHIFT};
        draw_line8_cn(img, w, h, stride, cn, a, b, col);
        return;
    }
    const double INV_XY_ONE = 1. / XY_ONE;
    pt64 pt[4], dp = {0, 0};
    const double dx = (double)(p0.x - p1.x) * INV_XY_ONE, dy = (double)(p1.y - p0.y) * INV_XY_ONE;
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    thickness <<= XY_SHIFT - 1;
    if (fabs(r) > 2.220446049250313e-16) {
        r = (thickness + odd * XY_ONE * 0.5) / sqrt(r);
        dp.x = cv_round_d(dy * r);
        dp.y = cv_round_d(dx * r);
        pt[0].x = p0.x + dp.x; pt[0].y = p0.y + dp.y;
        pt[1].x = p0.x - dp.x; pt[1].y = p0.y - dp.y;
        pt[2].x = p1.x - dp.x; pt[2].y = p1.y - dp.y;
        pt[3].x = p1.x + dp.x; pt[3].y = p1.y + dp.y;
        fill_convex_poly16(img, w, h, stride, cn, pt, 4, col);
    }
    for (int i = 0; i < 2; i++) {
        if (flags & (i + 1)) {
            const int cx = (int)((p0.x + (XY_ONE >> 1)) >> XY_SHIFT), cy = (int)((p0.y + (XY_ONE >> 1)) >> XY_SHIFT);
            circle_filled(img, w, h, stride, cn, cx, cy, (thickness + (XY_ONE >> 1)) >> XY_SHIFT, col);
        }
        p0 = p1;
    }
}

static int color_bytes(const int *color, int cn, uint8_t *col)
{
    if (!color || cn < 1 || cn > 4) return LO_EINVAL;
    for (int k = 0; k < cn; k++) col[k] = (uint8_t)(color[k] < 0 ? 0 : (color[k] > 255 ? 255 : color[k]));
    return LO_OK;
}

/* cv2.line(img, pt1, pt2, color, thickness) */
int lo_line(uint8_t *img, int w, int h, long stride, int cn, int x1, int y1, int x2, int y2, const int *color, int thickness)
{
    uint8_t col[4];
    if (!img || w <= 0 || h <= 0 || thickness < 1 || color_bytes(color, cn, col)) return LO_EINVAL;
    pt64 a = {x1, y1}, b = {x2, y2};
    thick_line(img, w, h, stride, cn, a, b, col, thickness, 3);
    return LO_OK;
}

/* cv2.polylines(img, pts, isClosed, color, thickness): PolyLine per contour */
int lo_polylines(uint8_t *img, int w, int h, long stride, int cn, const int32_t *pts, const int *npts, int ncontours,
                 int is_closed, const int *color, int thickness)
{
    uint8_t col[4];
    if (!img || w <= 0 || h <= 0 || thickness < 1 || ncontours < 0 || (ncontours && (!pts || !npts)) || color_bytes(color, cn, col))
        return LO_EINVAL;
    const int32_t *v = pts;
    for (int c = 0; c < ncontours; v += 2 * npts[c], c++) {
        const int count = npts[c];
        if (count <= 0) continue;
        int i = is_closed ? count - 1 : 0;
        int flags = 2 + !is_closed;
        pt64 p0 = {v[2 * i], v[2 * i + 1]};
        for (i = !is_closed; i < count; i++) {
            pt64 p = {v[2 * i], v[2 * i + 1]};
            thick_line(img, w, h, stride, cn, p0, p, col, thickness, flags);
            p0 = p;
            flags = 2;
        }
    }
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* hough.cpp: HoughLinesProbabilistic and its trig table.              */
/* ------------------------------------------------------------------ */
static int compute_numangle(double min_theta, double max_theta, double theta_step)
{
    int numangle = (int)floor((max_theta - min_theta) / theta_step) + 1;
    if (numangle > 1 && fabs(M_PI - (numangle - 1) * theta_step) < theta_step / 2)
        --numangle;
    return numangle;
}

/* Fills tab[2n] = (float)(cos(n*theta)/rho), tab[2n+1] = (float)(sin(n*theta)/rho)
 * exactly as HoughLinesProbabilistic does; rho and theta are narrowed to float
 * first, as the HoughLinesP wrapper does.  tab may be NULL to query sizes. */
int lo_hough_geometry(int w, int h, double rho_d, double theta_d,
                      int *numangle, int *numrho, float *tab, int tab_cap)
{
    if (w <= 0 || h <= 0 || !(rho_d > 0) || !(theta_d > 0)) return LO_EINVAL;
    const float rho = (float)rho_d, theta = (float)theta_d;
    const float irho = 1 / rho;
    int na = compute_numangle(0.0, M_PI, theta);
    int nr = cv_round_f(((w + h) * 2 + 1) / rho);
    if (numangle) *numangle = na;
    if (numrho) *numrho = nr;
    if (tab) {
        if (tab_cap < 2 * na) return LO_ECAP;
        for (int n = 0; n < na; n++) {
            tab[2 * n]     = (float)(cos((double)n * theta) * irho);
            tab[2 * n + 1] = (float)(sin((double)n * theta) * irho);
        }
    }
    return LO_OK;
}

/* cv::RNG (multiply-with-carry) as HoughLinesProbabilistic seeds it. */
typedef struct { uint64_t state; } lo_rng;
static uint32_t rng_next(lo_rng *r)
{
    r->state = (uint64_t)(uint32_t)r->state * 4164903690u + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}
/* first `n` outputs of RNG((uint64)-1), for known-answer tests */
void lo_rng_stream(uint32_t *out, int n)
{
    lo_rng r = { 0xFFFFFFFFFFFFFFFFull };
    for (int i = 0; i < n; i++) out[i] = rng_next(&r);
}

/* Full-vote accumulator: every non-zero pixel votes once in each of the
 * numangle rows, with HoughLinesProbabilistic's table and bin index, and no
 * vote is ever removed.  acc must hold numangle*numrho int32 (zeroed here). */
int lo_hough_accum(const uint8_t *img, int w, int h, long stride,
                   double rho, double theta, int32_t *acc, int acc_cap,
                   int *numangle_out, int *numrho_out)
{
    int na, nr;
    if (!img || !acc) return LO_EINVAL;
    int rc = lo_hough_geometry(w, h, rho, theta, &na, &nr, 0, 0);
    if (rc) return rc;
    if (acc_cap < na * nr) return LO_ECAP;
    float *tab = (float *)malloc(sizeof(float) * 2 * na);
    if (!tab) return LO_ENOMEM;
    lo_hough_geometry(w, h, rho, theta, 0, 0, tab, 2 * na);
    memset(acc, 0, sizeof(int32_t) * (size_t)na * nr);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            if (!img[(long)i * stride + j]) continue;
            for (int n = 0; n < na; n++) {
                int r = cv_round_f(j * tab[2 * n] + i * tab[2 * n + 1]) + (nr - 1) / 2;
                if ((unsigned)r >= (unsigned)nr) { free(tab); return LO_ERANGE; }
                acc[(size_t)n * nr + r]++;
            }
        }
    free(tab);
    if (numangle_out) *numangle_out = na;
    if (numrho_out) *numrho_out = nr;
    return LO_OK;
}

/* HoughLinesP(image, rho, theta, threshold, minLineLength, maxLineGap).
 * out: cap x 4 int32 (x1,y1,x2,y2) in detection order; *n_out = segments found
 * (if more than cap are found the call returns LO_ECAP with *n_out = total).
 * acc_final (optional): the accumulator as the call leaves it. */
int lo_hough_lines_p(const uint8_t *image, int width, int height, long stride,
                     double rho_d, double theta_d, int threshold,
                     double min_len_d, double max_gap_d,
                     int32_t *out, int cap, int *n_out,
                     int32_t *acc_final, int acc_cap)
{
    int numangle, numrho;
    if (!image || !n_out || (cap > 0 && !out)) return LO_EINVAL;
    int rc = lo_hough_geometry(width, height, rho_d, theta_d, &numangle, &numrho, 0, 0);
    if (rc) return rc;
    const int lineLength = cv_round_d(min_len_d), lineGap = cv_round_d(max_gap_d);
    int32_t *accum = (int32_t *)calloc((size_t)numangle * numrho, sizeof(int32_t));
    uint8_t *mask = (uint8_t *)malloc((size_t)width * height);
    float *ttab = (float *)malloc(sizeof(float) * 2 * numangle);
    int32_t *nz = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)width * height);
    int nlines = 0;
    if (!accum || !mask || !ttab || !nz) { rc = LO_ENOMEM; goto done; }
    lo_hough_geometry(width, height, rho_d, theta_d, 0, 0, ttab, 2 * numangle);

    /* stage 1: collect non-zero points, row-major */
    int count = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            if (image[(long)y * stride + x]) {
                mask[(size_t)y * width + x] = 1;
                nz[2 * count] = x; nz[2 * count + 1] = y; count++;
            } else
                mask[(size_t)y * width + x] = 0;
        }

    /* stage 2: process the points in random order */
    lo_rng rng = { 0xFFFFFFFFFFFFFFFFull };
    const int shift = 16;
    for (; count > 0; count--) {
        int idx = (int)(rng_next(&rng) % (uint32_t)count);
        int max_val = threshold - 1, max_n = 0;
        int j = nz[2 * idx], i = nz[2 * idx + 1];
        int line_end[2][2] = { { 0, 0 }, { 0, 0 } }; /* [k] = {x, y} */
        int k, x0, y0, dx0, dy0, xflag, good_line;
        float a, b;
        nz[2 * idx] = nz[2 * (count - 1)];
        nz[2 * idx + 1] = nz[2 * (count - 1) + 1];
        if (!mask[(size_t)i * width + j]) continue;

        for (int n = 0; n < numangle; n++) {
            int r = cv_round_f(j * ttab[n * 2] + i * ttab[n * 2 + 1]);
            r += (numrho - 1) / 2;
            if ((unsigned)r >= (unsigned)numrho) { rc = LO_ERANGE; goto done; }
            int val = ++accum[(size_t)n * numrho + r];
            if (max_val < val) { max_val = val; max_n = n; }
        }
        if (max_val < threshold) continue;

        a = -ttab[max_n * 2 + 1];
        b = ttab[max_n * 2];
        x0 = j; y0 = i;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = cv_round_f(b * (1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = cv_round_f(a * (1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        for (k = 0; k < 2; k++) {
            int gap = 0, x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) { dx = -dx; dy = -dy; }
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) { j1 = x; i1 = y >> shift; }
                else       { j1 = x >> shift; i1 = y; }
                if (j1 < 0 || j1 >= width || i1 < 0 || i1 >= height) break;
                if (mask[(size_t)i1 * width + j1]) {
                    gap = 0;
                    line_end[k][1] = i1;
                    line_end[k][0] = j1;
                } else if (++gap > lineGap)
                    break;
            }
        }
        good_line = abs(line_end[1][0] - line_end[0][0]) >= lineLength ||
                    abs(line_end[1][1] - line_end[0][1]) >= lineLength;
        for (k = 0; k < 2; k++) {
            int x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) { dx = -dx; dy = -dy; }
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) { j1 = x; i1 = y >> shift; }
                else       { j1 = x >> shift; i1 = y; }
                uint8_t *m = mask + (size_t)i1 * width + j1;
                if (*m) {
                    if (good_line) {
                        for (int n = 0; n < numangle; n++) {
                            int r = cv_round_f(j1 * ttab[n * 2] + i1 * ttab[n * 2 + 1]);
                            r += (numrho - 1) / 2;
                            accum[(size_t)n * numrho + r]--;
                        }
                    }
                    *m = 0;
                }
                if (i1 == line_end[k][1] && j1 == line_end[k][0]) break;
            }
        }
        if (good_line) {
            if (nlines < cap) {
                out[4 * nlines + 0] = line_end[0][0];
                out[4 * nlines + 1] = line_end[0][1];
                out[4 * nlines + 2] = line_end[1][0];
                out[4 * nlines + 3] = line_end[1][1];
            }
            nlines++;
        }
    }
    if (acc_final) {
        if (acc_cap < numangle * numrho) rc = LO_ECAP;
        else memcpy(acc_final, accum, sizeof(int32_t) * (size_t)numangle * numrho);
    }
    if (nlines > cap && rc == LO_OK) rc = LO_ECAP;
done:
    *n_out = nlines;
    free(accum); free(mask); free(ttab); free(nz);
    return rc;
}

/* ------------------------------------------------------------------ */
/* Colour gate of the HSV-gated lane scripts                           */
/* (Simple Lane Detection with OpenCV_6(mp4).py:65-88):                 */
/*   cvtColor(BGR2HSV) -> inRange x2 -> bitwise_or ->                   */
/*   bitwise_and(image, image, mask=) -> cvtColor(BGR2GRAY)             */
/*                                                                      */
/* cvtColor -> HSV, 8U, hue range 180: color_hsv.simd.hpp RGB2HSV_b.    */
/* Fixed point with hsv_shift = 12 and two reciprocal tables            */
/*   sdiv[i] = cvRound((255 << 12) / (double)i),                        */
/*   hdiv[i] = cvRound((180 << 12) / (6. * i)),   sdiv[0] = hdiv[0] = 0 */
/*   v = max(b,g,r), diff = v - min(b,g,r)                              */
/*   s = (diff * sdiv[v] + 2^11) >> 12                                  */
/*   h = v==r ? g-b : v==g ? b-r+2diff : r-g+4diff                      */
/*   h = (h * hdiv[diff] + 2^11) >> 12;  h += 180 if h < 0              */
/* ------------------------------------------------------------------ */
#define LO_COLOR_BGR2HSV 40
#define LO_COLOR_RGB2HSV 41

static int lo_sdiv[256], lo_hdiv180[256], lo_hsv_ready;
static void lo_hsv_tables(void)
{
    if (lo_hsv_ready) return;
    lo_sdiv[0] = lo_hdiv180[0] = 0;
    for (int i = 1; i < 256; i++) {
        lo_sdiv[i] = cv_round_d((255 << 12) / (1. * i));
        lo_hdiv180[i] = cv_round_d((180 << 12) / (6. * i));
    }
    lo_hsv_ready = 1;
}

static void lo_hsv_px(int b, int g, int r, int *ho, int *so, int *vo)
{
    int v = b, vmin = b;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    const int diff = v - vmin;
    const int s = (diff * lo_sdiv[v] + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * lo_hdiv180[diff] + (1 << 11)) >> 12;      /* arithmetic shift of a negative h, as in C++ */
    if (h < 0) h += 180;
    *ho = h > 255 ? 255 : h; *so = s; *vo = v;
}

int lo_cvt_hsv(const uint8_t *src, int w, int h, long sstride, int code,
               uint8_t *dst, long dstride)
{
    if (!src || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    if (code != LO_COLOR_BGR2HSV && code != LO_COLOR_RGB2HSV) return LO_EINVAL;
    lo_hsv_tables();
    const int bi = code == LO_COLOR_BGR2HSV ? 0 : 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = src + y * sstride + 3 * x;
            int hh, ss, vv;
            lo_hsv_px(p[bi], p[1], p[bi ^ 2], &hh, &ss, &vv);
            uint8_t *q = dst + y * dstride + 3 * x;
            q[0] = (uint8_t)hh; q[1] = (uint8_t)ss; q[2] = (uint8_t)vv;
        }
    return LO_OK;
}

/* cv::inRange on 8UC3 with per-channel integer bounds (arithm.cpp):   */
/* 255 where lo[c] <= src[c] <= hi[c] for all three channels, else 0.  */
int lo_in_range3(const uint8_t *src, int w, int h, long sstride,
                 const int *lo, const int *hi, uint8_t *dst, long dstride)
{
    if (!src || !dst || !lo || !hi || w <= 0 || h <= 0) return LO_EINVAL;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = src + y * sstride + 3 * x;
            int ok = 1;
            for (int c = 0; c < 3; c++) ok &= (p[c] >= lo[c] && p[c] <= hi[c]);
            dst[y * dstride + x] = ok ? 255 : 0;
        }
    return LO_OK;
}

int lo_or(const uint8_t *a, long astride, const uint8_t *b, long bstride,
          uint8_t *dst, long dstride, int w, int h)
{
    if (!a || !b || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[y * dstride + x] = a[y * astride + x] | b[y * bstride + x];
    return LO_OK;
}

/* bitwise_and(src, src, mask=mask) on 8UC3 into a fresh (zero) dst.    */
int lo_and3_mask(const uint8_t *src, long sstride, const uint8_t *mask, long mstride,
                 uint8_t *dst, long dstride, int w, int h)
{
    if (!src || !mask || !dst || w <= 0 || h <= 0) return LO_EINVAL;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++)
                dst[y * dstride + 3 * x + c] = mask[y * mstride + x] ? src[y * sstride + 3 * x + c] : 0;
    return LO_OK;
}

/* The whole gate, call by call: gate = lo0[3], hi0[3], lo1[3], hi1[3] in  */
/* H, S, V order; gray_out = BGR2GRAY of the masked image.                */
int lo_gate_gray(const uint8_t *bgr, int w, int h, const int *gate, int gray_code, uint8_t *gray_out)
{
    if (!bgr || !gate || !gray_out || w <= 0 || h <= 0) return LO_EINVAL;
    size_t n = (size_t)w * h;
    uint8_t *hsv = (uint8_t *)malloc(3 * n), *m0 = (uint8_t *)malloc(n), *m1 = (uint8_t *)malloc(n);
    uint8_t *mk = (uint8_t *)malloc(3 * n);
    int rc = LO_ENOMEM;
    if (!hsv || !m0 || !m1 || !mk) goto done;
    if ((rc = lo_cvt_hsv(bgr, w, h, 3L * w, LO_COLOR_BGR2HSV, hsv, 3L * w))) goto done;
    if ((rc = lo_in_range3(hsv, w, h, 3L * w, gate, gate + 3, m0, w))) goto done;
    if ((rc = lo_in_range3(hsv, w, h, 3L * w, gate + 6, gate + 9, m1, w))) goto done;
    if ((rc = lo_or(m0, w, m1, w, m0, w, w, h))) goto done;
    if ((rc = lo_and3_mask(bgr, 3L * w, m0, w, mk, 3L * w, w, h))) goto done;
    rc = lo_cvt_gray(mk, w, h, 3L * w, gray_code, gray_out, w);
done:
    free(hsv); free(m0); free(m1); free(mk);
    return rc;
}

/* ------------------------------------------------------------------ */
/* cv::resize(src, dsize) with INTER_LINEAR on 8U, 1 or 3 channels       */
/* (imgproc/src/resize.cpp, generic path, no IPP):                       */
/*   scale = 1 / ((double)dsize / ssize)                                 */
/*   exact 2x2 decimation is rerouted to INTER_AREA (fast):              */
/*       D = (S00 + S01 + S10 + S11 + 2) >> 2                            */
/*   otherwise fixed point with INTER_RESIZE_COEF_BITS = 11:             */
/*       fx = (float)((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx */
/*       sx < 0 -> (sx, fx) = (0, 0); sx >= sw - 1 -> (sw - 1, 0)        */
/*       alpha = (short)cvRound((1.f - fx) * 2048), (short)cvRound(fx * 2048) */
/*       rows: fy, sy likewise but the two source rows are clipped to    */
/*       [0, sh - 1] instead of changing the weights                     */
/*       horizontal: H = S[sx] * a0 + S[sx + 1] * a1  (int)              */
/*       vertical:   D = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2 */
/* ------------------------------------------------------------------ */
static int lo_clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static short lo_sat_short(float v) { int r = cv_round_f(v); return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r)); }

int lo_resize_linear(const uint8_t *src, int sw, int sh, long sstride, int cn,
                     uint8_t *dst, int dw, int dh, long dstride)
{
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || (cn != 1 && cn != 3)) return LO_EINVAL;
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    const int area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (area_fast && iscale_x == 2 && iscale_y == 2) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const uint8_t *s0 = src + (2 * y) * sstride + (2 * x) * cn + c, *s1 = s0 + sstride;
                    dst[y * dstride + x * cn + c] = (uint8_t)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
                }
        return LO_OK;
    }
    int *xofs = (int *)malloc(sizeof(int) * dw), *yofs = (int *)malloc(sizeof(int) * dh);
    short *ia = (short *)malloc(sizeof(short) * 2 * dw), *ib = (short *)malloc(sizeof(short) * 2 * dh);
    if (!xofs || !yofs || !ia || !ib) { free(xofs); free(yofs); free(ia); free(ib); return LO_ENOMEM; }
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ia[2 * dx] = lo_sat_short((1.f - fx) * 2048);
        ia[2 * dx + 1] = lo_sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        ib[2 * dy] = lo_sat_short((1.f - fy) * 2048);
        ib[2 * dy + 1] = lo_sat_short(fy * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        const uint8_t *r0 = src + lo_clipi(yofs[dy], 0, sh - 1) * sstride;
        const uint8_t *r1 = src + lo_clipi(yofs[dy] + 1, 0, sh - 1) * sstride;
        const int b0 = ib[2 * dy], b1 = ib[2 * dy + 1];
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;   /* weight of sx1 is 0 when it would fall outside */
            const int a0 = ia[2 * dx], a1 = ia[2 * dx + 1];
            for (int c = 0; c < cn; c++) {
                const int h0 = r0[sx * cn + c] * a0 + r0[sx1 * cn + c] * a1;
                const int h1 = r1[sx * cn + c] * a0 + r1[sx1 * cn + c] * a1;
                dst[dy * dstride + dx * cn + c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xofs); free(yofs); free(ia); free(ib);
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* cv::addWeighted on 8-bit data (arithm): the weights are narrowed to  */
/* float, dst = saturate_cast<uchar>(a * alpha + (b * beta + gamma)),    */
/* i.e. cvRound (ties to even) and clamping to [0, 255].  Each product   */
/* and sum is rounded separately here; a build that fuses them can       */
/* differ only where the exact value lies within a float ulp of a .5     */
/* tie, which cannot happen for the scripts' weight pairs (0.8, 1.0),    */
/* (1.0, 0.8) and (1.0, 0.6): n * 4/5 and n * 3/5 are never k + 1/2.     */
/* ------------------------------------------------------------------ */
int lo_add_weighted(const uint8_t *a, double alpha, const uint8_t *b, double beta, double gamma,
                    uint8_t *dst, size_t n)
{
    if (!a || !b || !dst) return LO_EINVAL;
    const float fa = (float)alpha, fb = (float)beta, fg = (float)gamma;
    for (size_t i = 0; i < n; i++) {
        const float t = (float)a[i] * fa + ((float)b[i] * fb + fg);
        int v = t >= 255.5f ? 255 : (t <= -0.5f ? 0 : cv_round_f(t));
        dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    return LO_OK;
}

/* ------------------------------------------------------------------ */
/* One frame through a recipe (the reference's call sequence), for the  */
/* parity tests and the CPU baseline timing.                           */
/*   code      LO_COLOR_*                                              */
/*   blur      0/1: GaussianBlur 5x5 between gray and Canny             */
/*   roi       w*h mask (0/255) ANDed onto the edge map, or NULL        */
/*   car_mask  w*h mask applied first as bitwise_and(e,e,mask=), or NULL */
/* Outputs (each optional): gray, blurred, edges (pre-ROI), masked.     */
/* ------------------------------------------------------------------ */
int lo_lane_frame_gated(const uint8_t *bgr, int w, int h, int code, const int *gate, int blur,
                        double t1, double t2, const uint8_t *car_mask, const uint8_t *roi,
                        double rho, double theta, int threshold, double min_len, double max_gap,
                        uint8_t *gray_out, uint8_t *blur_out, uint8_t *edges_out, uint8_t *masked_out,
                        int32_t *acc_out, int acc_cap, int32_t *seg_out, int seg_cap, int *nseg_out);

int lo_lane_frame(const uint8_t *bgr, int w, int h, int code, int blur,
                  double t1, double t2, const uint8_t *car_mask, const uint8_t *roi,
                  double rho, double theta, int threshold, double min_len, double max_gap,
                  uint8_t *gray_out, uint8_t *blur_out, uint8_t *edges_out, uint8_t *masked_out,
                  int32_t *acc_out, int acc_cap, int32_t *seg_out, int seg_cap, int *nseg_out)
{
    return lo_lane_frame_gated(bgr, w, h, code, 0, blur, t1, t2, car_mask, roi, rho, theta, threshold, min_len, max_gap,
                               gray_out, blur_out, edges_out, masked_out, acc_out, acc_cap, seg_out, seg_cap, nseg_out);
}

/* gate: NULL, or the 12 bounds of lo_gate_gray (the colour gate in front of the gray conversion) */
int lo_lane_frame_gated(const uint8_t *bgr, int w, int h, int code, const int *gate, int blur,
                        double t1, double t2, const uint8_t *car_mask, const uint8_t *roi,
                        double rho, double theta, int threshold, double min_len, double max_gap,
                        uint8_t *gray_out, uint8_t *blur_out, uint8_t *edges_out, uint8_t *masked_out,
                        int32_t *acc_out, int acc_cap, int32_t *seg_out, int seg_cap, int *nseg_out)
{
    size_t n = (size_t)w * h;
    uint8_t *gray = (uint8_t *)malloc(n), *bl = (uint8_t *)malloc(n);
    uint8_t *ed = (uint8_t *)malloc(n), *mk = (uint8_t *)malloc(n);
    int rc = LO_ENOMEM;
    if (!gray || !bl || !ed || !mk) goto done;
    if (gate) { if ((rc = lo_gate_gray(bgr, w, h, gate, code, gray))) goto done; }
    else if ((rc = lo_cvt_gray(bgr, w, h, 3L * w, code, gray, w))) goto done;
    if (blur) { if ((rc = lo_gaussian5(gray, w, h, w, bl, w))) goto done; }
    else memcpy(bl, gray, n);
    if ((rc = lo_canny(bl, w, h, w, t1, t2, ed, w, 0))) goto done;
    memcpy(mk, ed, n);
    if (car_mask) lo_and(mk, w, mk, w, car_mask, w, mk, w, w, h);
    if (roi) lo_and(mk, w, roi, w, 0, 0, mk, w, w, h);
    if (gray_out) memcpy(gray_out, gray, n);
    if (blur_out) memcpy(blur_out, bl, n);
    if (edges_out) memcpy(edges_out, ed, n);
    if (masked_out) memcpy(masked_out, mk, n);
    if (acc_out) { if ((rc = lo_hough_accum(mk, w, h, w, rho, theta, acc_out, acc_cap, 0, 0))) goto done; }
    if (nseg_out)
        rc = lo_hough_lines_p(mk, w, h, w, rho, theta, threshold, min_len, max_gap,
                              seg_out, seg_cap, nseg_out, 0, 0);
done:
    free(gray); free(bl); free(ed); free(mk);
    return rc;
}
