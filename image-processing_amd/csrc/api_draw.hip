// The drawing calls of the C ABI: fillPoly, filled rectangle, thick line and polylines.  The host walks the set-up of
// OpenCV's drawing.cpp (clipLine, LineIterator, CollectPolyEdges, ThickLine, FillConvexPoly, Line2, Circle) and hands
// closed-form pieces to the painters in kernels_poly.hip.
#include "host.h"

#include <climits>

using namespace lanecv;

namespace {

// ---- polygon set-up on the host (drawing.cpp: clipLine, LineIterator, CollectPolyEdges) ----
struct P64 { long long x, y; };

bool clip_line(long long width, long long height, P64 &p1, P64 &p2) {
    const long long right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return false;
    long long &x1 = p1.x, &y1 = p1.y, &x2 = p2.x, &y2 = p2.y;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (x2 - x1) / (y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (x2 - x1) / (y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (y2 - y1) / (x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (y2 - y1) / (x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

bool inside(const P64 &p, int w, int h) {
    return (unsigned long long)p.x < (unsigned long long)w && (unsigned long long)p.y < (unsigned long long)h;
}

// the clipped, left-to-right 8-connected segment LineIterator would walk
bool make_line(int w, int h, P64 a, P64 b, LineDev &L) {
    if (!inside(a, w, h) || !inside(b, w, h)) {
        if (!clip_line(w, h, a, b)) return false;
    }
    int dx = (int)(b.x - a.x), dy = (int)(b.y - a.y);
    int sx = 1, sy = 1;
    if (dx < 0) { dx = -dx; dy = -dy; a = b; }
    if (dy < 0) { dy = -dy; sy = -1; }
    L.x0 = (int)a.x; L.y0 = (int)a.y;
    L.vert = dy > dx;
    if (L.vert) { L.dmaj = sy; L.dmin = sx; L.dmajor_len = dy; L.dminor_len = dx; }
    else        { L.dmaj = sx; L.dmin = sy; L.dmajor_len = dx; L.dminor_len = dy; }
    L.count = L.dmajor_len + 1;
    return true;
}

// ---- thick lines: drawing.cpp's ThickLine / FillConvexPoly / Line2 / Circle set-up on the host -----------------
// What one cv2.line / cv2.polylines call paints, as closed-form pieces for the device (kernels_poly.hip).
struct ThickPieces {
    std::vector<TrapDev> traps;
    std::vector<Line2Dev> lines2;
    std::vector<LineDev> thin;
    std::vector<int2> circles;
    int max_count = 0;          // longest Line2, in steps
    int radius = 0;             // of every cap circle of the call
};

// Line2 up to the point where its loop starts: clip to the image in 16.16 fixed point, order the end points along
// the major axis, fixed-point step of the minor one
void add_line2(int w, int h, P64 pt1, P64 pt2, ThickPieces &out) {
    if (!clip_line((long long)w << 16, (long long)h << 16, pt1, pt2)) return;
    long long dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
    const long long j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
    const long long i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
    Line2Dev L{};
    if (ax > ay) {
        dy = (dy ^ j) - j;
        if (j) std::swap(pt1, pt2);
        L.xmajor = 1;
        L.step = dy * (1 << 16) / (ax | 1);
        L.ecount = (int)((pt2.x - pt1.x) >> 16);
    } else {
        dx = (dx ^ i) - i;
        if (i) std::swap(pt1, pt2);
        L.xmajor = 0;
        L.step = dx * (1 << 16) / (ay | 1);
        L.ecount = (int)((pt2.y - pt1.y) >> 16);
    }
    pt1.x += 32768;
    pt1.y += 32768;
    L.endx = (pt2.x + 32768) >> 16;
    L.endy = (pt2.y + 32768) >> 16;
    if (L.xmajor) { L.major0 = pt1.x >> 16; L.minor0 = pt1.y; }
    else          { L.major0 = pt1.y >> 16; L.minor0 = pt1.x; }
    out.max_count = std::max(out.max_count, L.ecount);
    out.lines2.push_back(L);
}

// FillConvexPoly(v, npts, LINE_8, shift 16): the outline as Line2 pieces, the fill as the trapezoids between the
// events of its two-chain edge walk (the walk itself jumps from event to event instead of row to row)
void add_convex_poly(int w, int h, const P64 *v, int npts, ThickPieces &out) {
    struct { int idx, di; long long x, dx; int ye; } edge[2];
    const long long delta = 32768;
    int imin = 0, edges = npts;
    long long xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    P64 p0 = v[npts - 1];
    for (int i = 0; i < npts; i++) {
        const P64 p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        ymax = std::max(ymax, p.y);
        xmax = std::max(xmax, p.x);
        xmin = std::min(xmin, p.x);
        add_line2(w, h, p0, p, out);
        p0 = p;
    }
    xmin = (xmin + delta) >> 16; xmax = (xmax + delta) >> 16;
    ymin = (ymin + delta) >> 16; ymax = (ymax + delta) >> 16;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    ymax = std::min<long long>(ymax, h - 1);
    int y = (int)ymin;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1; edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -65536;
    edge[0].dx = edge[1].dx = 0;
    while (y <= (int)ymax) {
        for (int i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    const int ty = (int)((v[idx].y + delta) >> 16);
                    if (ty > y) {
                        const long long xs = v[idx0].x, xe = v[idx].x;
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
        // nothing changes until the next row at which a chain reaches the end of its edge
        const int y_next = std::max(y + 1, std::min(std::min(edge[0].ye, edge[1].ye), (int)ymax + 1));
        TrapDev t{y, y_next, edge[0].x, edge[0].dx, edge[1].x, edge[1].dx};
        out.traps.push_back(t);
        edge[0].x += edge[0].dx * (y_next - y);
        edge[1].x += edge[1].dx * (y_next - y);
        y = y_next;
    }
}

// ThickLine(p0, p1, thickness, LINE_8, flags, shift 0): flags bit 0 / 1 = cap at p0 / p1
void add_thick_line(int w, int h, P64 p0, P64 p1, int thickness, int flags, ThickPieces &out) {
    if (thickness <= 1) {
        LineDev L;
        if (make_line(w, h, p0, p1, L)) out.thin.push_back(L);
        return;
    }
    p0.x <<= 16; p0.y <<= 16; p1.x <<= 16; p1.y <<= 16;
    const double dx = (double)(p0.x - p1.x) * (1. / 65536), dy = (double)(p1.y - p0.y) * (1. / 65536);
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    thickness <<= 15;
    if (std::fabs(r) > 2.220446049250313e-16) {
        r = (thickness + odd * 65536 * 0.5) / std::sqrt(r);
        const long long dpx = lrint(dy * r), dpy = lrint(dx * r);
        const P64 pt[4] = {{p0.x + dpx, p0.y + dpy}, {p0.x - dpx, p0.y - dpy}, {p1.x - dpx, p1.y - dpy}, {p1.x + dpx, p1.y + dpy}};
        add_convex_poly(w, h, pt, 4, out);
    }
    out.radius = (thickness + 32768) >> 16;
    for (int i = 0; i < 2; i++) {
        if (flags & (i + 1)) out.circles.push_back(int2{(int)((p0.x + 32768) >> 16), (int)((p0.y + 32768) >> 16)});
        p0 = p1;
    }
}

// Circle(radius, fill): half-width of every row -radius .. radius (the midpoint walk emits, per step, spans of half-width
// dx on rows +-dy and of half-width dy on rows +-dx; spans of one row share the centre, so the widest is their union)
std::vector<int> circle_half_widths(int radius) {
    std::vector<int> hw(2 * radius + 1, -1);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hw[radius - dy] = std::max(hw[radius - dy], dx); hw[radius + dy] = std::max(hw[radius + dy], dx);
        hw[radius - dx] = std::max(hw[radius - dx], dy); hw[radius + dx] = std::max(hw[radius + dx], dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
    return hw;
}

template <typename T>
int upload_vec(DevBuf<T> &d, const std::vector<T> &v, hipStream_t st) {
    int rc;
    if ((rc = d.alloc(v.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return LC_OK;
}

// paints the pieces into a host image of cn channels, in place
int paint_thick(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int cn, const int *color, const ThickPieces &tp) {
    int rc;
    uint32_t col = 0;
    for (int k = 0; k < cn; k++) col |= (uint32_t)std::min(255, std::max(0, color[k])) << (8 * k);
    const size_t row_bytes = (size_t)w * cn, n = row_bytes * h;
    if (ctx->rs_dst.n < n && (rc = ctx->rs_dst.alloc(n))) return rc;
    if ((rc = up2d(ctx->rs_dst.p, img, row_bytes, h, stride, ctx->stream))) return rc;
    DevBuf<TrapDev> dt;
    DevBuf<Line2Dev> dl2;
    DevBuf<LineDev> dl;
    DevBuf<int2> dc;
    DevBuf<int> dhw;
    if (!tp.traps.empty()) {
        if ((rc = upload_vec(dt, tp.traps, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.traps.size(); o += 32768)
            launch_paint_traps(ctx->rs_dst.p, w, h, cn, dt.p + o, (int)std::min<size_t>(32768, tp.traps.size() - o), col, ctx->stream);
    }
    if (!tp.lines2.empty()) {
        if ((rc = upload_vec(dl2, tp.lines2, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.lines2.size(); o += 32768)
            launch_paint_line2(ctx->rs_dst.p, w, h, cn, dl2.p + o, (int)std::min<size_t>(32768, tp.lines2.size() - o), tp.max_count, col,
                               ctx->stream);
    }
    if (!tp.thin.empty()) {
        if ((rc = upload_vec(dl, tp.thin, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.thin.size(); o += 32768)
            launch_draw_lines_cn(ctx->rs_dst.p, w, h, cn, dl.p + o, (int)std::min<size_t>(32768, tp.thin.size() - o), col, ctx->stream);
    }
    if (!tp.circles.empty()) {
        const std::vector<int> hw = circle_half_widths(tp.radius);
        if ((rc = upload_vec(dc, tp.circles, ctx->stream)) || (rc = upload_vec(dhw, hw, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.circles.size(); o += 32768)
            launch_paint_circles(ctx->rs_dst.p, w, h, cn, dc.p + o, (int)std::min<size_t>(32768, tp.circles.size() - o), dhw.p, tp.radius,
                                 col, ctx->stream);
    }
    return down2d(img, stride, ctx->rs_dst.p, row_bytes, h, ctx->stream);   // synchronises: the piece buffers may go
}

int check_draw(const void *img, int w, int h, size_t stride, int cn, const int *color, int thickness, const char *what) {
    int rc;
    if (cn < 1 || cn > 4) return fail(LC_EUNSUPPORTED, "%s: images of 1 to 4 channels are supported, got %d", what, cn);
    if ((rc = check_image(img, w, h, stride, (size_t)w * cn, what))) return rc;
    if (!color) return fail(LC_EINVAL, "%s: null colour", what);
    if (thickness < 1) return fail(LC_EINVAL, "%s: thickness must be positive (filled shapes are not lines), got %d", what, thickness);
    if (thickness > 255) return fail(LC_EUNSUPPORTED, "%s: thickness above 255 is not supported", what);
    return LC_OK;
}

}  // namespace

extern "C" {

int lc_fill_poly(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, const int32_t *pts, const int *npts, int ncontours,
                 int color) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(img, w, h, stride, (size_t)w, "fillPoly img"))) return rc;
    if (ncontours < 0 || (ncontours > 0 && (!pts || !npts))) return fail(LC_EINVAL, "fillPoly: bad contour arguments");
    color = color < 0 ? 0 : (color > 255 ? 255 : color);
    std::vector<PolyEdgeDev> edges;
    std::vector<LineDev> lines;
    const int32_t *v = pts;
    for (int c = 0; c < ncontours; v += 2 * (size_t)npts[c], c++) {
        const int count = npts[c];
        if (count < 0) return fail(LC_EINVAL, "fillPoly: negative vertex count");
        if (count == 0) continue;
        P64 pt0 = {(long long)v[2 * (count - 1)] << 16, v[2 * (count - 1) + 1]}, pt1;
        for (int i = 0; i < count; i++, pt0 = pt1) {
            pt1.x = (long long)v[2 * i] << 16;
            pt1.y = v[2 * i + 1];
            P64 pt0c = pt0, pt1c = pt1;
            P64 t0 = {(pt0.x + 32768) >> 16, pt0.y}, t1 = {(pt1.x + 32768) >> 16, pt1.y};
            LineDev L;
            if (make_line(w, h, t0, t1, L)) lines.push_back(L);
            if (!inside(t0, w, h) || !inside(t1, w, h)) {
                clip_line(w, h, t0, t1);
                if (t0.y != t1.y) {
                    pt0c.y = t0.y; pt1c.y = t1.y;
                    pt0c.x = t0.x << 16; pt1c.x = t1.x << 16;
                }
            } else {
                pt0c.x += 32768;
                pt1c.x += 32768;
            }
            if (pt0.y == pt1.y) continue;
            PolyEdgeDev e;
            e.dx = (pt1c.x - pt0c.x) / (pt1c.y - pt0c.y);
            if (pt0.y < pt1.y) { e.y0 = (int)pt0.y; e.y1 = (int)pt1.y; e.x = pt0c.x + (pt0.y - pt0c.y) * e.dx; }
            else               { e.y0 = (int)pt1.y; e.y1 = (int)pt0.y; e.x = pt1c.x + (pt1.y - pt1c.y) * e.dx; }
            edges.push_back(e);
        }
    }
    if (edges.size() > 1024) return fail(LC_EUNSUPPORTED, "fillPoly: more than 1024 non-horizontal edges");
    // FillEdgeCollection's early outs
    bool spans = edges.size() >= 2;
    int y_min = 0, y_max = 0;
    if (spans) {
        long long x_max = -1, x_min = INT64_MAX;
        y_max = INT32_MIN; y_min = INT32_MAX;
        for (const PolyEdgeDev &e : edges) {
            long long x1 = e.x + (long long)(e.y1 - e.y0) * e.dx;
            y_min = std::min(y_min, e.y0); y_max = std::max(y_max, e.y1);
            x_min = std::min(x_min, std::min(e.x, x1)); x_max = std::max(x_max, std::max(e.x, x1));
        }
        if (y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((long long)w << 16)) spans = false;
        y_min = std::max(y_min, 0);
        y_max = std::min(y_max, h);
    }
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, img, w, h, stride, ctx->stream))) return rc;
    DevBuf<PolyEdgeDev> de;
    DevBuf<LineDev> dl;
    if (!lines.empty()) {
        if ((rc = dl.alloc(lines.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(dl.p, lines.data(), lines.size() * sizeof(LineDev), hipMemcpyHostToDevice, ctx->stream));
        launch_draw_lines(ctx->tmpA.p, w, h, dl.p, (int)lines.size(), color, ctx->stream);
    }
    if (spans) {
        if ((rc = de.alloc(edges.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(de.p, edges.data(), edges.size() * sizeof(PolyEdgeDev), hipMemcpyHostToDevice, ctx->stream));
        launch_fill_spans(ctx->tmpA.p, w, h, de.p, (int)edges.size(), y_min, y_max, color, ctx->stream);
    }
    return down2d(img, stride, ctx->tmpA.p, w, h, ctx->stream);
}

int lc_rectangle_filled(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int x1, int y1, int x2, int y2, int color) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(img, w, h, stride, (size_t)w, "rectangle img"))) return rc;
    color = color < 0 ? 0 : (color > 255 ? 255 : color);
    if (x1 > x2) std::swap(x1, x2);
    if (y1 > y2) std::swap(y1, y2);
    x1 = std::max(x1, 0); y1 = std::max(y1, 0);
    x2 = std::min(x2, w - 1); y2 = std::min(y2, h - 1);
    if (x2 < x1 || y2 < y1) return LC_OK;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, img, w, h, stride, ctx->stream))) return rc;
    launch_fill_rect(ctx->tmpA.p, w, h, x1, y1, x2, y2, color, ctx->stream);
    return down2d(img, stride, ctx->tmpA.p, w, h, ctx->stream);
}

int lc_line(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int channels, int x1, int y1, int x2, int y2,
            const int *color, int thickness) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_draw(img, w, h, stride, channels, color, thickness, "line"))) return rc;
    ThickPieces tp;
    add_thick_line(w, h, P64{x1, y1}, P64{x2, y2}, thickness, 3, tp);
    return paint_thick(ctx, img, w, h, stride, channels, color, tp);
}

int lc_polylines(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int channels, const int32_t *pts, const int *npts,
                 int ncontours, int is_closed, const int *color, int thickness) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_draw(img, w, h, stride, channels, color, thickness, "polylines"))) return rc;
    if (ncontours < 0 || (ncontours > 0 && (!pts || !npts))) return fail(LC_EINVAL, "polylines: bad contour arguments");
    ThickPieces tp;
    const int32_t *v = pts;
    for (int c = 0; c < ncontours; v += 2 * (size_t)npts[c], c++) {
        const int count = npts[c];
        if (count < 0) return fail(LC_EINVAL, "polylines: negative vertex count");
        if (count == 0) continue;
        int i = is_closed ? count - 1 : 0;
        int flags = 2 + !is_closed;                 // PolyLine: an open polyline caps its first point too
        P64 p0{v[2 * i], v[2 * i + 1]};
        for (i = !is_closed; i < count; i++) {
            const P64 p{v[2 * i], v[2 * i + 1]};
            add_thick_line(w, h, p0, p, thickness, flags, tp);
            p0 = p;
            flags = 2;
        }
    }
    return paint_thick(ctx, img, w, h, stride, channels, color, tp);
}

}  // extern "C"
