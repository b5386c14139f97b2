// fillPoly / filled rectangle painting for gfx950.  The host prepares OpenCV's
// polygon edges (16.16 fixed point, clipped end points) and clipped outline
// segments; the device paints them with closed forms instead of OpenCV's
// incremental loops:
//   edge x at row y      = x0 + (y - y0) * dx               (FillEdgeCollection)
//   Bresenham minor step = floor((2*dmin*i + dmaj - 1) / (2*dmaj))   (LineIterator, 8-connected)
#include "common.h"

namespace lanecv {
namespace {

constexpr int MAX_EDGES = 1024;

// one block per image row: thread 0 orders the active edges' x, all threads paint the spans
__global__ __launch_bounds__(256) void k_fill_spans(uint8_t *__restrict__ img, int W, int H,
                                                    const PolyEdgeDev *__restrict__ edges, int nedges, int ymin, int color) {
    __shared__ long long xs[MAX_EDGES];
    __shared__ int nact;
    const int y = ymin + blockIdx.x;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int e = 0; e < nedges; e++)
            if (edges[e].y0 <= y && y < edges[e].y1) {
                long long x = edges[e].x + (long long)(y - edges[e].y0) * edges[e].dx;
                int k = n++;
                while (k > 0 && xs[k - 1] > x) { xs[k] = xs[k - 1]; k--; }   // insertion sort
                xs[k] = x;
            }
        nact = n;
    }
    __syncthreads();
    if (y < 0 || y >= H) return;
    uint8_t *row = img + (size_t)y * W;
    for (int k = 0; k + 1 < nact; k += 2) {
        long long x1 = xs[k] >> 16, x2 = xs[k + 1] >> 16;
        if (x1 >= W || x2 < 0) continue;
        if (x1 < 0) x1 = 0;
        if (x2 >= W) x2 = W - 1;
        for (int x = (int)x1 + threadIdx.x; x <= (int)x2; x += 256) row[x] = (uint8_t)color;
    }
}

// one thread per outline pixel
__global__ __launch_bounds__(256) void k_draw_lines(uint8_t *__restrict__ img, int W, int H, const LineDev *__restrict__ lines,
                                                    int color) {
    const LineDev L = lines[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.count) return;
    long long minor = L.dmajor_len > 0 ? (2ll * L.dminor_len * i + L.dmajor_len - 1) / (2ll * L.dmajor_len) : 0;
    int px, py;
    if (!L.vert) { px = L.x0 + L.dmaj * i; py = L.y0 + L.dmin * (int)minor; }
    else         { py = L.y0 + L.dmaj * i; px = L.x0 + L.dmin * (int)minor; }
    if ((unsigned)px < (unsigned)W && (unsigned)py < (unsigned)H) img[(size_t)py * W + px] = (uint8_t)color;
}

__global__ __launch_bounds__(256) void k_fill_rect(uint8_t *__restrict__ img, int W, int x1, int y1, int x2, int y2, int color) {
    const int y = y1 + blockIdx.y;
    const int x = x1 + blockIdx.x * 256 + threadIdx.x;
    if (y <= y2 && x <= x2) img[(size_t)y * W + x] = (uint8_t)color;
}

// ---- thick lines (cv2.line / cv2.polylines with a thickness) on images of cn channels ----------------------------
// The host walks OpenCV's ThickLine set-up and hands over closed-form pieces; every pixel of every piece gets the
// one colour of the call, so overlaps need no order.
__device__ __forceinline__ void put_color(uint8_t *__restrict__ img, int W, int cn, int x, int y, uint32_t color) {
    uint8_t *p = img + ((size_t)y * W + x) * cn;
    for (int k = 0; k < cn; k++) p[k] = (uint8_t)(color >> (8 * k));
}

// FillConvexPoly between two events of its edge walk: rows [y0, y1), the two chains' x at y0 and their slopes
__global__ __launch_bounds__(256) void k_paint_traps(uint8_t *__restrict__ img, int W, int H, int cn,
                                                     const TrapDev *__restrict__ traps, uint32_t color) {
    const TrapDev t = traps[blockIdx.x];
    for (int y = t.y0; y < t.y1; y++) {
        if (y < 0 || y >= H) continue;
        const long long xa = t.xa + (long long)(y - t.y0) * t.dxa, xb = t.xb + (long long)(y - t.y0) * t.dxb;
        const long long xl = xa > xb ? xb : xa, xr = xa > xb ? xa : xb;      // the walk's left / right swap
        long long x1 = (xl + 32768) >> 16, x2 = (xr + 32768) >> 16;
        if (x2 < 0 || x1 >= W) continue;
        if (x1 < 0) x1 = 0;
        if (x2 >= W) x2 = W - 1;
        for (int x = (int)x1 + threadIdx.x; x <= (int)x2; x += 256) put_color(img, W, cn, x, y, color);
    }
}

// Line2: one pixel per step of the major axis, the minor coordinate in 16.16 fixed point; step -1 is the extra end point
__global__ __launch_bounds__(256) void k_paint_line2(uint8_t *__restrict__ img, int W, int H, int cn,
                                                     const Line2Dev *__restrict__ lines, uint32_t color) {
    const Line2Dev L = lines[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x - 1;
    if (i > L.ecount) return;
    long long x, y;
    if (i < 0) { x = L.endx; y = L.endy; }
    else if (L.xmajor) { x = L.major0 + i; y = (L.minor0 + (long long)i * L.step) >> 16; }
    else { y = L.major0 + i; x = (L.minor0 + (long long)i * L.step) >> 16; }
    if (x >= 0 && x < W && y >= 0 && y < H) put_color(img, W, cn, (int)x, (int)y, color);
}

// filled circles of one radius: row j of the circle spans cx - hw[j] .. cx + hw[j]
__global__ __launch_bounds__(64) void k_paint_circles(uint8_t *__restrict__ img, int W, int H, int cn,
                                                      const int2 *__restrict__ centres, const int *__restrict__ hw, int radius,
                                                      uint32_t color) {
    const int2 c = centres[blockIdx.x];
    const int j = blockIdx.y, y = c.y - radius + j;
    if (y < 0 || y >= H || hw[j] < 0) return;
    const int x1 = max(c.x - hw[j], 0), x2 = min(c.x + hw[j], W - 1);
    for (int x = x1 + threadIdx.x; x <= x2; x += 64) put_color(img, W, cn, x, y, color);
}

// LineIterator lines on cn channels (thickness 1)
__global__ __launch_bounds__(256) void k_draw_lines_cn(uint8_t *__restrict__ img, int W, int H, int cn,
                                                       const LineDev *__restrict__ lines, uint32_t color) {
    const LineDev L = lines[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L.count) return;
    long long minor = L.dmajor_len > 0 ? (2ll * L.dminor_len * i + L.dmajor_len - 1) / (2ll * L.dmajor_len) : 0;
    int px, py;
    if (!L.vert) { px = L.x0 + L.dmaj * i; py = L.y0 + L.dmin * (int)minor; }
    else         { py = L.y0 + L.dmaj * i; px = L.x0 + L.dmin * (int)minor; }
    if ((unsigned)px < (unsigned)W && (unsigned)py < (unsigned)H) put_color(img, W, cn, px, py, color);
}

}  // namespace

void launch_paint_traps(uint8_t *img, int W, int H, int cn, const TrapDev *traps, int ntraps, uint32_t color, hipStream_t st) {
    if (ntraps > 0) hipLaunchKernelGGL(k_paint_traps, dim3(ntraps), dim3(256), 0, st, img, W, H, cn, traps, color);
}
void launch_paint_line2(uint8_t *img, int W, int H, int cn, const Line2Dev *lines, int nlines, int maxcount, uint32_t color,
                        hipStream_t st) {
    if (nlines > 0)
        hipLaunchKernelGGL(k_paint_line2, dim3((maxcount + 2 + 255) / 256, nlines), dim3(256), 0, st, img, W, H, cn, lines, color);
}
void launch_paint_circles(uint8_t *img, int W, int H, int cn, const int2 *centres, int ncircles, const int *hw, int radius,
                          uint32_t color, hipStream_t st) {
    if (ncircles > 0)
        hipLaunchKernelGGL(k_paint_circles, dim3(ncircles, 2 * radius + 1), dim3(64), 0, st, img, W, H, cn, centres, hw, radius, color);
}
void launch_draw_lines_cn(uint8_t *img, int W, int H, int cn, const LineDev *lines, int nlines, uint32_t color, hipStream_t st) {
    if (nlines <= 0) return;
    int maxlen = (W > H ? W : H);
    hipLaunchKernelGGL(k_draw_lines_cn, dim3((maxlen + 255) / 256, nlines), dim3(256), 0, st, img, W, H, cn, lines, color);
}

void launch_fill_spans(uint8_t *img, int W, int H, const PolyEdgeDev *edges, int nedges, int ymin, int ymax, int color,
                       hipStream_t st) {
    if (ymax <= ymin || nedges < 2) return;
    hipLaunchKernelGGL(k_fill_spans, dim3(ymax - ymin), dim3(256), 0, st, img, W, H, edges, nedges, ymin, color);
}

void launch_draw_lines(uint8_t *img, int W, int H, const LineDev *lines, int nlines, int color, hipStream_t st) {
    if (nlines <= 0) return;
    int maxlen = (W > H ? W : H);
    hipLaunchKernelGGL(k_draw_lines, dim3((maxlen + 255) / 256, nlines), dim3(256), 0, st, img, W, H, lines, color);
}

void launch_fill_rect(uint8_t *img, int W, int H, int x1, int y1, int x2, int y2, int color, hipStream_t st) {
    (void)H;
    if (x2 < x1 || y2 < y1) return;
    hipLaunchKernelGGL(k_fill_rect, dim3((x2 - x1 + 256) / 256, y2 - y1 + 1), dim3(256), 0, st, img, W, x1, y1, x2, y2, color);
}

}  // namespace lanecv
