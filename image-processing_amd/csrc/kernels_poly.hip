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

}  // namespace

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
