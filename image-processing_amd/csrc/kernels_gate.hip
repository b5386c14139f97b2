// Colour gate of the HSV-gated lane script: cvtColor(BGR2HSV), inRange, bitwise_or and the masked
// bitwise_and as stand-alone kernels (the per-op entry points), and the gate folded into a gray
// conversion (the pre-pass of frames the streaming front kernel cannot take).
//
// bitwise_and(image, image, mask=m) followed by BGR2GRAY equals "gray of the pixel where m is set,
// else 0" (the gray of a black pixel is (2^14) >> 15 = 0), so the fused forms never build the masked
// colour image.
#include "gate.h"
#include <cmath>

namespace lanecv {
namespace {

__global__ __launch_bounds__(256) void k_hsv(const uint8_t *__restrict__ src, size_t npix, int swap_rb,
                                             const int *__restrict__ tab, uint8_t *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint8_t *p = src + 3 * i;
    int h, s, v;
    hsv_px(p[swap_rb ? 2 : 0], p[1], p[swap_rb ? 0 : 2], tab, h, s, v);
    uint8_t *q = dst + 3 * i;
    q[0] = (uint8_t)h; q[1] = (uint8_t)s; q[2] = (uint8_t)v;
}

__global__ __launch_bounds__(256) void k_in_range3(const uint8_t *__restrict__ src, size_t npix, GateDev box,
                                                   uint8_t *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint8_t *p = src + 3 * i;
    dst[i] = gate_pass(p[0], p[1], p[2], box) ? 255 : 0;
}

__global__ __launch_bounds__(256) void k_or(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                            uint8_t *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a[i] | b[i];
}

__global__ __launch_bounds__(256) void k_and3_mask(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask,
                                                   uint8_t *__restrict__ dst, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const bool keep = mask[i] != 0;
    dst[3 * i] = keep ? src[3 * i] : 0;
    dst[3 * i + 1] = keep ? src[3 * i + 1] : 0;
    dst[3 * i + 2] = keep ? src[3 * i + 2] : 0;
}

// gated gray, 4 pixels (12 source bytes) per thread; the gate always reads the frame as B, G, R
__global__ __launch_bounds__(256) void k_gate_gray(const uint8_t *__restrict__ src, size_t npix, int k0, int k1, int k2,
                                                   const int *__restrict__ tab_g, GateDev gate,
                                                   uint8_t *__restrict__ dst) {
    __shared__ int tab[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = tab_g[i];
    __syncthreads();
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t p0 = q * 4;
    if (p0 >= npix) return;
    auto one = [&](uint32_t px) -> uint32_t {   // px = c0 | c1 << 8 | c2 << 16
        const uint32_t gy = ((px & 255) * k0 + ((px >> 8) & 255) * k1 + ((px >> 16) & 255) * k2 + (1 << 14)) >> 15;
        return gate_keep(px, tab, gate) ? gy : 0u;
    };
    if (p0 + 4 <= npix) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src) + q * 3;
        const uint32_t a = s[0], b = s[1], c = s[2];
        const uint32_t g0 = one(a), g1 = one((a >> 24) | (b << 8)), g2 = one((b >> 16) | (c << 16)), g3 = one(c >> 8);
        reinterpret_cast<uint32_t *>(dst)[q] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (size_t p = p0; p < npix; p++)
            dst[p] = (uint8_t)one(src[3 * p] | ((uint32_t)src[3 * p + 1] << 8) | ((uint32_t)src[3 * p + 2] << 16));
    }
}

// cv::resize, INTER_LINEAR, 8-bit, cn = 1 or 3 interleaved channels, dense frames.  One thread per output
// pixel.  The host builds OpenCV's tables: per output column the left source column and the two 11-bit
// weights (a0 | a1 << 16), per output row the upper source row (unclipped) and its two weights.
//   H = S[sx] * a0 + S[sx + 1] * a1;   D = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2
// area2: exact 2x2 decimation, which OpenCV reroutes to its area code: D = (sum of the four + 2) >> 2.
__global__ __launch_bounds__(256) void k_resize_linear(const uint8_t *__restrict__ src, int sw, int sh, int cn,
                                                       uint8_t *__restrict__ dst, int dw, int dh,
                                                       const int *__restrict__ xofs, const int *__restrict__ ia,
                                                       const int *__restrict__ yofs, const int *__restrict__ ib, int area2) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, f = blockIdx.z;
    if (dx >= dw) return;
    const uint8_t *fs = src + (size_t)f * sw * sh * cn;
    uint8_t *out = dst + ((size_t)f * dh + dy) * (size_t)dw * cn + (size_t)dx * cn;
    if (area2) {
        const uint8_t *s0 = fs + ((size_t)(2 * dy) * sw + 2 * dx) * cn, *s1 = s0 + (size_t)sw * cn;
        for (int c = 0; c < cn; c++) out[c] = (uint8_t)((s0[c] + s0[cn + c] + s1[c] + s1[cn + c] + 2) >> 2);
        return;
    }
    const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1), a = ia[dx], a0 = (short)(a & 0xFFFF), a1 = a >> 16;
    const int sy = yofs[dy], b = ib[dy], b0 = (short)(b & 0xFFFF), b1 = b >> 16;
    const uint8_t *r0 = fs + (size_t)min(max(sy, 0), sh - 1) * sw * cn, *r1 = fs + (size_t)min(max(sy + 1, 0), sh - 1) * sw * cn;
    for (int c = 0; c < cn; c++) {
        const int h0 = r0[sx * cn + c] * a0 + r0[sx1 * cn + c] * a1;
        const int h1 = r1[sx * cn + c] * a0 + r1[sx1 * cn + c] * a1;
        out[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

// The same for 3-channel frames whose output width is a multiple of 4: a thread makes four output pixels
// (12 bytes, three dword stores) and reads its source bytes as whole dwords (three per pixel and row cover
// the six bytes of the two taps; 24 contiguous bytes per row for the 2x2 mean).  src_bytes bounds the reads:
// the few threads whose last dword would cross it take single bytes.
__device__ __forceinline__ void taps6(const uint8_t *row, size_t base, size_t src_end, uint32_t &lo, uint32_t &hi) {
    // lo = bytes base .. base+3, hi = bytes base+4, base+5 (upper half undefined)
    const size_t al = base & ~(size_t)3;
    if (al + 12 <= src_end) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(row + al);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], off = (uint32_t)(base - al);
        lo = __builtin_amdgcn_alignbyte(d1, d0, off);
        hi = __builtin_amdgcn_alignbyte(d2, d1, off);
    } else {
        auto at = [&](size_t i) -> uint32_t { return i < src_end ? row[i] : 0u; };   // only ever a zero-weight tap
        lo = at(base) | (at(base + 1) << 8) | (at(base + 2) << 16) | (at(base + 3) << 24);
        hi = at(base + 4) | (at(base + 5) << 8);
    }
}

__global__ __launch_bounds__(256) void k_resize_linear3x4(const uint8_t *__restrict__ src, int sw, int sh, size_t src_bytes,
                                                          uint8_t *__restrict__ dst, int dw, int dh,
                                                          const int *__restrict__ xofs, const int *__restrict__ ia,
                                                          const int *__restrict__ yofs, const int *__restrict__ ib, int area2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, f = blockIdx.z;
    const int dx0 = 4 * t;
    if (dx0 >= dw) return;
    const size_t frame = (size_t)sw * sh * 3, fbase = (size_t)f * frame;
    uint32_t o[12];
    if (area2) {
        const size_t b0 = fbase + ((size_t)(2 * dy) * sw + 2 * dx0) * 3, b1 = b0 + (size_t)sw * 3;   // 24 bytes each, 8-byte aligned rows apart
        uint32_t r0[6], r1[6];
        if ((b0 & 3) == 0 && (b1 & 3) == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) { r0[k] = reinterpret_cast<const uint32_t *>(src + b0)[k]; r1[k] = reinterpret_cast<const uint32_t *>(src + b1)[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                r0[k] = src[b0 + 4 * k] | (src[b0 + 4 * k + 1] << 8) | (src[b0 + 4 * k + 2] << 16) | ((uint32_t)src[b0 + 4 * k + 3] << 24);
                r1[k] = src[b1 + 4 * k] | (src[b1 + 4 * k + 1] << 8) | (src[b1 + 4 * k + 2] << 16) | ((uint32_t)src[b1 + 4 * k + 3] << 24);
            }
        }
        auto byte = [](const uint32_t (&r)[6], int i) -> uint32_t { return (r[i >> 2] >> (8 * (i & 3))) & 255u; };
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                o[3 * p + c] = (byte(r0, 6 * p + c) + byte(r0, 6 * p + 3 + c) + byte(r1, 6 * p + c) + byte(r1, 6 * p + 3 + c) + 2) >> 2;
    } else {
        const int sy = yofs[dy], b = ib[dy], wb0 = (short)(b & 0xFFFF), wb1 = b >> 16;
        const size_t row0 = fbase + (size_t)min(max(sy, 0), sh - 1) * sw * 3, row1 = fbase + (size_t)min(max(sy + 1, 0), sh - 1) * sw * 3;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int sx = xofs[dx0 + p], a = ia[dx0 + p], a0 = (short)(a & 0xFFFF), a1 = a >> 16;
            // the right tap's weight is 0 wherever it would fall outside the row (sx = sw - 1), so reading the
            // six bytes at sx is enough: what lies behind the row's end is multiplied by zero
            // the lower row's weight is 0 for every output row of an odd integer reduction (1080p -> 360p: the sample
            // point falls on a source row): that row is then not read at all -- a third of the source instead of two
            uint32_t l0, h0, l1 = 0, h1 = 0;
            taps6(src, row0 + (size_t)sx * 3, src_bytes, l0, h0);
            if (wb1 != 0) taps6(src, row1 + (size_t)sx * 3, src_bytes, l1, h1);
            const uint32_t t0[6] = {l0 & 255, (l0 >> 8) & 255, (l0 >> 16) & 255, l0 >> 24, h0 & 255, (h0 >> 8) & 255};
            const uint32_t t1[6] = {l1 & 255, (l1 >> 8) & 255, (l1 >> 16) & 255, l1 >> 24, h1 & 255, (h1 >> 8) & 255};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int hh0 = (int)t0[c] * a0 + (int)t0[3 + c] * a1;
                const int hh1 = (int)t1[c] * a0 + (int)t1[3 + c] * a1;
                o[3 * p + c] = (uint32_t)((((wb0 * (hh0 >> 4)) >> 16) + ((wb1 * (hh1 >> 4)) >> 16) + 2) >> 2) & 255u;
            }
        }
    }
    uint32_t *out = reinterpret_cast<uint32_t *>(dst + ((size_t)f * dh + dy) * (size_t)dw * 3 + (size_t)dx0 * 3);
    out[0] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
    out[1] = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
    out[2] = o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24);
}

// Reduction by exactly three in both directions (1080p -> 640 x 360): every sample point of the bilinear formula
// falls on a source pixel -- x = 3 dx + 1, y = 3 dy + 1 with weights (2048, 0) both ways, for which the fixed-point
// expression returns the source byte itself -- so the result is that pixel's three bytes.  A thread makes four
// output pixels from the 36 contiguous source bytes that hold them (two 16-byte loads and a 4-byte one at 4-byte
// alignment; pixel p sits at bytes 9 p + 3 .. 9 p + 5).  The host takes this path only after checking the tables.
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void k_decimate3_3x4(const uint8_t *__restrict__ src, int sw, int sh, uint8_t *__restrict__ dst,
                                                       int dw, int dh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, f = blockIdx.z;
    if (4 * t >= dw) return;
    const uint8_t *base = src + ((size_t)f * sh + (size_t)(3 * dy + 1)) * (size_t)sw * 3 + (size_t)36 * t;
    const u32x4a4 q0 = *reinterpret_cast<const u32x4a4 *>(base), q1 = *reinterpret_cast<const u32x4a4 *>(base + 16);
    const uint32_t d[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, *reinterpret_cast<const uint32_t *>(base + 32)};
    auto byte = [&](int i) -> uint32_t { return (d[i >> 2] >> (8 * (i & 3))) & 255u; };
    uint32_t *out = reinterpret_cast<uint32_t *>(dst + ((size_t)f * dh + dy) * (size_t)dw * 3 + (size_t)12 * t);
    out[0] = byte(3) | (byte(4) << 8) | (byte(5) << 16) | (byte(12) << 24);
    out[1] = byte(13) | (byte(14) << 8) | (byte(21) << 16) | (byte(22) << 24);
    out[2] = byte(23) | (byte(30) << 8) | (byte(31) << 16) | (byte(32) << 24);
}

// cv::addWeighted on 8-bit data: saturate(cvRound(a * alpha + (b * beta + gamma))) in float, every product and
// sum rounded on its own; four bytes per thread
__device__ __forceinline__ uint32_t blend1(uint32_t a, uint32_t b, float fa, float fb, float fg) {
    const float t = __fadd_rn(__fmul_rn((float)a, fa), __fadd_rn(__fmul_rn((float)b, fb), fg));
    const int v = __float2int_rn(fminf(fmaxf(t, -1.0f), 256.0f));
    return (uint32_t)min(max(v, 0), 255);
}
__global__ __launch_bounds__(256) void k_add_weighted(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, float fa,
                                                      float fb, float fg, uint8_t *__restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const uint32_t va = *reinterpret_cast<const uint32_t *>(a + i), vb = *reinterpret_cast<const uint32_t *>(b + i);
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) r |= blend1((va >> (8 * k)) & 255u, (vb >> (8 * k)) & 255u, fa, fb, fg) << (8 * k);
        *reinterpret_cast<uint32_t *>(dst + i) = r;
    } else {
        for (size_t j = i; j < n; j++) dst[j] = (uint8_t)blend1(a[j], b[j], fa, fb, fg);
    }
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

void hsv_tables_host(int *tab) {
    tab[0] = tab[256] = 0;
    for (int i = 1; i < 256; i++) {
        tab[i] = (int)std::lrint((255 << 12) / (double)i);          // cvRound: to nearest, ties to even
        tab[256 + i] = (int)std::lrint((180 << 12) / (6. * i));
    }
}

GateDev make_gate(const int *b, int nboxes) {
    GateDev g;
    for (int box = 0; box < 2; box++)
        for (int c = 0; c < 3; c++) {
            int lo = 1000, hi = 1000;
            if (box < nboxes) {
                lo = b[6 * box + c] < 0 ? 0 : b[6 * box + c];
                hi = b[6 * box + 3 + c] > 255 ? 255 : b[6 * box + 3 + c];
                if (lo > hi) lo = hi = 1000;
            }
            g.lo[3 * box + c] = lo;
            g.span[3 * box + c] = (uint32_t)(hi - lo);
        }
    g.hue_any = 0;
    g.vmin = 0;
    for (int box = 0; box < 2; box++)
        if (g.lo[3 * box] == 0 && g.span[3 * box] >= 179u) g.hue_any |= 1 << box;
    return g;
}

void gate_tables_host(GateDev &g, int *tab) {
    hsv_tables_host(tab);
    g.vmin = 256;
    for (int v = 0; v < 256; v++) {
        int dmin[2], dmax[2];
        for (int box = 0; box < 2; box++) {
            dmin[box] = 255; dmax[box] = 0;
            const long long vlo = g.lo[3 * box + 2], vhi = vlo + g.span[3 * box + 2];
            if (v < vlo || v > vhi) continue;
            const long long slo = g.lo[3 * box + 1], shi = slo + g.span[3 * box + 1];
            int first = -1, last = -1;
            for (int d = 0; d <= v; d++) {                       // diff never exceeds v
                const int s = (d * tab[v] + (1 << 11)) >> 12;
                if (s >= slo && s <= shi) { if (first < 0) first = d; last = d; }
            }
            if (first >= 0) { dmin[box] = first; dmax[box] = last; if (v < g.vmin) g.vmin = v; }
        }
        tab[512 + v] = (int)((uint32_t)dmin[0] | ((uint32_t)dmax[0] << 8) | ((uint32_t)dmin[1] << 16) | ((uint32_t)dmax[1] << 24));
    }
}

// true when the tables of a (sw, sh) -> (dw, dh) resize describe the plain reduction by three: every output pixel is
// the source pixel (3 dx + 1, 3 dy + 1) with the whole weight
bool resize_is_decimate3(int sw, int sh, int dw, int dh, const int *xofs, const int *ia, const int *yofs, const int *ib) {
    if (sw != 3 * dw || sh != 3 * dh) return false;
    for (int dx = 0; dx < dw; dx++) if (xofs[dx] != 3 * dx + 1 || ia[dx] != 2048) return false;
    for (int dy = 0; dy < dh; dy++) if (yofs[dy] != 3 * dy + 1 || ib[dy] != 2048) return false;
    return true;
}

bool resize_tables_host(int sw, int sh, int dw, int dh, int *xofs, int *ia, int *yofs, int *ib) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int iscale_x = (int)std::lrint(scale_x), iscale_y = (int)std::lrint(scale_y);
    const bool area2 = std::fabs(scale_x - iscale_x) < 2.220446049250313e-16 && std::fabs(scale_y - iscale_y) < 2.220446049250313e-16 &&
                       iscale_x == 2 && iscale_y == 2;
    auto sat_short = [](float v) { long r = std::lrintf(v); return (int)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r)); };
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ia[dx] = (sat_short((1.f - fx) * 2048) & 0xFFFF) | (sat_short(fx * 2048) << 16);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ib[dy] = (sat_short((1.f - fy) * 2048) & 0xFFFF) | (sat_short(fy * 2048) << 16);
    }
    return area2;
}

void launch_resize_linear(const uint8_t *src, int sw, int sh, int cn, int nframes, uint8_t *dst, int dw, int dh,
                          const int *tabs, bool area2, hipStream_t st, bool decimate3) {
    // tabs: xofs[dw], ia[dw], yofs[dh], ib[dh] back to back
    if (cn == 3 && dw % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
        if (decimate3) {
            const int tpr3 = dw / 4, bs3 = std::min(256, (tpr3 + 63) / 64 * 64);
            hipLaunchKernelGGL(k_decimate3_3x4, dim3((tpr3 + bs3 - 1) / bs3, dh, nframes), dim3(bs3), 0, st, src, sw, sh, dst, dw, dh);
            return;
        }
        // a row of dw / 4 threads: workgroups as wide as the row needs, in whole wavefronts (640 columns: 160 threads
        // in a workgroup of 192, not 256)
        const int tpr = dw / 4, bs = std::min(256, (tpr + 63) / 64 * 64);
        hipLaunchKernelGGL(k_resize_linear3x4, dim3((tpr + bs - 1) / bs, dh, nframes), dim3(bs), 0, st, src, sw, sh,
                           (size_t)sw * sh * 3 * nframes, dst, dw, dh, tabs, tabs + dw, tabs + 2 * dw, tabs + 2 * dw + dh, area2 ? 1 : 0);
        return;
    }
    hipLaunchKernelGGL(k_resize_linear, dim3((dw + 255) / 256, dh, nframes), dim3(256), 0, st, src, sw, sh, cn, dst, dw, dh, tabs,
                       tabs + dw, tabs + 2 * dw, tabs + 2 * dw + dh, area2 ? 1 : 0);
}

void launch_add_weighted(const uint8_t *a, float alpha, const uint8_t *b, float beta, float gamma, uint8_t *dst, size_t n,
                         hipStream_t st) {
    hipLaunchKernelGGL(k_add_weighted, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, st, a, b, alpha, beta, gamma, dst, n);
}

void launch_hsv(const uint8_t *src, size_t npix, int code, const int *tab, uint8_t *dst, hipStream_t st) {
    hipLaunchKernelGGL(k_hsv, dim3(blocks_for(npix)), dim3(256), 0, st, src, npix, code == 41 ? 1 : 0, tab, dst);
}
void launch_in_range3(const uint8_t *src, size_t npix, GateDev box0, uint8_t *dst, hipStream_t st) {
    hipLaunchKernelGGL(k_in_range3, dim3(blocks_for(npix)), dim3(256), 0, st, src, npix, box0, dst);
}
void launch_or(const uint8_t *a, const uint8_t *b, uint8_t *dst, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_or, dim3(blocks_for(n)), dim3(256), 0, st, a, b, dst, n);
}
void launch_and3_mask(const uint8_t *src, const uint8_t *mask, uint8_t *dst, size_t npix, hipStream_t st) {
    hipLaunchKernelGGL(k_and3_mask, dim3(blocks_for(npix)), dim3(256), 0, st, src, mask, dst, npix);
}
void launch_gate_gray(const uint8_t *bgr, size_t npix, int gray_code, const int *tab, GateDev gate, uint8_t *gray,
                      hipStream_t st) {
    const int k0 = gray_code == 6 ? 3735 : 9798, k2 = gray_code == 6 ? 9798 : 3735;
    hipLaunchKernelGGL(k_gate_gray, dim3(blocks_for((npix + 3) / 4)), dim3(256), 0, st, bgr, npix, k0, 19235, k2, tab, gate, gray);
}

}  // namespace lanecv
