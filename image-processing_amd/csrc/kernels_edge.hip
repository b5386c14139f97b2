// Edge stage of the lane pipeline for gfx950: gray conversion, 5x5 binomial
// blur, Sobel, non-maximum suppression, hysteresis and the ROI mask.
//
// Arithmetic contract (bit-exact with OpenCV's 8U fixed-point paths):
//   gray   (b*k0 + g*k1 + r*k2 + 2^14) >> 15, weights 3735/19235/9798
//   blur   [1 4 6 4 1]^2, BORDER_REFLECT_101, (S + 128) >> 8 computed as
//          hline << 4 (16 bit) then (vline + 2^11) >> 12
//   Sobel  3x3, BORDER_REPLICATE; magnitude |dx| + |dy|, zero outside the image
//   NMS    three sectors split at tan(22.5 deg) = 13573 / 2^15 with OpenCV's
//          asymmetric > / >= comparisons; strict thresholds
//   hysteresis: 8-connected closure from strong through weak pixels
#include "hysteresis.h"
#include <cstdlib>
#include <algorithm>

namespace lanecv {
namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}
__device__ __forceinline__ int clampi(int p, int len) { return p < 0 ? 0 : (p >= len ? len - 1 : p); }

// ---------------------------------------------------------------------------
// stand-alone gray conversion (cvtColor drop-in): 4 pixels per thread
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gray(const uint8_t *__restrict__ src, size_t npix, int k0, int k1, int k2,
                                              uint8_t *__restrict__ dst) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 pixels = 12 source bytes
    size_t p0 = q * 4;
    if (p0 >= npix) return;
    if (p0 + 4 <= npix) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src) + q * 3;
        uint32_t a = s[0], b = s[1], c = s[2];
        uint32_t g0 = ((a & 255) * k0 + ((a >> 8) & 255) * k1 + ((a >> 16) & 255) * k2 + (1 << 14)) >> 15;
        uint32_t g1 = ((a >> 24) * k0 + (b & 255) * k1 + ((b >> 8) & 255) * k2 + (1 << 14)) >> 15;
        uint32_t g2 = (((b >> 16) & 255) * k0 + (b >> 24) * k1 + (c & 255) * k2 + (1 << 14)) >> 15;
        uint32_t g3 = (((c >> 8) & 255) * k0 + ((c >> 16) & 255) * k1 + (c >> 24) * k2 + (1 << 14)) >> 15;
        reinterpret_cast<uint32_t *>(dst)[q] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (size_t p = p0; p < npix; p++)
            dst[p] = (uint8_t)((src[3 * p] * k0 + src[3 * p + 1] * k1 + src[3 * p + 2] * k2 + (1 << 14)) >> 15);
    }
}

// ---------------------------------------------------------------------------
// stand-alone 5x5 blur (GaussianBlur drop-in): one 64x16 tile per block
// ---------------------------------------------------------------------------
constexpr int BT_W = 64, BT_H = 16;
__global__ __launch_bounds__(256) void k_blur5(const uint8_t *__restrict__ src, int W, int H, uint8_t *__restrict__ dst) {
    __shared__ uint8_t sg[BT_H + 4][BT_W + 4];
    __shared__ uint16_t sh[BT_H + 4][BT_W];
    const int tx0 = blockIdx.x * BT_W, ty0 = blockIdx.y * BT_H;
    const size_t fo = (size_t)blockIdx.z * W * H;
    for (int i = threadIdx.x; i < (BT_H + 4) * (BT_W + 4); i += 256) {
        int r = i / (BT_W + 4), c = i % (BT_W + 4);
        int y = reflect101(ty0 - 2 + r, H), x = reflect101(tx0 - 2 + c, W);
        sg[r][c] = src[fo + (size_t)y * W + x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (BT_H + 4) * BT_W; i += 256) {
        int r = i / BT_W, c = i % BT_W;
        int v = sg[r][c] + sg[r][c + 4] + 4 * (sg[r][c + 1] + sg[r][c + 3]) + 6 * sg[r][c + 2];
        sh[r][c] = (uint16_t)(v << 4);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BT_H * BT_W; i += 256) {
        int r = i / BT_W, c = i % BT_W;
        int y = ty0 + r, x = tx0 + c;
        if (y < H && x < W) {
            uint32_t v = (uint32_t)sh[r][c] + sh[r + 4][c] + 4u * ((uint32_t)sh[r + 1][c] + sh[r + 3][c]) + 6u * sh[r + 2][c];
            dst[fo + (size_t)y * W + x] = (uint8_t)((v + (1u << 11)) >> 12);
        }
    }
}

// ---------------------------------------------------------------------------
// fused front kernel: (BGR ->) gray (-> blur) -> Sobel -> NMS -> class bit planes
// One block = one 64 x FT_H output tile; every stage is staged through LDS with
// the halo the later stages need (blur 2 + Sobel 1 + NMS 1 = 4 pixels).
// Only in-image coordinates are computed; reads that leave the image are
// remapped with the border rule of the stage that reads.
// ---------------------------------------------------------------------------
constexpr int FT_W = 64, FT_H = 32, FNT = 256;

template <bool IN_BGR, bool BLUR>
__global__ __launch_bounds__(FNT) void k_front(const uint8_t *__restrict__ src, size_t src_bytes, Geo g, int k0, int k1,
                                               int k2, int low, int high, uint64_t *__restrict__ planeS,
                                               uint64_t *__restrict__ planeW, uint32_t *__restrict__ wake,
                                               uint8_t *__restrict__ gray_out, uint8_t *__restrict__ blur_out) {
    constexpr int HALO = BLUR ? 4 : 2;
    constexpr int GW = FT_W + 2 * HALO, GH = FT_H + 2 * HALO;   // gray region
    constexpr int SW = FT_W + 4, SH = FT_H + 4;                 // Sobel-input region (halo 2)
    constexpr int MW = FT_W + 2, MH = FT_H + 2;                 // magnitude region (halo 1)
    constexpr int RAWW = IN_BGR ? (3 * GW + 6) / 4 + 1 : 1;     // staging words per row

    __shared__ uint32_t sraw[IN_BGR ? GH : 1][RAWW];
    __shared__ uint8_t sg[GH][GW];
    __shared__ uint16_t sh[BLUR ? GH : 1][BLUR ? SW : 1];
    __shared__ uint8_t sblur[BLUR ? SH : 1][BLUR ? SW : 1];
    __shared__ uint16_t smag[MH][MW];
    __shared__ uint8_t sdir[MH][MW];

    const int W = g.W, H = g.H;
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * FT_W, ty0 = blockIdx.y * FT_H, f = blockIdx.z;
    const int gx0 = tx0 - HALO, gy0 = ty0 - HALO;
    // bit planes are row-major: one word per (row, 64-pixel column group)
    const size_t plane_base = ((size_t)f * H + ty0) * g.TX + blockIdx.x;

    // ---- 1. gray into LDS -------------------------------------------------
    const int cx0 = gx0 < 0 ? 0 : gx0;                       // first in-image column of the region
    const int cx1 = (gx0 + GW < W) ? gx0 + GW : W;           // one past the last
    if constexpr (IN_BGR) {
        const size_t fbase = (size_t)f * 3 * W * H;
        for (int i = tid; i < GH * RAWW; i += FNT) {
            int r = i / RAWW, k = i % RAWW;
            int y = gy0 + r;
            if ((unsigned)y >= (unsigned)H) continue;
            size_t a0 = fbase + (size_t)y * 3 * W + 3 * (size_t)cx0;
            size_t aw = (a0 & ~(size_t)3) + 4 * (size_t)k;
            size_t aend = a0 + 3 * (size_t)(cx1 - cx0);
            if (aw >= aend) continue;
            uint32_t v;
            if (aw + 4 <= src_bytes) v = *reinterpret_cast<const uint32_t *>(src + aw);
            else {
                v = 0;
                for (int b = 0; b < 4 && aw + b < src_bytes; b++) v |= (uint32_t)src[aw + b] << (8 * b);
            }
            sraw[r][k] = v;
        }
        __syncthreads();
        for (int i = tid; i < GH * GW; i += FNT) {
            int r = i / GW, c = i % GW;
            int y = gy0 + r, x = gx0 + c;
            if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
            size_t a0 = fbase + (size_t)y * 3 * W + 3 * (size_t)cx0;
            int bo = (int)(a0 & 3) + 3 * (x - cx0);
            const uint8_t *rb = reinterpret_cast<const uint8_t *>(&sraw[r][0]);
            sg[r][c] = (uint8_t)((rb[bo] * k0 + rb[bo + 1] * k1 + rb[bo + 2] * k2 + (1 << 14)) >> 15);
        }
    } else {
        const size_t fbase = (size_t)f * W * H;
        for (int i = tid; i < GH * GW; i += FNT) {
            int r = i / GW, c = i % GW;
            int y = gy0 + r, x = gx0 + c;
            if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
            sg[r][c] = src[fbase + (size_t)y * W + x];
        }
    }
    __syncthreads();
    if (IN_BGR && gray_out) {
        for (int i = tid; i < FT_H * FT_W; i += FNT) {
            int r = i / FT_W, c = i % FT_W;
            int y = ty0 + r, x = tx0 + c;
            if (y < H && x < W) gray_out[(size_t)f * W * H + (size_t)y * W + x] = sg[r + HALO][c + HALO];
        }
    }

    // ---- 2. optional 5x5 blur ----------------------------------------------
    if constexpr (BLUR) {
        for (int i = tid; i < GH * SW; i += FNT) {
            int r = i / SW, c = i % SW;
            int y = gy0 + r, x = tx0 - 2 + c;
            if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
            int a = sg[r][reflect101(x - 2, W) - gx0], b = sg[r][reflect101(x - 1, W) - gx0];
            int cc = sg[r][x - gx0];
            int d = sg[r][reflect101(x + 1, W) - gx0], e = sg[r][reflect101(x + 2, W) - gx0];
            sh[r][c] = (uint16_t)((a + e + 4 * (b + d) + 6 * cc) << 4);
        }
        __syncthreads();
        for (int i = tid; i < SH * SW; i += FNT) {
            int r = i / SW, c = i % SW;
            int y = ty0 - 2 + r, x = tx0 - 2 + c;
            if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
            uint32_t v0 = sh[reflect101(y - 2, H) - gy0][c], v1 = sh[reflect101(y - 1, H) - gy0][c];
            uint32_t v2 = sh[y - gy0][c];
            uint32_t v3 = sh[reflect101(y + 1, H) - gy0][c], v4 = sh[reflect101(y + 2, H) - gy0][c];
            uint8_t o = (uint8_t)((v0 + v4 + 4u * (v1 + v3) + 6u * v2 + (1u << 11)) >> 12);
            sblur[r][c] = o;
            if (blur_out && r >= 2 && r < SH - 2 && c >= 2 && c < SW - 2)
                blur_out[(size_t)f * W * H + (size_t)y * W + x] = o;
        }
        __syncthreads();
    }

    // ---- 3. Sobel + magnitude + direction sector ---------------------------
    for (int i = tid; i < MH * MW; i += FNT) {
        int r = i / MW, c = i % MW;
        int y = ty0 - 1 + r, x = tx0 - 1 + c;
        uint16_t m = 0;
        uint8_t dir = 0;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            // Sobel-input region starts at (ty0-2, tx0-2)
            int ym = clampi(y - 1, H) - (ty0 - 2), y0 = y - (ty0 - 2), yp = clampi(y + 1, H) - (ty0 - 2);
            int xm = clampi(x - 1, W) - (tx0 - 2), x0 = x - (tx0 - 2), xp = clampi(x + 1, W) - (tx0 - 2);
            int p00, p01, p02, p10, p12, p20, p21, p22;
            if constexpr (BLUR) {
                p00 = sblur[ym][xm]; p01 = sblur[ym][x0]; p02 = sblur[ym][xp];
                p10 = sblur[y0][xm];                      p12 = sblur[y0][xp];
                p20 = sblur[yp][xm]; p21 = sblur[yp][x0]; p22 = sblur[yp][xp];
            } else {
                p00 = sg[ym][xm]; p01 = sg[ym][x0]; p02 = sg[ym][xp];
                p10 = sg[y0][xm];                   p12 = sg[y0][xp];
                p20 = sg[yp][xm]; p21 = sg[yp][x0]; p22 = sg[yp][xp];
            }
            int dx = (p02 + 2 * p12 + p22) - (p00 + 2 * p10 + p20);
            int dy = (p20 + 2 * p21 + p22) - (p00 + 2 * p01 + p02);
            int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
            m = (uint16_t)(ax + ay);
            int ys = ay << 15, tg22x = ax * 13573;
            if (ys < tg22x) dir = 0;
            else if (ys > tg22x + (ax << 16)) dir = 1;
            else dir = ((dx ^ dy) < 0) ? 3 : 2;
        }
        smag[r][c] = m;
        sdir[r][c] = dir;
    }
    __syncthreads();

    // ---- 4. non-maximum suppression + double threshold -> bit planes --------
    for (int i = tid; i < FT_H * FT_W; i += FNT) {
        int r = i / FT_W, c = i % FT_W;   // r is uniform across the wavefront, c = lane
        int y = ty0 + r, x = tx0 + c;
        bool strong = false, weak = false;
        if (y < H && x < W) {
            int m = smag[r + 1][c + 1];
            if (m > low) {
                bool keep;
                switch (sdir[r + 1][c + 1]) {
                    case 0: keep = m > smag[r + 1][c] && m >= smag[r + 1][c + 2]; break;
                    case 1: keep = m > smag[r][c + 1] && m >= smag[r + 2][c + 1]; break;
                    case 2: keep = m > smag[r][c] && m > smag[r + 2][c + 2]; break;
                    default: keep = m > smag[r][c + 2] && m > smag[r + 2][c]; break;
                }
                strong = keep && m > high;
                weak = keep && !strong;
            }
        }
        uint64_t bs = __ballot(strong), bw = __ballot(weak);
        if (c == 0 && y < H) {
            planeS[plane_base + (size_t)r * g.TX] = bs;
            planeW[plane_base + (size_t)r * g.TX] = bw;
            if (bw) wake[((size_t)f * g.TY + (y >> 6)) * g.TX + (tx0 >> 6)] = 1u;
        }
    }
}

// The stand-alone hysteresis kernel (device code in hysteresis.h).  status (may be null): the words the kernels behind
// this one on the stream raise with plain stores; a launch over all frames zeroes them first (see launch_hysteresis in
// common.h for their meaning).
#ifndef LC_HC_THREADS
#define LC_HC_THREADS 1024
#endif
constexpr int HC_THREADS = LC_HC_THREADS;
__global__ __launch_bounds__(HC_THREADS) void k_hysteresis(HystArgs h) {
    __shared__ int any, count;
    __shared__ uint32_t px[RING_LIST];
    // every frame: the workgroup's one frame.  Flagged frames only (only_frames): a few workgroups that go through the
    // frames looking the flags up -- nearly always none is set, and a workgroup of this size per frame that only reads a
    // flag still has to find a CU with room for it
    for (int f = blockIdx.x; f < h.nframes; f += gridDim.x) hysteresis_one_frame<HC_THREADS>(h, f, any, count, px);
}

// ---------------------------------------------------------------------------
// finalize: strong plane (& ROI bits) -> 0/255 bytes, row-major bit plane for the
// Hough stage, per-row point counts.  One wavefront per 64x64 tile; a lane
// expands 16 pixels at a time so that stores are 16 bytes wide.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand4(uint32_t bits4) {
    // 4 bits -> 4 bytes of 0x00 / 0xFF
    uint32_t r = (bits4 * 0x00204081u) & 0x01010101u;
    return r * 255u;
}

__global__ __launch_bounds__(256) void k_finalize(Geo g, int ntiles_total, const uint64_t *__restrict__ planeS,
                                                  const uint64_t *__restrict__ roi_bits, uint8_t *__restrict__ edges_out,
                                                  uint8_t *__restrict__ masked_out, uint64_t *__restrict__ rowplane,
                                                  int *__restrict__ rowcount) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles_total) return;
    const int tiles_per_frame = g.TX * g.TY;
    const int f = t / tiles_per_frame, tt = t % tiles_per_frame;
    const int ty = tt / g.TX, tx = tt % g.TX;
    const int W = g.W, H = g.H;
    // lane = row: masked bits, row-major plane and row counts
    const int yrow = ty * TILE + lane;
    uint64_t S = yrow < H ? planeS[((size_t)f * H + yrow) * g.TX + tx] : 0;
    uint64_t M = (roi_bits && yrow < H) ? (S & roi_bits[(size_t)yrow * g.TX + tx]) : S;
    if (yrow < H) {
        if (rowplane) rowplane[((size_t)f * H + yrow) * g.TX + tx] = M;
        if (rowcount) {
            int c = __popcll(M);
            if (c) atomicAdd(&rowcount[(size_t)f * H + yrow], c);
        }
    }
    // byte expansion: iteration k covers rows 4k..4k+3, lane covers 16 pixels
    const bool vec_ok = (W % 16) == 0;
    const int sub = lane & 3, rsel = lane >> 2;   // 16-pixel column group, row within the group of 16 rows
    for (int k = 0; k < 4; k++) {
        int r = k * 16 + rsel;
        uint64_t Sr = shfl_u64(S, r), Mr = shfl_u64(M, r);
        int y = ty * TILE + r, x = tx * TILE + sub * 16;
        if (y >= H || x >= W) continue;
        uint32_t sb = (uint32_t)(Sr >> (sub * 16)) & 0xFFFFu, mb = (uint32_t)(Mr >> (sub * 16)) & 0xFFFFu;
        size_t o = (size_t)f * W * H + (size_t)y * W + x;
        uint4 es, ms;
        es.x = expand4(sb & 15); es.y = expand4((sb >> 4) & 15); es.z = expand4((sb >> 8) & 15); es.w = expand4(sb >> 12);
        ms.x = expand4(mb & 15); ms.y = expand4((mb >> 4) & 15); ms.z = expand4((mb >> 8) & 15); ms.w = expand4(mb >> 12);
        if (vec_ok) {
            if (edges_out) *reinterpret_cast<uint4 *>(edges_out + o) = es;
            if (masked_out) *reinterpret_cast<uint4 *>(masked_out + o) = ms;
        } else {
            int n = W - x < 16 ? W - x : 16;
            for (int j = 0; j < n; j++) {
                if (edges_out) edges_out[o + j] = (sb >> j) & 1 ? 255 : 0;
                if (masked_out) masked_out[o + j] = (mb >> j) & 1 ? 255 : 0;
            }
        }
    }
}

// finalize with dirty tracking, for a pipeline slot that writes the same masked-map buffer batch after batch.
// prevbits holds the masked bits this kernel wrote last time (all zero at first, like the buffer): a 16-pixel
// group whose bits were zero then and are zero now already holds its sixteen 0 bytes, so only groups that had
// or have an edge pixel are written -- a few per cent of the map.  The same holds for the plane word handed to
// the Hough stage (which only ever clears bits of it) and for the row counts (their consumer clears them).
// One thread per 64-pixel word, words in memory order.
// one plane word: M = its masked bits now, P = the bits written last time
__device__ __forceinline__ void finalize_word(const Geo &g, size_t f, size_t wf, size_t i, uint64_t M, uint64_t P,
                                              uint64_t *__restrict__ prevbits, uint8_t *__restrict__ masked_out,
                                              uint64_t *__restrict__ rowplane) {
    const uint64_t D = M | P;
    if (D == 0) return;
    if (rowplane) rowplane[i] = M;
    if (M != P) prevbits[i] = M;
    const int y = (int)(wf / g.TX), tx = (int)(wf % g.TX);
    const int W = g.W;
    const bool vec_ok = (W % 16) == 0;
    const size_t o = f * (size_t)W * g.H + (size_t)y * W + (size_t)tx * TILE;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t d = (uint32_t)(D >> (16 * q)) & 0xFFFFu, mb = (uint32_t)(M >> (16 * q)) & 0xFFFFu;
        const int x = tx * TILE + 16 * q;
        if (d == 0 || x >= W) continue;
        if (vec_ok) {
            uint4 ms;
            ms.x = expand4(mb & 15); ms.y = expand4((mb >> 4) & 15); ms.z = expand4((mb >> 8) & 15); ms.w = expand4(mb >> 12);
            *reinterpret_cast<uint4 *>(masked_out + o + 16 * q) = ms;
        } else {
            const int n = W - x < 16 ? W - x : 16;
            for (int j = 0; j < n; j++) masked_out[o + 16 * q + j] = (mb >> j) & 1 ? 255 : 0;
        }
    }
}

__global__ __launch_bounds__(256) void k_finalize_sparse(Geo g, size_t nwords, const uint64_t *__restrict__ planeS,
                                                         const uint64_t *__restrict__ roi_bits, uint64_t *__restrict__ prevbits,
                                                         uint8_t *__restrict__ masked_out, uint64_t *__restrict__ rowplane,
                                                         int *__restrict__ rowcount, FrameSet fs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nwords) return;
    const size_t words_per_frame = (size_t)g.TX * g.H;
    const size_t f = i / words_per_frame, wf = i % words_per_frame;   // wf = y * TX + tx
    if (!fs.has((int)f)) return;
    uint64_t M = planeS[i];
    if (roi_bits) M &= roi_bits[wf];
    const uint64_t P = prevbits[i];
    if (M && rowcount) atomicAdd(&rowcount[f * g.H + wf / g.TX], __popcll(M));
    finalize_word(g, f, wf, i, M, P, prevbits, masked_out, rowplane);
}

// The same over the plane words that hold ROI pixels only (listed once per pipeline as row << 8 | word of the row): the
// other words' masked bits are zero now and were zero last time, so there is nothing to look at -- with the
// triangle three quarters of a frame's words.  One thread per listed word, blockIdx.y = frame.
__global__ __launch_bounds__(256) void k_finalize_roi(Geo g, const uint32_t *__restrict__ roi_words, int nlist,
                                                      const uint64_t *__restrict__ planeS, const uint64_t *__restrict__ roi_bits,
                                                      uint64_t *__restrict__ prevbits, uint8_t *__restrict__ masked_out,
                                                      uint64_t *__restrict__ rowplane, int *__restrict__ rowcount, FrameSet fs) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x), f = (int)blockIdx.y;
    if (k >= nlist || !fs.has(f)) return;
    const uint32_t e = roi_words[k];
    const int y = (int)(e >> 8);
    const size_t wf = (size_t)y * g.TX + (e & 255u), i = (size_t)f * g.TX * g.H + wf;
    const uint64_t M = planeS[i] & roi_bits[wf], P = prevbits[i];
    if (M && rowcount) atomicAdd(&rowcount[(size_t)f * g.H + y], __popcll(M));
    finalize_word(g, (size_t)f, wf, i, M, P, prevbits, masked_out, rowplane);
}

// bitwise_and with an optional mask (dst = 0 where mask == 0)
__global__ __launch_bounds__(256) void k_and(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                             const uint8_t *__restrict__ mask, uint8_t *__restrict__ dst, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        uint32_t va = *reinterpret_cast<const uint32_t *>(a + i), vb = *reinterpret_cast<const uint32_t *>(b + i);
        uint32_t v = va & vb;
        if (mask) {
            uint32_t m = *reinterpret_cast<const uint32_t *>(mask + i);
            // per byte: 0xFF where the mask byte is non-zero
            uint32_t nzb = ((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu | m) & 0x80808080u;
            v &= (nzb >> 7) * 255u;
        }
        *reinterpret_cast<uint32_t *>(dst + i) = v;
    } else {
        for (; i < n; i++) dst[i] = (mask && !mask[i]) ? 0 : (a[i] & b[i]);
    }
}

// non-zero bytes -> row-major bit plane + row counts (front end of the stand-alone Hough calls)
__global__ __launch_bounds__(256) void k_pack_nonzero(const uint8_t *__restrict__ img, Geo g, int nframes,
                                                      uint64_t *__restrict__ rowplane, int *__restrict__ rowcount) {
    // one wavefront per (frame, row); lane l tests pixels l, 64+l, ...
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)nframes * g.H) return;
    const uint8_t *p = img + (size_t)row * g.W;
    int cnt = 0;
    for (int tx = 0; tx < g.TX; tx++) {
        int x = tx * 64 + lane;
        uint64_t b = __ballot(x < g.W && p[x] != 0);
        if (lane == 0) rowplane[(size_t)row * g.TX + tx] = b;
        cnt += __popcll(b);
    }
    if (lane == 0) rowcount[row] = cnt;
}

}  // namespace

// ---------------------------------------------------------------------------
void launch_gray(const uint8_t *bgr, int W, int H, int nframes, int code, uint8_t *gray, hipStream_t st) {
    size_t npix = (size_t)W * H * nframes;
    int k0 = code == 6 ? 3735 : 9798, k2 = code == 6 ? 9798 : 3735;
    size_t groups = (npix + 3) / 4;
    hipLaunchKernelGGL(k_gray, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, bgr, npix, k0, 19235, k2, gray);
}

void launch_blur5(const uint8_t *src, int W, int H, int nframes, uint8_t *dst, hipStream_t st) {
    dim3 grid((W + BT_W - 1) / BT_W, (H + BT_H - 1) / BT_H, nframes);
    hipLaunchKernelGGL(k_blur5, grid, dim3(256), 0, st, src, W, H, dst);
}

void launch_front(const uint8_t *src, size_t src_bytes, int in_kind, int code, bool blur, Geo g, int nframes, int low,
                  int high, uint64_t *planeS, uint64_t *planeW, uint32_t *wake, uint8_t *gray_out, uint8_t *blur_out,
                  hipStream_t st, const GateDev *gate, const int *tab, uint8_t *gate_gray, uint8_t *rowflags, bool dec3,
                  int rows_per_unit, const FrontSplit *split, const FrameSet *fs, const uint8_t *const *src_parts, const HystLaunch *lead) {
    static const char *force_tiled = getenv("LANECV_FRONT_TILED");       // diagnostics: use the tiled kernel everywhere
    if (in_kind == 1 && front_stream_supported(g) && !(force_tiled && force_tiled[0] == '1')) {
        launch_front_stream(src, code, blur, g, nframes, low, high, planeS, planeW, wake, gray_out, blur_out,
                            rows_per_unit, gate, tab, st, rowflags, dec3, split, fs, src_parts, lead);
        return;
    }
    // (the tiled kernel knows no frame sets: its callers launch it over one part's frames, all of them)
    if (fs && src_parts) src = src_parts[0];
    if (gate && in_kind == 1) {   // gated gray first, then the tiled kernel on that gray image
        const size_t npix = (size_t)g.W * g.H * nframes;
        launch_gate_gray(src, npix, code, tab, *gate, gate_gray, st);
        src = gate_gray; src_bytes = npix; in_kind = 0;
        if (gray_out == gate_gray) gray_out = nullptr;   // already there
    }
    int k0 = code == 6 ? 3735 : 9798, k2 = code == 6 ? 9798 : 3735;
    dim3 grid(g.TX, (g.H + FT_H - 1) / FT_H, nframes), block(FNT);
#define LC_FRONT(B, L) hipLaunchKernelGGL((k_front<B, L>), grid, block, 0, st, src, src_bytes, g, k0, 19235, k2, low, high, planeS, planeW, wake, gray_out, blur_out)
    if (in_kind == 1) { if (blur) LC_FRONT(true, true); else LC_FRONT(true, false); }
    else { if (blur) LC_FRONT(false, true); else LC_FRONT(false, false); }
#undef LC_FRONT
}

void launch_hysteresis(Geo g, int nframes, uint64_t *planeS, const uint64_t *planeW, uint32_t *wake, int *rounds,
                       hipStream_t st, int *status, const int *only_frames, const uint32_t *ring_words, int nring_words,
                       const uint64_t *ring_code, int *redo, const FrameSet *fs, int wake_frames) {
    const HystArgs h = hysteresis_args(g, nframes, planeS, planeW, wake, rounds, status, only_frames, ring_words, nring_words, ring_code, redo, fs, wake_frames);
    hipLaunchKernelGGL(k_hysteresis, dim3(only_frames ? std::min(nframes, 32) : nframes), dim3(HC_THREADS), 0, st, h);
}

void launch_finalize(Geo g, int nframes, const uint64_t *planeS, const uint64_t *roi_bits, uint8_t *edges_out,
                     uint8_t *masked_out, uint64_t *rowplane, int *rowcount, hipStream_t st, uint64_t *prevbits, const FrameSet *fs,
                     const uint32_t *roi_words, int n_roi_words) {
    if (prevbits && !edges_out && masked_out && roi_bits && roi_words && n_roi_words > 0) {
        hipLaunchKernelGGL(k_finalize_roi, dim3((unsigned)((n_roi_words + 255) / 256), (unsigned)nframes), dim3(256), 0, st, g, roi_words,
                           n_roi_words, planeS, roi_bits, prevbits, masked_out, rowplane, rowcount, fs ? *fs : FrameSet{});
        return;
    }
    if (prevbits && !edges_out && masked_out) {
        const size_t nwords = (size_t)g.TX * g.H * nframes;
        hipLaunchKernelGGL(k_finalize_sparse, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, g, nwords, planeS, roi_bits,
                           prevbits, masked_out, rowplane, rowcount, fs ? *fs : FrameSet{});
        return;
    }
    int nt = g.TX * g.TY * nframes;
    hipLaunchKernelGGL(k_finalize, dim3((nt + 3) / 4), dim3(256), 0, st, g, nt, planeS, roi_bits, edges_out, masked_out,
                       rowplane, rowcount);
}

void launch_and(const uint8_t *a, const uint8_t *b, const uint8_t *mask, uint8_t *dst, size_t n, hipStream_t st) {
    size_t groups = (n + 3) / 4;
    hipLaunchKernelGGL(k_and, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, a, b, mask, dst, n);
}

void launch_pack_nonzero(const uint8_t *img, Geo g, int nframes, uint64_t *rowplane, int *rowcount, hipStream_t st) {
    long long rows = (long long)nframes * g.H;
    hipLaunchKernelGGL(k_pack_nonzero, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, img, g, nframes, rowplane, rowcount);
}

}  // namespace lanecv
