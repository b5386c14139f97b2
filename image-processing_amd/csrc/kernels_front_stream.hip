// Streaming front kernel for gfx950: BGR -> gray (-> 5x5 blur) -> Sobel -> NMS -> class bit planes,
// one pass over the frame with everything between the 12-byte-per-lane HBM load and the bit-plane
// store held in registers.
//
// A wavefront owns a strip of 256 pixel columns (lane l holds 4 consecutive pixels, packed) and
// streams down a segment of rows.  Horizontal neighbours come from the adjacent lanes through
// DPP wave shifts, vertical neighbours from short register windows (5 rows of horizontal blur
// sums, 3 rows of Sobel partial sums, 3 rows of magnitudes).  The outer lanes 0 and 63 are halo
// (blur 2 + Sobel 1 + NMS 1 = 4 pixels = one lane each side), so a strip emits 248 columns = 31
// whole bytes of each bit plane per row.
//
// Arithmetic (bit-exact with OpenCV's 8U paths):
//   gray   (c0*k0 + c1*k1 + c2*k2 + 2^14) >> 15 as two v_dot4_u32_u8 over the low / high coefficient bytes
//   blur   [1 4 6 4 1] horizontally by v_dot4 (<= 4080), vertically in packed u16 (<= 65280), (v + 128) >> 8
//   Sobel  separable, packed i16: d = b(x+1) - b(x-1), s = b(x-1) + 2 b(x) + b(x+1),
//          dx = d(y-1) + 2 d(y) + d(y+1), dy = s(y+1) - s(y-1); magnitude |dx| + |dy|
//   borders: gray reflects (101) for the blur, the Sobel input replicates, magnitudes outside the image are 0
// Requires W % 4 == 0 (lanes are then wholly inside or wholly outside the image); other widths and
// gray input take the tiled kernel in kernels_edge.hip.
#include "common.h"

namespace lanecv {
namespace {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 U2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ ss2 S2(uint32_t x) { return __builtin_bit_cast(ss2, x); }
__device__ __forceinline__ uint32_t R32(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t R32(ss2 v) { return __builtin_bit_cast(uint32_t, v); }

// lane i receives lane i-1 (wave_shr:1) / lane i+1 (wave_shl:1); the wave's ends receive 0
__device__ __forceinline__ uint32_t from_left(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_right(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}
__device__ __forceinline__ int reflect101s(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

struct FrontArgs {
    const uint8_t *src;
    Geo g;
    uint32_t k_lo, k_hi;     // low / high bytes of the three 15-bit gray weights, in memory channel order
    int low, high;
    uint64_t *planeS, *planeW;
    uint8_t *gray_out, *blur_out;
    int rows_per_unit, nseg, nstrips, nunits;
};

struct Pair { uint32_t e, o; };   // 4 packed 16-bit values of a lane: e = (px0, px2), o = (px1, px3)

constexpr int STRIP = 248;        // output columns of a wavefront

template <bool BLUR>
__global__ __launch_bounds__(256) void k_front_stream(FrontArgs a) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= a.nunits) return;
    const int s = u % a.nstrips;
    const int sg = (u / a.nstrips) % a.nseg;
    const int f = u / (a.nstrips * a.nseg);
    const int W = a.g.W, H = a.g.H, TX = a.g.TX;
    const int X0 = STRIP * s - 4;                  // column of lane 0's first pixel
    const int y0 = sg * a.rows_per_unit, y1 = min(y0 + a.rows_per_unit, H);
    const int px = X0 + 4 * lane;
    const bool inimg = px >= 0 && px < W;
    const bool lb = px == -4;                      // the lane left of column 0 (strip 0 only)
    const int jW = (W - X0) >> 2;                  // the lane that starts at column W
    const bool rb = lane == jW;
    const bool has_border = X0 < 0 || jW <= 63;    // wave-uniform
    const bool emits = lane >= 1 && lane <= 62;
    const uint8_t *fsrc = a.src + (size_t)f * 3 * W * H;
    constexpr int DB = BLUR ? 2 : 0;               // rows the blur delays the pipeline by
    constexpr uint32_t M8 = 0x00FF00FFu;

    Pair Hw[5] = {};                               // horizontal blur sums of rows v-4 .. v
    Pair Dw[3] = {}, Sw[3] = {};                   // Sobel partial sums of Sobel-input rows b-2 .. b
    Pair Mw[3] = {};                               // magnitudes of rows m-2 .. m
    Pair DXc = {}, DYc = {}, DXn = {}, DYn = {};   // gradients of row n (centre of the NMS) and row m

    auto load_row = [&](int v, uint32_t &r0, uint32_t &r1, uint32_t &r2) {
        int rs = BLUR ? reflect101s(v, H) : (v < 0 ? 0 : (v >= H ? H - 1 : v));
        r0 = r1 = r2 = 0;
        if (inimg) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(fsrc + ((size_t)rs * W + px) * 3);
            r0 = p[0]; r1 = p[1]; r2 = p[2];
        }
    };

    int v = y0 - (DB + 2);
    const int vend = y1 - 1 + DB + 2;
    uint32_t n0, n1, n2;
    load_row(v, n0, n1, n2);
    for (; v <= vend; v++) {
        const uint32_t w0 = n0, w1 = n1, w2 = n2;
        if (v < vend) load_row(v + 1, n0, n1, n2);

        // ---- gray: 4 pixels -> one packed dword ------------------------------------------------
        const uint32_t p0 = w0, p1 = alignbyte(w1, w0, 3), p2 = alignbyte(w2, w1, 2), p3 = w2 >> 8;
        const uint32_t g0 = ((dot4(p0, a.k_hi, 0) << 8) + dot4(p0, a.k_lo, 1u << 14)) >> 15;
        const uint32_t g1 = ((dot4(p1, a.k_hi, 0) << 8) + dot4(p1, a.k_lo, 1u << 14)) >> 15;
        const uint32_t g2 = ((dot4(p2, a.k_hi, 0) << 8) + dot4(p2, a.k_lo, 1u << 14)) >> 15;
        const uint32_t g3 = ((dot4(p3, a.k_hi, 0) << 8) + dot4(p3, a.k_lo, 1u << 14)) >> 15;
        uint32_t G = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        if (a.gray_out && v >= y0 && v < y1 && inimg && emits)
            *reinterpret_cast<uint32_t *>(a.gray_out + (size_t)f * W * H + (size_t)v * W + px) = G;

        // ---- Sobel input row B (blurred row v-2, or the gray row itself) ------------------------
        uint32_t B;
        if constexpr (BLUR) {
            if (has_border) {   // reflect 101: columns -1,-2 mirror 1,2; columns W,W+1 mirror W-2,W-3
                const uint32_t GR0 = from_right(G), GL0 = from_left(G);
                if (lb) G = (GR0 & 0x00FF0000u) | ((GR0 << 16) & 0xFF000000u);
                if (rb) G = ((GL0 >> 16) & 0xFFu) | (GL0 & 0xFF00u);
            }
            const uint32_t GL = from_left(G), GR = from_right(G);
            const uint32_t q0 = alignbyte(G, GL, 2), q1 = alignbyte(G, GL, 3), q3 = alignbyte(GR, G, 1);
            const uint32_t h0 = dot4(q0, 0x04060401u, dot4(G, 0x00010000u, 0));
            const uint32_t h1 = dot4(q1, 0x04060401u, dot4(G, 0x01000000u, 0));
            const uint32_t h2 = dot4(G, 0x04060401u, dot4(GR, 0x00000001u, 0));
            const uint32_t h3 = dot4(q3, 0x04060401u, dot4(GR, 0x00000100u, 0));
            Hw[0] = Hw[1]; Hw[1] = Hw[2]; Hw[2] = Hw[3]; Hw[3] = Hw[4];
            Hw[4].e = h0 | (h2 << 16);
            Hw[4].o = h1 | (h3 << 16);
            const us2 ve = U2(Hw[0].e) + U2(Hw[4].e) + (U2(Hw[1].e) + U2(Hw[3].e)) * (unsigned short)4 +
                           U2(Hw[2].e) * (unsigned short)6;
            const us2 vo = U2(Hw[0].o) + U2(Hw[4].o) + (U2(Hw[1].o) + U2(Hw[3].o)) * (unsigned short)4 +
                           U2(Hw[2].o) * (unsigned short)6;
            const uint32_t be = R32((ve + (unsigned short)128) >> (unsigned short)8);
            const uint32_t bo = R32((vo + (unsigned short)128) >> (unsigned short)8);
            B = be | (bo << 8);
            const int bv = v - 2;
            if (a.blur_out && bv >= y0 && bv < y1 && inimg && emits)
                *reinterpret_cast<uint32_t *>(a.blur_out + (size_t)f * W * H + (size_t)bv * W + px) = B;
        } else {
            B = G;
        }
        if (has_border) {       // replicate: column -1 copies column 0, column W copies column W-1
            const uint32_t BR0 = from_right(B), BL0 = from_left(B);
            if (lb) B = (B & 0x00FFFFFFu) | (BR0 << 24);
            if (rb) B = (B & 0xFFFFFF00u) | (BL0 >> 24);
        }

        // ---- Sobel, horizontal part of row b = v - DB -------------------------------------------
        {
            const uint32_t BL = from_left(B), BR = from_right(B);
            const uint32_t bm = alignbyte(B, BL, 3), bp = alignbyte(BR, B, 1);
            const uint32_t ce = B & M8, co = (B >> 8) & M8;
            const uint32_t me = bm & M8, mo = (bm >> 8) & M8;
            const uint32_t pe = bp & M8, po = (bp >> 8) & M8;
            Dw[0] = Dw[1]; Dw[1] = Dw[2];
            Sw[0] = Sw[1]; Sw[1] = Sw[2];
            Dw[2].e = R32(S2(pe) - S2(me));
            Dw[2].o = R32(S2(po) - S2(mo));
            Sw[2].e = R32(U2(me) + U2(pe) + U2(ce) * (unsigned short)2);
            Sw[2].o = R32(U2(mo) + U2(po) + U2(co) * (unsigned short)2);
        }

        // ---- Sobel, vertical part, and magnitude of row m = b - 1 --------------------------------
        const int m_row = v - DB - 1;
        {
            const int ip = m_row == 0 ? 1 : 0, in = m_row == H - 1 ? 1 : 2;   // replicate above / below
            const Pair dP = ip ? Dw[1] : Dw[0], dN = in == 1 ? Dw[1] : Dw[2];
            const Pair sP = ip ? Sw[1] : Sw[0], sN = in == 1 ? Sw[1] : Sw[2];
            DXc = DXn; DYc = DYn;
            const ss2 dxe = S2(dP.e) + S2(Dw[1].e) * (short)2 + S2(dN.e);
            const ss2 dxo = S2(dP.o) + S2(Dw[1].o) * (short)2 + S2(dN.o);
            const ss2 dye = S2(sN.e) - S2(sP.e);
            const ss2 dyo = S2(sN.o) - S2(sP.o);
            DXn.e = R32(dxe); DXn.o = R32(dxo); DYn.e = R32(dye); DYn.o = R32(dyo);
            const bool live = inimg && m_row >= 0 && m_row < H;
            const ss2 mge = __builtin_elementwise_max(dxe, -dxe) + __builtin_elementwise_max(dye, -dye);
            const ss2 mgo = __builtin_elementwise_max(dxo, -dxo) + __builtin_elementwise_max(dyo, -dyo);
            Mw[0] = Mw[1]; Mw[1] = Mw[2];
            Mw[2].e = live ? R32(mge) : 0u;
            Mw[2].o = live ? R32(mgo) : 0u;
        }

        // ---- non-maximum suppression + thresholds of row n = m - 1 -----------------------------
        const int n_row = m_row - 1;
        if (n_row >= y0 && n_row < y1) {
            const uint32_t ce = Mw[1].e, co = Mw[1].o;
            const us2 mx = __builtin_elementwise_max(U2(ce), U2(co));
            const int mmax = max((int)(R32(mx) & 0xFFFFu), (int)(R32(mx) >> 16));
            uint32_t nib = 0;    // bits 0-3 strong, bits 8-11 weak
            if (__ballot(mmax > a.low) != 0) {
                int up[6], cu[6], dn[6];
                {
                    const uint32_t l0 = from_left(Mw[0].o), r0 = from_right(Mw[0].e);
                    const uint32_t l1 = from_left(co), r1 = from_right(ce);
                    const uint32_t l2 = from_left(Mw[2].o), r2 = from_right(Mw[2].e);
                    up[0] = l0 >> 16; up[1] = Mw[0].e & 0xFFFF; up[2] = Mw[0].o & 0xFFFF; up[3] = Mw[0].e >> 16; up[4] = Mw[0].o >> 16; up[5] = r0 & 0xFFFF;
                    cu[0] = l1 >> 16; cu[1] = ce & 0xFFFF; cu[2] = co & 0xFFFF; cu[3] = ce >> 16; cu[4] = co >> 16; cu[5] = r1 & 0xFFFF;
                    dn[0] = l2 >> 16; dn[1] = Mw[2].e & 0xFFFF; dn[2] = Mw[2].o & 0xFFFF; dn[3] = Mw[2].e >> 16; dn[4] = Mw[2].o >> 16; dn[5] = r2 & 0xFFFF;
                }
                int dxs[4], dys[4];
                dxs[0] = (short)(DXc.e & 0xFFFF); dxs[1] = (short)(DXc.o & 0xFFFF); dxs[2] = (int)DXc.e >> 16; dxs[3] = (int)DXc.o >> 16;
                dys[0] = (short)(DYc.e & 0xFFFF); dys[1] = (short)(DYc.o & 0xFFFF); dys[2] = (int)DYc.e >> 16; dys[3] = (int)DYc.o >> 16;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int m = cu[k + 1];
                    const int ax = abs(dxs[k]), ay = abs(dys[k]) << 15;
                    const int tg22x = ax * 13573;
                    const bool horiz = ay < tg22x;
                    const bool vert = !horiz && ay > tg22x + (ax << 16);
                    const bool opp = (dxs[k] ^ dys[k]) < 0;      // gradient along the anti-diagonal
                    const int a1 = horiz ? cu[k] : (vert ? up[k + 1] : (opp ? up[k + 2] : up[k]));
                    const int a2 = horiz ? cu[k + 2] : (vert ? dn[k + 1] : (opp ? dn[k] : dn[k + 2]));
                    const bool second_ge = horiz || vert;        // OpenCV: '>' then '>=' on the axes, '>' twice on diagonals
                    const bool keep = m > a.low && m > a1 && (second_ge ? m >= a2 : m > a2);
                    const bool strong = keep && m > a.high;
                    nib |= (strong ? 1u : 0u) << k;
                    nib |= ((keep && !strong) ? 1u : 0u) << (8 + k);
                }
            }
            // two lanes make a byte of each plane: odd lane 2i+1 takes lane 2i+2's nibble
            const uint32_t nb = from_right(nib);
            const uint32_t both = nib | (nb << 4);
            const int bi = STRIP / 8 * s + ((lane - 1) >> 1);
            if ((lane & 1) && lane <= 61 && bi < 8 * TX) {
                const size_t rowb = ((size_t)f * H + n_row) * TX * 8 + bi;
                reinterpret_cast<uint8_t *>(a.planeS)[rowb] = (uint8_t)(both & 0xFF);
                reinterpret_cast<uint8_t *>(a.planeW)[rowb] = (uint8_t)((both >> 8) & 0xFF);
            }
        }
    }
}

}  // namespace

bool front_stream_supported(Geo g) { return g.W % 4 == 0 && g.W >= 8; }

void launch_front_stream(const uint8_t *src, int code, bool blur, Geo g, int nframes, int low, int high,
                         uint64_t *planeS, uint64_t *planeW, uint8_t *gray_out, uint8_t *blur_out, int rows_per_unit,
                         hipStream_t st) {
    FrontArgs a;
    a.src = src; a.g = g; a.low = low; a.high = high;
    a.planeS = planeS; a.planeW = planeW; a.gray_out = gray_out; a.blur_out = blur_out;
    const uint32_t k0 = code == 6 ? 3735 : 9798, k1 = 19235, k2 = code == 6 ? 9798 : 3735;
    a.k_lo = (k0 & 255) | ((k1 & 255) << 8) | ((k2 & 255) << 16);
    a.k_hi = (k0 >> 8) | ((k1 >> 8) << 8) | ((k2 >> 8) << 16);
    a.nstrips = (TILE * g.TX + STRIP - 1) / STRIP;   // every byte of every plane row is written
    if (rows_per_unit <= 0) {
        // enough units to load every CU a few times over, few enough to keep the 8 halo rows cheap
        int want = 16384 / (nframes * a.nstrips > 0 ? nframes * a.nstrips : 1);
        if (want < 1) want = 1;
        rows_per_unit = (g.H + want - 1) / want;
        if (rows_per_unit < 64) rows_per_unit = 64;
    }
    if (rows_per_unit > g.H) rows_per_unit = g.H;
    a.rows_per_unit = rows_per_unit;
    a.nseg = (g.H + rows_per_unit - 1) / rows_per_unit;
    a.nunits = nframes * a.nseg * a.nstrips;
    dim3 grid((a.nunits + 3) / 4), block(256);
    if (blur) hipLaunchKernelGGL(k_front_stream<true>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_front_stream<false>, grid, block, 0, st, a);
}

}  // namespace lanecv
