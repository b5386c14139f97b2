// Streaming front kernel for gfx950: BGR -> gray (-> 5x5 blur) -> Sobel -> NMS -> class bit planes,
// one pass over the frame with everything between the 12-byte-per-lane HBM load and the bit-plane
// store held in registers.
//
// A wavefront owns a strip of 256 pixel columns (lane l holds 4 consecutive pixels) and streams down
// a segment of rows.  Horizontal neighbours come from the adjacent lanes through DPP wave shifts,
// vertical neighbours from three-row register windows.  The outer lanes 0 and 63 are halo (blur 2 +
// Sobel 1 + NMS 1 = 4 pixels = one lane each side), so a strip emits 248 columns = 31 whole bytes of
// each bit plane per row.
//
// Everything that depends only on the row is wave-uniform (the unit index goes through
// v_readfirstlane), so row tests are scalar branches.  The row loop is unrolled three times with the
// three-row windows indexed by the phase, so no register is ever copied to age a window.  Sixteen-bit
// values live as pairs: `e` = (pixel 0, pixel 2), `o` = (pixel 1, pixel 3) of the lane.
//
// Instruction choice.  The kernel is bound by vector instruction issue, not by memory: what counts is the number of
// vector instructions per row, then the number of scalar ones (the four SIMDs of a CU share one scalar unit).  On
// gfx950 a SIMD takes the plain 32-bit v_add / v_sub / v_and / v_or / v_xor / v_lshrrev / v_bitop3 at twice the rate
// of everything else used here (packed 16-bit math, v_dot4, v_perm, v_alignbyte, DPP moves, three-operand integer
// forms: 4 cycles per wavefront instruction against 2 -- tools/valu_rate.hip), but only among their like: mixed
// with the others, as here, they come out about a tenth cheaper.  So a pair of 16-bit halves is handled by ONE
// 32-bit instruction wherever the halves cannot disturb each other (sums of non-negative halves that stay below
// 2^16) and the count is the same as for the packed form; signed quantities stay packed.
//
// Arithmetic (bit-exact with OpenCV's 8U paths):
//   gray   with doubled weights 2k = 256 k_hi + k_lo:  S = sum c_i k_hi_i + ((sum c_i k_lo_i + 2^15) >> 8) by two
//          chained v_dot4_u32_u8; gray = byte 1 of S  (= (sum c_i k_i + 2^14) >> 15)
//   blur   [1 4 6 4 1] horizontally by v_dot4 (+8); vertically as four running pair sums
//          A(v) = H(v-1) + H(v), T(v) = A(v-1) + A(v), C(v) = T(v-1) + T(v), B(v) = C(v-1) + C(v) = 16 * ([1 4 6 4 1]
//          over rows v-4 .. v) + 128, all below 2^16; blurred value = B >> 8 (one v_perm takes both halves' high bytes)
//   Sobel  separable, packed i16: d = b(x+1) - b(x-1), s = b(x-1) + 2 b(x) + b(x+1) per row;
//          P(b) = d(b-1) + d(b), dx = P(m) + P(m+1); dy = s(m+1) - s(m-1); magnitude |dx| + |dy|
//   NMS    direction by the exact 16-bit form of OpenCV's 15-bit tan(22.5) test:
//          q = floor(|dx| * 13573 / 2^15) = (|dx| * 53 + (|dx| * 5 >> 8)) >> 7   (13573 = 53 * 256 + 5)
//          horizontal  <=>  |dy| < q + min(|dx|, 1);   vertical  <=>  |dy| > 2 |dx| + q
//          "m > x" is bit 15 of m + 0x7FFF - x (one 32-bit subtraction per pair); selections are v_bitop3
//   borders: gray reflects (101) for the blur, the Sobel input replicates, magnitudes outside the image are 0
// Requires W % 4 == 0 (lanes are then wholly inside or wholly outside the image); other widths and
// gray input take the tiled kernel in kernels_edge.hip.
#include "gate.h"
#include "hysteresis.h"
#include <type_traits>
#include <cstdlib>
#include <algorithm>

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
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t from_right(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}
// three-input bit operations (v_bitop3_b32): T = truth table with a = 0xF0, b = 0xCC, c = 0xAA
template <int T> __device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, T);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {   // mask ? a : b, bitwise
    return bitop3<0xCA>(mask, a, b);
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return bitop3<0xEA>(a, b, c); }   // (a & b) | c
__device__ __forceinline__ uint32_t or_and(uint32_t a, uint32_t b, uint32_t c) { return bitop3<0xA8>(a, b, c); }   // (a | b) & c
// 32-bit add / subtract on a pair of 16-bit halves that cannot carry or borrow into each other.  Inline assembly
// so that the optimiser does not merge them into three-operand or shifting forms (no fewer instructions, and the
// slower kind).  (Their results never feed a DPP move or a lane read directly, and they never consume a v_dot4 result
// directly: the compiler's hazard bookkeeping does not look inside assembly statements.)
__device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sub2(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the same with a wave-uniform operand held in a scalar register: k + v, k - v, v - k
__device__ __forceinline__ uint32_t add2k(uint32_t v, uint32_t k) {
    uint32_t r;
    asm("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t sub2k(uint32_t v, uint32_t k) {
    uint32_t r;
    asm("v_subrev_u32 %0, %1, %2" : "=v"(r) : "s"(k), "v"(v));
    return r;
}
// The next three are inline assembly so that the optimiser keeps them packed (it otherwise splits the
// halves again into 16-bit compares and selects).  None of their results feeds a DPP move directly.
// 0xFFFF in every half whose signed 16-bit value is negative
__device__ __forceinline__ uint32_t pksign(uint32_t a) {
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
// 1 in every half that is non-zero
__device__ __forceinline__ uint32_t pknz(uint32_t a) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
// packed 16-bit a * 2 + c (wraps; the same bits for signed and unsigned halves)
__device__ __forceinline__ uint32_t pkmad2(uint32_t a, uint32_t c) {
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pkadd(uint32_t a, uint32_t b) { return R32(U2(a) + U2(b)); }
__device__ __forceinline__ uint32_t pksub(uint32_t a, uint32_t b) { return R32(U2(a) - U2(b)); }
__device__ __forceinline__ uint32_t pkabs(uint32_t a) {
    const ss2 v = S2(a);
    return R32(__builtin_elementwise_max(v, -v));
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
    const uint8_t *src[MAX_PARTS];   // where the frames of each part start (one part without a frame set)
    FrameSet fs;
    Geo g;
    uint32_t k_lo, k_hi;     // low / high bytes of twice the three 15-bit gray weights, in memory channel order
    int low, high;
    uint64_t *planeS, *planeW;
    uint8_t *gray_out, *blur_out;
    uint32_t *wake;          // per 64x64 tile: set where a weak pixel is stored (the hysteresis kernel's work list)
    uint8_t *rowflags;       // [frame][strip][row]: the plane bytes last stored for that strip row were not all zero; or null
    int rows_per_unit, nseg, nstrips, nunits;
    // rows [0, row_split) form a region of their own with nseg_top units of rows_top rows (0: no such region); the
    // launch covers the units seg_first .. seg_first + nseg - 1 of a frame's nseg_top + lower units.  clear_top: the
    // launch only clears what a full pass left in the plane rows of the top region (see FrontSplit).
    int row_split, nseg_top, rows_top, seg_first, clear_top;
    // if set: only the frames whose word is non-zero are done, and the grid is not one workgroup per unit of every
    // frame but REDO_GROUPS per unit of ONE frame (nearly always no frame is flagged: 20 000 workgroups that only
    // look at a flag would cost more than the check is worth); group j does the flagged frames j, j + GROUPS, ...
    const int *only_frames;
    int nframes;
    // if set: the launch covers, per frame, only the nlist units listed (unit = segment | strip << 8) -- the units an
    // ROI needs (FrontSplit) or, for the clearing launch, the others
    const uint16_t *unit_list;
    int nlist;
    int wg_chunk;            // logical workgroups per XCD (see the kernel)
    const int *gate_tab;     // GATE only: the gate's tables in device memory (gate_tables_host)
    GateDev gate;
    // k_front_lead only: the first lead_blocks workgroups run the hysteresis of another batch (HystLaunch in common.h),
    // one frame each; the units follow behind them
    int lead_blocks;
    HystArgs lead;
    uint32_t *lead_px;
    uint8_t shift8[FRONT_SHIFT_SEGS];   // segment sg's strips start at column 248 * strip - 8 * shift8[sg] (FrontShift; all zero: the plain grid)
};

struct Pair { uint32_t e, o; };   // 4 packed 16-bit values of a lane: e = (px0, px2), o = (px1, px3)

constexpr int STRIP = 248;        // output columns of a wavefront
constexpr int REDO_GROUPS = 8;    // workgroups per unit of a launch that does flagged frames only

template <int P> using Phase = std::integral_constant<int, P>;

// GATE != 0: the colour gate of the HSV-gated script in front of the gray conversion -- a pixel outside both
// HSV boxes counts as black, so its gray value is 0 (gate_keep4 in gate.h).  Its tables are read from global
// memory (3 KiB, resident in every L1): the gray conversion of a row group sits between the wait for the group's
// pixels and the issue of the next group's loads, so the lookups of its 12 pixels go out together and cost one
// cache round trip per group.  In LDS they would cost less per lookup, but beside a HoughLinesP workgroup a CU
// has about 3 KiB of LDS left: one front-end workgroup per CU instead of five.
// INTER: also store the gray (and blurred) images -- parity runs only; the variant without them keeps their
// per-lane addresses out of the register budget.
// One unit: strip s (its first output column is 248 s - 8 d8, d8 the segment's shift), rows [y0, y1) of frame f.  Two properties of a unit are compile-time switches, so that the
// three quarters of the units that have neither carry none of the per-row tests (scalar instructions, which the
// four SIMDs of a CU share one issue port for):
//   HB  the strip holds a lane left of column 0 or right of column W - 1 (first and last strip of a frame row)
//   FE  the unit touches the top or bottom of the frame: rows to load may lie outside it, Sobel rows replicate,
//       magnitude rows outside the image are zero
//   DEC3  the frames arrive three times as wide and as high and are reduced by three first, as cv2.resize's bilinear
//       interpolation does it: every sample point falls on a source pixel with the whole weight, so pixel (x, y) IS
//       source pixel (3x + 1, 3y + 1).  The lane reads its four pixels out of 36 source bytes of every third row.
template <bool BLUR, int GATE, bool INTER, bool HB, bool FE, bool DEC3 = false>   // GATE: 0 none, 1 the colour gate
__device__ __forceinline__ void front_unit(const FrontArgs &a, const int f, const int s, const int d8, const int y0, const int y1) {
    const int *__restrict__ gtab = a.gate_tab;
    const int lane = threadIdx.x & 63;
    const int W = a.g.W, H = a.g.H, TX = a.g.TX;
    const int X0 = STRIP * s - 4 - 8 * d8;         // column of lane 0's first pixel
    const int px = X0 + 4 * lane;
    const bool inimg = px >= 0 && px < W;
    const bool lb = px == -4;                      // the lane left of column 0 (strip 0 only)
    const int jW = (W - X0) >> 2;                  // the lane that starts at column W
    const bool rb = lane == jW;
    constexpr bool has_border = HB;
    static_assert(!DEC3 || GATE == 0, "the colour gate is not combined with the fused reduction");
    constexpr int RS = DEC3 ? 3 : 1;               // source pixels per pixel, each way
    constexpr int NL = DEC3 ? 6 : 3;               // dwords a lane keeps of a source row
    const size_t frame_bytes = (size_t)3 * (RS * W) * (RS * H);
    const int part = a.fs.part(f);
    const uint8_t *fsrc = a.src[part] + (size_t)(f - part * a.fs.per_part) * frame_bytes;
    // lanes outside the image load a valid pixel group instead of nothing: what they hold never reaches
    // an output (border lanes are rewritten below, magnitudes outside the image are forced to zero)
    const uint32_t lane_off = (uint32_t)(3 * RS) * (uint32_t)min(max(px, 0), W - 4);
    constexpr int DB = BLUR ? 2 : 0;               // rows the blur delays the pipeline by
    constexpr uint32_t M8 = 0x00FF00FFu;
    // this lane's byte of the two bit planes (odd lanes 1..61 store: their nibble and the next lane's)
    const int bi = STRIP / 8 * s - d8 + ((lane - 1) >> 1);
    const bool stores = (lane & 1) && lane <= 61 && bi >= 0 && bi < 8 * TX;
    const bool emits = lane >= 1 && lane <= 62;

    // thresholds: no magnitude exceeds 2040, so anything above behaves like 2041
    const uint32_t low2 = (uint32_t)min(max(a.low, 0), 2041) * 0x10001u;
    const uint32_t high2 = (uint32_t)min(max(a.high, 0), 2041) * 0x10001u;
    // m + trigK has bit 15 set  <=>  m > low.  Only the emitting lanes decide whether a row needs the NMS block: the
    // halo lanes' magnitudes come from a truncated neighbourhood and would trigger every row, so they add nothing.
    const uint32_t trigK = emits ? 0x7FFF7FFFu - low2 : 0u;
    // running pair sums of the vertical blur (previous row's values)
    Pair Hp = {}, Ap = {}, Tp = {}, Cp = {};
    // Sobel: previous row's d and pair sum P; three-row windows (slot = phase of the iteration that wrote it) of s
    // and of the magnitudes; gradients of the current and the previous magnitude row
    Pair Dp = {}, Pp = {};
    Pair Sw[3] = {};
    Pair Mw[3] = {};
    Pair DXw[3] = {}, DYw[3] = {};

    uint32_t L[3][NL] = {};                        // source rows of a group: consumed by the gray conversion at the
                                                   // top of the group, then reloaded with the next group's rows
    uint32_t pend[3] = {};                         // plane bytes of the group just finished (strong | weak << 8)
    // A row is one 12-byte buffer load per lane: the frame as the buffer, the lane's column offset in a vector
    // register that never changes, the row's offset in a scalar register -- no vector arithmetic per load.
    const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fsrc), 0, (int)frame_bytes, 0x00020000);
    auto load_row = [&](int v, uint32_t (&r)[NL]) {
        const int rs = !FE ? v : BLUR ? reflect101s(v, H) : (v < 0 ? 0 : (v >= H ? H - 1 : v));
        if constexpr (DEC3) {
            // pixel i of the lane is source bytes 9 i + 3 .. 9 i + 5 of the lane's 36: dwords 0-1, 3, 5 and 7-8 hold them
            const uint32_t row = (uint32_t)(3 * rs + 1) * (uint32_t)(9 * W);
            const auto q0 = __builtin_amdgcn_raw_buffer_load_b64(frame, lane_off, row, 0);
            const uint32_t q1 = __builtin_amdgcn_raw_buffer_load_b32(frame, lane_off + 12u, row, 0);
            const uint32_t q2 = __builtin_amdgcn_raw_buffer_load_b32(frame, lane_off + 20u, row, 0);
            const auto q3 = __builtin_amdgcn_raw_buffer_load_b64(frame, lane_off + 28u, row, 0);
            r[0] = q0[0]; r[1] = q0[1]; r[2] = q1; r[3] = q2; r[4] = q3[0]; r[5] = q3[1];
        } else {
            const auto q = __builtin_amdgcn_raw_buffer_load_b96(frame, lane_off, (uint32_t)rs * (uint32_t)(3 * W), 0);
            r[0] = q[0]; r[1] = q[1]; r[2] = q[2];
        }
    };
    // Dirty tracking of the plane rows (rowflags != null; the planes then start all zero and only this kernel and the
    // hysteresis kernel, which adds strong bits where weak ones are, write them): strip rows whose bytes were zero
    // last time and are zero now are not stored again.  Most are, and a 31-byte store costs one or two whole sectors.
    // The unit of the bookkeeping is the group of three rows the loop below works in (one flag, one test, up to
    // three rows stored): per row it cost as much scalar and vector work as a sixth of the arithmetic.  A group's
    // flag sits at the group's middle row, clamped into the unit (groups differ in that row: see the loop).
    const size_t flag_base = ((size_t)f * a.nstrips + s) * H;
    auto flag_index = [&](int n0) -> size_t { return flag_base + (size_t)min(max(n0 + 1, y0), y1 - 1); };
    auto load_flag = [&](int n0) -> uint32_t {
        if (a.rowflags && n0 + 2 >= y0 && n0 < y1) return a.rowflags[flag_index(n0)];
        return 1u;
    };
    auto store_group = [&](int n0, const uint32_t (&pd)[3], uint32_t pflag) {
        if (n0 + 2 >= y0 && n0 < y1) {
            const bool nz = __builtin_amdgcn_ballot_w64(stores && ((pd[0] | pd[1] | pd[2]) & 0xFFFFu) != 0) != 0;
            const bool was = __builtin_amdgcn_readfirstlane(pflag) != 0;
            if (nz || was) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int n_row = n0 + k;
                    if (n_row < y0 || n_row >= y1) continue;
                    uint8_t *rowS = reinterpret_cast<uint8_t *>(a.planeS) + ((size_t)f * H + n_row) * TX * 8;
                    uint8_t *rowW = reinterpret_cast<uint8_t *>(a.planeW) + ((size_t)f * H + n_row) * TX * 8;
                    if (stores) {
                        const uint32_t wb = (pd[k] >> 8) & 0xFF;
                        rowS[bi] = (uint8_t)(pd[k] & 0xFF);
                        rowW[bi] = (uint8_t)wb;
                        if (wb) a.wake[((size_t)f * a.g.TY + (n_row >> 6)) * TX + (bi >> 3)] = 1u;
                    }
                }
            }
            if (a.rowflags && nz != was && lane == 0) a.rowflags[flag_index(n0)] = nz ? 1 : 0;
        }
    };
    uint32_t pf = 1u;                              // flag of the group the next group top stores

    const int vend = y1 - 1 + DB + 2;

    // ---- gray: 4 pixels (12 bytes) of source row v -> the gray bytes packed in one dword (BLUR, GATE, INTER) or
    // ---- directly as the pairs e = (px0, px2), o = (px1, px3) ------------------------------------------------
    constexpr bool PACKED_GRAY = BLUR || GATE != 0 || INTER;
    auto gray_row = [&](const uint32_t (&r)[NL], const int v) -> Pair {
        // the four pixels as dwords whose low three bytes are the channels (pixel 2 and, without the reduction, pixel
        // 3 sit a byte up in their dword: they take the weights moved up a byte instead of a shifted copy)
        uint32_t p0, p1, p2, p3, k2lo = a.k_lo, k2hi = a.k_hi, k3lo = a.k_lo << 8, k3hi = a.k_hi << 8;
        if constexpr (DEC3) {
            p0 = alignbyte(r[1], r[0], 3); p1 = r[2]; p2 = r[3]; p3 = alignbyte(r[5], r[4], 2);
            k2lo = a.k_lo << 8; k2hi = a.k_hi << 8; k3lo = a.k_lo; k3hi = a.k_hi;
        } else {
            p0 = r[0]; p1 = alignbyte(r[1], r[0], 3); p2 = alignbyte(r[2], r[1], 2); p3 = r[2];
        }
        // the four low-byte sums first, then their shifts, then the four high-byte sums: a v_dot4 result is not
        // ready for a different kind of instruction at once
        const uint32_t l0 = dot4(p0, a.k_lo, 1u << 15), l1 = dot4(p1, a.k_lo, 1u << 15);
        const uint32_t l2 = dot4(p2, k2lo, 1u << 15), l3 = dot4(p3, k3lo, 1u << 15);
        const uint32_t t0 = dot4(p0, a.k_hi, l0 >> 8), t1 = dot4(p1, a.k_hi, l1 >> 8);
        const uint32_t t2 = dot4(p2, k2hi, l2 >> 8), t3 = dot4(p3, k3hi, l3 >> 8);
        if constexpr (GATE != 0) p3 >>= 8;
        Pair g;
        if constexpr (PACKED_GRAY) {
            uint32_t G = __builtin_amdgcn_perm(t1, t0, 0x0c0c0501u) | __builtin_amdgcn_perm(t3, t2, 0x05010c0cu);
            if constexpr (GATE != 0) G &= gate_keep4(p0, p1, p2, p3, gtab, a.gate);
            if (INTER && a.gray_out && v >= y0 && v < y1) {
                if (inimg && emits) *reinterpret_cast<uint32_t *>(a.gray_out + (size_t)f * W * H + (size_t)v * W + px) = G;
            }
            g.e = G; g.o = 0;
        } else {
            g.e = __builtin_amdgcn_perm(t2, t0, 0x0c050c01u);
            g.o = __builtin_amdgcn_perm(t3, t1, 0x0c050c01u);
        }
        return g;
    };

    auto step = [&](auto phase, const int v, const Pair gray) {
        constexpr int P = decltype(phase)::value, PREV = (P + 2) % 3, OLD = (P + 1) % 3;

        // ---- Sobel input row b = v - DB (blurred row v-2, or the gray row itself) as pairs -------------------
        uint32_t be, bo;
        if constexpr (BLUR) {
            uint32_t G = gray.e;
            if constexpr (has_border) {   // reflect 101: columns -1,-2 mirror 1,2; columns W,W+1 mirror W-2,W-3
                const uint32_t GR0 = from_right(G), GL0 = from_left(G);
                if (lb) G = (GR0 & 0x00FF0000u) | ((GR0 << 16) & 0xFF000000u);
                if (rb) G = ((GL0 >> 16) & 0xFFu) | (GL0 & 0xFF00u);
            }
            const uint32_t GL = from_left(G), GR = from_right(G);
            const uint32_t q0 = alignbyte(G, GL, 2), q1 = alignbyte(G, GL, 3), q3 = alignbyte(GR, G, 1);
            const uint32_t h0 = dot4(q0, 0x04060401u, dot4(G, 0x00010000u, 8));
            const uint32_t h1 = dot4(q1, 0x04060401u, dot4(G, 0x01000000u, 8));
            const uint32_t h2 = dot4(G, 0x04060401u, dot4(GR, 0x00000001u, 8));
            const uint32_t h3 = dot4(q3, 0x04060401u, dot4(GR, 0x00000100u, 8));
            Pair Hc, Ac, Tc, Cc;
            Hc.e = h0 | (h2 << 16);
            Hc.o = h1 | (h3 << 16);
            // running pair sums (see the head of the file); every sum stays below 2^16 in each half
            Ac.e = add2(Hp.e, Hc.e); Ac.o = add2(Hp.o, Hc.o);
            Tc.e = add2(Ap.e, Ac.e); Tc.o = add2(Ap.o, Ac.o);
            Cc.e = add2(Tp.e, Tc.e); Cc.o = add2(Tp.o, Tc.o);
            const uint32_t Be = add2(Cp.e, Cc.e), Bo = add2(Cp.o, Cc.o);
            Hp = Hc; Ap = Ac; Tp = Tc; Cp = Cc;
            be = __builtin_amdgcn_perm(0u, Be, 0x0c030c01u);        // the high byte of each half = B >> 8
            bo = __builtin_amdgcn_perm(0u, Bo, 0x0c030c01u);
            const int bv = v - 2;
            if (INTER && a.blur_out && bv >= y0 && bv < y1) {
                if (inimg && emits) *reinterpret_cast<uint32_t *>(a.blur_out + (size_t)f * W * H + (size_t)bv * W + px) = be | (bo << 8);
            }
        } else if constexpr (PACKED_GRAY) {
            be = gray.e & M8;
            bo = (gray.e >> 8) & M8;
        } else {
            be = gray.e;
            bo = gray.o;
        }
        if constexpr (has_border) {       // replicate: column -1 copies column 0, column W copies column W-1
            const uint32_t r = from_right(be), l = from_left(bo);
            if (lb) bo = (bo & 0x0000FFFFu) | (r << 16);
            if (rb) be = (be & 0xFFFF0000u) | (l >> 16);
        }

        // ---- Sobel, horizontal part of row b: d = b(x+1) - b(x-1) (signed), s = b(x-1) + 2 b(x) + b(x+1) ------
        const int b_row = v - DB;
        Pair Dc;
        {
            const uint32_t le = alignbyte(bo, from_left(bo), 2);    // (px-1, px1): left neighbours of the e pixels
            const uint32_t ro = alignbyte(from_right(be), be, 2);   // (px2, px4): right neighbours of the o pixels
            Dc.e = pksub(bo, le);
            Dc.o = pksub(ro, be);
            Sw[P].e = pkmad2(be, le + bo);
            Sw[P].o = pkmad2(bo, be + ro);
        }
        // (the empty asm statements keep these two rare cases scalar branches instead of per-row selects)
        if (FE && b_row == 0) { asm volatile(""); Dp = Dc; Sw[PREV] = Sw[P]; }       // row -1 replicates row 0
        if (FE && b_row == H) { asm volatile(""); Dc = Dp; Sw[P] = Sw[PREV]; }       // row H replicates row H-1

        // ---- Sobel, vertical part, and magnitude of row m = b - 1 --------------------------------------------
        const int m_row = b_row - 1;
        {
            Pair Pc;
            Pc.e = pkadd(Dp.e, Dc.e); Pc.o = pkadd(Dp.o, Dc.o);     // d(b-1) + d(b)
            DXw[P].e = pkadd(Pp.e, Pc.e);                           // dx = d(m-1) + 2 d(m) + d(m+1)
            DXw[P].o = pkadd(Pp.o, Pc.o);
            DYw[P].e = pksub(Sw[P].e, Sw[OLD].e);                   // dy = s(m+1) - s(m-1)
            DYw[P].o = pksub(Sw[P].o, Sw[OLD].o);
            Dp = Dc; Pp = Pc;
            Mw[P].e = pkabs(DXw[P].e) + pkabs(DYw[P].e);            // <= 2040: a plain add cannot carry between the halves
            Mw[P].o = pkabs(DXw[P].o) + pkabs(DYw[P].o);
            // magnitudes outside the image are zero: whole rows above / below it (rare, a scalar branch) and the
            // lanes left / right of it in the first and last strips
            if (FE && (m_row < 0 || m_row >= H)) { asm volatile(""); Mw[P].e = 0u; Mw[P].o = 0u; }
            if constexpr (has_border) { Mw[P].e = inimg ? Mw[P].e : 0u; Mw[P].o = inimg ? Mw[P].o : 0u; }
        }

        // ---- non-maximum suppression + thresholds of row n = m - 1 -----------------------------
        // (not for the rows above y0 the unit warms up on: their half-filled windows look like strong gradients and
        // would run the NMS block every time.  The two rows below y1 - 1 it drains on do come through here: what
        // they yield is never stored, and their occasional NMS block costs less than a second test per row.)
        if (m_row > y0) {
            const uint32_t ce = Mw[PREV].e, co = Mw[PREV].o;
            uint32_t nib = 0;    // bits 0-3 strong, bits 8-11 weak
            const uint32_t over = or_and(ce + trigK, co + trigK, 0x80008000u);   // bit 15 of a half: its m > low
            if (__builtin_amdgcn_ballot_w64(over != 0) != 0) {
                // one pair of pixels: m against its two neighbours along the gradient (all magnitudes biased alike)
                auto nms = [&](uint32_t m, uint32_t dx, uint32_t dy, uint32_t cl, uint32_t cr, uint32_t ul, uint32_t uc,
                               uint32_t ur, uint32_t dl, uint32_t dc, uint32_t dr, uint32_t &sflag, uint32_t &wflag) {
                    const uint32_t ax = pkabs(dx), ay = pkabs(dy);          // <= 1020
                    const uint32_t ax5 = R32(U2(ax) * (unsigned short)5);
                    const uint32_t q = (R32(U2(ax) * (unsigned short)53 + U2((ax5 >> 8) & M8)) >> 7) & 0x01FF01FFu;
                    const uint32_t hz = add2(q, pknz(ax));
                    const uint32_t vt = add2(add2(ax, ax), q);
                    const uint32_t Hm = pksign(pksub(ay, hz));      // 0xFFFF where |dy| < hz
                    const uint32_t Vm = pksign(pksub(vt, ay));      // 0xFFFF where |dy| > vt
                    const uint32_t Om = pksign(dx ^ dy);            // 0xFFFF where dx, dy differ in sign
                    const uint32_t a1 = bfi(Hm, cl, bfi(Vm, uc, bfi(Om, ur, ul)));
                    const uint32_t a2 = bfi(Hm, cr, bfi(Vm, dc, bfi(Om, dl, dr)));
                    // bit 15 of (m | 0x8000) - 1 - x  <=>  m > x; on the axes the second neighbour is '>=': m + 1 > x
                    const uint32_t mk = add2k(m, 0x7FFF7FFFu);
                    const uint32_t mk2 = add2(mk, or_and(Hm, Vm, 0x00010001u));
                    const uint32_t keep = bitop3<0x80>(sub2k(mk, low2), sub2(mk, a1), sub2(mk2, a2));
                    const uint32_t fh = sub2k(mk, high2);
                    sflag = keep & fh;                              // bits 15 and 31
                    wflag = bitop3<0x30>(keep, fh, 0u);             // keep & ~fh
                };
                // magnitudes of the neighbouring lanes' pixels, for the three rows: .e = left neighbours of the e
                // pixels (px-1, px1), .o = right neighbours of the o pixels (px2, px4)
                Pair Nw[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    Nw[k].e = alignbyte(Mw[k].o, from_left(Mw[k].o), 2);
                    Nw[k].o = alignbyte(from_right(Mw[k].e), Mw[k].e, 2);
                }
                uint32_t se, we, so, wo;
                nms(ce, DXw[PREV].e, DYw[PREV].e, Nw[PREV].e, co, Nw[OLD].e, Mw[OLD].e, Mw[OLD].o, Nw[P].e, Mw[P].e, Mw[P].o, se, we);
                nms(co, DXw[PREV].o, DYw[PREV].o, ce, Nw[PREV].o, Mw[OLD].e, Mw[OLD].o, Nw[OLD].o, Mw[P].e, Mw[P].o, Nw[P].o, so, wo);
                // flags sit in bits 15 (px0 / px1) and 31 (px2 / px3): strong nibble in bits 0-3, weak in bits 8-11
                const uint32_t xs = and_or(se >> 15, 0x00010001u, (so >> 14) & 0x00020002u);
                const uint32_t xw = and_or(we >> 15, 0x00010001u, (wo >> 14) & 0x00020002u);
                nib = ((xs | (xs >> 14)) & 0xFu) | (((xw | (xw >> 14)) & 0xFu) << 8);
                // two lanes make a byte of each plane: odd lane 2i+1 takes lane 2i+2's nibble
                nib |= from_right(nib) << 4;
            }
            pend[P] = nib;
        }
    };

    // Memory traffic is issued once per group of three rows, near the top of the group: the gray conversion
    // first consumes the group's source rows, then the loads of the NEXT group's rows go out into the same
    // registers, together with the plane bytes of the PREVIOUS group.  The only wait for memory (the hardware
    // counts loads and stores in one counter) then covers operations issued a whole group earlier, so neither
    // the HBM latency nor the store acknowledgements stall the arithmetic.
    int v = y0 - (DB + 2);
    load_row(v, L[0]); load_row(v + 1, L[1]); load_row(v + 2, L[2]);
    for (; v <= vend; v += 3) {
        const Pair G0 = gray_row(L[0], v), G1 = gray_row(L[1], v + 1), G2 = gray_row(L[2], v + 2);
        // unconditional (past the unit's last row the last row is fetched again, from cache): a conditional reload
        // makes the compiler keep a second copy of the nine registers
        load_row(min(v + 3, vend), L[0]); load_row(min(v + 4, vend), L[1]); load_row(min(v + 5, vend), L[2]);
        store_group(v - 3 - DB - 2, pend, pf);
        pf = load_flag(v - DB - 2);
        step(Phase<0>{}, v, G0);
        step(Phase<1>{}, v + 1, G1);
        step(Phase<2>{}, v + 2, G2);
    }
    store_group(v - 3 - DB - 2, pend, pf);
}

// A unit of the skipped top region: nothing is computed, but plane rows an earlier full pass over this frame slot
// stored there (flagged in rowflags, group by group as front_unit does it) are zeroed again, so that the planes
// above row_split read as empty.
template <bool BLUR>
__device__ __forceinline__ void clear_unit(const FrontArgs &a, const int f, const int s, const int d8, const int y0, const int y1) {
    if (!a.rowflags) return;
    constexpr int DB = BLUR ? 2 : 0;
    const int lane = threadIdx.x & 63, H = a.g.H, TX = a.g.TX;
    const int bi = STRIP / 8 * s - d8 + ((lane - 1) >> 1);
    const bool stores = (lane & 1) && lane <= 61 && bi >= 0 && bi < 8 * TX;
    const size_t flag_base = ((size_t)f * a.nstrips + s) * H;
    const int vend = y1 - 1 + DB + 2;
    for (int v = y0 - (DB + 2); v <= vend + 3; v += 3) {
        const int n0 = v - 3 - DB - 2;
        if (n0 + 2 < y0 || n0 >= y1) continue;
        const size_t fi = flag_base + (size_t)min(max(n0 + 1, y0), y1 - 1);
        if (a.rowflags[fi] == 0) continue;
        for (int k = 0; k < 3; k++) {
            const int n_row = n0 + k;
            if (n_row < y0 || n_row >= y1 || !stores) continue;
            reinterpret_cast<uint8_t *>(a.planeS)[((size_t)f * H + n_row) * TX * 8 + bi] = 0;
            reinterpret_cast<uint8_t *>(a.planeW)[((size_t)f * H + n_row) * TX * 8 + bi] = 0;
        }
        if (lane == 0) a.rowflags[fi] = 0;
    }
}

// The kernel proper.  (k_front_again is the same code under a name of its own: the launch that classifies flagged
// frames a second time, so that profiles tell the two apart.)
template <bool BLUR, int GATE, bool INTER, bool DEC3>
__device__ __forceinline__ void front_stream_body(const FrontArgs &a) {
    // Workgroups go round-robin over the 8 XCDs, each with its own L2.  Neighbouring strips share the cache lines
    // their rows meet in (a strip row is 744 bytes, not a multiple of 128) and the plane bytes they write, so
    // consecutive units are handed to ONE XCD: workgroup b works on logical workgroup (b % 8) * chunk + b / 8.
    const unsigned bid = blockIdx.x - (unsigned)a.lead_blocks;
    const int wgl = (int)(bid & 7u) * a.wg_chunk + (int)(bid >> 3);
    const int u = __builtin_amdgcn_readfirstlane(wgl * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));   // wave-uniform
    if (u >= a.nunits) return;
    const int per_frame = a.unit_list ? a.nlist : a.nstrips * a.nseg;
    const int ui = u % per_frame;
    int s, sg;
    if (a.unit_list) { const int e = a.unit_list[ui]; sg = e & 255; s = e >> 8; }
    else { s = ui % a.nstrips; sg = a.seg_first + ui / a.nstrips; }
    const int H = a.g.H;
    constexpr int DB = BLUR ? 2 : 0;
    int y0, y1;
    if (sg < a.nseg_top) {
        y0 = sg * a.rows_top; y1 = min(y0 + a.rows_top, a.row_split);
    } else {
        y0 = a.row_split + (sg - a.nseg_top) * a.rows_per_unit; y1 = min(y0 + a.rows_per_unit, H);
    }
    const int d8 = a.shift8[sg & (FRONT_SHIFT_SEGS - 1)];
    if (STRIP * s - 8 * d8 >= TILE * a.g.TX) return;    // (a shifted segment's spare strip, past the end of the plane row)
    const bool hb = STRIP * s - 8 * d8 - 4 < 0 || ((a.g.W - (STRIP * s - 8 * d8 - 4)) >> 2) <= 63;
    const bool fe = y0 == 0 || y1 + DB + 1 >= H;        // rows y0 - DB - 2 .. y1 + DB + 1 are loaded
    auto unit = [&](const int f) {
        if (!a.fs.has(f)) return;
        if (hb) {
            if (fe) front_unit<BLUR, GATE, INTER, true, true, DEC3>(a, f, s, d8, y0, y1);
            else front_unit<BLUR, GATE, INTER, true, false, DEC3>(a, f, s, d8, y0, y1);
        } else {
            if (fe) front_unit<BLUR, GATE, INTER, false, true, DEC3>(a, f, s, d8, y0, y1);
            else front_unit<BLUR, GATE, INTER, false, false, DEC3>(a, f, s, d8, y0, y1);
        }
    };
    // every frame: the workgroup's one frame.  Flagged frames only: this wavefront's share of the flags in one
    // load, then frame after frame (one call site either way: the unit's code is inlined there)
    const int j = u / per_frame;
    for (int base = j; base < a.nframes; base += REDO_GROUPS * 64) {
        uint64_t todo = 1;
        if (a.only_frames) {
            const int fl = base + REDO_GROUPS * (int)(threadIdx.x & 63);
            todo = __ballot(fl < a.nframes && a.only_frames[fl] != 0);
        }
        while (todo) {
            const int f = base + REDO_GROUPS * __builtin_ctzll(todo);
            todo &= todo - 1;
            unit(f);
        }
        if (!a.only_frames) break;
    }
}

template <bool BLUR, int GATE, bool INTER, bool DEC3 = false>
__global__ __launch_bounds__(256) void k_front_stream(FrontArgs a) { front_stream_body<BLUR, GATE, INTER, DEC3>(a); }
template <bool BLUR, int GATE, bool DEC3>
__global__ __launch_bounds__(256) void k_front_again(FrontArgs a) { front_stream_body<BLUR, GATE, false, DEC3>(a); }
// The front kernel with the hysteresis of the batch before as its first workgroups (LEAD_THREADS threads each, a frame
// each): those start at once and run beside the units, instead of a hysteresis kernel between two front kernels -- which is
// latency-bound, leaves most of the GPU idle, and cannot run beside a front kernel as a launch of its own (its large
// workgroups find no room while the front kernel's small ones keep arriving).  The ring check's pixel list lives in
// device memory here: the kernel's LDS (two words and the workgroup reduction's scratch) is charged to every
// workgroup of the launch, and beside HoughLinesP workgroups a CU has only a few KiB left.
constexpr int LEAD_THREADS = 128;
template <bool BLUR, int GATE, bool DEC3>
__global__ __launch_bounds__(LEAD_THREADS) void k_front_lead(FrontArgs a) {
    if ((int)blockIdx.x < a.lead_blocks) {
        __shared__ int any, count;
        const int f = (int)blockIdx.x;
        hysteresis_one_frame<LEAD_THREADS, true>(a.lead, f, any, count, a.lead_px + (size_t)f * RING_LIST);
        return;
    }
    front_stream_body<BLUR, GATE, false, DEC3>(a);
}

// clears, for the flagged frames, the plane rows above row_split that a full pass stored (FrontSplit mode 2)
template <bool BLUR>
__global__ __launch_bounds__(256) void k_front_clear(FrontArgs a) {
    const int wgl = (int)(blockIdx.x & 7u) * a.wg_chunk + (int)(blockIdx.x >> 3);
    const int u = __builtin_amdgcn_readfirstlane(wgl * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
    if (u >= a.nunits) return;
    const int e = a.unit_list[u % a.nlist], sg = e & 255, s = e >> 8, j = u / a.nlist;
    int y0, y1;
    if (sg < a.nseg_top) { y0 = sg * a.rows_top; y1 = min(y0 + a.rows_top, a.row_split); }
    else { y0 = a.row_split + (sg - a.nseg_top) * a.rows_per_unit; y1 = min(y0 + a.rows_per_unit, a.g.H); }
    const int d8 = a.shift8[sg & (FRONT_SHIFT_SEGS - 1)];
    if (STRIP * s - 8 * d8 >= TILE * a.g.TX) return;
    for (int base = j; base < a.nframes; base += REDO_GROUPS * 64) {
        const int fl = base + REDO_GROUPS * (int)(threadIdx.x & 63);
        uint64_t todo = __ballot(fl < a.nframes && a.only_frames[fl] != 0);
        while (todo) {
            const int f = base + REDO_GROUPS * __builtin_ctzll(todo);
            if (a.fs.has(f)) clear_unit<BLUR>(a, f, s, d8, y0, y1);
            todo &= todo - 1;
        }
    }
}

}  // namespace

bool front_stream_supported(Geo g) { return g.W % 4 == 0 && g.W >= 8; }
bool front_takes_lead(Geo g, int in_kind) {
    static const char *force_tiled = getenv("LANECV_FRONT_TILED");
    return in_kind == 1 && front_stream_supported(g) && !(force_tiled && force_tiled[0] == '1');
}
int front_stream_strips(Geo g) { return (TILE * g.TX + STRIP - 1) / STRIP; }

// rows per unit of the streaming kernel for batches of up to `max_batch` frames: enough units to load every CU many
// times over (80 per CU measured best beside HoughLinesP: finer units balance better around its resident workgroups),
// few enough to keep the 8 halo rows cheap.  A pipeline fixes it once: the kernel's change tracking is kept per unit.
int front_stream_rows_per_unit(Geo g, int max_batch, int rows_classified) {
    const int nstrips = front_stream_strips(g);
    int want = 20480 / (max_batch * nstrips > 0 ? max_batch * nstrips : 1);
    if (want < 1) want = 1;
    if (rows_classified <= 0 || rows_classified > g.H) rows_classified = g.H;
    int rows = (rows_classified + want - 1) / want;
    if (rows < 64) rows = 64;
    return rows > g.H ? g.H : rows;
}

// the unit grid of a frame: rows [0, row_split) and [row_split, H) each in units of about rows_per_unit rows
FrontGrid front_stream_grid(Geo g, int rows_per_unit, int row_split) {
    FrontGrid fg{};
    if (rows_per_unit > g.H) rows_per_unit = g.H;
    if (rows_per_unit < 1) rows_per_unit = 1;
    const int rows_low = g.H - row_split;
    fg.nseg_low = (rows_low + rows_per_unit - 1) / rows_per_unit;
    fg.rows_low = rows_per_unit;
    if (row_split > 0) {
        fg.nseg_top = (row_split + rows_per_unit - 1) / rows_per_unit;
        fg.rows_top = (row_split + fg.nseg_top - 1) / fg.nseg_top;
        fg.rows_low = (rows_low + fg.nseg_low - 1) / fg.nseg_low;
        fg.nseg_low = (rows_low + fg.rows_low - 1) / fg.rows_low;
    }
    fg.nstrips = front_stream_strips(g);
    fg.strip_cols = STRIP;
    return fg;
}

void launch_front_stream(const uint8_t *src, int code, bool blur, Geo g, int nframes, int low, int high,
                         uint64_t *planeS, uint64_t *planeW, uint32_t *wake, uint8_t *gray_out, uint8_t *blur_out,
                         int rows_per_unit, const GateDev *gate, const int *tab, hipStream_t st, uint8_t *rowflags, bool dec3,
                         const FrontSplit *split, const FrameSet *fs, const uint8_t *const *src_parts, const HystLaunch *lead) {
    FrontArgs a;
    a.lead_blocks = 0;
    a.lead = HystArgs{};
    a.lead_px = nullptr;
    a.fs = fs ? *fs : FrameSet{};
    for (int j = 0; j < MAX_PARTS; j++) a.src[j] = (fs && src_parts) ? src_parts[j] : src;
    a.wake = wake;
    a.rowflags = rowflags;
    a.gate_tab = tab;
    if (gate) a.gate = *gate; else a.gate = GateDev{};
    a.g = g; a.low = low; a.high = high;
    a.planeS = planeS; a.planeW = planeW; a.gray_out = gray_out; a.blur_out = blur_out;
    // twice the 15-bit weights, so that the gray value lands on a byte boundary of the sum
    const uint32_t k0 = 2 * (code == 6 ? 3735 : 9798), k1 = 2 * 19235, k2 = 2 * (code == 6 ? 9798 : 3735);
    a.k_lo = (k0 & 255) | ((k1 & 255) << 8) | ((k2 & 255) << 16);
    a.k_hi = (k0 >> 8) | ((k1 >> 8) << 8) | ((k2 >> 8) << 16);
    a.nstrips = front_stream_strips(g);              // every byte of every plane row is covered
    for (int k = 0; k < FRONT_SHIFT_SEGS; k++) a.shift8[k] = 0;
    if (split && split->shift) {
        a.nstrips = split->shift->nstrips;
        for (int k = 0; k < FRONT_SHIFT_SEGS; k++) a.shift8[k] = split->shift->by8[k];
    }
    if (rows_per_unit <= 0) rows_per_unit = front_stream_rows_per_unit(g, nframes, 0);
    if (rows_per_unit > g.H) rows_per_unit = g.H;
    a.rows_per_unit = rows_per_unit;
    a.row_split = split ? split->row_split : 0;
    a.only_frames = split ? split->only_frames : nullptr;
    a.unit_list = nullptr;
    a.nlist = 0;
    a.seg_first = a.clear_top = 0;
    const FrontGrid fg = front_stream_grid(g, rows_per_unit, a.row_split);
    a.nseg_top = fg.nseg_top; a.rows_top = fg.rows_top; a.rows_per_unit = fg.rows_low;
    a.nseg = fg.nseg_top + fg.nseg_low;
    if (split && split->mode == 1) {
        if (split->needed) { a.unit_list = split->needed; a.nlist = split->nneeded; }
        else { a.seg_first = fg.nseg_top; a.nseg = fg.nseg_low; }
    }
    if (split && split->mode == 2) { a.unit_list = split->unneeded; a.nlist = split->nunneeded; a.clear_top = 1; }
    const int per_frame = a.unit_list ? a.nlist : a.nseg * a.nstrips;
    if (per_frame == 0 || (a.clear_top && !a.unit_list)) return;
    a.nframes = nframes;
    a.nunits = (a.only_frames ? std::min(REDO_GROUPS, nframes) : nframes) * per_frame;
    // no LDS and no barrier: one wavefront per workgroup, so that a unit can start wherever a single wave slot
    // is free (LANECV_FRONT_WAVES overrides: waves per workgroup, 1..4)
    static const char *wv_env = getenv("LANECV_FRONT_WAVES");
    int wpg = 1;
    if (wv_env && wv_env[0] >= '1' && wv_env[0] <= '4') wpg = wv_env[0] - '0';
    const bool inter = gray_out || blur_out;
    const bool with_lead = lead && !inter && !a.clear_top && !a.only_frames;
    if (with_lead) wpg = LEAD_THREADS / 64;
    const int nwg = (a.nunits + wpg - 1) / wpg;
    a.wg_chunk = (nwg + 7) / 8;
    dim3 grid(8 * a.wg_chunk), block(64 * wpg);
    if (with_lead) {
        a.lead = hysteresis_args(lead->g, lead->nframes, lead->planeS, lead->planeW, lead->wake, lead->rounds, lead->status, nullptr,
                                 lead->ring_words, lead->nring_words, lead->ring_code, lead->redo, &lead->fs, lead->wake_frames);
        a.lead_blocks = lead->nframes;
        a.lead_px = lead->px_scratch;
        grid = dim3(8 * a.wg_chunk + a.lead_blocks);
#define LC_FL(B, G, D) hipLaunchKernelGGL((k_front_lead<B, G, D>), grid, block, 0, st, a)
        if (dec3) { if (blur) LC_FL(true, 0, true); else LC_FL(false, 0, true); }
        else if (gate) { if (blur) LC_FL(true, 1, false); else LC_FL(false, 1, false); }
        else { if (blur) LC_FL(true, 0, false); else LC_FL(false, 0, false); }
#undef LC_FL
        return;
    }
    if (a.clear_top) {
        if (!a.only_frames || !a.rowflags) return;
        if (blur) hipLaunchKernelGGL((k_front_clear<true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_front_clear<false>), grid, block, 0, st, a);
        return;
    }
    if (a.only_frames) {      // the second pass over flagged frames (never with intermediates)
#define LC_FA(B, G, D) hipLaunchKernelGGL((k_front_again<B, G, D>), grid, block, 0, st, a)
        if (dec3) { if (blur) LC_FA(true, 0, true); else LC_FA(false, 0, true); }
        else if (gate) { if (blur) LC_FA(true, 1, false); else LC_FA(false, 1, false); }
        else { if (blur) LC_FA(true, 0, false); else LC_FA(false, 0, false); }
#undef LC_FA
        return;
    }
#define LC_FS(B, G, I) hipLaunchKernelGGL((k_front_stream<B, G, I>), grid, block, 0, st, a)
#define LC_FS2(B, G) do { if (inter) LC_FS(B, G, true); else LC_FS(B, G, false); } while (0)
    if (dec3) {   // src holds frames of 3 W x 3 H (the caller checked: no gate, no intermediates)
        if (blur) hipLaunchKernelGGL((k_front_stream<true, 0, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_front_stream<false, 0, false, true>), grid, block, 0, st, a);
    } else if (gate) { if (blur) LC_FS2(true, 1); else LC_FS2(false, 1); }
    else { if (blur) LC_FS2(true, 0); else LC_FS2(false, 0); }
#undef LC_FS2
#undef LC_FS
}

}  // namespace lanecv
