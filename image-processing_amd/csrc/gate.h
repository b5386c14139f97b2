// Device side of the colour gate (8-bit HSV conversion and range tests), shared by the per-op kernels
// and the front kernels that fuse the gate into the gray conversion.
#pragma once
#include "common.h"

namespace lanecv {

// OpenCV's 8U RGB2HSV (hue range 180) in fixed point, hsv_shift = 12:
//   v = max(b,g,r), diff = v - min(b,g,r), s = (diff * sdiv[v] + 2^11) >> 12,
//   h = v==r ? g-b : v==g ? b-r+2diff : r-g+4diff;  h = (h * hdiv[diff] + 2^11) >> 12;  h += 180 if h < 0
// tab: sdiv[256] followed by hdiv[256] (global or LDS)
__device__ __forceinline__ void hsv_px(int b, int g, int r, const int *tab, int &h, int &s, int &v) {
    v = max(b, max(g, r));
    const int diff = v - min(b, min(g, r));
    const int sd = tab[v], hd = tab[256 + diff];
    s = (diff * sd + (1 << 11)) >> 12;
    const int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (hh * hd + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
}

// true when (h, s, v) lies inside either of the gate's two boxes
__device__ __forceinline__ bool gate_pass(int h, int s, int v, const GateDev &gt) {
    const bool in0 = (uint32_t)(h - gt.lo[0]) <= gt.span[0] && (uint32_t)(s - gt.lo[1]) <= gt.span[1] &&
                     (uint32_t)(v - gt.lo[2]) <= gt.span[2];
    const bool in1 = (uint32_t)(h - gt.lo[3]) <= gt.span[3] && (uint32_t)(s - gt.lo[4]) <= gt.span[4] &&
                     (uint32_t)(v - gt.lo[5]) <= gt.span[5];
    return in0 || in1;
}

// p = B | G << 8 | R << 16 (bits 24-31 ignored): does the gate keep the pixel?  (the arithmetic form: full HSV
// conversion and two box tests; the gated gray pre-pass of frames the streaming kernel cannot take uses it)
__device__ __forceinline__ bool gate_keep(uint32_t p, const int *tab, const GateDev &gt) {
    const int b = (int)(p & 255u), g = (int)((p >> 8) & 255u), r = (int)((p >> 16) & 255u);
    int h, s, v;
    hsv_px(b, g, r, tab, h, s, v);
    return gate_pass(h, s, v, gt);
}

// Table form, four pixels at a time (the streaming kernel).  For a given v the saturation
// s = (diff * sdiv[v] + 2^11) >> 12 does not decrease with diff, so "s and v inside box k" is "diff inside an
// interval that depends on v": the host tabulates the interval of either box per v (gate_tables_host: tab[512 + v]
// = dmin0 | dmax0 << 8 | dmin1 << 16 | dmax1 << 24; an empty interval is 255..0), and a pixel costs max, min, the
// look-up and four compares.  The look-up is what bounds the kernel -- a per-lane load, 64 addresses per
// instruction, which the texture addresser takes four a clock: 0.21 of the kernel's 0.73 ms, whatever the
// arithmetic around it (the same table in registers behind ds_bpermute, one per quarter of the table, was
// slower still: 0.89 ms).  Only lanes whose v reaches the smallest v with a non-empty interval (gt.vmin) load.
// The hue is needed only by pixels that pass the s/v test of a box whose hue range is not the whole circle
// (gt.hue_any, one bit per box) -- the yellow box of the lane script, a few pixels per frame -- and is computed
// (as hsv_px does, with the wrap) only by wavefronts in which some lane needs it.
__device__ __forceinline__ uint32_t gate_keep4(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, const int *tab, const GateDev &gt) {
    const uint32_t p[4] = {p0, p1, p2, p3};
    int v[4], diff[4];
    bool s0[4], s1[4];
    bool need = false;
    const bool any0 = (gt.hue_any & 1) != 0, any1 = (gt.hue_any & 2) != 0;
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int b = (int)(p[k] & 255u), g = (int)((p[k] >> 8) & 255u), r = (int)((p[k] >> 16) & 255u);
        v[k] = max(b, max(g, r));
        diff[k] = v[k] - min(b, min(g, r));
        e[k] = 0x00FF00FFu;                                 // both intervals empty
        if (v[k] >= gt.vmin) e[k] = (uint32_t)tab[512 + v[k]];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t d = (uint32_t)diff[k];
        s0[k] = d >= (e[k] & 255u) && d <= ((e[k] >> 8) & 255u);
        s1[k] = d >= ((e[k] >> 16) & 255u) && d <= (e[k] >> 24);
        need |= (s0[k] && !any0) || (s1[k] && !any1);
    }
    uint32_t km = 0;
    if (__ballot(need) != 0) {
        asm volatile("" ::: "memory");     // stays a branch: the hue arithmetic must not be hoisted in front of it
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int b = (int)(p[k] & 255u), g = (int)((p[k] >> 8) & 255u), r = (int)((p[k] >> 16) & 255u);
            const int hd = tab[256 + diff[k]];
            const int hh = v[k] == r ? g - b : (v[k] == g ? b - r + 2 * diff[k] : r - g + 4 * diff[k]);
            int h = (hh * hd + (1 << 11)) >> 12;
            h += h < 0 ? 180 : 0;
            const bool in0 = s0[k] && (any0 || (uint32_t)(h - gt.lo[0]) <= gt.span[0]);
            const bool in1 = s1[k] && (any1 || (uint32_t)(h - gt.lo[3]) <= gt.span[3]);
            km |= (in0 || in1) ? (0xFFu << (8 * k)) : 0u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) km |= (s0[k] || s1[k]) ? (0xFFu << (8 * k)) : 0u;
    }
    return km;
}

}  // namespace lanecv
