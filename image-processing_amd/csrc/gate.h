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

// p = B | G << 8 | R << 16 (bits 24-31 ignored): does the gate keep the pixel?
// SHAPE 1 is the general form.  SHAPE 2 is the form of the lane script's two boxes, whatever their numbers
// (gate_shape decides): box 0 = "any hue, saturation at most s0, value at least v0", box 1 = "hue inside
// [a, b] with b < 150, saturation at least s1, value at least v1".  One-sided tests are a single compare, and
// a hue interval below 150 needs no wrap of negative hues (they become 150..179, outside it; tested unwrapped
// they are negative, which the unsigned compare rejects as well).
template <int SHAPE>
__device__ __forceinline__ bool gate_keep(uint32_t p, const int *tab, const GateDev &gt) {
    const int b = (int)(p & 255u), g = (int)((p >> 8) & 255u), r = (int)((p >> 16) & 255u);
    if constexpr (SHAPE == 2) {
        const int v = max(b, max(g, r)), diff = v - min(b, min(g, r));
        const int sd = tab[v], hd = tab[256 + diff];
        const int s = (diff * sd + (1 << 11)) >> 12;
        const int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
        const int h = (hh * hd + (1 << 11)) >> 12;                       // not wrapped
        const bool in0 = s <= (int)gt.span[1] && v >= gt.lo[2];          // box 0: lo[1] is 0, so span[1] is the upper bound
        const bool in1 = (uint32_t)(h - gt.lo[3]) <= gt.span[3] && s >= gt.lo[4] && v >= gt.lo[5];
        return in0 || in1;
    } else {
        int h, s, v;
        hsv_px(b, g, r, tab, h, s, v);
        return gate_pass(h, s, v, gt);
    }
}

// 2 when the (normalised) gate has the lane script's form described above, else 1
inline int gate_shape(const GateDev &gt) {
    auto upper_open = [&](int i) { return gt.lo[i] <= 255 && gt.lo[i] + (long long)gt.span[i] >= 255; };   // no upper bound
    const bool box0 = gt.lo[0] == 0 && gt.span[0] >= 179 && gt.lo[1] == 0 && gt.lo[2] <= 255 && upper_open(2);
    const bool box1 = gt.lo[3] <= 255 && gt.lo[3] + (long long)gt.span[3] < 150 && upper_open(4) && upper_open(5);
    return box0 && box1 ? 2 : 1;
}

}  // namespace lanecv
