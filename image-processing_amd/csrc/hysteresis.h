// Hysteresis of the Canny classes held as bit planes, as device code shared by the stand-alone kernel (kernels_edge.hip)
// and the streaming front kernel, whose launches can carry the hysteresis of the batch before as their first workgroups
// (kernels_front_stream.hip).
#pragma once
#include "common.h"

namespace lanecv {
namespace {

// ---------------------------------------------------------------------------
// hysteresis: one wavefront owns one 64x64 tile as 2 x 64 bit rows in registers
// (lane = row).  A round floods strong bits through weak bits inside the tile,
// seeded by the strong pixels of the one-pixel ring around it; tiles whose
// border gained a strong pixel wake the neighbours that touch it for the next
// round.  Rounds repeat (host side) until no tile is woken.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t flood_row(uint64_t seed, uint64_t cand) {
    // every run of 1s in cand that holds a seed bit becomes all seed
    seed &= cand;
    uint64_t up = seed | (cand & ~(cand + seed));              // towards bit 63
    uint64_t cr = __brevll(cand), sr = __brevll(up);
    uint64_t dn = sr | (cr & ~(cr + sr));                      // towards bit 0
    return __brevll(dn);
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// L2-served accesses of the strong plane: tiles read their neighbours' words while other
// wavefronts (possibly on other CUs) are adding bits to them.  A stale read only delays a
// promotion to a later round, it never loses one; bits are only ever added.
__device__ __forceinline__ uint64_t ldS(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stS(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wavefront floods tile (ty, tx) of frame f.  Returns an 8-bit mask of the neighbours
// (0 T, 1 B, 2 L, 3 R, 4 TL, 5 TR, 6 BL, 7 BR) that touch a pixel this call made strong.
__device__ __forceinline__ unsigned hyst_tile(const Geo &g, int f, int ty, int tx, uint64_t *__restrict__ planeS,
                                              const uint64_t *__restrict__ planeW, int lane) {
    // word of (row y, column group x) of this frame; rows below the image read as empty
    const size_t fb = (size_t)f * g.H * g.TX;
    const int y = ty * TILE + lane;
    const bool yin = y < g.H;
    auto widx = [&](int yy, int xx) -> size_t { return fb + (size_t)yy * g.TX + xx; };
    uint64_t Wk = yin ? planeW[widx(y, tx)] : 0;
    if (__ballot(Wk != 0) == 0) return 0;
    uint64_t S = yin ? ldS(planeS + widx(y, tx)) : 0;
    Wk &= ~S;
    if (__ballot(Wk != 0) == 0) return 0;
    const uint64_t S_in = S;
    const uint64_t cand = S | Wk;

    // strong bits of the one-pixel ring around the tile
    const bool hasL = tx > 0, hasR = tx + 1 < g.TX;
    uint64_t Lw = (hasL && yin) ? ldS(planeS + widx(y, tx - 1)) : 0;
    uint64_t Rw = (hasR && yin) ? ldS(planeS + widx(y, tx + 1)) : 0;
    uint64_t lcol = Lw >> 63, rcol = Rw & 1;     // (row, -1) and (row, 64)
    // lanes 0..2: row 63 of tiles (ty-1, tx-1..tx+1); lanes 3..5: row 0 of tiles (ty+1, tx-1..tx+1)
    uint64_t edge = 0;
    if (lane < 6) {
        int dy = lane < 3 ? -1 : 1, dx = (lane % 3) - 1;
        int yy = ty + dy, xx = tx + dx;
        const int ey = dy < 0 ? ty * TILE - 1 : ty * TILE + TILE;
        if (ey >= 0 && ey < g.H && xx >= 0 && xx < g.TX) edge = ldS(planeS + widx(ey, xx));
        (void)yy;
    }
    uint64_t tl = shfl_u64(edge, 0) >> 63, tm = shfl_u64(edge, 1), tr = shfl_u64(edge, 2) & 1;
    uint64_t bl = shfl_u64(edge, 3) >> 63, bm = shfl_u64(edge, 4), br = shfl_u64(edge, 5) & 1;
    // column neighbours of row r: rows r-1, r, r+1 of the left / right ring
    uint64_t l_up = shfl_u64(lcol, lane - 1), l_dn = shfl_u64(lcol, lane + 1);
    uint64_t r_up = shfl_u64(rcol, lane - 1), r_dn = shfl_u64(rcol, lane + 1);
    if (lane == 0) { l_up = tl; r_up = tr; }
    if (lane == 63) { l_dn = bl; r_dn = br; }
    uint64_t seed = ((lcol | l_up | l_dn) & 1) | (((rcol | r_up | r_dn) & 1) << 63);
    if (lane == 0) seed |= tm | (tm << 1) | (tm >> 1) | tl | (tr << 63);
    if (lane == 63) seed |= bm | (bm << 1) | (bm >> 1) | bl | (br << 63);
    S |= Wk & seed;

    for (;;) {
        uint64_t Sf = flood_row(S, cand);
        uint64_t up = shfl_u64(Sf, lane - 1), dn = shfl_u64(Sf, lane + 1);
        if (lane == 0) up = 0;
        if (lane == 63) dn = 0;
        uint64_t v = up | dn;
        uint64_t Sn = Sf | (cand & (v | (v << 1) | (v >> 1)));
        bool ch = Sn != S;
        S = Sn;
        if (__ballot(ch) == 0) break;
    }

    const uint64_t added = S & ~S_in;
    const uint64_t any = __ballot(added != 0);
    if (any == 0) return 0;
    if (added) stS(planeS + widx(y, tx), S);
    const bool top = (any & 1) != 0, bot = (any >> 63) != 0;
    const bool left = __ballot((added & 1) != 0) != 0, right = __ballot((added >> 63) != 0) != 0;
    const uint64_t a0 = shfl_u64(added, 0), a63 = shfl_u64(added, 63);
    unsigned m = 0;
    if (top && ty > 0) m |= 1u;
    if (bot && ty + 1 < g.TY) m |= 2u;
    if (left && hasL) m |= 4u;
    if (right && hasR) m |= 8u;
    if (top && (a0 & 1) && ty > 0 && hasL) m |= 16u;
    if (top && (a0 >> 63) && ty > 0 && hasR) m |= 32u;
    if (bot && (a63 & 1) && ty + 1 < g.TY && hasL) m |= 64u;
    if (bot && (a63 >> 63) && ty + 1 < g.TY && hasR) m |= 128u;
    return m;
}

__device__ __forceinline__ int nb_tile(const Geo &g, int ty, int tx, int k) {
    const int dys[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
    const int dxs[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    return (ty + dys[k]) * g.TX + (tx + dxs[k]);
}

// One workgroup per frame runs the rounds.  The first round's work list comes from the front kernel, which
// flags every tile it stores a weak pixel into; later rounds take the tiles woken by their neighbours, until a
// round wakes none.  All cross-tile hand-offs stay inside the workgroup (barriers + L2-served words), so the
// fixed point is reached on the device with no host round trip.  The flags live in global memory (two arrays
// per frame, swapped every round) and the kernel keeps 4 bytes of LDS, so it can share a CU with kernels of
// other batches that fill the LDS.  A wavefront reads 64 flags at once, lane l that of tile 16 l + wave:
// neighbouring tiles (a lane marking crosses several in a row) go to different wavefronts.
__device__ __forceinline__ uint32_t ldF(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stF(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the rounds of one frame, run by the HC_THREADS threads of a workgroup; `any` is a word of LDS
template <int HC_THREADS>
__device__ __forceinline__ int hysteresis_frame(const Geo &g, int f, uint64_t *__restrict__ planeS,
                                                 const uint64_t *__restrict__ planeW, uint32_t *__restrict__ wakeA,
                                                 uint32_t *__restrict__ wakeB, int &any) {
    constexpr int HC_WAVES = HC_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = g.TX * g.TY;
    uint32_t *cur = wakeA + (size_t)f * ntile, *nxt = wakeB + (size_t)f * ntile;   // nxt starts all zero
    int mine = 0;
    for (int i = tid; i < ntile; i += HC_THREADS) mine |= (int)cur[i];
    if (tid == 0) any = 0;
    __syncthreads();
    if (mine) any = 1;
    __syncthreads();
    int rounds = 0;
    while (any) {
        __syncthreads();
        if (tid == 0) any = 0;
        __syncthreads();
        for (int base = 0; base < ntile; base += HC_THREADS) {
            const int t = base + lane * HC_WAVES + wave;
            const bool flagged = t < ntile && ldF(cur + t) != 0;
            if (flagged) stF(cur + t, 0);             // this array is next round's target
            uint64_t todo = __ballot(flagged);
            while (todo) {
                const int tt = base + __builtin_ctzll(todo) * HC_WAVES + wave;
                todo &= todo - 1;
                const int ty = tt / g.TX, tx = tt % g.TX;
                const unsigned m = hyst_tile(g, f, ty, tx, planeS, planeW, lane);
                if (lane < 8 && ((m >> lane) & 1)) { stF(nxt + nb_tile(g, ty, tx, lane), 1); any = 1; }
            }
        }
        __threadfence_block();
        __syncthreads();
        uint32_t *tmp = cur; cur = nxt; nxt = tmp;
        rounds++;
    }
    return rounds;
}

// ---------------------------------------------------------------------------
// Was it enough to classify the units near the ROI?  (See launch_hysteresis in common.h.)  The frame's workgroup asks
// once its hysteresis has settled.  A pixel outside the classified units can only matter through a chain of weak
// pixels that stayed weak, leading from the ROI outwards through every ring.  The workgroup lists the leftover weak
// pixels of the rings (there are few), takes those of ring 1 as reached -- as if every one of them touched such a pixel
// of the ROI -- and floods from them through 8-neighbours in the list (a thread per listed pixel); only if the flood
// arrives in ring 15, or does not settle, or the list overflows, is the frame classified again in full.
// ---------------------------------------------------------------------------
constexpr int RING_LIST = RING_LIST_WORDS, RING_ROUNDS = 48;   // beyond either the frame is classified in full rather than looked at longer
struct RingArgs {
    const uint32_t *ring_words;     // indices of the plane words that hold ring pixels
    int nring_words;
    const uint64_t *code;           // four bit planes: the ring number (1..15) of every pixel, 0 = in no ring
    int *redo;                      // per frame: 1 = classify again in full
};

// px: RING_LIST words of LDS, count: one; returns (to every thread) whether frame f must be classified again
// PXG: the list lives in device memory (read and written through L2) instead of LDS
template <bool PXG> __device__ __forceinline__ uint32_t ld_px(const uint32_t *p) { if constexpr (PXG) return ldF(p); else return *p; }
template <bool PXG> __device__ __forceinline__ void st_px(uint32_t *p, uint32_t v) { if constexpr (PXG) stF(p, v); else *p = v; }
template <int HC_THREADS, bool PXG = false>
__device__ __forceinline__ bool ring_check_frame(const Geo &g, int f, const uint64_t *__restrict__ planeS,
                                                 const uint64_t *__restrict__ planeW, const RingArgs &ra, uint32_t *px, int &count) {
    const int tid = threadIdx.x;
    const size_t words = g.plane_words();
    if (tid == 0) count = 0;
    __syncthreads();
    for (int i = tid; i < ra.nring_words; i += HC_THREADS) {
        const uint32_t w = ra.ring_words[i];
        uint64_t left = planeW[(size_t)f * words + w];
        if (left) left &= ~ldS(planeS + (size_t)f * words + w);
        if (!left) continue;
        const uint64_t c0 = ra.code[w], c1 = ra.code[words + w], c2 = ra.code[2 * words + w], c3 = ra.code[3 * words + w];
        left &= c0 | c1 | c2 | c3;                                       // ring pixels only
        if (!left) continue;
        const int at = atomicAdd(&count, __popcll(left));
        const uint32_t y = w / g.TX, xb = (w % g.TX) * 64;
        int k = 0;
        for (uint64_t m = left; m; m &= m - 1, k++) {
            const int b = __builtin_ctzll(m);
            const uint32_t ring = (uint32_t)((c0 >> b) & 1) | (uint32_t)((c1 >> b) & 1) << 1 | (uint32_t)((c2 >> b) & 1) << 2 |
                                  (uint32_t)((c3 >> b) & 1) << 3;
            // x | y << 13 | ring << 26 | reached << 31
            if (at + k < RING_LIST) st_px<PXG>(px + at + k, (xb + b) | (y << 13) | (ring << 26) | ((ring == 1 ? 1u : 0u) << 31));
        }
    }
    __syncthreads();
    const int n = count;
    if (n == 0) return false;
    if (n > RING_LIST) return true;
    // a thread per listed pixel (several pixels per thread in workgroups smaller than the list)
    constexpr int PER = (RING_LIST + HC_THREADS - 1) / HC_THREADS;
    uint32_t p0[PER];
    bool mine[PER], reached[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int idx = tid + q * HC_THREADS;
        mine[q] = idx < n;
        p0[q] = mine[q] ? ld_px<PXG>(px + idx) : 0u;
        reached[q] = (p0[q] >> 31) != 0;
    }
    bool settled = false;
    for (int round = 0; round < RING_ROUNDS && !settled; round++) {
        bool changed = false;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            if (!mine[q] || reached[q]) continue;
            const int x = p0[q] & 0x1FFF, y = (p0[q] >> 13) & 0x1FFF;
            bool hit = false;
            for (int j = 0; j < n && !hit; j++) {
                const uint32_t v = ld_px<PXG>(px + j);        // (another thread may be setting its bit 31 right now: sooner or later is the same)
                hit = (v >> 31) && abs((int)(v & 0x1FFF) - x) <= 1 && abs((int)((v >> 13) & 0x1FFF) - y) <= 1;
            }
            if (hit) { reached[q] = true; st_px<PXG>(px + tid + q * HC_THREADS, p0[q] | 0x80000000u); changed = true; }
        }
        settled = __syncthreads_or(changed ? 1 : 0) == 0;
    }
    // still spreading after RING_ROUNDS: do not look longer
    bool far = false;
#pragma unroll
    for (int q = 0; q < PER; q++) far = far || (mine[q] && reached[q] && ((p0[q] >> 26) & 15) == 15);
    return __syncthreads_or((!settled || far) ? 1 : 0) != 0;
}

// Everything a hysteresis launch is told (see launch_hysteresis in common.h).
struct HystArgs {
    Geo g;
    uint64_t *planeS;
    const uint64_t *planeW;
    uint32_t *wakeA, *wakeB;
    int *rounds_out, *status;
    const int *only_frames;
    RingArgs ra;
    FrameSet fs;
    int nframes;
};

// One frame of a hysteresis launch, by the HC_THREADS threads of a workgroup: the rounds, the proof that the units
// classified settle the ROI's edges (ra.redo set), the frame's words of the result block.  any, count: a word of LDS
// each; px: RING_LIST words.  All threads of the workgroup must call it together.
template <int HC_THREADS, bool PXG = false>
__device__ __forceinline__ void hysteresis_one_frame(const HystArgs &h, int f, int &any, int &count, uint32_t *px) {
    if (!h.fs.has(f)) return;
    int *st = h.status ? h.status + 8 * h.fs.part(f) : nullptr;            // every part has status words of its own
    if (st && !h.only_frames && h.fs.first_of_part(f) && threadIdx.x < 5) st[threadIdx.x] = 0;
    if (h.only_frames && h.only_frames[f] == 0) return;
    __syncthreads();                                                       // (the LDS words of the frame before)
    const int rounds = hysteresis_frame<HC_THREADS>(h.g, f, h.planeS, h.planeW, h.wakeA, h.wakeB, any);
    bool bad = false;
    if (h.ra.redo) {
        __syncthreads();
        bad = ring_check_frame<HC_THREADS, PXG>(h.g, f, h.planeS, h.planeW, h.ra, px, count);
    }
    if (threadIdx.x == 0) {
        if (h.ra.redo) h.ra.redo[f] = bad ? 1 : 0;
        // bit 30 of the rounds word: the frame is, or (second pass over flagged frames) was, flagged
        if (h.rounds_out) h.rounds_out[f] = rounds | ((bad || h.only_frames) ? 0x40000000 : 0);
    }
}

inline HystArgs hysteresis_args(Geo g, int nframes, uint64_t *planeS, const uint64_t *planeW, uint32_t *wake, int *rounds, int *status,
                                const int *only_frames, const uint32_t *ring_words, int nring_words, const uint64_t *ring_code,
                                int *redo, const FrameSet *fs, int wake_frames = 0) {
    // wake holds two flag arrays of nframes * tiles words each.  The second is all zero on entry and both are on
    // exit (zeroed when allocated): the kernel clears every flag it consumes and stops when a round sets none.
    HystArgs h;
    h.g = g;
    h.planeS = planeS; h.planeW = planeW;
    h.wakeA = wake; h.wakeB = wake + (size_t)g.TX * g.TY * (wake_frames > 0 ? wake_frames : nframes);
    h.rounds_out = rounds; h.status = status; h.only_frames = only_frames;
    h.ra = RingArgs{ring_words, nring_words, ring_code, ring_code ? redo : nullptr};
    h.fs = fs ? *fs : FrameSet{};
    h.nframes = nframes;
    return h;
}

}  // namespace
}  // namespace lanecv
