// Hough stage for gfx950: ordered compaction of the edge points, the full-vote
// accumulator (LDS-privatised per angle group) and HoughLinesP as an exact-order
// progressive probabilistic Hough transform, one wavefront per frame.
//
// Float contract: the rho bin is cvRound(x * tab[2n] + y * tab[2n+1]) with two
// separately rounded float multiplies, one float add and round-half-even, so
// every product/sum below goes through __fmul_rn/__fadd_rn (no FMA contraction)
// and __float2int_rn.  The trig table and the walk steps come from the host.
#include "common.h"
#include <climits>
#include <cstdlib>
#include <algorithm>
#include <mutex>

namespace lanecv {
namespace {

__device__ __forceinline__ int rho_bin(int x, int y, float c, float s, int half) {
    return __float2int_rn(__fadd_rn(__fmul_rn((float)x, c), __fmul_rn((float)y, s))) + half;
}

// Ordered compaction of a frame's edge points: (x | y << 16) in row-major scan order, what OpenCV's loops
// over the edge map produce.  One 16-wavefront workgroup per frame.  A wavefront owns a contiguous band of
// rows; the per-row counts give its total (and are cleared: this kernel is their last reader), a scan over
// the 16 totals its first position, and then the band's plane words -- contiguous in memory and already in
// scan order -- are taken 64 at a time: popcount, wave scan, scatter.  Bands without points leave at once.
#ifndef LC_POINTS_WAVES
#define LC_POINTS_WAVES 16
#endif
constexpr int POINTS_WAVES = LC_POINTS_WAVES;

__global__ __launch_bounds__(64 * POINTS_WAVES) void k_points(Geo g, const uint64_t *__restrict__ rowplane,
                                                               int *__restrict__ rowcount, uint32_t *__restrict__ nz, int nz_cap,
                                                               int *__restrict__ npoints, int *__restrict__ status, FrameSet fs) {
    __shared__ int tot[POINTS_WAVES];
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (!fs.has(f)) return;
    if (status) status += 8 * fs.part(f);
    const int H = g.H, TX = g.TX;
    const int rpw = (H + POINTS_WAVES - 1) / POINTS_WAVES;
    const int r0 = min(H, wv * rpw), r1 = min(H, r0 + rpw);
    int s = 0;
    for (int r = r0 + lane; r < r1; r += 64) {
        const int c = rowcount[(size_t)f * H + r];
        if (c) rowcount[(size_t)f * H + r] = 0;
        s += c;
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) tot[wv] = s;
    __syncthreads();
    int off = 0, all = 0;
    for (int i = 0; i < POINTS_WAVES; i++) {
        const int t = tot[i];
        off += i < wv ? t : 0;
        all += t;
    }
    if (threadIdx.x == 0) {
        npoints[f] = all;
        if (status && all > nz_cap) status[0] = 1;   // the list is cut short: whoever consumes it must not trust it
    }
    if (s == 0) return;
    const uint64_t *wp = rowplane + ((size_t)f * H + r0) * TX;
    const int nwords = (r1 - r0) * TX;
    uint32_t *out = nz + (size_t)f * nz_cap;
    constexpr int UNR = 8;   // words per lane in flight: the loads of a batch go out together
    for (int k0 = 0; k0 < nwords; k0 += 64 * UNR) {
        uint64_t wq[UNR];
#pragma unroll
        for (int j = 0; j < UNR; j++) {
            const int idx = k0 + 64 * j + lane;
            wq[j] = idx < nwords ? wp[idx] : 0;
        }
#pragma unroll
        for (int j = 0; j < UNR; j++) {
            uint64_t w = wq[j];
            if (__ballot(w != 0) == 0) continue;
            const int idx = k0 + 64 * j + lane;
            const int c = __popcll(w);
            int incl = c;   // inclusive scan over the lanes
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            int pos = off + incl - c;
            const int rr = idx / TX;
            const uint32_t hi = (uint32_t)(r0 + rr) << 16, xb = (uint32_t)(idx - rr * TX) * 64u;
            while (w) {
                const int b = __builtin_ctzll(w);
                if (pos < nz_cap) out[pos] = (xb + b) | hi;
                pos++;
                w &= w - 1;
            }
            off += __shfl(incl, 63);
        }
    }
}

// full-vote accumulator: block (group, frame) owns `apg` angle rows in LDS.
// The points come sorted by row and column, and rho moves by less than one bin per pixel: neighbours of the
// list mostly vote for the same bin of an angle, and lanes that add to one LDS word in one instruction are
// served one after the other.  The 64 lanes of a wavefront therefore take 64 equal stretches of the whole list
// (n / 64 points each, split between the workgroup's wavefronts): lanes are then many rows apart and their
// votes spread over the angle's row, while a lane's own consecutive points, which do share bins, arrive in
// different instructions.  Points are taken four at a time, their coordinates converted to float once for all
// angles of the group.  A vote is branch-free: two multiplies and an add (each rounded on its own, as
// OpenCV's float expression is), round-to-nearest-even by adding 1.5 * 2^23 (exact for |rho| < 2^22; the
// result's low bits are the integer), one subtraction that also adds the row's centre, an unsigned minimum that
// sends anything outside the row to a spare word behind it, and the atomic add.  The ragged ends of the
// stretches go point by point with absent points masked off (adds of 0 for them would all meet in one word).
// LDS: apg rows of numrho + 1 words.  (a_first, a_step: the angle groups this workgroup takes are a_first,
// a_first + a_step, ...)
__device__ __forceinline__ void vote_frame(int *lacc, int f, int a_first, int a_step, const uint32_t *__restrict__ nz, int nz_cap,
                                           const int *__restrict__ npoints, const AngleEntry *__restrict__ tab, int numangle,
                                           int numrho, int apg, int32_t *__restrict__ acc) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = min(npoints[f], nz_cap);
    const int half = (numrho - 1) / 2, stride = numrho + 1;
    const uint32_t *pts = nz + (size_t)f * nz_cap;
    // a lane's stretch and a wavefront's part of it, both multiples of four points: with a list that starts on a
    // 16-byte boundary a lane's four points are one 16-byte load (a wavefront's loads touch 64 different cache lines
    // either way; single-word loads touch them four times)
    const int S64 = ((n + 63) / 64 + 3) & ~3, nw = nt >> 6, per = ((S64 + nw - 1) / nw + 3) & ~3;
    const bool vec = ((reinterpret_cast<uintptr_t>(pts) & 15) == 0);
    const int w0 = (tid >> 6) * per, wend = min(w0 + per, S64);
    const uint32_t unbias = 0x4B400000u - (uint32_t)half;                       // bits of 1.5 * 2^23, less the row's centre
    for (int a0 = a_first * apg; a0 < numangle; a0 += a_step * apg) {
        const int na = min(apg, numangle - a0);
        __syncthreads();
        for (int i = tid; i < na * stride; i += nt) lacc[i] = 0;
        __syncthreads();
        for (int i = w0; i < wend; i += 4) {
            const int p = (tid & 63) * S64 + i;
            const int cnt = min(4, wend - i);
            if (cnt == 4 && p + 3 < n) {                 // all lanes but those at the very end of the list
                float fx[4], fy[4];
                uint32_t pt[4];
                if (vec) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(pts + p);
                    pt[0] = q.x; pt[1] = q.y; pt[2] = q.z; pt[3] = q.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) pt[k] = pts[p + k];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    fx[k] = (float)(pt[k] & 0xFFFF);
                    fy[k] = (float)(pt[k] >> 16);
                }
                for (int a = 0; a < na; a++) {
                    const float c = tab[a0 + a].c, sn = tab[a0 + a].s;
                    int *row = lacc + a * stride;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float t = __fadd_rn(__fadd_rn(__fmul_rn(fx[k], c), __fmul_rn(fy[k], sn)), 12582912.0f);
                        atomicAdd(&row[min(__float_as_uint(t) - unbias, (uint32_t)numrho)], 1);
                    }
                }
            } else {                                     // ragged end of the stretches: point by point, absent ones masked off
                for (int k = 0; k < cnt; k++) {
                    if (p + k >= n) continue;
                    const uint32_t pt = pts[p + k];
                    const float fx = (float)(pt & 0xFFFF), fy = (float)(pt >> 16);
                    for (int a = 0; a < na; a++) {
                        const float t = __fadd_rn(__fadd_rn(__fmul_rn(fx, tab[a0 + a].c), __fmul_rn(fy, tab[a0 + a].s)), 12582912.0f);
                        atomicAdd(&lacc[a * stride + min(__float_as_uint(t) - unbias, (uint32_t)numrho)], 1);
                    }
                }
            }
        }
        __syncthreads();
        int32_t *out = acc + ((size_t)f * numangle + a0) * numrho;
        for (int a = 0; a < na; a++)
            for (int i = tid; i < numrho; i += nt) out[a * numrho + i] = lacc[a * stride + i];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_vote(const uint32_t *__restrict__ nz, int nz_cap, const int *__restrict__ npoints,
                                               const AngleEntry *__restrict__ tab, int numangle, int numrho, int apg,
                                               int32_t *__restrict__ acc, const int *__restrict__ flags, FrameSet fs) {
    extern __shared__ int lacc[];
    const int f = blockIdx.y;
    if (!fs.has(f)) return;
    if (flags && !(flags[f] & 2)) return;   // HoughLinesP already produced this frame's accumulator
    // a block takes the angle groups blockIdx.x, blockIdx.x + gridDim.x, ...
    vote_frame(lacc, f, blockIdx.x, gridDim.x, nz, nz_cap, npoints, tab, numangle, numrho, apg, acc);
}

// ---------------------------------------------------------------------------
// HoughLinesP, exact order.  One wavefront per frame walks OpenCV's sequence:
// MWC random pick with swap-remove, 60-lane vote with returned counts, first
// maximum wins, fixed-point walk in both directions with the gap rule, vote
// removal and mask clearing along accepted lines.  Lanes parallelise the angles
// of a vote and the steps of a walk; everything that orders is wave-uniform.
// All mutated state (accumulator, mask bits, point list) is touched with
// agent-scope atomics / L2-served loads only, so nothing can be read stale from
// the vector L1.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld_l2(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_l2(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_l2(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long shfl_xor_i64(long long v, int m) {
    int lo = __shfl_xor((int)(v & 0xFFFFFFFFll), m), hi = __shfl_xor((int)(v >> 32), m);
    return ((long long)hi << 32) | (unsigned)lo;
}

// maximum of a signed 32-bit key over the wavefront, returned wave-uniform: two quad permutes and two
// row rotates (DPP, no LDS traffic) leave each row of 16 lanes with its maximum, four readlanes finish.
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));   // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));   // row_ror:8
    int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

struct Walk {
    int xflag, x0, y0, dx, dy;   // fixed-point start and step of direction 0
    __device__ __forceinline__ void at(int t, int sgn, int &j1, int &i1) const {
        int x = x0 + sgn * t * dx, y = y0 + sgn * t * dy;
        if (xflag) { j1 = x; i1 = y >> 16; }
        else       { j1 = x >> 16; i1 = y; }
    }
};

#ifdef LC_STAMPS
__device__ long long g_stamps[8];   // diagnostic build only: cycle sums of frame 0
__device__ long long g_more[8];     // frame 0: vote groups, alive points at batch starts, cycles between trigger and walk, batches with a trigger, segments
__device__ unsigned long long g_dbg[4];
__device__ int g_simd[8];
__device__ long long g_setup[16];             // frame 0: cycles since kernel start at the marks of the set-up
#define SETUP_MARK(k) do { if (f == 0 && tid == 0) g_setup[k] = STAMP() - t_kernel0; } while (0)
__device__ long long g_frame_cycles[1024];   // consumer loop and whole kernel (set-up included) per frame: [2f], [2f + 1]
__device__ long long g_frame_real[512];      // whole kernel per frame in ticks of the constant 100 MHz counter (s_memrealtime)
#define STAMP() __builtin_amdgcn_s_memtime()
#endif
#ifndef LC_STAMPS
#define SETUP_MARK(k) do { } while (0)
#endif
#ifndef PPHT_Q
#define PPHT_Q 4   // points voting speculatively per group in the LDS path
#endif
// Offset of the 16-bit accumulator bins kept in LDS.  A bin's count lies in (-n, n) for n points (a line takes the votes
// of all its pixels back, voted or not, so counts go negative), and up to PPHT_Q - 1 votes ahead of OpenCV's order are
// in a bin for a moment: with fewer than LDS_BIAS - 8 points a biased bin stays inside (0, 0xFFFF], and the two bins
// of a 32-bit word never carry into each other.
constexpr int LDS_BIAS = 0x8000;
// Dynamic LDS of the fast HoughLinesP kernel.  MAX: a frame alone on a CU (152 of its 160 KiB; the rest covers the
// kernel's own static tables and the little the short kernels of other batches need beside it).  HALF: two frames
// per CU -- 1080p road frames need 55 - 60 KiB -- so that the kernels of two batches run side by side.
constexpr int PPHT_LDS_MAX = 152 * 1024, PPHT_LDS_HALF = 72 * 1024;

// Per-frame constants of one HoughLinesP run.
struct Ppht {
    int W, H, TX, numangle, numrho, half, threshold, line_length, line_gap, seg_cap;
    uint64_t *mask;            // row-major bit plane of the frame (bits are cleared as lines are found)
    const AngleEntry *tab;
    int16_t *lacc;             // LDS accumulator (fast path)
    int32_t *acc;              // global accumulator (general path)
    int32_t *seg;
    // fast path: the frame's edge bits live in LDS, only the 64-pixel words that hold a point
    const AngleEntry *ltab;    // LDS copy of tab (walk steps of every angle)
    uint64_t *lmask;           // those words, row after row
    // The row table: per row ymin + r one entry (frames up to 2048 pixels wide: wshift = 0) or two (wider ones, 32 image
    // words each: wshift = 1) -- x = LDS index of the first kept word the entry covers, y = which of its 32 words are kept
    const uint2 *rowtab;
    int wshift;
    int ymin, nrows;
};

// the entry of row ymin + r that covers image word wi
__device__ __forceinline__ uint32_t row_slot(const Ppht &c, uint32_t r, int wi) {
    return (r << c.wshift) + (((uint32_t)wi >> 5) & (uint32_t)c.wshift);
}
// LDS word of image word wi of its row-table entry; false if that word holds no point (all zero)
__device__ __forceinline__ bool row_word(const uint2 &e, int wi, uint32_t &widx) {
    const uint32_t b = (uint32_t)wi & 31u;
    widx = e.x + (uint32_t)__popc(e.y & ((1u << b) - 1u));
    return (e.y >> b) & 1u;
}

// Edge bit of pixel (j, i) in the LDS copy; widx is the LDS word that holds it (valid when the result is true).
__device__ __forceinline__ bool lds_mask_bit(const Ppht &c, int j, int i, uint32_t &widx) {
    const unsigned r = (unsigned)(i - c.ymin);
    if (r >= (unsigned)c.nrows) return false;
    const uint2 e = c.rowtab[row_slot(c, r, j >> 6)];
    if (!row_word(e, j >> 6, widx)) return false;
    return (c.lmask[widx] >> (j & 63)) & 1;
}

// ---- vote removal spread over the wavefronts of the workgroup (fast path) --------------------
// The line-walking wavefront (wave 0) posts one job per stretch of LINE_NC * 64 steps of an accepted
// line (most lines are one job): the walk parameters and, per direction and 64-step chunk, the ballot
// of the edge pixels whose votes go.  JOB_WORKERS wavefronts -- wave 0 itself and the helper waves --
// share a job by pixel: worker p takes the bits p, p + JOB_WORKERS, ... of every chunk, derives each
// pixel from the walk with scalar arithmetic and subtracts its vote with one angle per lane, exactly
// the shape of the voting step.  One job is in flight at a time and every worker reports before the
// next is posted, so helpers see every job exactly once.
// 64-step chunks per direction fetched (and posted) together.  One at a time with the state in LDS: the staged
// fetch of four kept 72 VGPRs alive (103 for the kernel, 77 with one -- which lets a fourth front-end wave per SIMD
// run beside this kernel), and a job per 64 steps lets the helpers take votes back while the walker clears the
// next chunk.  From global memory a round trip is expensive, so the general kernel takes four.
constexpr int LINE_NC = 1, LINE_NC_GLOBAL = 4;
// Wavefronts of a fast-kernel workgroup (template parameter WAVES): 0 consumer, 1 producer, 2.. helpers; the job
// workers are wave 0 and the helpers.  Eight when a frame has a CU to itself (the vote removal of a line is spread
// widest); four when two frames share a CU: twice eight wavefronts of 72 registers leave a SIMD room for two of
// the edge kernel's wavefronts instead of four, and everything beside HoughLinesP crawls.
struct Jobs {
    int seq;                     // jobs posted so far; -1 tells the helpers to leave
    int done;                    // workers that finished job `seq`
    int par[6];                  // xflag, x0, y0, dx, dy, first step of the stretch
    uint32_t bits[2][LINE_NC][2];
};

// What a lane needs to vote for its angle in the LDS accumulator (two biased 16-bit bins per word).
struct VoteLane {
    float c, s;                  // trig pair of this lane's angle (0 past the last angle)
    uint32_t base;               // first bin of this lane's angle row in the compact accumulator
    uint32_t one;                // 1 for a real angle, 0 past the last angle
    uint32_t rfix;               // turns the rounded float's bits into (bin - first reachable bin of this angle)
    uint32_t wid;                // reachable bins of this angle - 1
    uint32_t *words;
    uint32_t kfold;              // rfix + base: turns the rounded float's bits straight into the bin's index in the compact array
};
// cvRound of |v| < 2^22: adding 1.5 * 2^23 rounds v half-to-even into the low mantissa bits
constexpr float ROUND_MAGIC = 12582912.0f;

// bin (as a word index and a shift) this lane's angle gives the pixel (fj, fi); over is non-zero if the
// bin fell outside the angle's reachable range (cannot happen for pixels inside the points' bounding box)
__device__ __forceinline__ void vote_bin(const VoteLane &vl, float fj, float fi, uint32_t &widx, uint32_t &sh, uint32_t &over) {
    const float v = __fadd_rn(__fmul_rn(fj, vl.c), __fmul_rn(fi, vl.s));
    uint32_t r = __float_as_uint(__fadd_rn(v, ROUND_MAGIC)) + vl.rfix;   // cvRound(v) + (numrho - 1) / 2 - first bin
    const uint32_t rc = min(r, vl.wid);
    over = r - rc;
    r = rc;
    const uint32_t a = vl.base + r;
    sh = (a & 1u) << 4;
    widx = a >> 1;
}

// The same for a point of the frame, without the range test: the point lies in the box of its band of rows, x * c and
// y * s are each monotone in the coordinate and so are the rounded sum and the rounded bin, so the bin lies between those of
// that box's corners, which the range the angle's row was sized for includes (see the set-up of k_ppht_fast).
__device__ __forceinline__ void vote_bin_inside(const VoteLane &vl, float fj, float fi, uint32_t &widx, uint32_t &sh) {
    const float v = __fadd_rn(__fmul_rn(fj, vl.c), __fmul_rn(fi, vl.s));
    const uint32_t a = __float_as_uint(__fadd_rn(v, ROUND_MAGIC)) + vl.kfold;
    sh = a << 4;                 // (a shift takes the low five bits: (a & 1) << 4)
    widx = a >> 1;
}

template <int WAVES>
__device__ __forceinline__ void job_work(Jobs *jb, int part, const VoteLane &vl, int lane) {
    constexpr int JOB_WORKERS = WAVES - 1;
    Walk wk;
    wk.xflag = __builtin_amdgcn_readfirstlane(jb->par[0]); wk.x0 = __builtin_amdgcn_readfirstlane(jb->par[1]);
    wk.y0 = __builtin_amdgcn_readfirstlane(jb->par[2]); wk.dx = __builtin_amdgcn_readfirstlane(jb->par[3]);
    wk.dy = __builtin_amdgcn_readfirstlane(jb->par[4]);
    const int base = __builtin_amdgcn_readfirstlane(jb->par[5]);
    uint64_t pat = 0;
    for (int bb = part; bb < 64; bb += JOB_WORKERS) pat |= 1ull << bb;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int q = 0; q < LINE_NC; q++) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb->bits[k][q][0]);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb->bits[k][q][1]);
            uint64_t m = (((uint64_t)hi << 32) | lo) & pat;
            // two pixels at a time as straight-line code: positions first, then the bins, then the subtractions
            while (m) {
                const int t0 = base + q * 64 + __builtin_ctzll(m);
                m &= m - 1;
                const bool two = m != 0;
                const int t1 = two ? base + q * 64 + __builtin_ctzll(m) : t0;
                m &= m - 1;          // (0 & anything stays 0)
                int j0, i0, j1, i1;
                wk.at(t0, k ? -1 : 1, j0, i0);
                wk.at(t1, k ? -1 : 1, j1, i1);
                // (the pixels of a walked line whose votes go are edge pixels: points of the frame)
                uint32_t w0, s0, w1, s1;
                vote_bin_inside(vl, (float)j0, (float)i0, w0, s0);
                vote_bin_inside(vl, (float)j1, (float)i1, w1, s1);
                atomicSub(&vl.words[w0], vl.one << (s0 & 31u));
                atomicSub(&vl.words[w1], (two ? vl.one : 0u) << (s1 & 31u));
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&jb->done, 1);
}

// wave 0: do its share of the job in flight (if any) and wait for the helpers
template <int WAVES>
__device__ __forceinline__ void job_finish(Jobs *jb, bool inflight, const VoteLane &vl, int lane) {
    constexpr int JOB_WORKERS = WAVES - 1;
    if (!inflight) return;
    job_work<WAVES>(jb, 0, vl, lane);
    while (*(volatile int *)&jb->done < JOB_WORKERS) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// helper wavefronts (parts 1 .. JOB_WORKERS - 1): serve every job until told to leave
template <int WAVES>
__device__ __forceinline__ void job_serve(Jobs *jb, int part, const VoteLane &vl, int lane) {
    int seen = 0;
    for (;;) {
        int sq;
        while ((sq = *(volatile int *)&jb->seq) == seen) __builtin_amdgcn_s_sleep(2);
        if (sq < 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        seen = sq;
        job_work<WAVES>(jb, part, vl, lane);
    }
}

// N chunks (64 steps each) of both directions of a walk, fetched in one go: every load of a stage is
// issued before the first result is used (one LDS or memory round trip per stage, not per chunk).
// want[k] switches a finished direction off.  nz gets the ballots of the edge pixels, inbm of the steps
// inside the image; wword, per chunk, the LDS word of the lane's step (fast path; meaningful where the
// step's bit is set).
template <bool LDS_STATE, int N>
__device__ __forceinline__ void fetch_chunks(const Ppht &c, const Walk &wk, int lane, int sbase, const bool (&want)[2],
                                             uint64_t (&nz)[2][N], uint64_t (&inbm)[2][N], uint32_t (&wword)[2][N]) {
    uint64_t wv[2][N];
    bool inb[2][N];
    int sh[2][N];
    if constexpr (LDS_STATE) {
        uint2 ent[2][N];
        int wcol[2][N];
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < N; q++) {
                int j1, i1;
                wk.at(sbase + q * 64 + lane, k ? -1 : 1, j1, i1);
                inb[k][q] = want[k] && (unsigned)j1 < (unsigned)c.W && (unsigned)i1 < (unsigned)c.H;
                sh[k][q] = j1 & 63;
                wcol[k][q] = j1 >> 6;
                const unsigned r = (unsigned)(i1 - c.ymin);
                ent[k][q] = c.rowtab[row_slot(c, r < (unsigned)c.nrows ? r : 0u, wcol[k][q] & 63)];
                if (r >= (unsigned)c.nrows) wcol[k][q] = -1;          // no points in that row
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < N; q++) {
                uint32_t widx;
                const bool kept = row_word(ent[k][q], wcol[k][q] & 63, widx) && wcol[k][q] >= 0;
                wword[k][q] = kept ? widx : 0u;
                wv[k][q] = c.lmask[wword[k][q]];
                if (!kept) sh[k][q] = -1;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < N; q++) {
                inbm[k][q] = __ballot(inb[k][q]);
                nz[k][q] = __ballot(inb[k][q] && sh[k][q] >= 0 && ((wv[k][q] >> (sh[k][q] & 63)) & 1));
            }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < N; q++) {
                int j1, i1;
                wk.at(sbase + q * 64 + lane, k ? -1 : 1, j1, i1);
                inb[k][q] = want[k] && (unsigned)j1 < (unsigned)c.W && (unsigned)i1 < (unsigned)c.H;
                sh[k][q] = j1 & 63;
                wword[k][q] = 0;
                wv[k][q] = inb[k][q] ? ld_l2(c.mask + (size_t)i1 * c.TX + (j1 >> 6)) : 0;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < N; q++) {
                inbm[k][q] = __ballot(inb[k][q]);
                nz[k][q] = __ballot(inb[k][q] && ((wv[k][q] >> sh[k][q]) & 1));
            }
    }
}

// What the caller needs to know about a line that was walked: its pixels are on `wk` at the steps
// -end1 .. end0 (negative = the second direction).
struct LineOut {
    Walk wk;
    int end0, end1;
};

// A point reached the threshold at angle max_n: walk the line both ways, decide whether it is long
// enough, clear its mask bits and, if accepted, take its votes back and emit the segment.
// (Inlined at its single call site per kernel so that every LDS access keeps its address space and the
// constants stay in registers.)
// Returns (accepted ? 1 : 0) | jobs posted so far << 1 (segment number nseg is written if it fits).
// jobseq is the caller's count of vote-removal jobs posted so far; it lives in registers, not LDS.
// LDS_STATE: accumulator, angle table and vote-removal jobs in LDS (the fast kernel); MASK_LDS: the edge bits in LDS too (else in
// the frame's bit plane in device memory: the fast kernel's form for frames whose mask words do not fit beside the rest).
template <bool LDS_STATE, int WAVES = 8, bool MASK_LDS = LDS_STATE>
__device__ __forceinline__ int extract_line(const Ppht &c, int lane, int j, int i, int max_n, int nseg, Jobs *jb, int jobseq,
                                            const VoteLane &vl, LineOut &lo) {
    static_assert(LDS_STATE || !MASK_LDS, "the mask is never in LDS without the accumulator");
    constexpr int NC = LDS_STATE ? LINE_NC : LINE_NC_GLOBAL;
    bool inflight = false;   // a vote-removal job of this line is with the helpers
    const AngleEntry e = LDS_STATE ? c.ltab[max_n] : c.tab[max_n];
    Walk wk;
    wk.xflag = e.xflag;
    if (e.xflag) { wk.x0 = j; wk.y0 = (i << 16) + 32768; wk.dx = e.dmaj; wk.dy = e.dmin; }
    else         { wk.x0 = (j << 16) + 32768; wk.y0 = i; wk.dx = e.dmin; wk.dy = e.dmaj; }

#ifdef LC_STAMPS
    long long ts0 = STAMP();
#endif
    // pass 1: how far the line reaches in each direction.  With the mask in LDS a round trip is cheap, so
    // the walk goes chunk by chunk and stops early (most candidate lines are short); from memory it takes
    // NC chunks per round trip.  The ballots of the first NC chunks are kept: pass 2 needs them again.
    constexpr int NG = MASK_LDS ? 1 : NC;
    int end_t[2] = {0, 0};
    uint64_t nzm[2][NC] = {};
    uint32_t wword0[2][NC] = {};
    {
        bool going[2] = {true, true};
        // OpenCV stops at the first edge pixel more than line_gap + 1 steps after the previous one.
        // All 64 steps at once: `reach` has the steps within that distance of an earlier edge pixel (of
        // this chunk, or of the last one before it); the first edge pixel outside it ends the line.
        auto scan = [&](int k, int base, uint64_t nz_m, uint64_t inb_m) {
            int last_t = end_t[k];
            const int first_oob = inb_m == ~0ull ? 64 : __builtin_ctzll(~inb_m);
            if (first_oob < 64) nz_m &= (1ull << first_oob) - 1;
            bool done = false;
            if (nz_m) {
                const int G = c.line_gap + 1;
                uint64_t reach;
                if (G <= 0) {
                    reach = 0;
                } else if (G >= 64) {
                    reach = ~(nz_m ^ (nz_m - 1));            // everything above the lowest edge pixel
                } else {
                    reach = nz_m << 1;                       // distances 1 .. span
                    int span = 1;
                    while (span * 2 <= G) { reach |= reach << span; span *= 2; }
                    if (span < G) reach |= reach << (G - span);
                }
                const int from_last = G - (base - last_t);   // last step of this chunk still within reach of last_t
                if (from_last >= 63) reach = ~0ull;
                else if (from_last >= 0) reach |= (2ull << from_last) - 1;
                const uint64_t fail = nz_m & ~reach;
                if (fail) {
                    done = true;
                    nz_m &= (fail & (0 - fail)) - 1;         // the edge pixels before the first failure
                }
                if (nz_m) last_t = base + 63 - __builtin_clzll(nz_m);
            }
            if (first_oob < 64 || base + 63 - last_t > c.line_gap) done = true;
            end_t[k] = last_t;
            if (done) going[k] = false;
        };
#pragma unroll
        for (int g0 = 0; g0 < NC; g0 += NG) {
            if (going[0] || going[1]) {
                uint64_t nz[2][NG], inbm[2][NG];
                uint32_t ww[2][NG];
                fetch_chunks<MASK_LDS, NG>(c, wk, lane, g0 * 64, going, nz, inbm, ww);
#pragma unroll
                for (int k = 0; k < 2; k++)
#pragma unroll
                    for (int q = 0; q < NG; q++) {
                        nzm[k][g0 + q] = nz[k][q];
                        wword0[k][g0 + q] = ww[k][q];
                        if (going[k]) scan(k, (g0 + q) * 64, nz[k][q], inbm[k][q]);
                    }
            }
        }
        for (int sbase = NC * 64; going[0] || going[1]; sbase += NG * 64) {
            uint64_t nz[2][NG], inbm[2][NG];
            uint32_t ww[2][NG];
            fetch_chunks<MASK_LDS, NG>(c, wk, lane, sbase, going, nz, inbm, ww);
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int q = 0; q < NG; q++)
                    if (going[k]) scan(k, sbase + q * 64, nz[k][q], inbm[k][q]);
        }
    }
    int ex0, ey0, ex1, ey1;
    wk.at(end_t[0], 1, ex0, ey0);
    wk.at(end_t[1], -1, ex1, ey1);
    const bool good = abs(ex1 - ex0) >= c.line_length || abs(ey1 - ey0) >= c.line_length;
    lo.wk = wk; lo.end0 = end_t[0]; lo.end1 = end_t[1];
#ifdef LC_STAMPS
    long long ts1 = STAMP();
#endif

    // pass 2: clear the mask along the line; an accepted line takes its votes back
    const int end_max = max(end_t[0], end_t[1]);
    for (int sbase = 0; sbase <= end_max; sbase += NC * 64) {
        uint64_t bm[2][NC];
        uint32_t ww[2][NC];
        if (sbase == 0) {
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int q = 0; q < NC; q++) { bm[k][q] = nzm[k][q]; ww[k][q] = wword0[k][q]; }
        } else {
            const bool want[2] = {sbase <= end_t[0], sbase <= end_t[1]};
            uint64_t inbm[2][NC];
            fetch_chunks<MASK_LDS, NC>(c, wk, lane, sbase, want, bm, inbm, ww);
        }
        uint64_t any = 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const int base = sbase + q * 64;
                const int nvalid = end_t[k] - base + 1;                 // steps of this chunk on the line
                uint64_t m = bm[k][q];
                if (nvalid <= 0) m = 0;
                else if (nvalid < 64) m &= (1ull << nvalid) - 1;
                if (k == 1 && base == 0) m &= ~1ull;                    // t = 0 belongs to direction 0
                bm[k][q] = m;
                any |= m;
                if (m == 0) continue;
                if ((m >> lane) & 1) {
                    int j1, i1;
                    wk.at(base + lane, k ? -1 : 1, j1, i1);
                    if constexpr (MASK_LDS) {
                        atomicAnd((unsigned long long *)(c.lmask + ww[k][q]), ~(1ull << (j1 & 63)));
                    } else {
                        atomicAnd((unsigned long long *)(c.mask + (size_t)i1 * c.TX + (j1 >> 6)), ~(1ull << (j1 & 63)));
                    }
                    if constexpr (!LDS_STATE) {
                        if (good) {
                            for (int nn = 0; nn < c.numangle; nn++) {
                                int r = rho_bin(j1, i1, c.tab[nn].c, c.tab[nn].s, c.half);
                                if ((unsigned)r < (unsigned)c.numrho) atomicSub(&c.acc[(size_t)nn * c.numrho + r], 1);
                            }
                        }
                    }
                }
            }
        if constexpr (LDS_STATE) {
            if (good && any) {   // hand the stretch's vote removal to the workgroup
                job_finish<WAVES>(jb, inflight, vl, lane);       // one job in flight at a time
                ++jobseq;
                if (lane == 0) {
                    jb->par[0] = wk.xflag; jb->par[1] = wk.x0; jb->par[2] = wk.y0; jb->par[3] = wk.dx; jb->par[4] = wk.dy;
                    jb->par[5] = sbase;
#pragma unroll
                    for (int k = 0; k < 2; k++)
#pragma unroll
                        for (int q = 0; q < NC; q++) {
                            jb->bits[k][q][0] = (uint32_t)bm[k][q];
                            jb->bits[k][q][1] = (uint32_t)(bm[k][q] >> 32);
                        }
                    jb->done = 0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    *(volatile int *)&jb->seq = jobseq;
                }
                inflight = true;
            }
        }
    }
#ifdef LC_STAMPS
    long long ts2 = STAMP();
#endif
    if constexpr (LDS_STATE) job_finish<WAVES>(jb, inflight, vl, lane);   // every vote is back before the next pick
#ifdef LC_STAMPS
    long long ts3 = STAMP();
#endif
    if constexpr (!MASK_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the cleared bits are in L2 before the next pick reads them
#ifdef LC_STAMPS
    if (blockIdx.x == 0 && lane == 0) {
        g_dbg[0] += ts1 - ts0; g_dbg[1] += ts2 - ts1; g_dbg[2] += ts3 - ts2; g_dbg[3] += STAMP() - ts3;
    }
#endif
    if (good && lane == 0 && nseg < c.seg_cap) {
        c.seg[4 * nseg + 0] = ex0; c.seg[4 * nseg + 1] = ey0;
        c.seg[4 * nseg + 2] = ex1; c.seg[4 * nseg + 3] = ey1;
    }
    return (good ? 1 : 0) | (jobseq << 1);
}

// General path: every piece of state in global memory behind agent-scope atomics, any number of
// angles and points.  After the fast path it only runs the frames that path flagged.
__device__ __forceinline__ void ppht_general_frame(const int f, const int fl, int *lacc_vote, Geo g, uint32_t *__restrict__ nz_all, int nz_cap,
                                                      const int *__restrict__ npoints, uint64_t *__restrict__ rowplane,
                                                      const AngleEntry *__restrict__ tab, int numangle, int numrho,
                                                      int32_t *__restrict__ acc_all, int threshold, int line_length,
                                                      int line_gap, int32_t *__restrict__ seg_all, int seg_cap,
                                                      int *__restrict__ seg_count, int *__restrict__ status,
                                                      int32_t *__restrict__ vote_acc, int vote_apg) {
    const int tid = threadIdx.x, lane = tid & 63;
    // the full-vote accumulator of a frame the fast kernel could not do, before this kernel permutes the point list
    if (vote_acc && (fl & 2)) vote_frame(lacc_vote, f, 0, 1, nz_all, nz_cap, npoints, tab, numangle, numrho, vote_apg, vote_acc);
    if (!(fl & 1)) return;
    uint32_t *nz = nz_all + (size_t)f * nz_cap;
    Ppht c;
    c.W = g.W; c.H = g.H; c.TX = g.TX; c.numangle = numangle; c.numrho = numrho; c.half = (numrho - 1) / 2;
    c.threshold = threshold; c.line_length = line_length; c.line_gap = line_gap; c.seg_cap = seg_cap;
    c.mask = rowplane + (size_t)f * g.rowplane_words();
    c.tab = tab; c.lacc = nullptr;
    c.acc = acc_all + (size_t)f * numangle * numrho;
    c.seg = seg_all + (size_t)f * seg_cap * 4;
    const int n = npoints[f];
    if (n > nz_cap) {   // the point list was truncated: refuse rather than differ
        if (tid == 0) { seg_count[f] = 0; status[0] = 1; }
        return;
    }
    for (int i = tid; i < numangle * numrho; i += 256) c.acc[i] = 0;
    __threadfence();
    __syncthreads();
    if (tid >= 64) return;

    unsigned long long state = 0xFFFFFFFFFFFFFFFFull;
    int nseg = 0;
    for (int count = n; count > 0; count--) {
        state = (unsigned long long)(uint32_t)state * 4164903690ull + (state >> 32);
        const uint32_t idx = (uint32_t)state % (uint32_t)count;
        const uint32_t pt = ld_l2(nz + idx);
        const uint32_t last = ld_l2(nz + count - 1);
        if (lane == 0) st_l2(nz + idx, last);
        const int j = pt & 0xFFFF, i = pt >> 16;
        const uint64_t mw = ld_l2(c.mask + (size_t)i * c.TX + (j >> 6));
        if (!((mw >> (j & 63)) & 1)) continue;
        // vote; the key orders by count first, then by smaller angle index
        long long best = LLONG_MIN;
        for (int nn = lane; nn < numangle; nn += 64) {
            int r = rho_bin(j, i, tab[nn].c, tab[nn].s, c.half);
            if ((unsigned)r >= (unsigned)numrho) { status[1] = 1; continue; }
            int val = atomicAdd(&c.acc[(size_t)nn * numrho + r], 1) + 1;
            long long key = (long long)val * 4294967296ll + (long long)(0x7FFFFFFF - nn);
            best = key > best ? key : best;
        }
        for (int m = 32; m >= 1; m >>= 1) {
            long long o = shfl_xor_i64(best, m);
            best = o > best ? o : best;
        }
        if (best == LLONG_MIN) continue;
        const int max_n = 0x7FFFFFFF - (int)(unsigned)(best & 0xFFFFFFFFll);
        const int max_val = (int)((best - (long long)(0x7FFFFFFF - max_n)) / 4294967296ll);
        if (max_val < threshold) continue;
        VoteLane unused_vl{};
        LineOut unused_lo;
        if (extract_line<false>(c, lane, j, i, max_n, nseg, nullptr, 0, unused_vl, unused_lo) & 1) nseg++;
    }
    if (lane == 0) seg_count[f] = nseg;
}

// One workgroup per frame when every frame is to be done (flags null); otherwise a few workgroups that look up which
// frames the fast kernel flagged -- a wavefront's share of the flags in one load -- and do those one after the other.
// Nearly always none is flagged, and a workgroup per frame that only reads its flag still has to find a place on a
// GPU that the other kernels keep full.
__global__ __launch_bounds__(256) void k_ppht_general(Geo g, int nframes, uint32_t *__restrict__ nz_all, int nz_cap,
                                                      const int *__restrict__ npoints, uint64_t *__restrict__ rowplane,
                                                      const AngleEntry *__restrict__ tab, int numangle, int numrho,
                                                      int32_t *__restrict__ acc_all, int threshold, int line_length,
                                                      int line_gap, int32_t *__restrict__ seg_all, int seg_cap,
                                                      int *__restrict__ seg_count, int *__restrict__ status,
                                                      const int *__restrict__ fallback, int32_t *__restrict__ vote_acc,
                                                      int vote_apg, int status_frames) {
    extern __shared__ int lacc_vote[];
    const int G = gridDim.x;
    for (int base = blockIdx.x; base < nframes; base += G * 64) {
        uint64_t todo = 1;
        if (fallback) {
            const int fi = base + G * (int)(threadIdx.x & 63);
            todo = __ballot(fi < nframes && (fallback[fi] & 3) != 0);
        }
        while (todo) {
            const int f = base + G * __builtin_ctzll(todo);
            todo &= todo - 1;
            ppht_general_frame(f, fallback ? fallback[f] : 3, lacc_vote, g, nz_all, nz_cap, npoints, rowplane, tab, numangle, numrho,
                               acc_all, threshold, line_length, line_gap, seg_all, seg_cap, seg_count,
                               status + (status_frames ? 8 * (f / status_frames) : 0), vote_acc, vote_apg);
            __syncthreads();
        }
        if (!fallback) break;
    }
}

// Fast path (up to 64 angles, point list and int16 accumulator fit the CU's LDS):
//   wavefront 1 runs the data-independent part of OpenCV's loop -- the MWC random picks with
//   swap-remove -- ahead of time, leaving the visiting order in the point array itself (the removed
//   point goes to the slot the swap frees, so the array read from the back is the order);
//   wavefront 0 follows it, with the mask words of the next four points in flight, votes in LDS with
//   one angle per lane and reduces the maximum with DPP.
//   MASK_LDS = false: the same with the edge bits left in the frame's bit plane in device memory (no row table, no mask
//   words in LDS): for frames whose points are so scattered that nearly each sits in a mask word of its own -- points and
//   accumulator still fit, and a round trip to the L2 per 64 points and per 64 steps of a walk costs a fraction of what the
//   general kernel's accumulator in device memory does.
//   POINTS_LDS = false (with MASK_LDS = false): the point list stays in device memory too and is permuted in place there,
//   a round trip to the L2 per 64 picks and per 64 points; only the accumulator is in LDS.  For frames with more points
//   than fit beside it.
template <int PPHT_WAVES, bool MASK_LDS = true, bool POINTS_LDS = true>
__global__ __launch_bounds__(64 * PPHT_WAVES) void k_ppht_fast(Geo g, const uint32_t *__restrict__ nz_all, int nz_cap,
                                                   const int *__restrict__ npoints, uint64_t *__restrict__ rowplane,
                                                   const AngleEntry *__restrict__ tab, int numangle, int numrho,
                                                   int threshold, int line_length, int line_gap,
                                                   int32_t *__restrict__ seg_all, int seg_cap, int *__restrict__ seg_count,
                                                   int *__restrict__ status, int *__restrict__ fallback, int lds_points,
                                                   const uint32_t *__restrict__ rng_out, int lds_bytes,
                                                   int32_t *__restrict__ acc_full, int only_flagged, int status_frames, FrameSet fs,
                                                   FlagList fl) {
    constexpr int PPHT_THREADS = 64 * PPHT_WAVES;
    extern __shared__ __align__(16) uint32_t lds32[];
    __shared__ int progress;
    __shared__ Jobs jobs;
    __shared__ int s_maskwords, s_accbins;
    // column range of the points in every band of 2^bshift rows (at most PPHT_BANDS bands), and from those the extreme
    // bins of every angle: what sizes the angle's row of the compact accumulator
    // (these four arrays of 64 words live in the dynamic array, where the accumulator will be: they are dead before it is set up)
    constexpr int PPHT_BANDS = 64;
    __shared__ int s_part[PPHT_THREADS];
    __shared__ uint32_t a_base[64], a_rfix[64], a_wid[64];
    __shared__ int a_rmin[64];
    __shared__ AngleEntry s_tab[64];
    int f_ = blockIdx.x;
    // (the launches over flagged frames -- eight wavefronts, a whole CU's LDS -- may name their frames; the four-wavefront kernel of the shared-CU launches never does)
    if constexpr (!MASK_LDS || PPHT_WAVES == 8) { if (fl.n) f_ = fl.f[blockIdx.x]; }
    const int f = f_, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!fs.has(f)) return;
    // one launch may cover the batches of several callers (parts of a pipeline slot): every status_frames frames have
    // status words of their own
    if (status_frames) status += 8 * (f / status_frames);
#ifdef LC_STAMPS
    if (f == 0 && lane == 0) g_simd[wave] = __builtin_amdgcn_s_getreg(2308);
    const long long t_kernel0 = STAMP();
    const long long t_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    static_assert(POINTS_LDS || !MASK_LDS, "the mask words in LDS are found through the points in LDS");
    const uint32_t *nz = nz_all + (size_t)f * nz_cap;
    uint32_t *gnz = const_cast<uint32_t *>(nz);       // (POINTS_LDS = false) the list, permuted in place: L2-served accesses only
    if (only_flagged && !(fallback[f] & 1)) return;   // second launch (more LDS per frame): the frames the first one left
    const int n = npoints[f];
    if (n > nz_cap) {
        if (tid == 0) { seg_count[f] = 0; fallback[f] = 2; status[0] = 1; }
        return;
    }
    if (n == 0) {                            // no edge points: no segments, an accumulator of zeros
        if (acc_full) {
            int32_t *out = acc_full + (size_t)f * numangle * numrho;
            for (int i = tid; i < numangle * numrho; i += PPHT_THREADS) out[i] = 0;
        }
        if (tid == 0) { seg_count[f] = 0; fallback[f] = 0; }
        return;
    }
    // status[4]: some frame is left to the general kernel -- which the caller may then launch only when this is raised
    if (n > lds_points) {                    // (the host's limit: what the biased 16-bit bins and the table of random numbers hold)
        if (tid == 0) { fallback[f] = 3; seg_count[f] = -1; status[4] = 1; }
        return;
    }
    Ppht c;
    c.W = g.W; c.H = g.H; c.TX = g.TX; c.numangle = numangle; c.numrho = numrho; c.half = (numrho - 1) / 2;
    c.threshold = threshold; c.line_length = line_length; c.line_gap = line_gap; c.seg_cap = seg_cap;
    c.mask = rowplane + (size_t)f * g.rowplane_words();
    c.tab = tab; c.acc = nullptr;
    c.seg = seg_all + (size_t)f * seg_cap * 4;

    // ---- set-up: everything the loop touches moves into LDS ------------------------------------
    // layout: [points n][row table, 8 B an entry][accumulator (first the full vote, then HoughLinesP's own)][mask words]
    // The points arrive sorted by row, then column.  Of every angle's accumulator row only the bins are kept that a
    // pixel of the boxes around the points of each band of rows can reach (bands of 2^bshift rows: the points' bounding
    // box cut into at most 64 slices, which follows a triangle of lane markings far more closely than the one box does),
    // and of the mask only the 64-pixel words that hold a point (a 64-bit map per row says which) -- together a
    // fraction of the full arrays.
    uint32_t *lnz = lds32;
    const int ymin = (int)(nz[0] >> 16), ymax = (int)(nz[n - 1] >> 16);
    const int nrows = ymax - ymin + 1;
    const size_t rt_off = POINTS_LDS ? ((size_t)n + 3) & ~(size_t)3 : 0;          // 16-byte aligned
    uint32_t *rowtab = lds32 + rt_off;
    const int wshift = g.TX > 32 ? 1 : 0;
    const int nent = MASK_LDS ? nrows << wshift : 0;             // row-table entries, two words each
    const size_t rt_words = ((size_t)2 * nent + 3) & ~(size_t)3;
    if ((rt_off + rt_words + 4 * PPHT_BANDS) * 4 > (size_t)lds_bytes) {   // (the row table and the set-up's scratch alone do not fit)
        if (tid == 0) { fallback[f] = 3; seg_count[f] = -1; status[2] = 1; status[4] = 1; }
        return;
    }
    if (tid == 0) {
        progress = 0;
        jobs.seq = 0; jobs.done = 0;
    }
    int *b_lo = reinterpret_cast<int *>(lds32 + rt_off + rt_words), *b_hi = b_lo + PPHT_BANDS, *a_lo = b_hi + PPHT_BANDS, *a_hi = a_lo + 64;
    if (tid < PPHT_BANDS) { b_lo[tid] = INT_MAX; b_hi[tid] = -1; }
    if (tid < 64) { a_lo[tid] = INT_MAX; a_hi[tid] = INT_MIN; }
    int bshift = 0;
    while (((nrows - 1) >> bshift) >= PPHT_BANDS) bshift++;
    for (int r = tid; r < 2 * nent; r += PPHT_THREADS) rowtab[r] = 0;
    if (tid < numangle) s_tab[tid] = tab[tid];
    __syncthreads();
    for (int i = tid; POINTS_LDS && i < n; i += PPHT_THREADS) lnz[i] = nz[i];
    __syncthreads();
    SETUP_MARK(0);
    const uint32_t *snz = POINTS_LDS ? lnz : nz;      // the sorted list, as the set-up reads it
    for (int i = tid; i < n; i += PPHT_THREADS) {
        // the first point of every 64-pixel word marks the word in its row's map; the first and the last point of a
        // row (its leftmost and rightmost) widen the column range of the row's band
        const uint32_t pt = snz[i];
        const int x = (int)(pt & 0xFFFF), wi = x >> 6, y = (int)(pt >> 16);
        const uint32_t pv = i ? snz[i - 1] : 0xFFFFFFFFu, nx = i + 1 < n ? snz[i + 1] : 0xFFFFFFFFu;
        const bool row_first = (int)(pv >> 16) != y;
        if (MASK_LDS && (row_first || ((int)(pv & 0xFFFF) >> 6) != wi))
            atomicOr(&rowtab[2 * ((((uint32_t)(y - ymin)) << wshift) + ((uint32_t)(wi >> 5) & (uint32_t)wshift)) + 1], 1u << (wi & 31));
        if (row_first) atomicMin(&b_lo[(y - ymin) >> bshift], x);
        if ((int)(nx >> 16) != y) atomicMax(&b_hi[(y - ymin) >> bshift], x);
    }
    __syncthreads();
    SETUP_MARK(1);
    {   // exclusive scan of the kept-word counts over the rows -> LDS index of every row's first word
        const int per = (nent + PPHT_THREADS - 1) / PPHT_THREADS;
        const int r0 = tid * per, r1 = min(nent, r0 + per);
        int sum = 0;
        for (int r = r0; r < r1; r++) sum += __popc(rowtab[2 * r + 1]);
        s_part[tid] = sum;
        __syncthreads();
        for (int d = 1; d < PPHT_THREADS; d <<= 1) {
            const int v = tid >= d ? s_part[tid - d] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        int off = s_part[tid] - sum;
        for (int r = r0; r < r1; r++) {
            rowtab[2 * r] = (uint32_t)off;
            off += __popc(rowtab[2 * r + 1]);
        }
        if (tid == PPHT_THREADS - 1) s_maskwords = s_part[tid];
    }
    SETUP_MARK(2);
    {   // Reachable bins of every angle (one per lane): x * c and y * s are each monotone in the coordinate and so are the
        // rounded sum and the rounded bin, so over a box the extremes are at its corners -- here the boxes of the bands
        // (the wavefronts share them), each as wide as its points' columns and as high as the band's rows.
        float ac = 0.f, as = 0.f;
        if (lane < numangle) { ac = tab[lane].c; as = tab[lane].s; }
        float vlo = 3.0e38f, vhi = -3.0e38f;
        for (int b = wave; b < PPHT_BANDS; b += PPHT_WAVES) {
            const int xl = b_lo[b], xr = b_hi[b];
            if (xr < 0) continue;                   // (no point in the band)
            const float y0 = (float)(ymin + (b << bshift)), y1 = (float)min(ymin + ((b + 1) << bshift) - 1, ymax);
            const float cl = __fmul_rn((float)xl, ac), cr = __fmul_rn((float)xr, ac);
            const float s0 = __fmul_rn(y0, as), s1 = __fmul_rn(y1, as);
            const float v0 = __fadd_rn(cl, s0), v1 = __fadd_rn(cr, s0), v2 = __fadd_rn(cl, s1), v3 = __fadd_rn(cr, s1);
            vlo = fminf(vlo, fminf(fminf(v0, v1), fminf(v2, v3)));
            vhi = fmaxf(vhi, fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
        }
        if (lane < numangle && vlo <= vhi) {
            atomicMin(&a_lo[lane], (int)(__float_as_uint(__fadd_rn(vlo, ROUND_MAGIC)) - 0x4B400000u));
            atomicMax(&a_hi[lane], (int)(__float_as_uint(__fadd_rn(vhi, ROUND_MAGIC)) - 0x4B400000u));
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int rmin = a_lo[lane < numangle ? lane : 0] + c.half, rmax = a_hi[lane < numangle ? lane : 0] + c.half;
        const int nb = lane < numangle ? rmax - rmin + 1 : 0;
        int incl = nb;
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        a_base[lane] = lane < numangle ? (uint32_t)(incl - nb) : 0u;
        a_wid[lane] = lane < numangle ? (uint32_t)(rmax - rmin) : 0u;
        a_rfix[lane] = (uint32_t)(c.half - rmin) - 0x4B400000u;
        a_rmin[lane] = rmin;
        if (lane == 63) s_accbins = incl;
        // a table whose bins leave [0, numrho) is not OpenCV's: refuse
        if (lane < numangle && (rmin < 0 || rmax >= numrho)) status[1] = 1;
    }
    __syncthreads();
    const int accbins = s_accbins, maskwords = s_maskwords;
    const size_t acc_off = rt_off + rt_words;                                      // in 32-bit words, 16-byte aligned
    const size_t acc_words = ((size_t)accbins + 1) / 2;
    // The full-vote accumulator, when wanted, is made first, in the same compact array, by all wavefronts: every point
    // votes once per angle (one angle per lane, plain 16-bit counts: there are fewer than 2^15 points), the array goes
    // out in OpenCV's [angle][rho] shape and is then handed to HoughLinesP.  A few per cent of the frame's time (the
    // sequential loop below dominates), against a second array of the same size that would keep a CU's LDS from
    // holding two frames.
    const size_t mask_off = (acc_off + acc_words + 1) & ~(size_t)1;
    if ((mask_off + 2 * (size_t)maskwords) * 4 > (size_t)lds_bytes) {
        if (tid == 0) { fallback[f] = 3; seg_count[f] = -1; status[2] = 1; status[4] = 1; }   // does not fit: another kernel (and its vote) take the frame
        return;
    }
    if (tid == 0) fallback[f] = 0;
    c.lacc = reinterpret_cast<int16_t *>(lds32 + acc_off);
    c.lmask = reinterpret_cast<uint64_t *>(lds32 + mask_off);
    c.rowtab = reinterpret_cast<const uint2 *>(rowtab); c.wshift = wshift; c.ymin = ymin; c.nrows = nrows;
    c.ltab = s_tab;
    // lane l votes for angle l; lanes past the last angle add zero to a harmless bin instead of branching
    VoteLane vl;
    vl.c = vl.s = 0.f;
    if (lane < numangle) { vl.c = tab[lane].c; vl.s = tab[lane].s; }
    vl.base = a_base[lane];
    vl.one = lane < numangle ? 1u : 0u;
    vl.rfix = a_rfix[lane];
    vl.wid = a_wid[lane];
    vl.words = reinterpret_cast<uint32_t *>(c.lacc);
    vl.kfold = vl.rfix + vl.base;
    SETUP_MARK(3);
    if (acc_full != nullptr) {
        uint32_t *fw = vl.words;
        for (int i = tid; i < (int)acc_words; i += PPHT_THREADS) fw[i] = 0;
        __syncthreads();
        for (int p0 = wave * 64; p0 < n; p0 += PPHT_WAVES * 64) {
            const int m = min(64, n - p0);
            const uint32_t ptv = lane < m ? snz[p0 + lane] : 0u;
            const float fjv = (float)(ptv & 0xFFFF), fiv = (float)(ptv >> 16);
            // four points at a time as straight-line code (coordinates out of the lanes first, then the four bins, then
            // the four adds): one point per iteration left the wavefront waiting on every step of its own chain
            int t = 0;
            for (; t + 4 <= m; t += 4) {
                float fj[4], fi[4];
                uint32_t widx[4], sh[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    fj[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fjv), t + q));
                    fi[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fiv), t + q));
                }
#pragma unroll
                for (int q = 0; q < 4; q++) vote_bin_inside(vl, fj[q], fi[q], widx[q], sh[q]);
#pragma unroll
                for (int q = 0; q < 4; q++) atomicAdd(&fw[widx[q]], vl.one << (sh[q] & 31u));
            }
            for (; t < m; t++) {
                const float fj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fjv), t));
                const float fi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fiv), t));
                uint32_t widx, sh;
                vote_bin_inside(vl, fj, fi, widx, sh);
                atomicAdd(&fw[widx], vl.one << (sh & 31u));
            }
        }
        __syncthreads();
        SETUP_MARK(4);
        // write-out in OpenCV's [angle][rho] shape: zeros outside the reachable range of an angle
        for (int a = wave; a < numangle; a += PPHT_WAVES) {
            const int rmin = a_rmin[a], wid = (int)a_wid[a];
            const uint32_t base = a_base[a];
            int32_t *out = acc_full + ((size_t)f * numangle + a) * numrho;
            for (int r = lane; r < numrho; r += 64) {
                const int k = r - rmin;
                int v = 0;
                if ((unsigned)k <= (unsigned)wid) {
                    const uint32_t bin = base + (uint32_t)k;
                    v = (int)((fw[bin >> 1] >> ((bin & 1u) << 4)) & 0xFFFFu);
                }
                out[r] = v;
            }
        }
        __syncthreads();
    }
    SETUP_MARK(5);
    for (int i = tid; i < (int)acc_words; i += PPHT_THREADS) lds32[acc_off + i] = (uint32_t)LDS_BIAS * 0x10001u;
    // the kept mask words: the first point of every 64-pixel word fetches the word.  Four points per thread at a time,
    // address arithmetic first, then the loads together, then the stores: one round trip to memory per four words,
    // not one per word (a loop over the rows, a row per wavefront, spent a quarter of the set-up waiting for them).
    for (int i0 = tid; MASK_LDS && i0 < n; i0 += 4 * PPHT_THREADS) {
        uint32_t widx[4];
        const uint64_t *src[4];
        bool first[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * PPHT_THREADS;
            first[k] = false;
            widx[k] = 0;
            src[k] = c.mask;
            if (i < n) {
                const uint32_t pt = lnz[i];
                const int wi = (int)(pt & 0xFFFF) >> 6, y = (int)(pt >> 16);
                const uint32_t pv = i ? lnz[i - 1] : 0u;
                first[k] = i == 0 || (int)(pv >> 16) != y || ((int)(pv & 0xFFFF) >> 6) != wi;
                if (first[k]) {
                    (void)row_word(c.rowtab[row_slot(c, (uint32_t)(y - ymin), wi)], wi, widx[k]);
                    src[k] = c.mask + (size_t)y * c.TX + wi;
                }
            }
        }
        uint64_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = first[k] ? *src[k] : 0ull;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (first[k]) c.lmask[widx[k]] = w[k];
    }
    __syncthreads();
    SETUP_MARK(6);
    if (wave >= 2) {                // helpers: vote removal for accepted lines
        job_serve<PPHT_WAVES>(&jobs, wave - 1, vl, lane);
        return;
    }

    if (wave == 1) {   // producer: visiting order, in place
#ifdef LC_STAMPS
        long long tp0 = STAMP();
#endif
        // rng_out[k] is the k-th output of cv::RNG(2^64 - 1) (fixed seed, so a host-made table).
        // 64 picks at a time: lane t holds pick t's index and the two array elements its swap-remove
        // reads, as they were before the batch.  The swaps then run in order on those registers -- a
        // write of an earlier pick is forwarded to the later lanes that read the same slot -- and the
        // array is updated with two vector stores, instead of 64 dependent LDS round trips.
        uint32_t rnd = lane < n ? rng_out[lane] : 0u;         // the random numbers are fetched a batch ahead
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            const int m = min(64, n - k0);
            const bool on = lane < m;
            const uint32_t rcur = rnd;
            if (k + 64 < n) rnd = rng_out[k + 64];
            uint32_t idx = 0xFFFFFFFFu, cpos = 0xFFFFFFFEu;   // slot picked / last slot of the shrinking array
            uint32_t acur = 0, bcur = 0;
            if (on) {
                idx = rcur % (uint32_t)(n - k);
                cpos = (uint32_t)(n - k - 1);
                acur = POINTS_LDS ? lnz[idx] : ld_l2(gnz + idx);
                bcur = POINTS_LDS ? lnz[cpos] : ld_l2(gnz + cpos);
            }
            uint64_t overwritten = 0;   // picks whose slot a later pick of the batch writes again
            for (int t = 0; t < m; t++) {
                const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)idx, t);
                const uint64_t later = ~((2ull << t) - 1);
                // later picks that read the slot this pick writes: as their own pick / as their last element
                const uint64_t same_a = __ballot(idx == it) & later, same_b = __ballot(cpos == it) & later;
                if ((same_a | same_b) == 0) continue;       // the common case: nobody to forward to
                const uint32_t lastv = (uint32_t)__builtin_amdgcn_readlane((int)bcur, t);
                acur = ((same_a >> lane) & 1) ? lastv : acur;     // the slot now holds what was the last element
                bcur = ((same_b >> lane) & 1) ? lastv : bcur;
                if (same_a) overwritten |= 1ull << t;
            }
            // lane t: acur = the point pick t removes, bcur = the last element that takes its slot
            if constexpr (POINTS_LDS) {
                if (on && !((overwritten >> lane) & 1)) lnz[idx] = bcur;
                if (on) lnz[cpos] = acur;             // after the slot writes: a later pick's own slot wins
                if (lane == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    *(volatile int *)&progress = k0 + m;
                }
            } else {
                // in device memory no two stores of a batch may meet in a word (nothing orders them there): a picked slot
                // among the batch's last m slots is some later pick's own slot, whose store wins -- it is left out here
                if (on && !((overwritten >> lane) & 1) && idx < (uint32_t)(n - k0 - m)) st_l2(gnz + idx, bcur);
                if (on) st_l2(gnz + cpos, acur);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the stores are in the L2 before the consumer (and the next batch's loads) look
                if (lane == 0) *(volatile int *)&progress = k0 + m;
            }
        }
#ifdef LC_STAMPS
        if (f == 0 && lane == 0) g_stamps[0] = STAMP() - tp0;
#endif
        return;
    }

    // consumer.  Points are taken 64 at a time: lane t loads point k + t and its mask word, a ballot
    // says which are still alive (dead ones cost nothing).  Alive points vote Q at a time with
    // returning LDS atomics issued back to back; if one of them reaches the threshold the votes of the
    // later ones are taken back and the line is extracted.  The batch then goes on behind that point
    // without touching memory: a batch point died iff it lies on the walked stretch of the line, which
    // every lane checks for its own point.  Every count any point sees is therefore the one OpenCV's
    // strictly sequential loop would see.
    uint32_t *words = vl.words;
    const uint64_t lanes_mask = numangle >= 64 ? ~0ull : ((1ull << numangle) - 1);
    // count after the vote >= threshold  <=>  biased 16-bit bin before the vote >= threshold - 1 + bias
    const uint32_t thr_b = (uint32_t)min(max((long long)threshold - 1 + LDS_BIAS, 0ll), 0x10000ll);
    uint32_t r_over = 0;   // non-zero if a vote ever left its angle's reachable range (a bad table)
    int nseg = 0, jobseq = 0;
#ifdef LC_STAMPS
    long long t_wait = 0, t_vote = 0, t_line = 0, n_alive = 0, n_line = 0, tc0 = STAMP();
    long long n_groups = 0, t_pre = 0, n_trig_batches = 0;
#endif
    for (int k = 0; k < n; k += 64) {
        const int m = min(64, n - k);
#ifdef LC_STAMPS
        long long tw = STAMP();
#endif
        while (*(volatile int *)&progress < k + m) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t ptv = 0;
        bool alive_l = false;
        if (lane < m) {
            ptv = POINTS_LDS ? lnz[n - 1 - (k + lane)] : ld_l2(gnz + (n - 1 - (k + lane)));
            uint32_t widx;
            if constexpr (MASK_LDS) alive_l = lds_mask_bit(c, (int)(ptv & 0xFFFF), (int)(ptv >> 16), widx);
            else alive_l = (ld_l2(c.mask + (size_t)(ptv >> 16) * c.TX + ((ptv & 0xFFFF) >> 6)) >> (ptv & 63)) & 1;
        }
        const int pjv = (int)(ptv & 0xFFFF), piv = (int)(ptv >> 16);
        const float fjv = (float)pjv, fiv = (float)piv;
        uint64_t alive = __ballot(alive_l);
#ifdef LC_STAMPS
        t_wait += STAMP() - tw;
        n_alive += __popcll(alive);
        bool trig_here = false;
#endif
        while (alive) {
            constexpr int Q = PPHT_Q;
            const uint64_t alive0 = alive;
            int tq[Q];
            uint32_t inc[Q], widx[Q], sh[Q], old[Q];
            uint64_t vmask[Q];
#ifdef LC_STAMPS
            long long tv = STAMP();
#endif
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const bool valid = alive != 0;
                tq[q] = valid ? __builtin_ctzll(alive) : 0;
                alive &= alive - 1;
                vmask[q] = valid ? lanes_mask : 0ull;
                const float fj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fjv), tq[q]));
                const float fi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fiv), tq[q]));
                uint32_t over;
                vote_bin(vl, fj, fi, widx[q], sh[q], over);
                r_over |= over;
                inc[q] = (valid ? vl.one : 0u) << sh[q];
                old[q] = atomicAdd(&words[widx[q]], inc[q]);
            }
            // a point starts a line iff some angle's count reaches the threshold: one compare and a ballot
            // per point; the arg-max over the angles is only needed for the rare point that does
            uint64_t trig[Q], any = 0;
            uint32_t val[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                val[q] = (old[q] >> sh[q]) & 0xFFFFu;
                trig[q] = __ballot(val[q] >= thr_b) & vmask[q];
                any |= trig[q];
            }
#ifdef LC_STAMPS
            t_vote += STAMP() - tv;
            n_groups++;
            long long tpre = STAMP();
#endif
            if (any) {
                int hit = Q - 1;   // first point of the group, in OpenCV's order, that starts a line
#pragma unroll
                for (int q = Q - 2; q >= 0; q--)
                    if (trig[q] != 0) hit = q;
                uint32_t hval = val[0];
                int ht = tq[0];
#pragma unroll
                for (int q = 1; q < Q; q++)
                    if (hit == q) { hval = val[q]; ht = tq[q]; }
                const uint32_t hpt = (uint32_t)__builtin_amdgcn_readlane((int)ptv, ht);
                // count first, then the smaller angle index (OpenCV keeps the first maximum)
                const int kmax = wave_max_i32(lane < numangle ? (int)(hval * 64u) + (63 - lane) : -1);
                const int max_n = 63 - (kmax & 63);
                // the points behind it have not voted yet in OpenCV's order: take their votes back
#pragma unroll
                for (int e = 1; e < Q; e++)
                    if (e > hit) atomicSub(&words[widx[e]], inc[e]);
#ifdef LC_STAMPS
                long long tl = STAMP();
                t_pre += tl - tpre;
                n_line++;
                if (!trig_here) { trig_here = true; n_trig_batches++; }
#endif
                LineOut lo;
                const int lr = extract_line<true, PPHT_WAVES, MASK_LDS>(c, lane, (int)(hpt & 0xFFFF), (int)(hpt >> 16), max_n, nseg, &jobs, jobseq, vl, lo);
                jobseq = lr >> 1;
                if (lr & 1) nseg++;
                // which points of the batch the line took: signed step of the lane's point along the major axis
                int st, minor;
                if (lo.wk.xflag) { st = (pjv - lo.wk.x0) * lo.wk.dx; minor = ((lo.wk.y0 + st * lo.wk.dy) >> 16) - piv; }
                else             { st = (piv - lo.wk.y0) * lo.wk.dy; minor = ((lo.wk.x0 + st * lo.wk.dx) >> 16) - pjv; }
                const bool on_line = minor == 0 && st <= lo.end0 && st >= -lo.end1;
                alive = alive0 & ~((2ull << ht) - 1) & ~__ballot(on_line);
#ifdef LC_STAMPS
                t_line += STAMP() - tl;
#endif
            }
        }
    }
    if (__ballot(r_over != 0 && lane < numangle) != 0 && lane == 0) status[1] = 1;
    if (lane == 0) *(volatile int *)&jobs.seq = -1;   // helpers leave
    if (lane == 0) seg_count[f] = nseg;
#ifdef LC_STAMPS
    if (f == 0 && lane == 0) {
        g_stamps[1] = STAMP() - tc0; g_stamps[2] = t_wait; g_stamps[3] = t_vote; g_stamps[4] = t_line;
        g_stamps[5] = n_alive; g_stamps[6] = n_line; g_stamps[7] = n;
        g_more[0] = n_groups; g_more[1] = n_alive; g_more[2] = t_pre; g_more[3] = n_trig_batches; g_more[4] = nseg;
    }
    if (f < 512 && lane == 0) {
        g_frame_cycles[2 * f] = STAMP() - tc0; g_frame_cycles[2 * f + 1] = STAMP() - t_kernel0;
        g_frame_real[f] = __builtin_amdgcn_s_memrealtime() - t_real0;
    }
#endif
}

}  // namespace

hipError_t LdsOptIn::ensure(const void *kernel, size_t bytes) {
    static std::mutex mu;
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    const int slot = dev & 63;
    if (have[slot] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have[slot] = bytes;
    return e;
}

void launch_points(Geo g, int nframes, const uint64_t *rowplane, int *rowcount, uint32_t *nz, int nz_cap, int *npoints,
                   int *status, hipStream_t st, const FrameSet *fs) {
    hipLaunchKernelGGL(k_points, dim3(nframes), dim3(64 * POINTS_WAVES), 0, st, g, rowplane, rowcount, nz, nz_cap, npoints, status,
                       fs ? *fs : FrameSet{});
}

// angle rows per workgroup of the vote: what fits 6 KiB of LDS, at least one.  Small on purpose: the kernel then
// shares CUs with the HoughLinesP kernel of another batch, which leaves only a few KiB of LDS free.
static int vote_apg(int numangle, int numrho, int &groups, int budget = 6 * 1024) {
    int apg = budget / ((numrho + 1) * 4);
    if (apg < 1) apg = 1;
    if (apg > numangle) apg = numangle;
    groups = (numangle + apg - 1) / apg;
    return (numangle + groups - 1) / groups;
}

void launch_vote(Geo g, int nframes, const uint32_t *nz, int nz_cap, const int *npoints, const AngleEntry *tab,
                 int numangle, int numrho, int32_t *acc, const int *flags, hipStream_t st, const FrameSet *fs) {
    (void)g;
    int groups;
    // on its own (no flags) the kernel does not share CUs with HoughLinesP: 16 KiB of rows per workgroup (eight
    // workgroups fill a CU's wave slots), so that a workgroup's pass over the frame's points serves more angles;
    // 32 KiB was slower (0.054 against 0.040 ms for 256 1080p frames), 8 KiB too (0.079)
    const int budget = 16 * 1024;
    const int apg = flags ? vote_apg(numangle, numrho, groups) : vote_apg(numangle, numrho, groups, budget);
    size_t lds = (size_t)apg * (numrho + 1) * 4;
    static LdsOptIn optin;
    (void)optin.ensure(reinterpret_cast<const void *>(k_vote), lds);
    // the flagged launch (often no frame has work) is at most 8 blocks per frame: it must not queue thousands of
    // blocks that only look at a flag behind the waves of the kernels it shares the GPU with
    hipLaunchKernelGGL(k_vote, dim3(flags ? std::min(groups, 8) : groups, nframes), dim3(256), lds, st, nz, nz_cap, npoints, tab, numangle,
                       numrho, apg, acc, flags, fs ? *fs : FrameSet{});
}

int ppht_lds_half() {
    // LANECV_PPHT_LDS_KB (diagnostics, 16 .. 72, halves allowed): another size for the shared-CU launches, e.g. 49.5 for three frames per CU
    static const int half_kb = [] {      // in units of 512 bytes
        const char *e = getenv("LANECV_PPHT_LDS_KB");
        const double v = e ? atof(e) : 0.0;
        return (v >= 16.0 && v <= 72.0) ? (int)(2.0 * v) : 0;
    }();
    return half_kb ? half_kb * 512 : PPHT_LDS_HALF;
}

// HoughLinesP in two steps, so that the caller can put the flagged vote kernel between them (the general
// kernel permutes the point list in place).  flags[f]: bit 0 = the general kernel must do frame f,
// bit 1 = the frame's full-vote accumulator (acc_full, may be null) was not produced here.
bool launch_ppht_fast(Geo g, int nframes, uint32_t *nz, int nz_cap, const int *npoints, uint64_t *rowplane,
                      const AngleEntry *tab, int numangle, int numrho, int threshold, int line_length, int line_gap,
                      int32_t *seg, int seg_cap, int *seg_count, int *status, int *flags, const uint32_t *rng_out,
                      int rng_len, int32_t *acc_full, hipStream_t st, int lds_bytes, bool only_flagged, int waves,
                      int status_frames, const FrameSet *fs, int device_state, const FlagList *list) {
    // points, compact int16 accumulator and the mask spans in LDS; a frame whose set does not fit lds_bytes (the
    // kernel decides, per frame) is flagged.
    if (numangle > 64 || g.W > 4096) return false;   // one lane per angle; one map bit per 64-pixel word of a row
    if (lds_bytes <= 0 || lds_bytes > PPHT_LDS_MAX) lds_bytes = PPHT_LDS_MAX;
    // A bin's count stays inside (-n, n) for n points: fewer than LDS_BIAS - 8 keep the biased 16-bit bins in range.
    int lds_points = std::min(rng_len, LDS_BIAS - 9);
    static LdsOptIn optin8, optin4, optin8g, optin8p;
    if (device_state) {       // the forms for many or scattered points: a frame alone on its CU, eight wavefronts
        static const bool off = getenv("LANECV_NO_SCATTER_KERNEL") != nullptr;   // diagnostics: straight to the general kernel
        if (off) return true;
        if (device_state == 1) {
            if (optin8g.ensure(reinterpret_cast<const void *>(k_ppht_fast<8, false>), (size_t)PPHT_LDS_MAX) != hipSuccess) return false;
            hipLaunchKernelGGL((k_ppht_fast<8, false>), dim3(list && list->n ? list->n : nframes), dim3(512), lds_bytes, st, g, nz, nz_cap, npoints, rowplane, tab, numangle,
                               numrho, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, flags, lds_points, rng_out,
                               lds_bytes, acc_full, only_flagged ? 1 : 0, status_frames, fs ? *fs : FrameSet{}, list ? *list : FlagList{});
            return true;
        }
        // Points in device memory: their number is no longer bounded by the LDS.  A bin's count is also bounded by the
        // pixels whose votes can land in it -- a stripe one bin (rho pixels) wide across the frame: per row (or per
        // column, whichever the stripe crosses more steeply) at most floor(sqrt(2) * rho) + 1 pixels -- and where that
        // bound keeps the 16-bit bins in range, the table of random numbers alone limits the points.
        const double rho_est = (2.0 * (g.W + g.H) + 1.0) / std::max(1.0, numrho - 0.5);      // numrho = round((2 (W + H) + 1) / rho)
        const long long stripe = (long long)std::max(g.W, g.H) * ((long long)std::floor(1.4143 * rho_est) + 3);
        if (stripe < LDS_BIAS - 8) lds_points = rng_len;
        if (optin8p.ensure(reinterpret_cast<const void *>(k_ppht_fast<8, false, false>), (size_t)PPHT_LDS_MAX) != hipSuccess) return false;
        hipLaunchKernelGGL((k_ppht_fast<8, false, false>), dim3(list && list->n ? list->n : nframes), dim3(512), lds_bytes, st, g, nz, nz_cap, npoints, rowplane, tab, numangle,
                           numrho, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, flags, lds_points, rng_out,
                           lds_bytes, acc_full, only_flagged ? 1 : 0, status_frames, fs ? *fs : FrameSet{}, list ? *list : FlagList{});
        return true;
    }
    static const char *wv_env = getenv("LANECV_PPHT_WAVES");           // diagnostics: 4 or 8 whatever the LDS size
    const bool four = wv_env ? wv_env[0] == '4' : (waves ? waves == 4 : lds_bytes <= PPHT_LDS_HALF);
    if (four) {
        if (optin4.ensure(reinterpret_cast<const void *>(k_ppht_fast<4>), (size_t)PPHT_LDS_MAX) != hipSuccess) return false;
        hipLaunchKernelGGL(k_ppht_fast<4>, dim3(nframes), dim3(256), lds_bytes, st, g, nz, nz_cap, npoints, rowplane, tab, numangle,
                           numrho, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, flags, lds_points, rng_out,
                           lds_bytes, acc_full, only_flagged ? 1 : 0, status_frames, fs ? *fs : FrameSet{}, list ? *list : FlagList{});
    } else {
        if (optin8.ensure(reinterpret_cast<const void *>(k_ppht_fast<8>), (size_t)PPHT_LDS_MAX) != hipSuccess) return false;
        hipLaunchKernelGGL(k_ppht_fast<8>, dim3(list && list->n ? list->n : nframes), dim3(512), lds_bytes, st, g, nz, nz_cap, npoints, rowplane, tab, numangle,
                           numrho, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, flags, lds_points, rng_out,
                           lds_bytes, acc_full, only_flagged ? 1 : 0, status_frames, fs ? *fs : FrameSet{}, list ? *list : FlagList{});
    }
    return true;
}

void launch_ppht_general(Geo g, int nframes, uint32_t *nz, int nz_cap, const int *npoints, uint64_t *rowplane,
                         const AngleEntry *tab, int numangle, int numrho, int32_t *acc, int threshold, int line_length,
                         int line_gap, int32_t *seg, int seg_cap, int *seg_count, int *status, const int *flags,
                         hipStream_t st, int32_t *vote_acc, int status_frames) {
    int groups, apg = 0;
    size_t lds = 0;
    if (vote_acc) {
        apg = vote_apg(numangle, numrho, groups);
        lds = (size_t)apg * (numrho + 1) * 4;
        static LdsOptIn optin;
        (void)optin.ensure(reinterpret_cast<const void *>(k_ppht_general), lds);
    }
    hipLaunchKernelGGL(k_ppht_general, dim3(flags ? std::min(nframes, 16) : nframes), dim3(256), lds, st, g, nframes, nz, nz_cap, npoints,
                       rowplane, tab, numangle, numrho, acc, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, flags,
                       vote_acc, apg, status_frames);
}

#ifdef LC_STAMPS
extern "C" int lc_debug_stamps(long long *out) {
    (void)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_dbg), sizeof(long long) * 4);
    (void)hipMemcpyFromSymbol(out + 12, HIP_SYMBOL(g_simd), sizeof(int) * 8);
    (void)hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_more), sizeof(long long) * 8);
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 8);
}
extern "C" int lc_debug_setup(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_setup), sizeof(long long) * 16); }
extern "C" int lc_debug_frame_cycles(long long *out, int nframes) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_frame_cycles), sizeof(long long) * 2 * (nframes < 512 ? nframes : 512));
}
extern "C" int lc_debug_frame_real(long long *out, int nframes) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_frame_real), sizeof(long long) * (nframes < 512 ? nframes : 512));
}
#endif

}  // namespace lanecv
