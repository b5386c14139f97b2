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
#include <algorithm>

namespace lanecv {
namespace {

__device__ __forceinline__ int rho_bin(int x, int y, float c, float s, int half) {
    return __float2int_rn(__fadd_rn(__fmul_rn((float)x, c), __fmul_rn((float)y, s))) + half;
}

// exclusive scan of the per-row point counts of one frame
__global__ __launch_bounds__(256) void k_rowscan(int H, const int *__restrict__ rowcount, int *__restrict__ rowoff,
                                                 int *__restrict__ npoints) {
    __shared__ int part[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int per = (H + 255) / 256;
    const int r0 = tid * per, r1 = min(H, r0 + per);
    int s = 0;
    for (int r = r0; r < r1; r++) s += rowcount[(size_t)f * H + r];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int off = part[tid] - s;
    for (int r = r0; r < r1; r++) {
        rowoff[(size_t)f * H + r] = off;
        off += rowcount[(size_t)f * H + r];
    }
    if (tid == 255) npoints[f] = part[255];
}

// one wavefront per (frame, row): points in left-to-right order at the row's offset.  Empty rows
// (most of the image outside the ROI) leave at once; otherwise lane l takes word l of the row, a
// wave scan of the popcounts places each word's points.
__global__ __launch_bounds__(256) void k_scatter(Geo g, int nframes, const uint64_t *__restrict__ rowplane,
                                                 const int *__restrict__ rowcount, const int *__restrict__ rowoff,
                                                 uint32_t *__restrict__ nz, int nz_cap) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)nframes * g.H) return;
    if (rowcount[row] == 0) return;
    const int f = (int)(row / g.H), y = (int)(row % g.H);
    int off = rowoff[row];
    uint32_t *out = nz + (size_t)f * nz_cap;
    for (int tx0 = 0; tx0 < g.TX; tx0 += 64) {
        const int tx = tx0 + lane;
        uint64_t w = tx < g.TX ? rowplane[(size_t)row * g.TX + tx] : 0;
        const int c = __popcll(w);
        int incl = c;   // inclusive scan over the lanes
        for (int d = 1; d < 64; d <<= 1) {
            int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        int pos = off + incl - c;
        while (w) {
            int b = __builtin_ctzll(w);
            if (pos < nz_cap) out[pos] = (uint32_t)(tx * 64 + b) | ((uint32_t)y << 16);
            pos++;
            w &= w - 1;
        }
        off += __shfl(incl, 63);
    }
}

// full-vote accumulator: block (group, frame) owns `apg` angle rows in LDS
__global__ __launch_bounds__(256) void k_vote(const uint32_t *__restrict__ nz, int nz_cap, const int *__restrict__ npoints,
                                               const AngleEntry *__restrict__ tab, int numangle, int numrho, int apg,
                                               int32_t *__restrict__ acc) {
    extern __shared__ int lacc[];
    const int f = blockIdx.y, a0 = blockIdx.x * apg;
    const int na = min(apg, numangle - a0);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < na * numrho; i += nt) lacc[i] = 0;
    __syncthreads();
    const int n = min(npoints[f], nz_cap);
    const int half = (numrho - 1) / 2;
    const uint32_t *pts = nz + (size_t)f * nz_cap;
    for (int p = tid; p < n; p += nt) {
        uint32_t pt = pts[p];
        int x = pt & 0xFFFF, y = pt >> 16;
        for (int a = 0; a < na; a++) {
            int r = rho_bin(x, y, tab[a0 + a].c, tab[a0 + a].s, half);
            if ((unsigned)r < (unsigned)numrho) atomicAdd(&lacc[a * numrho + r], 1);
        }
    }
    __syncthreads();
    int32_t *out = acc + ((size_t)f * numangle + a0) * numrho;
    for (int i = tid; i < na * numrho; i += nt) out[i] = lacc[i];
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
__device__ unsigned long long g_dbg[4];
#define STAMP() __builtin_amdgcn_s_memtime()
#endif
#ifndef PPHT_Q
#define PPHT_Q 4   // points voting speculatively per group in the LDS path
#endif
constexpr int LDS_BIAS = 0x4000;   // offset of the 16-bit accumulator bins kept in LDS

// Per-frame constants of one HoughLinesP run.
struct Ppht {
    int W, H, TX, numangle, numrho, half, threshold, line_length, line_gap, seg_cap;
    uint64_t *mask;            // row-major bit plane of the frame (bits are cleared as lines are found)
    const AngleEntry *tab;
    int16_t *lacc;             // LDS accumulator (fast path)
    int32_t *acc;              // global accumulator (general path)
    int32_t *seg;
};

// ---- vote removal spread over the wavefronts of the workgroup (fast path) --------------------
// The line-walking wavefront (wave 0) posts one job per 64-step chunk of an accepted line: the
// chunk's walk parameters and the ballot of its edge pixels.  Three wavefronts share a job
// statically -- wave 0 and the two helper waves take angle groups p, p + 3, p + 6, ... -- and
// subtract the chunk's votes for their angles (lane = pixel).  One job is in flight at a time and
// every participant reports before the next is posted, so helpers see every job exactly once.
constexpr int JOB_ANGLES = 5, JOB_WAVES = 3;
struct Jobs {
    int seq;                     // jobs posted so far; -1 tells the helpers to leave
    int done;                    // participants that finished job `seq`
    int par[16];                 // xflag, x0, y0, dx, dy, sgn, base, bits lo, bits hi
    float tc[64], ts[64];        // trig table of the (at most 64) angles
};

__device__ __forceinline__ void job_work(Jobs *jb, int part, const Ppht &c, int lane) {
    Walk wk;
    wk.xflag = jb->par[0]; wk.x0 = jb->par[1]; wk.y0 = jb->par[2]; wk.dx = jb->par[3]; wk.dy = jb->par[4];
    const int sgn = jb->par[5], base = jb->par[6];
    const uint64_t bits = (uint32_t)jb->par[7] | ((uint64_t)(uint32_t)jb->par[8] << 32);
    const bool bit = (bits >> lane) & 1;
    int j1, i1;
    wk.at(base + lane, sgn, j1, i1);
    uint32_t *words = reinterpret_cast<uint32_t *>(c.lacc);
    if (bit) {
        for (int n0 = part * JOB_ANGLES; n0 < c.numangle; n0 += JOB_WAVES * JOB_ANGLES) {
            const int n1 = min(c.numangle, n0 + JOB_ANGLES);
            for (int nn = n0; nn < n1; nn++) {
                const int r = rho_bin(j1, i1, jb->tc[nn], jb->ts[nn], c.half);
                if ((unsigned)r < (unsigned)c.numrho) {
                    const int a = nn * c.numrho + r;
                    // two biased 16-bit bins per word; the bias keeps a half from borrowing out of the other
                    atomicSub(&words[a >> 1], (a & 1) ? 0x10000u : 1u);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&jb->done, 1);
}

// wave 0: do its share of the job in flight (if any) and wait for the helpers
__device__ __forceinline__ void job_finish(Jobs *jb, bool inflight, const Ppht &c, int lane) {
    if (!inflight) return;
    job_work(jb, 0, c, lane);
    while (*(volatile int *)&jb->done < JOB_WAVES) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// helper wavefronts (part 1 and 2): serve every job until told to leave
__device__ __forceinline__ void job_serve(Jobs *jb, int part, const Ppht &c, int lane) {
    int seen = 0;
    for (;;) {
        int sq;
        while ((sq = *(volatile int *)&jb->seq) == seen) __builtin_amdgcn_s_sleep(2);
        if (sq < 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        seen = sq;
        job_work(jb, part, c, lane);
    }
}

// A point reached the threshold at angle max_n: walk the line both ways, decide whether it is long
// enough, clear its mask bits and, if accepted, take its votes back and emit the segment.
// (Inlined at its single call site per kernel so that every LDS access keeps its address space and the
// constants stay in registers; the per-lane trig pair is read with v_readlane, not from memory.)
// Returns (accepted ? 1 : 0) | jobs posted so far << 1 (segment number nseg is written if it fits).
// jobseq is the caller's count of vote-removal jobs posted so far; it lives in registers, not LDS.
template <bool LDS_STATE>
__device__ __forceinline__ int extract_line(const Ppht &c, int lane, int j, int i, int max_n, int nseg, Jobs *jb, int jobseq) {
    bool inflight = false;   // a vote-removal job of this line is with the helpers
    const AngleEntry e = c.tab[max_n];
    Walk wk;
    wk.xflag = e.xflag;
    if (e.xflag) { wk.x0 = j; wk.y0 = (i << 16) + 32768; wk.dx = e.dmaj; wk.dy = e.dmin; }
    else         { wk.x0 = (j << 16) + 32768; wk.y0 = i; wk.dx = e.dmin; wk.dy = e.dmaj; }

    // The first NC chunks (64 steps each) of both directions are fetched in one go -- all mask words in
    // flight together, one memory round trip for most lines -- and their ballots are kept: pass 2 needs
    // the same bits (nothing else clears this line's pixels in between), so it reloads nothing.
    constexpr int NC = 4;
    uint64_t nzm[2][NC], inbm[2][NC];
    {
        uint64_t wv[2][NC];
        bool inb[2][NC];
        int sh[2][NC];
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < NC; q++) {
                int j1, i1;
                wk.at(q * 64 + lane, k ? -1 : 1, j1, i1);
                inb[k][q] = (unsigned)j1 < (unsigned)c.W && (unsigned)i1 < (unsigned)c.H;
                sh[k][q] = j1 & 63;
                wv[k][q] = inb[k][q] ? ld_l2(c.mask + (size_t)i1 * c.TX + (j1 >> 6)) : 0;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int q = 0; q < NC; q++) {
                inbm[k][q] = __ballot(inb[k][q]);
                nzm[k][q] = __ballot(inb[k][q] && ((wv[k][q] >> sh[k][q]) & 1));
            }
    }

    // pass 1: how far the line reaches in each direction
    int end_t[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int sgn = k ? -1 : 1;
        int last_t = 0;
        bool done = false;
        auto scan = [&](int base, uint64_t inb_m, uint64_t nz_m) {
            const int first_oob = inb_m == ~0ull ? 64 : __builtin_ctzll(~inb_m);
            if (first_oob < 64) nz_m &= (1ull << first_oob) - 1;
            while (nz_m) {
                int tt = base + __builtin_ctzll(nz_m);
                if (tt > last_t + c.line_gap + 1) { done = true; break; }
                last_t = tt;
                nz_m &= nz_m - 1;
            }
            if (first_oob < 64 || base + 63 - last_t > c.line_gap) done = true;
        };
#pragma unroll
        for (int q = 0; q < NC; q++)
            if (!done) scan(q * 64, inbm[k][q], nzm[k][q]);
        for (int base = NC * 64; !done; base += 64) {   // rare: longer than NC chunks
            int j1, i1;
            wk.at(base + lane, sgn, j1, i1);
            const bool inb = (unsigned)j1 < (unsigned)c.W && (unsigned)i1 < (unsigned)c.H;
            bool bit = false;
            if (inb) bit = (ld_l2(c.mask + (size_t)i1 * c.TX + (j1 >> 6)) >> (j1 & 63)) & 1;
            scan(base, __ballot(inb), __ballot(bit));
        }
        end_t[k] = last_t;
    }
    int ex0, ey0, ex1, ey1;
    wk.at(end_t[0], 1, ex0, ey0);
    wk.at(end_t[1], -1, ex1, ey1);
    const bool good = abs(ex1 - ex0) >= c.line_length || abs(ey1 - ey0) >= c.line_length;

    // pass 2: clear the mask along the line; an accepted line takes its votes back
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int sgn = k ? -1 : 1;
        for (int base = 0; base <= end_t[k]; base += 64) {
            const int t = base + lane;
            int j1, i1;
            wk.at(t, sgn, j1, i1);
            uint64_t bm;
            if (base < NC * 64) {
                bm = base == 0 ? nzm[k][0] : (base == 64 ? nzm[k][1] : (base == 128 ? nzm[k][2] : nzm[k][3]));
            } else {
                bool bit = false;
                if (t <= end_t[k]) bit = (ld_l2(c.mask + (size_t)i1 * c.TX + (j1 >> 6)) >> (j1 & 63)) & 1;
                bm = __ballot(bit);
            }
            const int nvalid = end_t[k] - base + 1;                     // steps of this chunk on the line
            if (nvalid < 64) bm &= (1ull << nvalid) - 1;
            if (k == 1 && base == 0) bm &= ~1ull;                       // t = 0 belongs to direction 0
            const bool bit = (bm >> lane) & 1;
            if constexpr (LDS_STATE) {
                if (good && bm) {   // hand the chunk's vote removal to the workgroup, keep walking
                    job_finish(jb, inflight, c, lane);       // one job in flight at a time
                    ++jobseq;
                    if (lane == 0) {
                        jb->par[0] = wk.xflag; jb->par[1] = wk.x0; jb->par[2] = wk.y0; jb->par[3] = wk.dx; jb->par[4] = wk.dy;
                        jb->par[5] = sgn; jb->par[6] = base;
                        jb->par[7] = (int)(uint32_t)bm; jb->par[8] = (int)(uint32_t)(bm >> 32);
                        jb->done = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        *(volatile int *)&jb->seq = jobseq;
                    }
                    inflight = true;
                }
            } else if (bit && good) {
                for (int nn = 0; nn < c.numangle; nn++) {
                    int r = rho_bin(j1, i1, c.tab[nn].c, c.tab[nn].s, c.half);
                    if ((unsigned)r < (unsigned)c.numrho) atomicSub(&c.acc[(size_t)nn * c.numrho + r], 1);
                }
            }
            if (bit) atomicAnd((unsigned long long *)(c.mask + (size_t)i1 * c.TX + (j1 >> 6)), ~(1ull << (j1 & 63)));
        }
    }
    if constexpr (LDS_STATE) job_finish(jb, inflight, c, lane);   // every vote is back before the next pick
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (good && lane == 0 && nseg < c.seg_cap) {
        c.seg[4 * nseg + 0] = ex0; c.seg[4 * nseg + 1] = ey0;
        c.seg[4 * nseg + 2] = ex1; c.seg[4 * nseg + 3] = ey1;
    }
    return (good ? 1 : 0) | (jobseq << 1);
}

// General path: every piece of state in global memory behind agent-scope atomics, any number of
// angles and points.  After the fast path it only runs the frames that path flagged.
__global__ __launch_bounds__(256) void k_ppht_general(Geo g, uint32_t *__restrict__ nz_all, int nz_cap,
                                                      const int *__restrict__ npoints, uint64_t *__restrict__ rowplane,
                                                      const AngleEntry *__restrict__ tab, int numangle, int numrho,
                                                      int32_t *__restrict__ acc_all, int threshold, int line_length,
                                                      int line_gap, int32_t *__restrict__ seg_all, int seg_cap,
                                                      int *__restrict__ seg_count, int *__restrict__ status,
                                                      const int *__restrict__ fallback) {
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (fallback && !fallback[f]) return;
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
        if (tid == 0) { seg_count[f] = 0; atomicOr(status, 1); }
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
            if ((unsigned)r >= (unsigned)numrho) { atomicOr(status, 2); continue; }
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
        if (extract_line<false>(c, lane, j, i, max_n, nseg, nullptr, 0) & 1) nseg++;
    }
    if (lane == 0) seg_count[f] = nseg;
}

// Fast path (up to 64 angles, point list and int16 accumulator fit the CU's LDS):
//   wavefront 1 runs the data-independent part of OpenCV's loop -- the MWC random picks with
//   swap-remove -- ahead of time, leaving the visiting order in the point array itself (the removed
//   point goes to the slot the swap frees, so the array read from the back is the order);
//   wavefront 0 follows it, with the mask words of the next four points in flight, votes in LDS with
//   one angle per lane and reduces the maximum with DPP.
__global__ __launch_bounds__(256) void k_ppht_fast(Geo g, const uint32_t *__restrict__ nz_all, int nz_cap,
                                                   const int *__restrict__ npoints, uint64_t *__restrict__ rowplane,
                                                   const AngleEntry *__restrict__ tab, int numangle, int numrho,
                                                   int threshold, int line_length, int line_gap,
                                                   int32_t *__restrict__ seg_all, int seg_cap, int *__restrict__ seg_count,
                                                   int *__restrict__ status, int *__restrict__ fallback, int lds_points,
                                                   const uint32_t *__restrict__ rng_out) {
    extern __shared__ int16_t lds16[];
    __shared__ int progress;
    __shared__ Jobs jobs;
    // this kernel is a few latency-bound wavefronts per CU: let them issue ahead of the streaming
    // kernels of other batches that share the SIMDs
    __builtin_amdgcn_s_setprio(3);
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *nz = nz_all + (size_t)f * nz_cap;
    const int n = npoints[f];
    if (n > nz_cap) {
        if (tid == 0) { seg_count[f] = 0; fallback[f] = 0; atomicOr(status, 1); }
        return;
    }
    if (n > lds_points || n >= LDS_BIAS) {   // biased 16-bit bins hold counts in (-LDS_BIAS, LDS_BIAS)
        if (tid == 0) fallback[f] = 1;
        return;
    }
    Ppht c;
    c.W = g.W; c.H = g.H; c.TX = g.TX; c.numangle = numangle; c.numrho = numrho; c.half = (numrho - 1) / 2;
    c.threshold = threshold; c.line_length = line_length; c.line_gap = line_gap; c.seg_cap = seg_cap;
    c.mask = rowplane + (size_t)f * g.rowplane_words();
    c.tab = tab; c.acc = nullptr;
    c.lacc = lds16;
    c.seg = seg_all + (size_t)f * seg_cap * 4;
    uint32_t *lnz = reinterpret_cast<uint32_t *>(lds16 + (((size_t)numangle * numrho + 1) & ~(size_t)1));
    {
        uint32_t *z = reinterpret_cast<uint32_t *>(lds16);
        const int zw = (numangle * numrho + 1) / 2;
        for (int i = tid; i < zw; i += 256) z[i] = (uint32_t)LDS_BIAS * 0x10001u;
        for (int i = tid; i < n; i += 256) lnz[i] = nz[i];
        if (tid == 0) {
            progress = 0; fallback[f] = 0;
            jobs.seq = 0; jobs.done = 0;
        }
        if (tid < 64) { jobs.tc[tid] = tid < numangle ? tab[tid].c : 0.f; jobs.ts[tid] = tid < numangle ? tab[tid].s : 0.f; }
    }
    __syncthreads();
    float my_c = 0.f, my_s = 0.f;   // lane l holds the trig pair of angle l
    if (lane < numangle) { my_c = tab[lane].c; my_s = tab[lane].s; }
    if (wave >= 2) {                // helpers: vote removal for accepted lines
        job_serve(&jobs, wave - 1, c, lane);
        return;
    }

    if (wave == 1) {   // producer: visiting order, in place
#ifdef LC_STAMPS
        long long tp0 = STAMP();
#endif
        // rng_out[k] is the k-th output of cv::RNG(2^64 - 1) (fixed seed, so a host-made table);
        // 64 lanes reduce 64 picks modulo their counts at once, the swap-removes then run in order
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            uint32_t idxv = 0;
            if (k < n) idxv = rng_out[k] % (uint32_t)(n - k);
            const int m = min(64, n - k0);
            for (int t = 0; t < m; t++) {
                const uint32_t idx = __shfl(idxv, t);
                const int cnt = n - (k0 + t);
                if (lane == 0) {
                    const uint32_t pt = lnz[idx], last = lnz[cnt - 1];
                    lnz[idx] = last;
                    lnz[cnt - 1] = pt;
                }
            }
            if (lane == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                *(volatile int *)&progress = k0 + m;
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
    // later ones are taken back, the line is extracted and the batch restarts behind that point, so
    // every count any point sees is the one OpenCV's strictly sequential loop would see.
    // The vote itself is straight-line code: a missing point of a short group and the lanes past the
    // last angle add zero to a harmless bin instead of branching around the atomic.
    uint32_t *words = reinterpret_cast<uint32_t *>(c.lacc);
    const uint32_t lane_base = lane < numangle ? (uint32_t)(lane * numrho) : (uint32_t)(lane * 2);
    const uint32_t lane_one = lane < numangle ? 1u : 0u;
    const uint64_t lanes_mask = numangle >= 64 ? ~0ull : ((1ull << numangle) - 1);
    // count after the vote >= threshold  <=>  biased 16-bit bin before the vote >= threshold - 1 + bias
    const uint32_t thr_b = (uint32_t)min(max((long long)threshold - 1 + LDS_BIAS, 0ll), 0x10000ll);
    // cvRound of |v| < 2^22: adding 1.5 * 2^23 rounds v half-to-even into the low mantissa bits
    constexpr float ROUND_MAGIC = 12582912.0f;
    const uint32_t rfix = (uint32_t)c.half - 0x4B400000u;
    const uint32_t rho_last = (uint32_t)(numrho - 1);
    uint32_t r_seen = 0;   // largest (unsigned) bin number any vote produced; out of range means a bad table
    int nseg = 0, jobseq = 0;
#ifdef LC_STAMPS
    long long t_wait = 0, t_vote = 0, t_line = 0, n_alive = 0, n_line = 0, tc0 = STAMP();
#endif
    int k = 0;
    while (k < n) {
        const int m = min(64, n - k);
#ifdef LC_STAMPS
        long long tw = STAMP();
#endif
        while (*(volatile int *)&progress < k + m) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t ptv = 0;
        bool alive_l = false;
        if (lane < m) {
            ptv = lnz[n - 1 - (k + lane)];
            const uint64_t w = ld_l2(c.mask + (size_t)(ptv >> 16) * c.TX + ((ptv & 0xFFFF) >> 6));
            alive_l = (w >> (ptv & 63)) & 1;
        }
        const float fjv = (float)(ptv & 0xFFFF), fiv = (float)(ptv >> 16);
        uint64_t alive = __ballot(alive_l);
#ifdef LC_STAMPS
        t_wait += STAMP() - tw;
#endif
        bool restarted = false;
        while (alive && !restarted) {
            constexpr int Q = PPHT_Q;
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
                const float v = __fadd_rn(__fmul_rn(fj, my_c), __fmul_rn(fi, my_s));
                uint32_t r = __float_as_uint(__fadd_rn(v, ROUND_MAGIC)) + rfix;   // cvRound(v) + half
                r_seen = max(r_seen, r);
                r = min(r, rho_last);
                const uint32_t a = lane_base + r;
                sh[q] = (a & 1u) << 4;               // two biased 16-bit bins per word
                widx[q] = a >> 1;
                inc[q] = (valid ? lane_one : 0u) << sh[q];
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
                n_line++;
#endif
                const int lr = extract_line<true>(c, lane, (int)(hpt & 0xFFFF), (int)(hpt >> 16), max_n, nseg, &jobs, jobseq);
                jobseq = lr >> 1;
                if (lr & 1) nseg++;
#ifdef LC_STAMPS
                t_line += STAMP() - tl;
#endif
                k += ht + 1;
                restarted = true;
            }
        }
        if (!restarted) k += m;
    }
    if (__ballot(r_seen > rho_last) != 0 && lane == 0) atomicOr(status, 2);
    if (lane == 0) *(volatile int *)&jobs.seq = -1;   // helpers leave
    if (lane == 0) seg_count[f] = nseg;
#ifdef LC_STAMPS
    if (f == 0 && lane == 0) {
        g_stamps[1] = STAMP() - tc0; g_stamps[2] = t_wait; g_stamps[3] = t_vote; g_stamps[4] = t_line;
        g_stamps[5] = n_alive; g_stamps[6] = n_line; g_stamps[7] = n;
    }
#endif
}

}  // namespace

void launch_rowscan(Geo g, int nframes, const int *rowcount, int *rowoff, int *npoints, hipStream_t st) {
    hipLaunchKernelGGL(k_rowscan, dim3(nframes), dim3(256), 0, st, g.H, rowcount, rowoff, npoints);
}

void launch_scatter(Geo g, int nframes, const uint64_t *rowplane, const int *rowcount, const int *rowoff, uint32_t *nz,
                    int nz_cap, hipStream_t st) {
    long long rows = (long long)nframes * g.H;
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, g, nframes, rowplane, rowcount, rowoff, nz,
                       nz_cap);
}

void launch_vote(Geo g, int nframes, const uint32_t *nz, int nz_cap, const int *npoints, const AngleEntry *tab,
                 int numangle, int numrho, int32_t *acc, hipStream_t st) {
    (void)g;
    // angle rows per block: what fits 6 KiB of LDS, at least one.  Small on purpose: the kernel then shares
    // CUs with the HoughLinesP kernel of the previous batch, which leaves only a few KiB of LDS free.
    int apg = (6 * 1024) / (numrho * 4);
    if (apg < 1) apg = 1;
    if (apg > numangle) apg = numangle;
    int groups = (numangle + apg - 1) / apg;
    apg = (numangle + groups - 1) / groups;
    size_t lds = (size_t)apg * numrho * 4;
    static size_t attr_set = 0;
    if (lds > 64 * 1024 && lds > attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_vote), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = lds;
    }
    hipLaunchKernelGGL(k_vote, dim3(groups, nframes), dim3(256), lds, st, nz, nz_cap, npoints, tab, numangle, numrho, apg, acc);
}

void launch_ppht(Geo g, int nframes, uint32_t *nz, int nz_cap, const int *npoints, uint64_t *rowplane,
                 const AngleEntry *tab, int numangle, int numrho, int32_t *acc, int threshold, int line_length,
                 int line_gap, int32_t *seg, int seg_cap, int *seg_count, int *status, int *fallback,
                 const uint32_t *rng_out, int rng_len, hipStream_t st) {
    // fast path: int16 accumulator + point list in LDS (160 KiB per CU); frames that do not fit fall back
    const size_t acc_bytes = (((size_t)numangle * numrho + 1) & ~(size_t)1) * 2;
    const size_t lds_max = 160 * 1024 - 1024;   // minus the kernel's static LDS (job block, progress word)
    bool fast = false;
    if (numangle <= 64 && acc_bytes + 4096 <= lds_max) {
        // the point list takes at most 32 KiB so that a few KiB of every CU's LDS stay free for the small
        // kernels of the next batch (hysteresis closure, vote) running beside this one
        const int lds_points = std::min(std::min((int)((lds_max - acc_bytes) / 4), rng_len), 8192);
        const size_t lds = acc_bytes + 4 * (size_t)lds_points;
        static size_t fast_attr = 0;
        if (lds > fast_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ppht_fast), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            fast_attr = lds;
        }
        hipLaunchKernelGGL(k_ppht_fast, dim3(nframes), dim3(256), lds, st, g, nz, nz_cap, npoints, rowplane, tab, numangle,
                           numrho, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, fallback, lds_points, rng_out);
        fast = true;
    }
    hipLaunchKernelGGL(k_ppht_general, dim3(nframes), dim3(256), 0, st, g, nz, nz_cap, npoints, rowplane, tab, numangle, numrho,
                       acc, threshold, line_length, line_gap, seg, seg_cap, seg_count, status, fast ? fallback : nullptr);
}

#ifdef LC_STAMPS
extern "C" int lc_debug_stamps(long long *out) {
    (void)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_dbg), sizeof(long long) * 4);
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 8);
}
#endif

}  // namespace lanecv
