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

// one wavefront per (frame, row): points in left-to-right order at the row's offset
__global__ __launch_bounds__(256) void k_scatter(Geo g, int nframes, const uint64_t *__restrict__ rowplane,
                                                 const int *__restrict__ rowoff, uint32_t *__restrict__ nz, int nz_cap) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)nframes * g.H) return;
    const int f = (int)(row / g.H), y = (int)(row % g.H);
    int off = rowoff[row];
    uint32_t *out = nz + (size_t)f * nz_cap;
    for (int tx = 0; tx < g.TX; tx++) {
        uint64_t w = rowplane[(size_t)row * g.TX + tx];
        if (w == 0) continue;
        if ((w >> lane) & 1) {
            int pos = off + __popcll(w & ((1ull << lane) - 1));
            if (pos < nz_cap) out[pos] = (uint32_t)(tx * 64 + lane) | ((uint32_t)y << 16);
        }
        off += __popcll(w);
    }
}

// full-vote accumulator: block (group, frame) owns `apg` angle rows in LDS
__global__ __launch_bounds__(1024) void k_vote(const uint32_t *__restrict__ nz, int nz_cap, const int *__restrict__ npoints,
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

struct Walk {
    int xflag, x0, y0, dx, dy;   // fixed-point start and step of direction 0
    __device__ __forceinline__ void at(int t, int sgn, int &j1, int &i1) const {
        int x = x0 + sgn * t * dx, y = y0 + sgn * t * dy;
        if (xflag) { j1 = x; i1 = y >> 16; }
        else       { j1 = x >> 16; i1 = y; }
    }
};

__global__ __launch_bounds__(64) void k_ppht(Geo g, uint32_t *__restrict__ nz_all, int nz_cap,
                                             const int *__restrict__ npoints, uint64_t *__restrict__ rowplane,
                                             const AngleEntry *__restrict__ tab, int numangle, int numrho,
                                             int32_t *__restrict__ acc_all, int threshold, int line_length, int line_gap,
                                             int32_t *__restrict__ seg_all, int seg_cap, int *__restrict__ seg_count,
                                             int *__restrict__ status) {
    const int f = blockIdx.x, lane = threadIdx.x;
    const int W = g.W, H = g.H, TX = g.TX;
    uint32_t *nz = nz_all + (size_t)f * nz_cap;
    uint64_t *mask = rowplane + (size_t)f * g.rowplane_words();
    int32_t *acc = acc_all + (size_t)f * numangle * numrho;
    int32_t *seg = seg_all + (size_t)f * seg_cap * 4;
    const int half = (numrho - 1) / 2;
    int n = npoints[f];
    if (n > nz_cap) {   // the point list was truncated: refuse rather than differ
        if (lane == 0) { seg_count[f] = 0; atomicOr(status, 1); }
        return;
    }
    unsigned long long state = 0xFFFFFFFFFFFFFFFFull;
    int nseg = 0;
    for (int count = n; count > 0; count--) {
        state = (unsigned long long)(uint32_t)state * 4164903690ull + (state >> 32);
        const uint32_t idx = (uint32_t)state % (uint32_t)count;
        const uint32_t pt = ld_l2(nz + idx);
        const uint32_t last = ld_l2(nz + count - 1);
        if (lane == 0) st_l2(nz + idx, last);
        const int j = pt & 0xFFFF, i = pt >> 16;
        const uint64_t mw = ld_l2(mask + (size_t)i * TX + (j >> 6));
        if (!((mw >> (j & 63)) & 1)) continue;

        // vote; key orders by count first, then by smaller angle index
        long long best = LLONG_MIN;
        for (int nn = lane; nn < numangle; nn += 64) {
            int r = rho_bin(j, i, tab[nn].c, tab[nn].s, half);
            if ((unsigned)r >= (unsigned)numrho) { atomicOr(status, 2); continue; }
            int val = atomicAdd(&acc[(size_t)nn * numrho + r], 1) + 1;
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

        const AngleEntry e = tab[max_n];
        Walk wk;
        wk.xflag = e.xflag;
        if (e.xflag) { wk.x0 = j; wk.y0 = (i << 16) + 32768; wk.dx = e.dmaj; wk.dy = e.dmin; }
        else         { wk.x0 = (j << 16) + 32768; wk.y0 = i; wk.dx = e.dmin; wk.dy = e.dmaj; }

        // pass 1: how far the line reaches in each direction
        int end_t[2];
        for (int k = 0; k < 2; k++) {
            const int sgn = k ? -1 : 1;
            int last_t = 0;
            for (int base = 0;; base += 64) {
                int j1, i1;
                wk.at(base + lane, sgn, j1, i1);
                const bool inb = (unsigned)j1 < (unsigned)W && (unsigned)i1 < (unsigned)H;
                bool bit = false;
                if (inb) bit = (ld_l2(mask + (size_t)i1 * TX + (j1 >> 6)) >> (j1 & 63)) & 1;
                const uint64_t inb_m = __ballot(inb);
                uint64_t nz_m = __ballot(bit);
                const int first_oob = inb_m == ~0ull ? 64 : __builtin_ctzll(~inb_m);
                if (first_oob < 64) nz_m &= (1ull << first_oob) - 1;
                bool done = false;
                while (nz_m) {
                    int tt = base + __builtin_ctzll(nz_m);
                    if (tt > last_t + line_gap + 1) { done = true; break; }
                    last_t = tt;
                    nz_m &= nz_m - 1;
                }
                if (done || first_oob < 64 || base + 63 - last_t > line_gap) break;
            }
            end_t[k] = last_t;
        }
        int ex0, ey0, ex1, ey1;
        wk.at(end_t[0], 1, ex0, ey0);
        wk.at(end_t[1], -1, ex1, ey1);
        const bool good = abs(ex1 - ex0) >= line_length || abs(ey1 - ey0) >= line_length;

        // pass 2: clear the mask along the line; an accepted line takes its votes back
        for (int k = 0; k < 2; k++) {
            const int sgn = k ? -1 : 1;
            for (int base = k; base <= end_t[k]; base += 64) {   // t = 0 is handled by direction 0
                const int t = base + lane;
                bool bit = false;
                int j1 = 0, i1 = 0;
                if (t <= end_t[k]) {
                    wk.at(t, sgn, j1, i1);
                    bit = (ld_l2(mask + (size_t)i1 * TX + (j1 >> 6)) >> (j1 & 63)) & 1;
                }
                if (bit) {
                    if (good) {
                        for (int nn = 0; nn < numangle; nn++) {
                            int r = rho_bin(j1, i1, tab[nn].c, tab[nn].s, half);
                            if ((unsigned)r < (unsigned)numrho) atomicSub(&acc[(size_t)nn * numrho + r], 1);
                        }
                    }
                    atomicAnd((unsigned long long *)(mask + (size_t)i1 * TX + (j1 >> 6)), ~(1ull << (j1 & 63)));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (good) {
            if (lane == 0 && nseg < seg_cap) {
                seg[4 * nseg + 0] = ex0; seg[4 * nseg + 1] = ey0;
                seg[4 * nseg + 2] = ex1; seg[4 * nseg + 3] = ey1;
            }
            nseg++;
        }
    }
    if (lane == 0) seg_count[f] = nseg;
}

}  // namespace

void launch_rowscan(Geo g, int nframes, const int *rowcount, int *rowoff, int *npoints, hipStream_t st) {
    hipLaunchKernelGGL(k_rowscan, dim3(nframes), dim3(256), 0, st, g.H, rowcount, rowoff, npoints);
}

void launch_scatter(Geo g, int nframes, const uint64_t *rowplane, const int *rowoff, uint32_t *nz, int nz_cap,
                    hipStream_t st) {
    long long rows = (long long)nframes * g.H;
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, g, nframes, rowplane, rowoff, nz, nz_cap);
}

void launch_vote(Geo g, int nframes, const uint32_t *nz, int nz_cap, const int *npoints, const AngleEntry *tab,
                 int numangle, int numrho, int32_t *acc, hipStream_t st) {
    (void)g;
    // angle rows per block: as many as fit 64 KiB of LDS (two blocks per CU), at least one
    int apg = (64 * 1024) / (numrho * 4);
    if (apg < 1) apg = 1;
    if (apg > numangle) apg = numangle;
    int groups = (numangle + apg - 1) / apg;
    apg = (numangle + groups - 1) / groups;
    size_t lds = (size_t)apg * numrho * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_vote), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_vote, dim3(groups, nframes), dim3(1024), lds, st, nz, nz_cap, npoints, tab, numangle, numrho, apg, acc);
}

void launch_ppht(Geo g, int nframes, uint32_t *nz, int nz_cap, const int *npoints, uint64_t *rowplane,
                 const AngleEntry *tab, int numangle, int numrho, int32_t *acc, int threshold, int line_length,
                 int line_gap, int32_t *seg, int seg_cap, int *seg_count, int *status, hipStream_t st) {
    hipLaunchKernelGGL(k_ppht, dim3(nframes), dim3(64), 0, st, g, nz, nz_cap, npoints, rowplane, tab, numangle, numrho, acc,
                       threshold, line_length, line_gap, seg, seg_cap, seg_count, status);
}

}  // namespace lanecv
