// Host-side declarations shared by the translation units behind the C ABI (api.hip, api_draw.hip, pipe_create.hip,
// pipe_run.hip): error reporting, device buffers, the per-frame-size working set and the context.
#pragma once
#include "../../include/lanecv.h"
#include "common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

namespace lanecv {

// records the message lc_last_error() returns on this thread and hands `code` back
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(LC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Length of the table of HoughLinesP's random numbers: at least every point count the forms of the LDS kernel with the points in
// LDS accept, at most RNG_TABLE_MAX (the form that leaves the points in device memory takes as many points as the table is long).
constexpr int RNG_TABLE = 32768, RNG_TABLE_MAX = 1 << 20;
constexpr int MAX_DIM = 8192;       // 16.16 walk arithmetic and the packed (x | y << 16) points need this bound

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;      // false: a slice of another buffer (alias), not freed here
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T) + 64);
        if (e != hipSuccess) { p = nullptr; return fail(LC_ENOMEM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e)); }
        n = count;
        return LC_OK;
    }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
    void alias(T *base, size_t count) { release(); p = base; n = count; owned = false; }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), owned(o.owned) { o.p = nullptr; o.n = 0; o.owned = true; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; owned = o.owned; o.p = nullptr; o.n = 0; o.owned = true; }
        return *this;
    }
    ~DevBuf() { release(); }
};

// ---- Hough geometry (HoughLinesP narrows rho/theta to float first) ---------
struct HoughGeo {
    int numangle = 0, numrho = 0;
    std::vector<AngleEntry> tab;
};
int hough_geo(int w, int h, double rho_d, double theta_d, HoughGeo &hg);

// ---- device working set for up to B frames of one size ---------------------
struct Work {
    Geo g{};
    int B = 0;
    size_t npix = 0;
    bool inter = false;
    DevBuf<uint8_t> bgr, gray, blur, edges, masked;
    DevBuf<uint64_t> planeS, planeW, rowplane;
    DevBuf<uint64_t> prevbits;    // masked bits last written into `masked` (pipeline slots without intermediates)
    DevBuf<uint8_t> src_in;       // frames at their source size when the pipeline resizes them first
    DevBuf<uint8_t> rowflags;     // streaming front kernel's dirty flags of the plane rows (pipeline slots)
    DevBuf<uint32_t> wake;        // two wake-flag arrays of B * tiles words (hysteresis closure)
    DevBuf<int> rounds, rowcount, npoints, segcount, status, fallback;
    DevBuf<int> redo;             // per frame: the pass over the lower rows only was not enough (pipelines that skip rows)
    DevBuf<uint32_t> ring_px;     // per frame: the ring check's pixel list when hysteresis leads a front launch (HystLaunch)
    DevBuf<uint32_t> nz;
    int nz_cap = 0;
    DevBuf<int32_t> acc_full, acc_ppht, seg;
    DevBuf<AngleEntry> tab;
    DevBuf<uint32_t> rng_out;     // first rng_len outputs of cv::RNG(2^64 - 1), HoughLinesP's fixed seed
    int rng_len = 0;
    HoughGeo hg;
    double rho = 0, theta = 0;
    int seg_cap = 0;
    // pipeline slots keep everything that goes back to the host after a batch in one block, so that one
    // copy fetches it: [segment counts B][hysteresis rounds B][8 status words per part][segments B * cap * 4]
    DevBuf<int32_t> resblock;
    size_t res_head = 0;
    int32_t *r_segcount = nullptr, *r_rounds = nullptr, *r_status = nullptr, *r_seg = nullptr;
    int init_results(int segcap, int nparts = 1) {
        int rc;
        res_head = 2 * (size_t)B + 8 * (size_t)nparts;
        if ((rc = resblock.alloc(res_head + (size_t)B * segcap * 4))) return rc;
        HIP_TRY(hipMemset(resblock.p, 0, sizeof(int32_t) * res_head));
        r_segcount = resblock.p; r_rounds = resblock.p + B; r_status = resblock.p + 2 * (size_t)B; r_seg = resblock.p + res_head;
        return LC_OK;
    }

    int init_images(int W, int H, int batch, bool own_input, bool intermediates, bool colour) {
        g = make_geo(W, H);
        B = batch;
        npix = (size_t)W * H;
        inter = intermediates;
        int rc;
        if (own_input && (rc = bgr.alloc((size_t)B * npix * (colour ? 3 : 1)))) return rc;
        if (intermediates) {
            if ((rc = gray.alloc((size_t)B * npix))) return rc;
            if ((rc = blur.alloc((size_t)B * npix))) return rc;
            if ((rc = edges.alloc((size_t)B * npix))) return rc;
        }
        if ((rc = masked.alloc((size_t)B * npix))) return rc;
        if ((rc = planeS.alloc((size_t)B * g.plane_words()))) return rc;
        if ((rc = planeW.alloc((size_t)B * g.plane_words()))) return rc;
        if ((rc = rowplane.alloc((size_t)B * g.rowplane_words()))) return rc;
        if (own_input && front_stream_supported(g) && !getenv("LANECV_FRONT_TILED")) {
            // the streaming front kernel then stores only the strip rows that hold or held a candidate pixel
            const size_t nf = (size_t)B * (front_stream_strips(g) + 1) * H;   // (+ 1: shifted strips, FrontShift)
            if ((rc = rowflags.alloc(nf))) return rc;
            HIP_TRY(hipMemset(rowflags.p, 0, nf));
            HIP_TRY(hipMemset(planeS.p, 0, sizeof(uint64_t) * (size_t)B * g.plane_words()));
            HIP_TRY(hipMemset(planeW.p, 0, sizeof(uint64_t) * (size_t)B * g.plane_words()));
        }
        if (own_input && !intermediates) {
            // finalize then writes only what changed since the slot's previous batch: everything starts at zero
            if ((rc = prevbits.alloc((size_t)B * g.rowplane_words()))) return rc;
            HIP_TRY(hipMemset(prevbits.p, 0, sizeof(uint64_t) * (size_t)B * g.rowplane_words()));
            HIP_TRY(hipMemset(rowplane.p, 0, sizeof(uint64_t) * (size_t)B * g.rowplane_words()));
            HIP_TRY(hipMemset(masked.p, 0, (size_t)B * npix));
        }
        if ((rc = wake.alloc(2 * (size_t)B * g.TX * g.TY))) return rc;
        if ((rc = rounds.alloc(B))) return rc;
        if ((rc = rowcount.alloc((size_t)B * H))) return rc;
        // both are left all zero by their last reader (hysteresis closure / point scatter), so once is enough
        HIP_TRY(hipMemset(wake.p, 0, sizeof(uint32_t) * 2 * (size_t)B * g.TX * g.TY));
        HIP_TRY(hipMemset(rowcount.p, 0, sizeof(int) * (size_t)B * H));
        if ((rc = npoints.alloc(B))) return rc;
        if ((rc = segcount.alloc(B))) return rc;
        if ((rc = status.alloc(8))) return rc;
        if ((rc = redo.alloc(B))) return rc;
        HIP_TRY(hipMemset(redo.p, 0, sizeof(int) * (size_t)B));
        if ((rc = ring_px.alloc((size_t)B * RING_LIST_WORDS))) return rc;
        if ((rc = fallback.alloc(B))) return rc;
        return LC_OK;
    }
    int init_points(int cap) {
        cap = (cap + 3) & ~3;     // every frame's list starts on a 16-byte boundary (the vote loads four points at once)
        nz_cap = cap;
        return nz.alloc((size_t)B * cap);
    }
    int init_hough(double rho_, double theta_, int segcap, bool full, bool ppht, hipStream_t st) {
        int rc;
        if (rho_ != rho || theta_ != theta || hg.numangle == 0) {
            if ((rc = hough_geo(g.W, g.H, rho_, theta_, hg))) return rc;
            rho = rho_; theta = theta_;
            if ((rc = tab.alloc(hg.numangle))) return rc;
            HIP_TRY(hipMemcpyAsync(tab.p, hg.tab.data(), sizeof(AngleEntry) * hg.numangle, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            acc_full.release();
            acc_ppht.release();
        }
        // one output per point a frame can have, up to RNG_TABLE_MAX (frames with more go to the general kernel, which steps the generator itself)
        const int rng_want = std::max(RNG_TABLE, std::min(nz_cap, RNG_TABLE_MAX));
        if (ppht && rng_len < rng_want) {
            std::vector<uint32_t> r(rng_want);
            unsigned long long state = 0xFFFFFFFFFFFFFFFFull;
            for (int k = 0; k < rng_want; k++) {
                state = (unsigned long long)(uint32_t)state * 4164903690ull + (state >> 32);
                r[k] = (uint32_t)state;
            }
            if ((rc = rng_out.alloc(rng_want))) return rc;
            HIP_TRY(hipMemcpyAsync(rng_out.p, r.data(), sizeof(uint32_t) * rng_want, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            rng_len = rng_want;
        }
        size_t accn = (size_t)B * hg.numangle * hg.numrho;
        if (full && acc_full.n < accn && (rc = acc_full.alloc(accn))) return rc;
        if (ppht && acc_ppht.n < accn && (rc = acc_ppht.alloc(accn))) return rc;
        if (ppht && (seg_cap != segcap || !seg.p)) {
            seg_cap = segcap;
            if ((rc = seg.alloc((size_t)B * segcap * 4))) return rc;
        }
        return LC_OK;
    }
};

// Working memory of lc_canny_noalloc (api.hip): the largest image it takes
constexpr int NOALLOC_MAX_W = 4096, NOALLOC_MAX_H = 2304;          // covers 3840 x 2160

}  // namespace lanecv

// (the C ABI's opaque types live at global scope)
using lanecv::DevBuf;
using lanecv::Work;

struct lc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned char *scratch_dev = nullptr;   // this device's copy of `scratch` (set by lc_create)
    Work work;            // batch-1 working set of the per-op calls, keyed by (w, h)
    int work_w = 0, work_h = 0;
    DevBuf<uint8_t> tmpA, tmpB, tmpC, tmpD;
    DevBuf<uint8_t> tmp3;     // a 3-channel image of the current size (colour gate per-op calls)
    DevBuf<int> hsv_tab;      // sdiv[256], hdiv[256] of the 8-bit HSV conversion
    DevBuf<uint8_t> rs_src, rs_dst;   // resize: source and destination images
    DevBuf<int> rs_tab;
};

namespace lanecv {

int ctx_bind(lc_ctx *ctx);
// per-op calls: bind, and create the context's stream on first use
int ctx_op(lc_ctx *ctx);
int check_image(const void *p, int w, int h, size_t stride, size_t min_stride, const char *what);
// the context's batch-1 working set for w x h images
int ctx_work(lc_ctx *ctx, int w, int h);
// Canny's integer thresholds: swapped if out of order, floored
int canny_thresholds(double t1, double t2, int &low, int &high);
int up2d(void *dst, const void *src, size_t row_bytes, int h, size_t stride, hipStream_t st);
// copies back and synchronises the stream
int down2d(void *dst, size_t stride, const void *src, size_t row_bytes, int h, hipStream_t st);

}  // namespace lanecv
