// C ABI of liblanecv.so (declared in include/lanecv.h): contexts, the per-op
// cv2-shaped entry points and the batched lane pipeline.  Host logic only; the
// kernels live in kernels_*.hip.
#include "../../include/lanecv.h"
#include "common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <mutex>
#include <algorithm>

using namespace lanecv;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(LC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int RNG_TABLE = 16384;    // covers every point count the LDS fast path of HoughLinesP accepts
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

int hough_geo(int w, int h, double rho_d, double theta_d, HoughGeo &hg) {
    if (!(rho_d > 0) || !(theta_d > 0)) return fail(LC_EINVAL, "HoughLinesP: rho and theta must be positive");
    const float rho = (float)rho_d, theta = (float)theta_d;
    const float irho = 1 / rho;
    // computeNumangle(0, pi, theta)
    int na = (int)std::floor((M_PI - 0.0) / (double)theta) + 1;
    if (na > 1 && std::fabs(M_PI - (na - 1) * (double)theta) < (double)theta / 2) --na;
    const int nr = (int)lrintf((float)((w + h) * 2 + 1) / rho);
    if (na < 1 || nr < 1) return fail(LC_EINVAL, "HoughLinesP: degenerate accumulator %d x %d", na, nr);
    // the vote kernels keep at least one angle's row of int32 bins in a CU's LDS
    if (nr > 38000) return fail(LC_EUNSUPPORTED, "HoughLinesP: rho %g gives %d distance bins for a %d x %d image; at most 38000 are supported",
                                rho_d, nr, w, h);
    hg.numangle = na;
    hg.numrho = nr;
    hg.tab.resize(na);
    for (int n = 0; n < na; n++) {
        AngleEntry e;
        e.c = (float)(std::cos((double)n * theta) * irho);
        e.s = (float)(std::sin((double)n * theta) * irho);
        const float a = -e.s, b = e.c;
        if (std::fabs(a) > std::fabs(b)) {
            e.xflag = 1;
            e.dmaj = a > 0 ? 1 : -1;
            e.dmin = (int)lrintf(b * (1 << 16) / std::fabs(a));
        } else {
            e.xflag = 0;
            e.dmaj = b > 0 ? 1 : -1;
            e.dmin = (int)lrintf(a * (1 << 16) / std::fabs(b));
        }
        e.pad = 0;
        hg.tab[n] = e;
    }
    return LC_OK;
}

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
    DevBuf<uint32_t> nz;
    int nz_cap = 0;
    DevBuf<int32_t> acc_full, acc_ppht, seg;
    DevBuf<AngleEntry> tab;
    DevBuf<uint32_t> rng_out;     // first RNG_TABLE outputs of cv::RNG(2^64 - 1), HoughLinesP's fixed seed
    HoughGeo hg;
    double rho = 0, theta = 0;
    int seg_cap = 0;
    // pipeline slots keep everything that goes back to the host after a batch in one block, so that one
    // copy fetches it: [segment counts B][hysteresis rounds B][status, 7 spare][segments B * cap * 4]
    DevBuf<int32_t> resblock;
    size_t res_head = 0;
    int32_t *r_segcount = nullptr, *r_rounds = nullptr, *r_status = nullptr, *r_seg = nullptr;
    int init_results(int segcap) {
        int rc;
        res_head = 2 * (size_t)B + 8;
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
            const size_t nf = (size_t)B * front_stream_strips(g) * H;
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
        if ((rc = status.alloc(4))) return rc;
        if ((rc = redo.alloc(B))) return rc;
        HIP_TRY(hipMemset(redo.p, 0, sizeof(int) * (size_t)B));
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
        if (ppht && !rng_out.p) {
            std::vector<uint32_t> r(RNG_TABLE);
            unsigned long long state = 0xFFFFFFFFFFFFFFFFull;
            for (int k = 0; k < RNG_TABLE; k++) {
                state = (unsigned long long)(uint32_t)state * 4164903690ull + (state >> 32);
                r[k] = (uint32_t)state;
            }
            if ((rc = rng_out.alloc(RNG_TABLE))) return rc;
            HIP_TRY(hipMemcpyAsync(rng_out.p, r.data(), sizeof(uint32_t) * RNG_TABLE, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
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

}  // namespace

// ---------------------------------------------------------------------------
// Working memory of lc_canny_noalloc: a fixed device array instead of hipMalloc'ed buffers, so that the call
// allocates nothing.  Every GPU the library runs on holds its own copy at its own address until the process ends.
// `scratch` itself is only a host-side handle of the symbol: the device address comes from hipGetSymbolAddress
// (scratch_address, per device) and is what host code passes to kernels and memsets.
// Layout for a w x h image (TX = ceil(w / 64), TY = ceil(h / 64)):
//   [planeS TX * h u64][planeW TX * h u64][wake 2 * TX * TY u32][rounds 1 i32]
constexpr int NOALLOC_MAX_W = 4096, NOALLOC_MAX_H = 2304;          // covers 3840 x 2160
constexpr size_t NOALLOC_PLANE_WORDS = (size_t)(NOALLOC_MAX_W / 64) * NOALLOC_MAX_H;
constexpr size_t NOALLOC_WAKE_WORDS = 2 * (size_t)(NOALLOC_MAX_W / 64) * ((NOALLOC_MAX_H + 63) / 64);
constexpr size_t NOALLOC_BYTES = 2 * NOALLOC_PLANE_WORDS * 8 + NOALLOC_WAKE_WORDS * 4 + 256;   // 2.4 MB
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

namespace {
// device address of `scratch` on the current device, looked up once per device
int scratch_address(int device, unsigned char **out) {
    static std::mutex mu;
    static unsigned char *addr[64] = {};
    if (device < 0 || device >= 64) return fail(LC_EUNSUPPORTED, "device %d: at most 64 devices are supported", device);
    std::lock_guard<std::mutex> lock(mu);
    if (!addr[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(LC_EHIP, "hipGetSymbolAddress returned a null address for the scratch array");
        addr[device] = static_cast<unsigned char *>(p);
    }
    *out = addr[device];
    return LC_OK;
}
}  // namespace

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

namespace {

int ctx_bind(lc_ctx *ctx) {
    if (!ctx) return fail(LC_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return LC_OK;
}

// per-op calls: bind, and create the context's stream on first use -- a process that only runs pipelines never
// owns it (HIP folds a process's streams onto a few hardware queues, and the pipeline's streams want them)
int ctx_op(lc_ctx *ctx) {
    int rc;
    if ((rc = ctx_bind(ctx))) return rc;
    if (!ctx->stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    return LC_OK;
}

int check_image(const void *p, int w, int h, size_t stride, size_t min_stride, const char *what) {
    if (!p) return fail(LC_EINVAL, "%s: null image pointer", what);
    if (w <= 0 || h <= 0) return fail(LC_EINVAL, "%s: empty image %d x %d", what, w, h);
    if (w > MAX_DIM || h > MAX_DIM) return fail(LC_EUNSUPPORTED, "%s: images above %d pixels a side are not supported", what, MAX_DIM);
    if (stride < min_stride) return fail(LC_EINVAL, "%s: stride %zu below row size %zu", what, stride, min_stride);
    return LC_OK;
}

int ctx_work(lc_ctx *ctx, int w, int h) {
    if (ctx->work_w == w && ctx->work_h == h) return LC_OK;
    ctx->work_w = ctx->work_h = 0;
    ctx->work = Work();
    int rc;
    if ((rc = ctx->work.init_images(w, h, 1, true, false, true))) return rc;
    if ((rc = ctx->work.init_points(w * h))) return rc;
    size_t n = (size_t)w * h;
    if ((rc = ctx->tmpA.alloc(n)) || (rc = ctx->tmpB.alloc(n)) || (rc = ctx->tmpC.alloc(n)) || (rc = ctx->tmpD.alloc(n))) return rc;
    HIP_TRY(hipDeviceSynchronize());   // the clears above ran on the null stream, which ctx->stream does not wait for
    ctx->work_w = w; ctx->work_h = h;
    return LC_OK;
}

// Canny's integer thresholds: swapped if out of order, floored
int canny_thresholds(double t1, double t2, int &low, int &high) {
    if (!(t1 == t1) || !(t2 == t2)) return fail(LC_EINVAL, "Canny: NaN threshold");
    if (t1 > t2) std::swap(t1, t2);
    t1 = std::max(-1e9, std::min(1e9, t1));
    t2 = std::max(-1e9, std::min(1e9, t2));
    low = (int)std::floor(t1);
    high = (int)std::floor(t2);
    return LC_OK;
}

int ctx_hsv_tab(lc_ctx *ctx) {
    if (ctx->hsv_tab.p) return LC_OK;
    int rc, tab[512];
    hsv_tables_host(tab);
    if ((rc = ctx->hsv_tab.alloc(512))) return rc;
    HIP_TRY(hipMemcpy(ctx->hsv_tab.p, tab, sizeof tab, hipMemcpyHostToDevice));
    return LC_OK;
}

int ctx_tmp3(lc_ctx *ctx, int w, int h) {
    const size_t n = 3 * (size_t)w * h;
    if (ctx->tmp3.n >= n) return LC_OK;
    return ctx->tmp3.alloc(n);
}

int up2d(void *dst, const void *src, size_t row_bytes, int h, size_t stride, hipStream_t st) {
    HIP_TRY(hipMemcpy2DAsync(dst, row_bytes, src, stride, row_bytes, h, hipMemcpyHostToDevice, st));
    return LC_OK;
}
int down2d(void *dst, size_t stride, const void *src, size_t row_bytes, int h, hipStream_t st) {
    HIP_TRY(hipGetLastError());   // a kernel launch of the call that failed (bad configuration, missing code object)
    HIP_TRY(hipMemcpy2DAsync(dst, stride, src, row_bytes, row_bytes, h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LC_OK;
}

// non-zero image already in tmpA -> point list of work (row-major order) and row plane
int points_from_bytes(lc_ctx *ctx, int *npoints_host) {
    Work &w = ctx->work;
    hipStream_t st = ctx->stream;
    launch_pack_nonzero(ctx->tmpA.p, w.g, 1, w.rowplane.p, w.rowcount.p, st);
    launch_points(w.g, 1, w.rowplane.p, w.rowcount.p, w.nz.p, w.nz_cap, w.npoints.p, nullptr, st);
    if (npoints_host) {
        HIP_TRY(hipMemcpyAsync(npoints_host, w.npoints.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return LC_OK;
}

// ---- polygon set-up on the host (drawing.cpp: clipLine, LineIterator, CollectPolyEdges) ----
struct P64 { long long x, y; };

bool clip_line(long long width, long long height, P64 &p1, P64 &p2) {
    const long long right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return false;
    long long &x1 = p1.x, &y1 = p1.y, &x2 = p2.x, &y2 = p2.y;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (x2 - x1) / (y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (x2 - x1) / (y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (y2 - y1) / (x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (y2 - y1) / (x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

bool inside(const P64 &p, int w, int h) {
    return (unsigned long long)p.x < (unsigned long long)w && (unsigned long long)p.y < (unsigned long long)h;
}

// the clipped, left-to-right 8-connected segment LineIterator would walk
bool make_line(int w, int h, P64 a, P64 b, LineDev &L) {
    if (!inside(a, w, h) || !inside(b, w, h)) {
        if (!clip_line(w, h, a, b)) return false;
    }
    int dx = (int)(b.x - a.x), dy = (int)(b.y - a.y);
    int sx = 1, sy = 1;
    if (dx < 0) { dx = -dx; dy = -dy; a = b; }
    if (dy < 0) { dy = -dy; sy = -1; }
    L.x0 = (int)a.x; L.y0 = (int)a.y;
    L.vert = dy > dx;
    if (L.vert) { L.dmaj = sy; L.dmin = sx; L.dmajor_len = dy; L.dminor_len = dx; }
    else        { L.dmaj = sx; L.dmin = sy; L.dmajor_len = dx; L.dminor_len = dy; }
    L.count = L.dmajor_len + 1;
    return true;
}

// ---- thick lines: drawing.cpp's ThickLine / FillConvexPoly / Line2 / Circle set-up on the host -----------------
// What one cv2.line / cv2.polylines call paints, as closed-form pieces for the device (kernels_poly.hip).
struct ThickPieces {
    std::vector<TrapDev> traps;
    std::vector<Line2Dev> lines2;
    std::vector<LineDev> thin;
    std::vector<int2> circles;
    int max_count = 0;          // longest Line2, in steps
    int radius = 0;             // of every cap circle of the call
};

// Line2 up to the point where its loop starts: clip to the image in 16.16 fixed point, order the end points along
// the major axis, fixed-point step of the minor one
void add_line2(int w, int h, P64 pt1, P64 pt2, ThickPieces &out) {
    if (!clip_line((long long)w << 16, (long long)h << 16, pt1, pt2)) return;
    long long dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
    const long long j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
    const long long i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
    Line2Dev L{};
    if (ax > ay) {
        dy = (dy ^ j) - j;
        if (j) std::swap(pt1, pt2);
        L.xmajor = 1;
        L.step = dy * (1 << 16) / (ax | 1);
        L.ecount = (int)((pt2.x - pt1.x) >> 16);
    } else {
        dx = (dx ^ i) - i;
        if (i) std::swap(pt1, pt2);
        L.xmajor = 0;
        L.step = dx * (1 << 16) / (ay | 1);
        L.ecount = (int)((pt2.y - pt1.y) >> 16);
    }
    pt1.x += 32768;
    pt1.y += 32768;
    L.endx = (pt2.x + 32768) >> 16;
    L.endy = (pt2.y + 32768) >> 16;
    if (L.xmajor) { L.major0 = pt1.x >> 16; L.minor0 = pt1.y; }
    else          { L.major0 = pt1.y >> 16; L.minor0 = pt1.x; }
    out.max_count = std::max(out.max_count, L.ecount);
    out.lines2.push_back(L);
}

// FillConvexPoly(v, npts, LINE_8, shift 16): the outline as Line2 pieces, the fill as the trapezoids between the
// events of its two-chain edge walk (the walk itself jumps from event to event instead of row to row)
void add_convex_poly(int w, int h, const P64 *v, int npts, ThickPieces &out) {
    struct { int idx, di; long long x, dx; int ye; } edge[2];
    const long long delta = 32768;
    int imin = 0, edges = npts;
    long long xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    P64 p0 = v[npts - 1];
    for (int i = 0; i < npts; i++) {
        const P64 p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        ymax = std::max(ymax, p.y);
        xmax = std::max(xmax, p.x);
        xmin = std::min(xmin, p.x);
        add_line2(w, h, p0, p, out);
        p0 = p;
    }
    xmin = (xmin + delta) >> 16; xmax = (xmax + delta) >> 16;
    ymin = (ymin + delta) >> 16; ymax = (ymax + delta) >> 16;
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= w || (int)ymin >= h) return;
    ymax = std::min<long long>(ymax, h - 1);
    int y = (int)ymin;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1; edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -65536;
    edge[0].dx = edge[1].dx = 0;
    while (y <= (int)ymax) {
        for (int i = 0; i < 2; i++) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    const int ty = (int)((v[idx].y + delta) >> 16);
                    if (ty > y) {
                        const long long xs = v[idx0].x, xe = v[idx].x;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        // nothing changes until the next row at which a chain reaches the end of its edge
        const int y_next = std::max(y + 1, std::min(std::min(edge[0].ye, edge[1].ye), (int)ymax + 1));
        TrapDev t{y, y_next, edge[0].x, edge[0].dx, edge[1].x, edge[1].dx};
        out.traps.push_back(t);
        edge[0].x += edge[0].dx * (y_next - y);
        edge[1].x += edge[1].dx * (y_next - y);
        y = y_next;
    }
}

// ThickLine(p0, p1, thickness, LINE_8, flags, shift 0): flags bit 0 / 1 = cap at p0 / p1
void add_thick_line(int w, int h, P64 p0, P64 p1, int thickness, int flags, ThickPieces &out) {
    if (thickness <= 1) {
        LineDev L;
        if (make_line(w, h, p0, p1, L)) out.thin.push_back(L);
        return;
    }
    p0.x <<= 16; p0.y <<= 16; p1.x <<= 16; p1.y <<= 16;
    const double dx = (double)(p0.x - p1.x) * (1. / 65536), dy = (double)(p1.y - p0.y) * (1. / 65536);
    double r = dx * dx + dy * dy;
    const int odd = thickness & 1;
    thickness <<= 15;
    if (std::fabs(r) > 2.220446049250313e-16) {
        r = (thickness + odd * 65536 * 0.5) / std::sqrt(r);
        const long long dpx = lrint(dy * r), dpy = lrint(dx * r);
        const P64 pt[4] = {{p0.x + dpx, p0.y + dpy}, {p0.x - dpx, p0.y - dpy}, {p1.x - dpx, p1.y - dpy}, {p1.x + dpx, p1.y + dpy}};
        add_convex_poly(w, h, pt, 4, out);
    }
    out.radius = (thickness + 32768) >> 16;
    for (int i = 0; i < 2; i++) {
        if (flags & (i + 1)) out.circles.push_back(int2{(int)((p0.x + 32768) >> 16), (int)((p0.y + 32768) >> 16)});
        p0 = p1;
    }
}

// Circle(radius, fill): half-width of every row -radius .. radius (the midpoint walk emits, per step, spans of half-width
// dx on rows +-dy and of half-width dy on rows +-dx; spans of one row share the centre, so the widest is their union)
std::vector<int> circle_half_widths(int radius) {
    std::vector<int> hw(2 * radius + 1, -1);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hw[radius - dy] = std::max(hw[radius - dy], dx); hw[radius + dy] = std::max(hw[radius + dy], dx);
        hw[radius - dx] = std::max(hw[radius - dx], dy); hw[radius + dx] = std::max(hw[radius + dx], dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
    return hw;
}

template <typename T>
int upload_vec(DevBuf<T> &d, const std::vector<T> &v, hipStream_t st) {
    int rc;
    if ((rc = d.alloc(v.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return LC_OK;
}

// paints the pieces into a host image of cn channels, in place
int paint_thick(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int cn, const int *color, const ThickPieces &tp) {
    int rc;
    uint32_t col = 0;
    for (int k = 0; k < cn; k++) col |= (uint32_t)std::min(255, std::max(0, color[k])) << (8 * k);
    const size_t row_bytes = (size_t)w * cn, n = row_bytes * h;
    if (ctx->rs_dst.n < n && (rc = ctx->rs_dst.alloc(n))) return rc;
    if ((rc = up2d(ctx->rs_dst.p, img, row_bytes, h, stride, ctx->stream))) return rc;
    DevBuf<TrapDev> dt;
    DevBuf<Line2Dev> dl2;
    DevBuf<LineDev> dl;
    DevBuf<int2> dc;
    DevBuf<int> dhw;
    if (!tp.traps.empty()) {
        if ((rc = upload_vec(dt, tp.traps, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.traps.size(); o += 32768)
            launch_paint_traps(ctx->rs_dst.p, w, h, cn, dt.p + o, (int)std::min<size_t>(32768, tp.traps.size() - o), col, ctx->stream);
    }
    if (!tp.lines2.empty()) {
        if ((rc = upload_vec(dl2, tp.lines2, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.lines2.size(); o += 32768)
            launch_paint_line2(ctx->rs_dst.p, w, h, cn, dl2.p + o, (int)std::min<size_t>(32768, tp.lines2.size() - o), tp.max_count, col,
                               ctx->stream);
    }
    if (!tp.thin.empty()) {
        if ((rc = upload_vec(dl, tp.thin, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.thin.size(); o += 32768)
            launch_draw_lines_cn(ctx->rs_dst.p, w, h, cn, dl.p + o, (int)std::min<size_t>(32768, tp.thin.size() - o), col, ctx->stream);
    }
    if (!tp.circles.empty()) {
        const std::vector<int> hw = circle_half_widths(tp.radius);
        if ((rc = upload_vec(dc, tp.circles, ctx->stream)) || (rc = upload_vec(dhw, hw, ctx->stream))) return rc;
        for (size_t o = 0; o < tp.circles.size(); o += 32768)
            launch_paint_circles(ctx->rs_dst.p, w, h, cn, dc.p + o, (int)std::min<size_t>(32768, tp.circles.size() - o), dhw.p, tp.radius,
                                 col, ctx->stream);
    }
    return down2d(img, stride, ctx->rs_dst.p, row_bytes, h, ctx->stream);   // synchronises: the piece buffers may go
}

int check_draw(const void *img, int w, int h, size_t stride, int cn, const int *color, int thickness, const char *what) {
    int rc;
    if (cn < 1 || cn > 4) return fail(LC_EUNSUPPORTED, "%s: images of 1 to 4 channels are supported, got %d", what, cn);
    if ((rc = check_image(img, w, h, stride, (size_t)w * cn, what))) return rc;
    if (!color) return fail(LC_EINVAL, "%s: null colour", what);
    if (thickness < 1) return fail(LC_EINVAL, "%s: thickness must be positive (filled shapes are not lines), got %d", what, thickness);
    if (thickness > 255) return fail(LC_EUNSUPPORTED, "%s: thickness above 255 is not supported", what);
    return LC_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

int lc_version(void) { return 100; }

const char *lc_last_error(void) { return g_err.c_str(); }

int lc_create(int device, lc_ctx **out) {
    if (!out) return fail(LC_EINVAL, "lc_create: null output");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(LC_EHIP, "lc_create: no HIP device available (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return fail(LC_EINVAL, "lc_create: device %d out of range [0, %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    lc_ctx *ctx = new lc_ctx();
    ctx->device = device;
    if (int rc = scratch_address(device, &ctx->scratch_dev)) { delete ctx; return rc; }
    *out = ctx;
    return LC_OK;
}

int lc_destroy(lc_ctx *ctx) {
    if (!ctx) return LC_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
    return LC_OK;
}

int lc_host_alloc_pinned(size_t bytes, void **out) {
    if (!out) return fail(LC_EINVAL, "null output");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return LC_OK;
}

int lc_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return LC_OK;
}

int lc_cvt_gray(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride, int code, uint8_t *dst, size_t dst_stride) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if (code != LC_COLOR_BGR2GRAY && code != LC_COLOR_RGB2GRAY)
        return fail(LC_EUNSUPPORTED, "cvtColor: only COLOR_BGR2GRAY (6) and COLOR_RGB2GRAY (7) are supported, got %d", code);
    if ((rc = check_image(src, w, h, src_stride, 3 * (size_t)w, "cvtColor src"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, (size_t)w, "cvtColor dst"))) return rc;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->work.bgr.p, src, 3 * (size_t)w, h, src_stride, ctx->stream))) return rc;
    launch_gray(ctx->work.bgr.p, w, h, 1, code, ctx->tmpA.p, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmpA.p, w, h, ctx->stream);
}

int lc_cvt_hsv(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride, int code, uint8_t *dst, size_t dst_stride) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if (code != LC_COLOR_BGR2HSV && code != LC_COLOR_RGB2HSV)
        return fail(LC_EUNSUPPORTED, "cvtColor: only COLOR_BGR2HSV (40) and COLOR_RGB2HSV (41) are supported here, got %d", code);
    if ((rc = check_image(src, w, h, src_stride, 3 * (size_t)w, "cvtColor src"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, 3 * (size_t)w, "cvtColor dst"))) return rc;
    if ((rc = ctx_work(ctx, w, h)) || (rc = ctx_tmp3(ctx, w, h)) || (rc = ctx_hsv_tab(ctx))) return rc;
    if ((rc = up2d(ctx->work.bgr.p, src, 3 * (size_t)w, h, src_stride, ctx->stream))) return rc;
    launch_hsv(ctx->work.bgr.p, (size_t)w * h, code, ctx->hsv_tab.p, ctx->tmp3.p, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmp3.p, 3 * (size_t)w, h, ctx->stream);
}

int lc_in_range3(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride, const int *lowerb, const int *upperb,
                 uint8_t *dst, size_t dst_stride) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if (!lowerb || !upperb) return fail(LC_EINVAL, "inRange: null bounds");
    if ((rc = check_image(src, w, h, src_stride, 3 * (size_t)w, "inRange src"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, (size_t)w, "inRange dst"))) return rc;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->work.bgr.p, src, 3 * (size_t)w, h, src_stride, ctx->stream))) return rc;
    const int b[6] = {lowerb[0], lowerb[1], lowerb[2], upperb[0], upperb[1], upperb[2]};
    launch_in_range3(ctx->work.bgr.p, (size_t)w * h, make_gate(b, 1), ctx->tmpA.p, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmpA.p, w, h, ctx->stream);
}

int lc_or(lc_ctx *ctx, const uint8_t *a, size_t a_stride, const uint8_t *b, size_t b_stride, uint8_t *dst, size_t dst_stride,
          int w, int h) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(a, w, h, a_stride, (size_t)w, "bitwise_or src1"))) return rc;
    if ((rc = check_image(b, w, h, b_stride, (size_t)w, "bitwise_or src2"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, (size_t)w, "bitwise_or dst"))) return rc;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, a, w, h, a_stride, ctx->stream))) return rc;
    if ((rc = up2d(ctx->tmpB.p, b, w, h, b_stride, ctx->stream))) return rc;
    launch_or(ctx->tmpA.p, ctx->tmpB.p, ctx->tmpD.p, (size_t)w * h, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmpD.p, w, h, ctx->stream);
}

int lc_and3_mask(lc_ctx *ctx, const uint8_t *src, size_t src_stride, const uint8_t *mask, size_t mask_stride, uint8_t *dst,
                 size_t dst_stride, int w, int h) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(src, w, h, src_stride, 3 * (size_t)w, "bitwise_and src"))) return rc;
    if ((rc = check_image(mask, w, h, mask_stride, (size_t)w, "bitwise_and mask"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, 3 * (size_t)w, "bitwise_and dst"))) return rc;
    if ((rc = ctx_work(ctx, w, h)) || (rc = ctx_tmp3(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->work.bgr.p, src, 3 * (size_t)w, h, src_stride, ctx->stream))) return rc;
    if ((rc = up2d(ctx->tmpC.p, mask, w, h, mask_stride, ctx->stream))) return rc;
    launch_and3_mask(ctx->work.bgr.p, ctx->tmpC.p, ctx->tmp3.p, (size_t)w * h, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmp3.p, 3 * (size_t)w, h, ctx->stream);
}

int lc_add_weighted(lc_ctx *ctx, const uint8_t *a, size_t a_stride, double alpha, const uint8_t *b, size_t b_stride, double beta,
                    double gamma, uint8_t *dst, size_t dst_stride, int row_bytes, int h) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if (!a || !b || !dst) return fail(LC_EINVAL, "addWeighted: null image pointer");
    if (row_bytes <= 0 || h <= 0) return fail(LC_EINVAL, "addWeighted: empty image");
    if (a_stride < (size_t)row_bytes || b_stride < (size_t)row_bytes || dst_stride < (size_t)row_bytes)
        return fail(LC_EINVAL, "addWeighted: stride below row size");
    if (!(alpha == alpha) || !(beta == beta) || !(gamma == gamma)) return fail(LC_EINVAL, "addWeighted: NaN weight");
    const size_t n = (size_t)row_bytes * h;
    if (ctx->rs_src.n < n && (rc = ctx->rs_src.alloc(n))) return rc;
    if (ctx->rs_dst.n < n && (rc = ctx->rs_dst.alloc(n))) return rc;
    if (ctx->tmp3.n < n && (rc = ctx->tmp3.alloc(n))) return rc;
    if ((rc = up2d(ctx->rs_src.p, a, row_bytes, h, a_stride, ctx->stream))) return rc;
    if ((rc = up2d(ctx->tmp3.p, b, row_bytes, h, b_stride, ctx->stream))) return rc;
    launch_add_weighted(ctx->rs_src.p, (float)alpha, ctx->tmp3.p, (float)beta, (float)gamma, ctx->rs_dst.p, n, ctx->stream);
    return down2d(dst, dst_stride, ctx->rs_dst.p, row_bytes, h, ctx->stream);
}

int lc_lds_atomic_rate(lc_ctx *ctx, int run, int returning, double *gadds_per_s) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if (!gadds_per_s) return fail(LC_EINVAL, "lds_atomic_rate: null result pointer");
    if (run < 0 || run > 64) return fail(LC_EINVAL, "lds_atomic_rate: run length %d outside [0, 64]", run);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 8;   // eight 256-thread workgroups (8 KiB of LDS each) per CU
    DevBuf<int> sink;
    if ((rc = sink.alloc(blocks))) return rc;
    float ms = 0.f;
    double adds = 0;
    // slow patterns serialise inside the LDS unit: fewer iterations keep every launch in the millisecond range
    const int iters = run == 1 ? 4000 : 1000;
    if (lds_atomic_probe(run, returning != 0, blocks, iters, sink.p, ctx->stream, &ms, &adds) || !(ms > 0.f))
        return fail(LC_EHIP, "lds_atomic_rate: probe launch failed: %s", hipGetErrorString(hipGetLastError()));
    *gadds_per_s = adds / (ms * 1e-3) / 1e9;
    return LC_OK;
}

int lc_resize_linear(lc_ctx *ctx, const uint8_t *src, int sw, int sh, size_t src_stride, int channels, uint8_t *dst, int dw, int dh,
                     size_t dst_stride) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if (channels != 1 && channels != 3) return fail(LC_EUNSUPPORTED, "resize: 1 or 3 channels are supported, got %d", channels);
    if ((rc = check_image(src, sw, sh, src_stride, (size_t)sw * channels, "resize src"))) return rc;
    if ((rc = check_image(dst, dw, dh, dst_stride, (size_t)dw * channels, "resize dst"))) return rc;
    const size_t sn = (size_t)sw * sh * channels, dn = (size_t)dw * dh * channels;
    if (ctx->rs_src.n < sn && (rc = ctx->rs_src.alloc(sn))) return rc;
    if (ctx->rs_dst.n < dn && (rc = ctx->rs_dst.alloc(dn))) return rc;
    const size_t tn = 2 * ((size_t)dw + dh);
    if (ctx->rs_tab.n < tn && (rc = ctx->rs_tab.alloc(tn))) return rc;
    std::vector<int> tab(tn);
    const bool area2 = resize_tables_host(sw, sh, dw, dh, tab.data(), tab.data() + dw, tab.data() + 2 * dw, tab.data() + 2 * dw + dh);
    HIP_TRY(hipMemcpyAsync(ctx->rs_tab.p, tab.data(), sizeof(int) * tn, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = up2d(ctx->rs_src.p, src, (size_t)sw * channels, sh, src_stride, ctx->stream))) return rc;
    const bool dec3 = !area2 && resize_is_decimate3(sw, sh, dw, dh, tab.data(), tab.data() + dw, tab.data() + 2 * dw, tab.data() + 2 * dw + dh);
    launch_resize_linear(ctx->rs_src.p, sw, sh, channels, 1, ctx->rs_dst.p, dw, dh, ctx->rs_tab.p, area2, ctx->stream, dec3);
    return down2d(dst, dst_stride, ctx->rs_dst.p, (size_t)dw * channels, dh, ctx->stream);   // synchronises: tab may go
}

int lc_gaussian5(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride, uint8_t *dst, size_t dst_stride) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(src, w, h, src_stride, (size_t)w, "GaussianBlur src"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, (size_t)w, "GaussianBlur dst"))) return rc;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, src, w, h, src_stride, ctx->stream))) return rc;
    launch_blur5(ctx->tmpA.p, w, h, 1, ctx->tmpB.p, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmpB.p, w, h, ctx->stream);
}

int lc_canny(lc_ctx *ctx, const uint8_t *src, int w, int h, size_t src_stride, double t1, double t2, uint8_t *dst,
             size_t dst_stride) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(src, w, h, src_stride, (size_t)w, "Canny image"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, (size_t)w, "Canny edges"))) return rc;
    int low, high;
    if ((rc = canny_thresholds(t1, t2, low, high))) return rc;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    Work &wk = ctx->work;
    if ((rc = up2d(ctx->tmpA.p, src, w, h, src_stride, ctx->stream))) return rc;
    launch_front(ctx->tmpA.p, (size_t)w * h, 0, LC_COLOR_BGR2GRAY, false, wk.g, 1, low, high, wk.planeS.p, wk.planeW.p,
                 wk.wake.p, nullptr, nullptr, ctx->stream);
    launch_hysteresis(wk.g, 1, wk.planeS.p, wk.planeW.p, wk.wake.p, wk.rounds.p, ctx->stream);
    launch_finalize(wk.g, 1, wk.planeS.p, nullptr, ctx->tmpB.p, nullptr, nullptr, nullptr, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmpB.p, w, h, ctx->stream);
}

int lc_canny_noalloc(lc_ctx *ctx, const void *dev_image, int w, int h, double t1, double t2, void *dev_edges, void *stream) {
    int rc;
    if ((rc = ctx_bind(ctx))) return rc;
    if (!dev_image || !dev_edges) return fail(LC_EINVAL, "Canny: null device pointer");
    if (w <= 0 || h <= 0) return fail(LC_EINVAL, "Canny: empty image %d x %d", w, h);
    if (w > NOALLOC_MAX_W || h > NOALLOC_MAX_H)
        return fail(LC_EUNSUPPORTED, "lc_canny_noalloc: %d x %d exceeds the fixed working memory (at most %d x %d); use lc_canny",
                    w, h, NOALLOC_MAX_W, NOALLOC_MAX_H);
    if (((uintptr_t)dev_edges & 15) != 0) return fail(LC_EINVAL, "lc_canny_noalloc: dev_edges must be 16-byte aligned");
    int low, high;
    if ((rc = canny_thresholds(t1, t2, low, high))) return rc;
    if (!ctx->scratch_dev) return fail(LC_EHIP, "lc_canny_noalloc: the context has no scratch address");
    const Geo g = make_geo(w, h);
    const size_t plane_words = g.plane_words(), wake_words = 2 * (size_t)g.TX * g.TY;
    unsigned char *base = ctx->scratch_dev;                                   // the device address, never `scratch` itself
    uint64_t *planeS = reinterpret_cast<uint64_t *>(base);
    uint64_t *planeW = planeS + plane_words;
    uint32_t *wake = reinterpret_cast<uint32_t *>(planeW + plane_words);
    int *rounds = reinterpret_cast<int *>(wake + wake_words);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the array keeps the previous call's contents: the wake flags must start at zero (a call that was cut short
    // may have left some set); both planes are written whole by the front kernel before anything reads them
    HIP_TRY(hipMemsetAsync(wake, 0, wake_words * sizeof(uint32_t), st));
    launch_front(static_cast<const uint8_t *>(dev_image), (size_t)w * h, 0, LC_COLOR_BGR2GRAY, false, g, 1, low, high, planeS, planeW,
                 wake, nullptr, nullptr, st);
    launch_hysteresis(g, 1, planeS, planeW, wake, rounds, st);
    launch_finalize(g, 1, planeS, nullptr, static_cast<uint8_t *>(dev_edges), nullptr, nullptr, nullptr, st);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

int lc_and(lc_ctx *ctx, const uint8_t *a, size_t a_stride, const uint8_t *b, size_t b_stride, const uint8_t *mask,
           size_t mask_stride, uint8_t *dst, size_t dst_stride, int w, int h) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(a, w, h, a_stride, (size_t)w, "bitwise_and src1"))) return rc;
    if ((rc = check_image(b, w, h, b_stride, (size_t)w, "bitwise_and src2"))) return rc;
    if ((rc = check_image(dst, w, h, dst_stride, (size_t)w, "bitwise_and dst"))) return rc;
    if (mask && (rc = check_image(mask, w, h, mask_stride, (size_t)w, "bitwise_and mask"))) return rc;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, a, w, h, a_stride, ctx->stream))) return rc;
    if ((rc = up2d(ctx->tmpB.p, b, w, h, b_stride, ctx->stream))) return rc;
    if (mask && (rc = up2d(ctx->tmpC.p, mask, w, h, mask_stride, ctx->stream))) return rc;
    launch_and(ctx->tmpA.p, ctx->tmpB.p, mask ? ctx->tmpC.p : nullptr, ctx->tmpD.p, (size_t)w * h, ctx->stream);
    return down2d(dst, dst_stride, ctx->tmpD.p, w, h, ctx->stream);
}

int lc_fill_poly(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, const int32_t *pts, const int *npts, int ncontours,
                 int color) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(img, w, h, stride, (size_t)w, "fillPoly img"))) return rc;
    if (ncontours < 0 || (ncontours > 0 && (!pts || !npts))) return fail(LC_EINVAL, "fillPoly: bad contour arguments");
    color = color < 0 ? 0 : (color > 255 ? 255 : color);
    std::vector<PolyEdgeDev> edges;
    std::vector<LineDev> lines;
    const int32_t *v = pts;
    for (int c = 0; c < ncontours; v += 2 * (size_t)npts[c], c++) {
        const int count = npts[c];
        if (count < 0) return fail(LC_EINVAL, "fillPoly: negative vertex count");
        if (count == 0) continue;
        P64 pt0 = {(long long)v[2 * (count - 1)] << 16, v[2 * (count - 1) + 1]}, pt1;
        for (int i = 0; i < count; i++, pt0 = pt1) {
            pt1.x = (long long)v[2 * i] << 16;
            pt1.y = v[2 * i + 1];
            P64 pt0c = pt0, pt1c = pt1;
            P64 t0 = {(pt0.x + 32768) >> 16, pt0.y}, t1 = {(pt1.x + 32768) >> 16, pt1.y};
            LineDev L;
            if (make_line(w, h, t0, t1, L)) lines.push_back(L);
            if (!inside(t0, w, h) || !inside(t1, w, h)) {
                clip_line(w, h, t0, t1);
                if (t0.y != t1.y) {
                    pt0c.y = t0.y; pt1c.y = t1.y;
                    pt0c.x = t0.x << 16; pt1c.x = t1.x << 16;
                }
            } else {
                pt0c.x += 32768;
                pt1c.x += 32768;
            }
            if (pt0.y == pt1.y) continue;
            PolyEdgeDev e;
            e.dx = (pt1c.x - pt0c.x) / (pt1c.y - pt0c.y);
            if (pt0.y < pt1.y) { e.y0 = (int)pt0.y; e.y1 = (int)pt1.y; e.x = pt0c.x + (pt0.y - pt0c.y) * e.dx; }
            else               { e.y0 = (int)pt1.y; e.y1 = (int)pt0.y; e.x = pt1c.x + (pt1.y - pt1c.y) * e.dx; }
            edges.push_back(e);
        }
    }
    if (edges.size() > 1024) return fail(LC_EUNSUPPORTED, "fillPoly: more than 1024 non-horizontal edges");
    // FillEdgeCollection's early outs
    bool spans = edges.size() >= 2;
    int y_min = 0, y_max = 0;
    if (spans) {
        long long x_max = -1, x_min = INT64_MAX;
        y_max = INT32_MIN; y_min = INT32_MAX;
        for (const PolyEdgeDev &e : edges) {
            long long x1 = e.x + (long long)(e.y1 - e.y0) * e.dx;
            y_min = std::min(y_min, e.y0); y_max = std::max(y_max, e.y1);
            x_min = std::min(x_min, std::min(e.x, x1)); x_max = std::max(x_max, std::max(e.x, x1));
        }
        if (y_max < 0 || y_min >= h || x_max < 0 || x_min >= ((long long)w << 16)) spans = false;
        y_min = std::max(y_min, 0);
        y_max = std::min(y_max, h);
    }
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, img, w, h, stride, ctx->stream))) return rc;
    DevBuf<PolyEdgeDev> de;
    DevBuf<LineDev> dl;
    if (!lines.empty()) {
        if ((rc = dl.alloc(lines.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(dl.p, lines.data(), lines.size() * sizeof(LineDev), hipMemcpyHostToDevice, ctx->stream));
        launch_draw_lines(ctx->tmpA.p, w, h, dl.p, (int)lines.size(), color, ctx->stream);
    }
    if (spans) {
        if ((rc = de.alloc(edges.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(de.p, edges.data(), edges.size() * sizeof(PolyEdgeDev), hipMemcpyHostToDevice, ctx->stream));
        launch_fill_spans(ctx->tmpA.p, w, h, de.p, (int)edges.size(), y_min, y_max, color, ctx->stream);
    }
    return down2d(img, stride, ctx->tmpA.p, w, h, ctx->stream);
}

int lc_rectangle_filled(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int x1, int y1, int x2, int y2, int color) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(img, w, h, stride, (size_t)w, "rectangle img"))) return rc;
    color = color < 0 ? 0 : (color > 255 ? 255 : color);
    if (x1 > x2) std::swap(x1, x2);
    if (y1 > y2) std::swap(y1, y2);
    x1 = std::max(x1, 0); y1 = std::max(y1, 0);
    x2 = std::min(x2, w - 1); y2 = std::min(y2, h - 1);
    if (x2 < x1 || y2 < y1) return LC_OK;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    if ((rc = up2d(ctx->tmpA.p, img, w, h, stride, ctx->stream))) return rc;
    launch_fill_rect(ctx->tmpA.p, w, h, x1, y1, x2, y2, color, ctx->stream);
    return down2d(img, stride, ctx->tmpA.p, w, h, ctx->stream);
}

int lc_line(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int channels, int x1, int y1, int x2, int y2,
            const int *color, int thickness) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_draw(img, w, h, stride, channels, color, thickness, "line"))) return rc;
    ThickPieces tp;
    add_thick_line(w, h, P64{x1, y1}, P64{x2, y2}, thickness, 3, tp);
    return paint_thick(ctx, img, w, h, stride, channels, color, tp);
}

int lc_polylines(lc_ctx *ctx, uint8_t *img, int w, int h, size_t stride, int channels, const int32_t *pts, const int *npts,
                 int ncontours, int is_closed, const int *color, int thickness) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_draw(img, w, h, stride, channels, color, thickness, "polylines"))) return rc;
    if (ncontours < 0 || (ncontours > 0 && (!pts || !npts))) return fail(LC_EINVAL, "polylines: bad contour arguments");
    ThickPieces tp;
    const int32_t *v = pts;
    for (int c = 0; c < ncontours; v += 2 * (size_t)npts[c], c++) {
        const int count = npts[c];
        if (count < 0) return fail(LC_EINVAL, "polylines: negative vertex count");
        if (count == 0) continue;
        int i = is_closed ? count - 1 : 0;
        int flags = 2 + !is_closed;                 // PolyLine: an open polyline caps its first point too
        P64 p0{v[2 * i], v[2 * i + 1]};
        for (i = !is_closed; i < count; i++) {
            const P64 p{v[2 * i], v[2 * i + 1]};
            add_thick_line(w, h, p0, p, thickness, flags, tp);
            p0 = p;
            flags = 2;
        }
    }
    return paint_thick(ctx, img, w, h, stride, channels, color, tp);
}

int lc_hough_geometry(int w, int h, double rho, double theta, int *numangle, int *numrho) {
    if (w <= 0 || h <= 0) return fail(LC_EINVAL, "empty image");
    HoughGeo hg;
    int rc = hough_geo(w, h, rho, theta, hg);
    if (rc) return rc;
    if (numangle) *numangle = hg.numangle;
    if (numrho) *numrho = hg.numrho;
    return LC_OK;
}

int lc_hough_accum(lc_ctx *ctx, const uint8_t *img, int w, int h, size_t stride, double rho, double theta, int32_t *acc,
                   int acc_capacity, int *numangle, int *numrho) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(img, w, h, stride, (size_t)w, "hough image"))) return rc;
    if (!acc) return fail(LC_EINVAL, "null accumulator");
    if ((rc = ctx_work(ctx, w, h))) return rc;
    Work &wk = ctx->work;
    if ((rc = wk.init_hough(rho, theta, 1, true, false, ctx->stream))) return rc;
    const int na = wk.hg.numangle, nr = wk.hg.numrho;
    if (numangle) *numangle = na;
    if (numrho) *numrho = nr;
    if ((long long)acc_capacity < (long long)na * nr) return fail(LC_ECAP, "accumulator needs %d x %d int32", na, nr);
    if ((rc = up2d(ctx->tmpA.p, img, w, h, stride, ctx->stream))) return rc;
    if ((rc = points_from_bytes(ctx, nullptr))) return rc;
    launch_vote(wk.g, 1, wk.nz.p, wk.nz_cap, wk.npoints.p, wk.tab.p, na, nr, wk.acc_full.p, nullptr, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(acc, wk.acc_full.p, sizeof(int32_t) * (size_t)na * nr, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LC_OK;
}

int lc_hough_lines_p(lc_ctx *ctx, const uint8_t *img, int w, int h, size_t stride, double rho, double theta, int threshold,
                     double min_line_length, double max_line_gap, int32_t *out, int capacity, int *n) {
    int rc;
    if ((rc = ctx_op(ctx))) return rc;
    if ((rc = check_image(img, w, h, stride, (size_t)w, "HoughLinesP image"))) return rc;
    if (!n || capacity < 0 || (capacity > 0 && !out)) return fail(LC_EINVAL, "HoughLinesP: bad output arguments");
    *n = 0;
    if ((rc = ctx_work(ctx, w, h))) return rc;
    Work &wk = ctx->work;
    const int cap = std::max(capacity, 1);
    if ((rc = wk.init_hough(rho, theta, cap, false, true, ctx->stream))) return rc;
    const int na = wk.hg.numangle, nr = wk.hg.numrho;
    if ((rc = up2d(ctx->tmpA.p, img, w, h, stride, ctx->stream))) return rc;
    if ((rc = points_from_bytes(ctx, nullptr))) return rc;
    HIP_TRY(hipMemsetAsync(wk.status.p, 0, 4 * sizeof(int), ctx->stream));
    const int line_length = (int)lrint(min_line_length), line_gap = (int)lrint(max_line_gap);
    const bool fast = launch_ppht_fast(wk.g, 1, wk.nz.p, wk.nz_cap, wk.npoints.p, wk.rowplane.p, wk.tab.p, na, nr, threshold, line_length,
                                       line_gap, wk.seg.p, cap, wk.segcount.p, wk.status.p, wk.fallback.p, wk.rng_out.p, RNG_TABLE,
                                       nullptr, ctx->stream);
    launch_ppht_general(wk.g, 1, wk.nz.p, wk.nz_cap, wk.npoints.p, wk.rowplane.p, wk.tab.p, na, nr, wk.acc_ppht.p, threshold, line_length,
                        line_gap, wk.seg.p, cap, wk.segcount.p, wk.status.p, fast ? wk.fallback.p : nullptr, ctx->stream);
    HIP_TRY(hipGetLastError());
    int host[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(&host[0], wk.segcount.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&host[1], wk.status.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (host[1] | host[2]) return fail(LC_EINVAL, "HoughLinesP: internal range error (status %d)", host[1] | (host[2] << 1));
    *n = host[0];
    const int ncopy = std::min(host[0], capacity);
    if (ncopy > 0) {
        HIP_TRY(hipMemcpyAsync(out, wk.seg.p, sizeof(int32_t) * 4 * (size_t)ncopy, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (host[0] > capacity) return fail(LC_ECAP, "HoughLinesP found %d segments, capacity is %d", host[0], capacity);
    return LC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// batched pipeline
// ---------------------------------------------------------------------------
struct Slot {
    Work w;
    hipEvent_t ev[12] = {};
    int nframes = 0;
    const uint8_t *input = nullptr;   // device pointer of the frames of the batch in flight
    size_t input_bytes = 0;
    bool busy = false, timed = false;
    int level = 0;                // ROI-scheme level of the batch in flight (or last); the pipeline's nlevels = classified in full
    int planes_level = 0;         // loosest level a batch of this slot ran at since its planes were last cleared down
    int group = 0, gpos = 0;      // group of slots whose HoughLinesP runs as one launch, and the slot's place in it
    bool lines_pending = false;   // the batch's points are queued, its HoughLinesP launch is not yet
    int *h_gstatus = nullptr;     // status words the group's HoughLinesP launch raises (the slot's own without groups)
    // pinned host mirror of the slot's result block
    int32_t *h_res = nullptr;
    bool zero_copy = false;       // the kernels write the result block straight into h_res (mapped pinned memory)
    int end_ev = 0;               // the event that marks the batch complete (index into ev)
    int *h_rounds = nullptr, *h_status = nullptr, *h_segcount = nullptr;
    int32_t *h_seg = nullptr;
};

struct lc_pipe {
    lc_ctx *ctx = nullptr;
    lc_pipe_params p{};
    Geo g{};
    int low = 0, high = 0, line_length = 0, line_gap = 0;
    DevBuf<uint64_t> roi_bits;
    bool has_roi = false;
    bool has_gate = false;        // colour gate in front of the gray conversion
    GateDev gate{};
    DevBuf<int> gate_tab;     // gate_tables_host: the HSV reciprocal tables and this gate's s/v intervals
    // cv2.resize in front of everything: submitted frames are src_w x src_h, the pipeline works on width x height
    bool resizes = false, resize_area2 = false, resize_dec3 = false;
    int src_w = 0, src_h = 0;
    size_t src_frame_bytes = 0;   // bytes of one submitted frame
    // the reduction by exactly three is read by the front kernel itself (no resized frame exists then): without
    // intermediates, without the colour gate, widths the streaming kernel takes
    bool fused_dec3() const {
        return resizes && resize_dec3 && !p.keep_intermediates && !has_gate && front_stream_supported(g) && !getenv("LANECV_FRONT_TILED") &&
               !getenv("LANECV_NO_FUSED_RESIZE");
    }
    DevBuf<int> resize_tab;
    // A pipeline owns three or four streams of equal priority (more HoughLinesP streams for small batches, below),
    // whatever its number of slots; a slot is a set of buffers and events.  HIP maps streams onto a few hardware
    // queues, so the count stays small, and the waits between streams run one way only -- st_up -> st_edge ->
    // st_post -> st_lines.
    //   st_up     host-to-device copies of submitted frames (created by the first host submit).
    //   st_edge   the front kernels (and the leading resize) of all slots, in submission order, back to back.
    //             They fill the GPU by themselves, so two of them gain nothing from running side by side.
    //   st_post   the short latency-bound kernels behind a front kernel: hysteresis, finalize, compaction (and the
    //             vote kernel of accumulator-only recipes); with a single slot it is st_edge itself.
    //   st_lines  HoughLinesP and the copy of the batch's results.
    hipStream_t st_up = nullptr, st_edge = nullptr, st_post = nullptr, st_lines = nullptr;
    // More HoughLinesP streams for small batches: the kernel is one sequential workgroup per frame that takes a CU
    // for itself, so a batch of 32 frames leaves 7/8 of the chip to whatever may run beside it -- and on a single
    // stream that is never another batch's HoughLinesP.  Batches take the streams in turn: CUs / max_batch of them,
    // at most three and one fewer than there are slots (720p x 32: 71 k frames/s with one stream, 124 k with two
    // of three slots, 146 k with three of six; four were slower again).  One stream when the batch fills the CUs (a
    // second one gained nothing there); LANECV_LINES_STREAMS overrides the count.
    std::vector<hipStream_t> st_lines_more;
    unsigned lines_turn = 0;
    // The streaming front kernel's unit height (fixed for the pipeline's largest batch: its change tracking is per unit)
    // and, for pipelines with an ROI, which units it classifies at all.  The final edge map is the hysteresis result
    // INSIDE the ROI, and that depends on pixels outside it only through weak pixels that connect the two.  The front
    // kernel classifies the units (strip x row segment) that come within 15 pixels of the ROI; launch_ring_check proves per
    // frame that nothing further out can matter (one of the 15 rings around the ROI without leftover weak pixels), and the
    // rare frame it cannot prove it for is classified again in full.  Not with intermediates (the pre-ROI edge map
    // must then be complete).  row_split: the first row of the lower region of the unit grid (rows above hold no
    // needed unit).
    int front_rows = 0, row_split = 0, roi_first_row = 0;
    // Levels of the scheme, tightest first: [0] the units within 15 pixels of the ROI, rings around the ROI; [1] all units
    // from row_split down, "rings" = the rows between row_split and the ROI's first row (the same proof around the band of
    // rows the ROI spans: for recipes whose weak pixels connect everything near the ROI, as the F1 tracker's low Canny
    // thresholds do, but not the sky above it).  `level` is what the next batch uses; nlevels = classify in full.  A batch
    // in which more than a sixteenth of the frames needed the full pass after all moves the pipeline one level out for
    // the next 256 batches, after which it tries one level in again.  A slot that goes back to a tighter level has the
    // plane rows of the units that level does not need cleared first (for every frame: `ones`).
    struct SkipLevel {
        DevBuf<uint16_t> needed, unneeded;      // units: segment | strip << 8
        int n_needed = 0, n_unneeded = 0, n_ring_words = 0;
        DevBuf<uint32_t> ring_words;            // plane words that hold ring pixels
        DevBuf<uint64_t> ring_code;             // four bit planes: the ring number (1..15) of every pixel, 0 = in no ring
        long long pixels = 0;                   // pixels per frame of the needed units
    };
    SkipLevel levels[2];
    int nlevels = 0, level = 0, level_hold = 0;
    DevBuf<int> ones;
    long long frame_pixels = 0;
    FrontSplit split(int lvl, int mode, const int *only) const {
        if (lvl >= nlevels) return FrontSplit{row_split, 0, only, nullptr, nullptr, 0, 0};
        const SkipLevel &L = levels[lvl];
        return FrontSplit{row_split, mode, only, L.needed.p, L.unneeded.p, L.n_needed, L.n_unneeded};
    }
    bool fuse_vote = true;        // HoughLinesP's kernel also produces the full-vote accumulator (in LDS, before its own)
    // HoughLinesP with half a CU's LDS per frame, so that two frames -- of two batches, on two streams -- share a CU:
    // the kernel is one sequential wavefront per frame and leaves most of a CU idle.  For frame sizes whose working
    // set usually fits (up to about 2.5 megapixels; 1080p road frames take 55 - 60 KiB).
    // A frame that needs more falls to the slow general kernel; lc_pipe_wait sees it (status word 2) and the next
    // `ppht_full_for` batches get a whole CU's LDS per frame again, after which the half size is tried anew.
    bool ppht_two_per_cu = false;
    int ppht_full_for = 0;
    int cu_count = 256;
    int shared_queues = 0;        // streams of this layout that found no hardware queue of their own (lc_pipe_create)

    // HoughLinesP and the front-end kernel slow each other down most (both want the vector ALUs).  The HoughLinesP
    // step of a batch is therefore not queued when the batch is submitted but when the NEXT one is, behind that
    // batch's front-end kernel: started at submit time it would run beside two front kernels in a row.  `deferred` is the slot whose HoughLinesP step is still to be queued (-1: none); waiting for
    // that slot queues it at once.  defer_after: 0 = never defer, 1 = behind the front end, 2 = behind hysteresis.
    int deferred = -1;
    int defer_after = 1;
    // Batches that leave most CUs free (HoughLinesP is one workgroup per frame) do not get a HoughLinesP launch each:
    // `group_size` consecutive slots form a group whose launch covers all their frames at once, queued when the
    // group's last slot has been submitted or one of its slots is waited for.  The buffers that launch indexes by frame
    // (point lists, counts, bit planes, accumulators, segments) are slices of one allocation per group, a slot's
    // frames following the previous slot's full capacity; 1 = no grouping.
    struct Group {
        DevBuf<uint32_t> nz;
        DevBuf<int> npoints, fallback;
        DevBuf<uint64_t> rowplane;
        DevBuf<int32_t> acc_full, acc_ppht;
        int32_t *h_res = nullptr;          // pinned: [segcount G*B][rounds G*B][status 8*G][group status 8][segments G*B*cap*4]
        int32_t *d_res = nullptr;          // the same block as the GPU addresses it
        int *h_gstatus = nullptr, *d_gstatus = nullptr;
        std::vector<int> pending;          // slots with points queued and HoughLinesP not yet, in submission order
    };
    int group_size = 1;
    std::vector<Group> groups;
    std::vector<Slot> slots;
};

namespace {

enum { EV_START = 0, EV_UPLOADED, EV_FRONT, EV_HYST, EV_EDGE, EV_POINTS, EV_VOTE, EV_LINES, EV_END, EV_ESTART, EV_LSTART, EV_VSTART };

void enqueue_lines(lc_pipe *pp, Slot &s, hipEvent_t after);
void flush_group(lc_pipe *pp, int g);

void enqueue_back(lc_pipe *pp, Slot &s) {
    Work &w = s.w;
    const lc_pipe_params &p = pp->p;
    const int n = s.nframes;
    hipStream_t es = pp->st_post;
    const bool hough = p.want_accumulator || p.want_segments;
    const bool points = hough || p.want_points;
    if (!points) (void)hipMemsetAsync(w.rowcount.p, 0, sizeof(int) * (size_t)n * w.g.H, es);   // nobody consumes (and clears) the counts
    launch_finalize(w.g, n, w.planeS.p, pp->has_roi ? pp->roi_bits.p : nullptr, w.inter ? w.edges.p : nullptr, w.masked.p,
                    w.rowplane.p, w.rowcount.p, es, w.inter ? nullptr : w.prevbits.p);
    if (s.level < pp->nlevels) {
        // frames that were classified in full: the plane rows of the units no frame needs are cleared again, now that
        // they have been read (a few workgroups that look at the flags and, nearly always, leave)
        const FrontSplit clear = pp->split(s.level, 2, w.redo.p);
        launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p, w.wake.p,
                     nullptr, nullptr, es, nullptr, nullptr, nullptr, w.rowflags.p, pp->fused_dec3(), pp->front_rows, &clear);
    }
    (void)hipEventRecord(s.ev[EV_EDGE], es);
    if (points) launch_points(w.g, n, w.rowplane.p, w.rowcount.p, w.nz.p, w.nz_cap, w.npoints.p, w.r_status, es);
    (void)hipEventRecord(s.ev[EV_POINTS], es);
    // The full-vote accumulator depends only on the point list.  With segments wanted, the HoughLinesP kernel
    // produces it on the side (its helper wavefronts, from the same LDS-resident set-up) and the vote kernel
    // only runs for the frames that kernel flags; it sits between the two HoughLinesP steps because the
    // general one permutes the point list.  Stage times: VOTE = that flagged launch, LINES = both steps.
    const bool vote_fused = p.want_accumulator && p.want_segments && pp->fuse_vote;
    if (p.want_accumulator && !vote_fused)
        launch_vote(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_full.p, nullptr, es);
    if (!vote_fused) (void)hipEventRecord(s.ev[EV_VOTE], es);
    if (p.want_segments && pp->group_size > 1) {
        // grouped slots: the HoughLinesP launch waits for the group's last slot (or for a wait on one of its slots).
        // Frames of the slot's capacity beyond this batch must read as empty to that launch.
        if (n < p.max_batch) (void)hipMemsetAsync(w.npoints.p + n, 0, sizeof(int) * (size_t)(p.max_batch - n), es);
        lc_pipe::Group &G = pp->groups[s.group];
        G.pending.push_back((int)(&s - &pp->slots[0]));
        s.lines_pending = true;
        const int first = s.group * pp->group_size, gs = std::min(pp->group_size, (int)pp->slots.size() - first);
        if ((int)G.pending.size() >= gs) flush_group(pp, s.group);
        return;
    }
    if (p.want_segments && pp->st_lines != es && pp->defer_after && pp->slots.size() > 1) pp->deferred = (int)(&s - &pp->slots[0]);
    else enqueue_lines(pp, s, nullptr);
}

// One HoughLinesP launch for the slots of a group whose points are queued: their frames lie back to back in the
// group's buffers (slot j at frame j * max_batch), the launch covers them all.
void flush_group(lc_pipe *pp, int g) {
    lc_pipe::Group &G = pp->groups[g];
    if (G.pending.empty()) return;
    const lc_pipe_params &p = pp->p;
    const int first = g * pp->group_size, m = (int)G.pending.size();
    bool prefix = true;
    for (int i = 0; i < m; i++) prefix = prefix && G.pending[i] == first + i;
    if (!prefix) {      // slots used out of order: a launch each
        for (int k : G.pending) { pp->slots[k].lines_pending = false; enqueue_lines(pp, pp->slots[k], nullptr); }
        G.pending.clear();
        return;
    }
    Slot &lead = pp->slots[first];
    Work &w = lead.w;
    const int n_total = (m - 1) * p.max_batch + pp->slots[first + m - 1].nframes;
    const bool vote_fused = p.want_accumulator && p.want_segments && pp->fuse_vote;
    const unsigned turn = pp->lines_turn++ % (unsigned)(1 + pp->st_lines_more.size());
    hipStream_t ls = turn ? pp->st_lines_more[turn - 1] : pp->st_lines;
    for (int i = 0; i < m; i++) {
        Slot &s = pp->slots[first + i];
        if (ls != pp->st_post) (void)hipStreamWaitEvent(ls, s.ev[(p.want_accumulator && !vote_fused) ? EV_VOTE : EV_POINTS], 0);
    }
    for (int k = 0; k < 8; k++) G.h_gstatus[k] = 0;     // (every slot of the group has been waited for since the last launch)
    for (int i = 0; i < m; i++) (void)hipEventRecord(pp->slots[first + i].ev[EV_LSTART], ls);
    const bool fast = launch_ppht_fast(w.g, n_total, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho,
                                       p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, G.d_gstatus,
                                       w.fallback.p, w.rng_out.p, RNG_TABLE, vote_fused ? w.acc_full.p : nullptr, ls,
                                       (pp->ppht_two_per_cu && pp->ppht_full_for == 0) ? ppht_lds_half() : 0, false,
                                       2 * n_total > pp->cu_count ? 0 : 8);
    if (pp->ppht_full_for > 0) pp->ppht_full_for--;
    launch_ppht_general(w.g, n_total, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_ppht.p,
                        p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, G.d_gstatus,
                        fast ? w.fallback.p : nullptr, ls, vote_fused ? w.acc_full.p : nullptr);
    for (int i = 0; i < m; i++) {
        Slot &s = pp->slots[first + i];
        (void)hipEventRecord(s.ev[EV_LINES], ls);
        s.end_ev = EV_LINES;
        s.lines_pending = false;
    }
    G.pending.clear();
}

// HoughLinesP step and the copy of the batch's results; `after` (may be null) is a later batch's event to start behind
void enqueue_lines(lc_pipe *pp, Slot &s, hipEvent_t after) {
    Work &w = s.w;
    const lc_pipe_params &p = pp->p;
    const int n = s.nframes;
    hipStream_t es = pp->st_post;
    const bool vote_fused = p.want_accumulator && p.want_segments && pp->fuse_vote;
    hipStream_t ls = es;   // the stream the batch finishes on
    if (p.want_segments) {
        const unsigned turn = pp->lines_turn++ % (unsigned)(1 + pp->st_lines_more.size());
        ls = turn ? pp->st_lines_more[turn - 1] : pp->st_lines;
        // (a stand-alone vote kernel reads the point list the general kernel permutes: behind it, then)
        if (ls != es) (void)hipStreamWaitEvent(ls, s.ev[(p.want_accumulator && !vote_fused) ? EV_VOTE : EV_POINTS], 0);
        if (ls != es && after) (void)hipStreamWaitEvent(ls, after, 0);
        (void)hipEventRecord(s.ev[EV_LSTART], ls);
        const bool fast = launch_ppht_fast(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho,
                                           p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status,
                                           w.fallback.p, w.rng_out.p, RNG_TABLE, vote_fused ? w.acc_full.p : nullptr, ls,
                                           (pp->ppht_two_per_cu && pp->ppht_full_for == 0) ? ppht_lds_half() : 0, false,
                                           2 * p.max_batch > pp->cu_count ? 0 : 8);
        if (pp->ppht_full_for > 0) pp->ppht_full_for--;
        // one launch for the frames the fast kernel flagged: their full-vote accumulator first (the general kernel
        // permutes the point list), then HoughLinesP with all state in global memory.  No events in between: every
        // packet on this stream is on the critical path.
        launch_ppht_general(w.g, n, w.nz.p, w.nz_cap, w.npoints.p, w.rowplane.p, w.tab.p, w.hg.numangle, w.hg.numrho, w.acc_ppht.p,
                            p.threshold, pp->line_length, pp->line_gap, w.r_seg, w.seg_cap, w.r_segcount, w.r_status,
                            fast ? w.fallback.p : nullptr, ls, vote_fused ? w.acc_full.p : nullptr);
        (void)hipEventRecord(s.ev[EV_LINES], ls);
    } else {
        (void)hipEventRecord(s.ev[EV_LSTART], ls);
        (void)hipEventRecord(s.ev[EV_LINES], ls);
    }
    s.end_ev = EV_LINES;
    if (!s.zero_copy) {   // one copy: counts, rounds, status and (if wanted) the segments of the n frames
        const size_t res_ints = w.res_head + (p.want_segments ? (size_t)n * w.seg_cap * 4 : 0);
        (void)hipMemcpyAsync(s.h_res, w.resblock.p, sizeof(int32_t) * res_ints, hipMemcpyDeviceToHost, ls);
        (void)hipEventRecord(s.ev[EV_END], ls);
        s.end_ev = EV_END;
    }
}

// from_host: the frames were just queued for copy on st_up (EV_START is recorded there)
int enqueue_all(lc_pipe *pp, Slot &s, bool from_host) {
    Work &w = s.w;
    const lc_pipe_params &p = pp->p;
    const int n = s.nframes;
    hipStream_t es = pp->st_edge;
    if (from_host) {
        HIP_TRY(hipEventRecord(s.ev[EV_UPLOADED], pp->st_up));
        HIP_TRY(hipStreamWaitEvent(es, s.ev[EV_UPLOADED], 0));
    } else {
        HIP_TRY(hipEventRecord(s.ev[EV_START], es));
        HIP_TRY(hipEventRecord(s.ev[EV_UPLOADED], es));
    }
    HIP_TRY(hipEventRecord(s.ev[EV_ESTART], es));
    // the exact reduction by three needs no pass of its own: the front kernel reads the source pixels it keeps
    const bool fused_resize = pp->fused_dec3();
    if (pp->resizes && !fused_resize) {
        launch_resize_linear(s.input, pp->src_w, pp->src_h, 3, n, w.bgr.p, w.g.W, w.g.H, pp->resize_tab.p, pp->resize_area2, es, pp->resize_dec3);
        s.input = w.bgr.p;
        s.input_bytes = (size_t)n * 3 * w.npix;
    }
    if (pp->level > 0 && pp->nlevels > 0) {
        if (pp->level_hold > 0) pp->level_hold--;
        else { pp->level--; pp->level_hold = 0; }     // try one level in again
    }
    s.level = pp->nlevels > 0 ? pp->level : 0;
    const bool skipping = s.level < pp->nlevels;
    const FrontSplit skip = pp->split(s.level, 1, nullptr), again = pp->split(pp->nlevels, 0, w.redo.p);
    if (pp->nlevels > 0) {
        if (s.level < s.planes_level) {      // back at a tighter level: the plane rows of the units it does not need go first
            const FrontSplit clear_all = pp->split(s.level, 2, pp->ones.p);
            launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p,
                         w.wake.p, nullptr, nullptr, es, nullptr, nullptr, nullptr, w.rowflags.p, pp->fused_dec3(), pp->front_rows,
                         &clear_all);
        }
        s.planes_level = s.level;
    }
    launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p,
                 w.wake.p, w.inter ? w.gray.p : nullptr, (w.inter && p.blur) ? w.blur.p : nullptr, es,
                 pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p, w.rowflags.p, fused_resize, pp->front_rows,
                 pp->nlevels > 0 ? &skip : nullptr);
    HIP_TRY(hipEventRecord(s.ev[EV_FRONT], es));
    if (pp->deferred >= 0 && pp->defer_after == 1) { enqueue_lines(pp, pp->slots[pp->deferred], s.ev[EV_FRONT]); pp->deferred = -1; }
    hipStream_t ps = pp->st_post;
    if (ps != es) HIP_TRY(hipStreamWaitEvent(ps, s.ev[EV_FRONT], 0));
    launch_hysteresis(w.g, n, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, ps, w.r_status);
    if (skipping) {
        // frames for which the units near the ROI alone do not settle its edges: all units again, and their hysteresis
        // (both launches leave at once for the other frames, which is nearly always all of them)
        const lc_pipe::SkipLevel &L = pp->levels[s.level];
        launch_ring_check(w.g, n, w.planeS.p, w.planeW.p, L.ring_words.p, L.n_ring_words, L.ring_code.p, w.redo.p, w.r_status, ps,
                          w.r_rounds);
        launch_front(s.input, s.input_bytes, 1, p.color_code, p.blur != 0, w.g, n, pp->low, pp->high, w.planeS.p, w.planeW.p,
                     w.wake.p, nullptr, nullptr, ps, pp->has_gate ? &pp->gate : nullptr, pp->gate_tab.p, w.gray.p, w.rowflags.p,
                     fused_resize, pp->front_rows, &again);
        launch_hysteresis(w.g, n, w.planeS.p, w.planeW.p, w.wake.p, w.r_rounds, ps, nullptr, w.redo.p);
    }
    HIP_TRY(hipEventRecord(s.ev[EV_HYST], ps));
    if (pp->deferred >= 0) { enqueue_lines(pp, pp->slots[pp->deferred], s.ev[EV_HYST]); pp->deferred = -1; }
    enqueue_back(pp, s);
    HIP_TRY(hipGetLastError());
    s.busy = true;
    s.timed = true;
    return LC_OK;
}

// The copy stream, created by the first host submit (a pipeline fed device frames never owns it).  Copies run on the
// DMA engines, but their packets still queue behind whatever shares the stream's hardware queue: picked like the rest.
int upload_stream(lc_pipe *pipe) {
    if (pipe->st_up) return LC_OK;
    for (Slot &s : pipe->slots)
        if (s.busy) { HIP_TRY(hipStreamCreateWithFlags(&pipe->st_up, hipStreamNonBlocking)); return LC_OK; }   // no experiments beside work in flight
    StreamPicker picker;
    std::vector<hipStream_t> mine = {pipe->st_edge};
    if (pipe->st_post != pipe->st_edge) mine.push_back(pipe->st_post);
    if (pipe->st_lines != pipe->st_edge) mine.push_back(pipe->st_lines);
    for (hipStream_t st : pipe->st_lines_more) mine.push_back(st);
    bool own = true;
    pipe->st_up = picker.pick(mine, &own);
    if (!pipe->st_up) return fail(LC_EHIP, "hipStreamCreate failed for the upload stream");
    if (!own) pipe->shared_queues++;
    return LC_OK;
}

int get_slot(lc_pipe *pipe, int slot, Slot **out) {
    if (!pipe) return fail(LC_EINVAL, "null pipeline");
    if (slot < 0 || slot >= (int)pipe->slots.size()) return fail(LC_EINVAL, "slot %d out of range", slot);
    HIP_TRY(hipSetDevice(pipe->ctx->device));
    *out = &pipe->slots[slot];
    return LC_OK;
}

}  // namespace

extern "C" {

int lc_pipe_create(lc_ctx *ctx, const lc_pipe_params *params, lc_pipe **out) {
    int rc;
    if (!out || !params) return fail(LC_EINVAL, "lc_pipe_create: null argument");
    *out = nullptr;
    if ((rc = ctx_bind(ctx))) return rc;
    const lc_pipe_params &p = *params;
    if (p.width <= 0 || p.height <= 0 || p.max_batch <= 0 || p.nslots <= 0) return fail(LC_EINVAL, "lc_pipe_create: bad sizes");
    if (p.width > MAX_DIM || p.height > MAX_DIM) return fail(LC_EUNSUPPORTED, "frames above %d pixels a side are not supported", MAX_DIM);
    if (p.color_code != LC_COLOR_BGR2GRAY && p.color_code != LC_COLOR_RGB2GRAY)
        return fail(LC_EUNSUPPORTED, "color_code must be COLOR_BGR2GRAY (6) or COLOR_RGB2GRAY (7)");
    if (!(p.canny_lo == p.canny_lo) || !(p.canny_hi == p.canny_hi)) return fail(LC_EINVAL, "NaN Canny threshold");
    const bool hough = p.want_accumulator || p.want_segments;
    if (p.want_segments && p.seg_capacity <= 0) return fail(LC_EINVAL, "seg_capacity must be positive");

    lc_pipe *pp = new lc_pipe();
    pp->ctx = ctx;
    pp->p = p;
    pp->p.roi_mask = pp->p.car_mask = nullptr;   // host pointers are not kept
    pp->g = make_geo(p.width, p.height);
    double t1 = std::min(p.canny_lo, p.canny_hi), t2 = std::max(p.canny_lo, p.canny_hi);
    t1 = std::max(-1e9, std::min(1e9, t1));
    t2 = std::max(-1e9, std::min(1e9, t2));
    pp->low = (int)std::floor(t1);
    pp->high = (int)std::floor(t2);
    pp->line_length = (int)lrint(p.min_line_length);
    pp->line_gap = (int)lrint(p.max_line_gap);
    if (const char *fv = getenv("LANECV_VOTE_FUSED")) pp->fuse_vote = fv[0] != '0';
    pp->ppht_two_per_cu = p.want_segments && (size_t)p.width * p.height <= 2600000 && p.nslots >= 3;
    if (const char *tp = getenv("LANECV_PPHT_TWO_PER_CU")) pp->ppht_two_per_cu = tp[0] != '0';   // diagnostics

    pp->src_w = p.src_width > 0 ? p.src_width : p.width;
    pp->src_h = p.src_height > 0 ? p.src_height : p.height;
    if (pp->src_w > MAX_DIM || pp->src_h > MAX_DIM) { delete pp; return fail(LC_EUNSUPPORTED, "source frames above %d pixels a side are not supported", MAX_DIM); }
    pp->resizes = pp->src_w != p.width || pp->src_h != p.height;
    pp->src_frame_bytes = (size_t)3 * pp->src_w * pp->src_h;
    if (pp->resizes) {
        const size_t tn = 2 * ((size_t)p.width + p.height);
        std::vector<int> tab(tn);
        pp->resize_area2 = resize_tables_host(pp->src_w, pp->src_h, p.width, p.height, tab.data(), tab.data() + p.width,
                                              tab.data() + 2 * p.width, tab.data() + 2 * p.width + p.height);
        pp->resize_dec3 = !pp->resize_area2 && resize_is_decimate3(pp->src_w, pp->src_h, p.width, p.height, tab.data(), tab.data() + p.width,
                                                                   tab.data() + 2 * p.width, tab.data() + 2 * p.width + p.height);
        if ((rc = pp->resize_tab.alloc(tn))) { delete pp; return rc; }
        hipError_t e = hipMemcpy(pp->resize_tab.p, tab.data(), sizeof(int) * tn, hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete pp; return fail(LC_EHIP, "resize table upload failed: %s", hipGetErrorString(e)); }
    }
    if (p.hsv_gate) {
        pp->has_gate = true;
        pp->gate = make_gate(p.hsv_bounds, 2);
        std::vector<int> tab(768);
        gate_tables_host(pp->gate, tab.data());
        if ((rc = pp->gate_tab.alloc(tab.size()))) { delete pp; return rc; }
        hipError_t e = hipMemcpy(pp->gate_tab.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete pp; return fail(LC_EHIP, "gate table upload failed: %s", hipGetErrorString(e)); }
    }

    // ROI and car mask fold into one bit plane: keep = (roi == 255) && (car != 0)
    std::vector<uint64_t> bits;
    if (p.roi_mask || p.car_mask) {
        const Geo g = pp->g;
        bits.assign(g.plane_words(), 0);
        int roi_first = -1;
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                size_t i = (size_t)y * g.W + x;
                bool keep = true;
                if (p.roi_mask) {
                    uint8_t v = p.roi_mask[i];
                    if (v != 0 && v != 255) {
                        delete pp;
                        return fail(LC_EUNSUPPORTED, "roi_mask must hold only 0 and 255 (found %d at %d,%d)", v, x, y);
                    }
                    keep = v != 0;
                }
                if (keep && p.car_mask) keep = p.car_mask[i] != 0;
                if (keep) {
                    bits[(size_t)y * g.TX + (x >> 6)] |= 1ull << (x & 63);
                    if (roi_first < 0) roi_first = y;
                }
            }
        pp->roi_first_row = roi_first < 0 ? g.H : roi_first;
        if ((rc = pp->roi_bits.alloc(bits.size()))) { delete pp; return rc; }
        hipError_t e = hipMemcpy(pp->roi_bits.p, bits.data(), bits.size() * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete pp; return fail(LC_EHIP, "roi upload failed: %s", hipGetErrorString(e)); }
        pp->has_roi = true;
    }

    {
        static const char *ns = getenv("LANECV_NO_ROW_SKIP");
        const Geo g = pp->g;
        if (pp->has_roi && !p.keep_intermediates && front_stream_supported(g) && !getenv("LANECV_FRONT_TILED") && !ns) {
            // rings 1..15 around the ROI by fifteen 3x3 dilations of its bit plane
            constexpr int NRING = 15;
            const int TX = g.TX, H = g.H;
            const uint64_t last_mask = (g.W & 63) ? ((1ull << (g.W & 63)) - 1) : ~0ull;
            auto dilate = [&](const std::vector<uint64_t> &in) {
                std::vector<uint64_t> hz(in.size()), out(in.size());
                for (int y = 0; y < H; y++)
                    for (int tx = 0; tx < TX; tx++) {
                        const uint64_t c = in[(size_t)y * TX + tx];
                        const uint64_t l = tx > 0 ? in[(size_t)y * TX + tx - 1] >> 63 : 0, r = tx + 1 < TX ? in[(size_t)y * TX + tx + 1] << 63 : 0;
                        uint64_t v = c | (c << 1) | (c >> 1) | l | r;
                        if (tx == TX - 1) v &= last_mask;
                        hz[(size_t)y * TX + tx] = v;
                    }
                for (int y = 0; y < H; y++)
                    for (int tx = 0; tx < TX; tx++) {
                        uint64_t v = hz[(size_t)y * TX + tx];
                        if (y > 0) v |= hz[(size_t)(y - 1) * TX + tx];
                        if (y + 1 < H) v |= hz[(size_t)(y + 1) * TX + tx];
                        out[(size_t)y * TX + tx] = v;
                    }
                return out;
            };
            std::vector<uint64_t> cur = bits, code(4 * bits.size(), 0);
            for (int k = 1; k <= NRING; k++) {
                std::vector<uint64_t> nxt = dilate(cur);
                for (size_t i = 0; i < bits.size(); i++) {
                    const uint64_t ring = nxt[i] & ~cur[i];
                    for (int b = 0; b < 4; b++)
                        if (k & (1 << b)) code[b * bits.size() + i] |= ring;
                }
                cur.swap(nxt);
            }
            int first = -1;
            for (int y = 0; y < H && first < 0; y++)
                for (int tx = 0; tx < TX; tx++)
                    if (cur[(size_t)y * TX + tx]) { first = y; break; }
            if (first >= 0) {
                pp->row_split = first >= 64 ? first : 0;
                pp->front_rows = front_stream_rows_per_unit(g, p.max_batch, H - pp->row_split);
                const FrontGrid fg = front_stream_grid(g, pp->front_rows, pp->row_split);
                // uploads one level: the units a region `reach` (bit plane) touches, and the ring code planes around it
                auto add_level = [&](const std::vector<uint64_t> &reach, const std::vector<uint64_t> &rcode) -> int {
                    std::vector<uint16_t> needed, unneeded;
                    long long npix = 0;
                    for (int sg = 0; sg < fg.nseg_top + fg.nseg_low; sg++)
                        for (int st = 0; st < fg.nstrips; st++) {
                            bool need = false;
                            if (sg >= fg.nseg_top) {
                                const int y0 = pp->row_split + (sg - fg.nseg_top) * fg.rows_low, y1 = std::min(y0 + fg.rows_low, H);
                                const int x0 = st * fg.strip_cols, x1 = std::min(x0 + fg.strip_cols, g.W);
                                for (int y = y0; y < y1 && !need; y++)
                                    for (int tx = x0 >> 6; tx <= (x1 - 1) >> 6 && tx < TX && !need; tx++) {
                                        uint64_t v = reach[(size_t)y * TX + tx];
                                        const int lo = std::max(x0 - tx * 64, 0), hi = std::min(x1 - tx * 64, 64);   // bits [lo, hi) of the word
                                        if (lo > 0) v &= ~((1ull << lo) - 1);
                                        if (hi < 64) v &= (1ull << hi) - 1;
                                        need = v != 0;
                                    }
                                if (need) npix += (long long)(y1 - y0) * (x1 - x0);
                            }
                            (need ? needed : unneeded).push_back((uint16_t)(sg | (st << 8)));
                        }
                    if (needed.empty() || unneeded.empty()) return LC_OK;          // nothing to skip at this level
                    for (int lv = 0; lv < pp->nlevels; lv++)
                        if (pp->levels[lv].n_needed == (int)needed.size()) return LC_OK;   // no looser than the level before it
                    std::vector<uint32_t> rw;
                    const size_t nw = bits.size();
                    for (size_t i = 0; i < nw; i++)
                        if (rcode[i] | rcode[nw + i] | rcode[2 * nw + i] | rcode[3 * nw + i]) rw.push_back((uint32_t)i);
                    lc_pipe::SkipLevel &L = pp->levels[pp->nlevels];
                    int rc2;
                    if ((rc2 = L.needed.alloc(needed.size())) || (rc2 = L.unneeded.alloc(unneeded.size())) ||
                        (rc2 = L.ring_words.alloc(rw.size())) || (rc2 = L.ring_code.alloc(rcode.size()))) return rc2;
                    hipError_t e = hipMemcpy(L.needed.p, needed.data(), needed.size() * 2, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(L.unneeded.p, unneeded.data(), unneeded.size() * 2, hipMemcpyHostToDevice);
                    if (e == hipSuccess && !rw.empty()) e = hipMemcpy(L.ring_words.p, rw.data(), rw.size() * 4, hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(L.ring_code.p, rcode.data(), rcode.size() * 8, hipMemcpyHostToDevice);
                    if (e != hipSuccess) return fail(LC_EHIP, "unit list upload failed: %s", hipGetErrorString(e));
                    L.n_needed = (int)needed.size(); L.n_unneeded = (int)unneeded.size(); L.n_ring_words = (int)rw.size();
                    L.pixels = npix;
                    pp->nlevels++;
                    return LC_OK;
                };
                if (fg.nseg_top + fg.nseg_low < 256 && fg.nstrips < 256) {
                    if ((rc = add_level(cur, code))) { delete pp; return rc; }
                    if (pp->row_split > 0) {
                        // the band of rows the ROI spans as the region: ring k = row (first ROI row) - k, all columns
                        std::vector<uint64_t> band(bits.size(), 0), bcode(4 * bits.size(), 0);
                        for (int y = pp->row_split; y < H; y++)
                            for (int tx = 0; tx < TX; tx++) band[(size_t)y * TX + tx] = tx == TX - 1 ? last_mask : ~0ull;
                        for (int k = 1; k <= NRING; k++) {
                            const int y = pp->roi_first_row - k;
                            if (y < 0) break;
                            for (int tx = 0; tx < TX; tx++)
                                for (int b = 0; b < 4; b++)
                                    if (k & (1 << b)) bcode[b * bits.size() + (size_t)y * TX + tx] = tx == TX - 1 ? last_mask : ~0ull;
                        }
                        if ((rc = add_level(band, bcode))) { delete pp; return rc; }
                    }
                }
                if (pp->nlevels > 0) {
                    std::vector<int> one(p.max_batch, 1);
                    if ((rc = pp->ones.alloc(one.size()))) { delete pp; return rc; }
                    if (hipMemcpy(pp->ones.p, one.data(), one.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                        delete pp;
                        return fail(LC_EHIP, "flag upload failed");
                    }
                } else {
                    pp->row_split = 0;
                }
            }
        }
        if (pp->nlevels == 0) pp->front_rows = front_stream_rows_per_unit(g, p.max_batch, g.H);
        pp->frame_pixels = (long long)g.W * g.H;
    }

    {
        // All streams have the same priority.  With the post and HoughLinesP streams at high priority their few
        // workgroups were placed ahead of a front kernel's thousands a little sooner (327 k against 320 k frames/s at
        // 1080p x 256), but both streams spend most of their time with a wait for another stream at the head of their
        // queue, and whenever the process owned more hardware queues than this pipeline's (RCCL initialised, a second
        // pipeline process on the GPU) the front kernel's queue was never served again: a hang at the first batch.
        // st_up is created by the first host submit: a pipeline that is only ever handed device frames keeps its
        // stream count (and with it the number of hardware queues it wants) at three
        //
        // HIP folds a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and kernels of
        // streams that share one run one after the other -- which undoes the overlap this layout exists for (198 k
        // instead of 327 k frames/s at 1080p x 256 when the post stream shared a queue).  Which streams share is
        // HIP's business and depends on what else the process created, so every stream below is picked by
        // experiment (StreamPicker): candidates are created until one runs side by side with the streams picked so
        // far.  In the order of what the overlap is worth: HoughLinesP beside the front kernels, then the short
        // kernels, then further HoughLinesP streams for small batches; a stream for which no queue of its own is
        // left falls back to the stream it would otherwise have run beside (counted in `shared_queues`).
        StreamPicker picker;
        std::vector<hipStream_t> mine;
        bool own = true;
        hipError_t e = hipStreamCreateWithFlags(&pp->st_edge, hipStreamNonBlocking);
        if (e == hipSuccess) mine.push_back(pp->st_edge);
        const char *da = getenv("LANECV_LINES_DEFER");      // diagnostics: 0 / 1 / 2, see defer_after
        if (da && da[0] >= '0' && da[0] <= '2') pp->defer_after = da[0] - '0';
        const char *pr = getenv("LANECV_LINES_PRIORITY");   // diagnostics: 0 keeps HoughLinesP on the edge stream
        if (e == hipSuccess && p.want_segments && !(pr && pr[0] == '0')) {
            pp->st_lines = picker.pick(mine, &own);
            if (!pp->st_lines) e = hipErrorOutOfMemory;
            else { mine.push_back(pp->st_lines); pp->shared_queues += own ? 0 : 1; }
        }
        if (e != hipSuccess) { lc_pipe_destroy(pp); return fail(LC_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
        const bool own_lines = pp->st_lines != nullptr;
        if (!pp->st_lines) pp->st_lines = pp->st_edge;
        // the short kernels get a stream of their own when batches are pipelined: the next batch's
        // front kernel then starts at once and they run beside it (LANECV_POST_STREAM=0: keep them on st_edge)
        pp->st_post = pp->st_edge;
        const char *po = getenv("LANECV_POST_STREAM");
        if (p.nslots > 1 && !(po && po[0] == '0')) {
            hipStream_t st = picker.pick(mine, &own);
            if (st && own) { pp->st_post = st; mine.push_back(st); }
            else {                                  // no queue of its own: behind the front kernels, as with one slot
                if (st) picker.rejected.push_back(st);
                pp->shared_queues++;
            }
        }
        if (own_lines) {
            hipDeviceProp_t prop;
            int want = 1;
            if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) pp->cu_count = prop.multiProcessorCount;
            if (pp->cu_count > 0)
                want = std::max(1, std::min(std::min((pp->ppht_two_per_cu ? 2 : 1) * pp->cu_count / p.max_batch, p.nslots - 1), 3));
            const char *ns = getenv("LANECV_LINES_STREAMS");
            if (ns && ns[0] >= '1' && ns[0] <= '3') want = ns[0] - '0';   // more than three hung lc_pipe_wait (queue sharing)
            for (int k = 1; k < want; k++) {
                hipStream_t st = picker.pick(mine, &own);
                if (st && own) { pp->st_lines_more.push_back(st); mine.push_back(st); }
                else { if (st) picker.rejected.push_back(st); pp->shared_queues++; break; }
            }
        }
        // with the post stream HoughLinesP starts as soon as its points are there (they are late enough: the short
        // kernels wait for room beside the next front kernel); deferring it behind the next front kernel on top
        // only ties it to the host's next submit
        if (pp->st_post != pp->st_edge && !(da && da[0] >= '0' && da[0] <= '2')) pp->defer_after = 0;
    }
    pp->slots.resize(p.nslots);
    for (Slot &s : pp->slots) {
        hipError_t e;
        for (auto &ev : s.ev) {
            e = hipEventCreate(&ev);
            if (e != hipSuccess) { lc_pipe_destroy(pp); return fail(LC_EHIP, "hipEventCreate failed: %s", hipGetErrorString(e)); }
        }
        if ((rc = s.w.init_images(p.width, p.height, p.max_batch, true, p.keep_intermediates != 0, true))) { lc_pipe_destroy(pp); return rc; }
        if (pp->resizes && (rc = s.w.src_in.alloc((size_t)p.max_batch * pp->src_frame_bytes))) { lc_pipe_destroy(pp); return rc; }
        // frames the streaming front kernel cannot take pass through a gated gray image
        if (pp->has_gate && !s.w.gray.p && !front_stream_supported(pp->g) &&
            (rc = s.w.gray.alloc((size_t)p.max_batch * p.width * p.height))) { lc_pipe_destroy(pp); return rc; }
        if (hough || p.want_points) {
            // point list capacity per frame: an eighth of the pixels (a Canny map inside an ROI stays far below)
            int cap = std::max(1024, (int)std::min<size_t>((size_t)p.width * p.height, (size_t)p.width * p.height / 8 + 1024));
            if ((rc = s.w.init_points(cap))) { lc_pipe_destroy(pp); return rc; }
        }
        if (hough) {
            if ((rc = s.w.init_hough(p.rho, p.theta, std::max(p.seg_capacity, 1), p.want_accumulator != 0, p.want_segments != 0, pp->st_edge))) {
                lc_pipe_destroy(pp);
                return rc;
            }
        }
        const int segcap = std::max(p.seg_capacity, 1);
        s.w.seg_cap = segcap;
        if ((rc = s.w.init_results(segcap))) { lc_pipe_destroy(pp); return rc; }
        void *hp = nullptr;
        const size_t res_ints = s.w.res_head + (size_t)p.max_batch * segcap * 4;
        // mapped and coherent: the kernels store a batch's results into it directly (see below), and the host reads
        // them after the batch's last event
        if (hipHostMalloc(&hp, sizeof(int32_t) * res_ints, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            hp = nullptr;
            (void)hipGetLastError();
            if ((rc = lc_host_alloc_pinned(sizeof(int32_t) * res_ints, &hp))) { lc_pipe_destroy(pp); return rc; }
        }
        s.h_res = (int32_t *)hp;
        memset(s.h_res, 0, sizeof(int32_t) * s.w.res_head);
        s.h_segcount = s.h_res; s.h_rounds = s.h_res + p.max_batch; s.h_status = s.h_res + 2 * (size_t)p.max_batch;
        s.h_seg = s.h_res + s.w.res_head;
        // What goes back per batch is a few hundred bytes per frame, written once: the kernels store it into the
        // pinned block directly, which takes a copy and its launch off the HoughLinesP stream
        // (LANECV_RESULT_COPY=1 keeps the device block and the copy).
        const char *rc_env = getenv("LANECV_RESULT_COPY");
        void *dp = nullptr;
        if (!(rc_env && rc_env[0] == '1') && hipHostGetDevicePointer(&dp, s.h_res, 0) == hipSuccess && dp) {
            int32_t *d = (int32_t *)dp;
            s.w.r_segcount = d; s.w.r_rounds = d + p.max_batch; s.w.r_status = d + 2 * (size_t)p.max_batch; s.w.r_seg = d + s.w.res_head;
            s.zero_copy = true;
        }
    }
    for (Slot &s : pp->slots) s.h_gstatus = s.h_status;
    {
        // groups of slots with one HoughLinesP launch (see lc_pipe::Group): for batches that leave CUs free, when there
        // are slots enough to fill one group while another one's launch runs
        bool all_zero_copy = true;
        for (Slot &s : pp->slots) all_zero_copy = all_zero_copy && s.zero_copy;
        const char *ge = getenv("LANECV_NO_GROUPS");
        int G = 1;
        if (p.want_segments && all_zero_copy && !ge && 2 * p.max_batch <= pp->cu_count && p.nslots >= 4)
            G = std::max(1, std::min(p.nslots / 2, pp->cu_count / p.max_batch));
        if (G > 1) {
            pp->group_size = G;
            const int B = p.max_batch, segcap = std::max(p.seg_capacity, 1);
            const size_t words = pp->g.rowplane_words();
            pp->groups.resize((p.nslots + G - 1) / G);
            for (size_t gi = 0; gi < pp->groups.size(); gi++) {
                lc_pipe::Group &grp = pp->groups[gi];
                const int first = (int)gi * G, gs = std::min(G, p.nslots - first);
                Work &w0 = pp->slots[first].w;
                const size_t accn = (size_t)w0.hg.numangle * w0.hg.numrho, cap = (size_t)w0.nz_cap, nfr = (size_t)gs * B;
                if ((rc = grp.nz.alloc(nfr * cap)) || (rc = grp.npoints.alloc(nfr)) || (rc = grp.fallback.alloc(nfr)) ||
                    (rc = grp.rowplane.alloc(nfr * words)) || (rc = grp.acc_ppht.alloc(nfr * accn)) ||
                    (p.want_accumulator && (rc = grp.acc_full.alloc(nfr * accn)))) { lc_pipe_destroy(pp); return rc; }
                HIP_TRY(hipMemset(grp.rowplane.p, 0, sizeof(uint64_t) * nfr * words));
                HIP_TRY(hipMemset(grp.npoints.p, 0, sizeof(int) * nfr));
                const size_t head = 2 * nfr + 8 * (size_t)gs + 8, ints = head + nfr * segcap * 4;
                void *hp = nullptr, *dp = nullptr;
                if (hipHostMalloc(&hp, sizeof(int32_t) * ints, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                    hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess || !dp) {
                    lc_pipe_destroy(pp);
                    return fail(LC_EHIP, "pinned result block of a slot group failed");
                }
                grp.h_res = (int32_t *)hp; grp.d_res = (int32_t *)dp;
                memset(grp.h_res, 0, sizeof(int32_t) * head);
                grp.h_gstatus = grp.h_res + 2 * nfr + 8 * (size_t)gs;
                grp.d_gstatus = grp.d_res + 2 * nfr + 8 * (size_t)gs;
                for (int j = 0; j < gs; j++) {
                    Slot &s = pp->slots[first + j];
                    Work &w = s.w;
                    const size_t fo = (size_t)j * B;      // the slot's first frame in the group's buffers
                    w.nz.alias(grp.nz.p + fo * cap, (size_t)B * cap);
                    w.npoints.alias(grp.npoints.p + fo, B);
                    w.fallback.alias(grp.fallback.p + fo, B);
                    w.rowplane.alias(grp.rowplane.p + fo * words, (size_t)B * words);
                    w.acc_ppht.alias(grp.acc_ppht.p + fo * accn, (size_t)B * accn);
                    if (p.want_accumulator) w.acc_full.alias(grp.acc_full.p + fo * accn, (size_t)B * accn);
                    if (s.h_res) { (void)hipHostFree(s.h_res); s.h_res = nullptr; }
                    s.group = (int)gi; s.gpos = j;
                    s.h_segcount = grp.h_res + fo; s.h_rounds = grp.h_res + nfr + fo; s.h_status = grp.h_res + 2 * nfr + 8 * (size_t)j;
                    s.h_seg = grp.h_res + head + fo * segcap * 4;
                    s.h_gstatus = grp.h_gstatus;
                    w.r_segcount = grp.d_res + fo; w.r_rounds = grp.d_res + nfr + fo; w.r_status = grp.d_res + 2 * nfr + 8 * (size_t)j;
                    w.r_seg = grp.d_res + head + fo * segcap * 4;
                }
            }
        }
    }
    // the buffers were cleared with hipMemset, which may still be running on the null stream, and the pipeline's
    // streams do not wait for that stream: nothing may be submitted before the clears have landed
    {
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { lc_pipe_destroy(pp); return fail(LC_EHIP, "pipeline set-up failed: %s", hipGetErrorString(e)); }
    }
    *out = pp;
    return LC_OK;
}

int lc_pipe_destroy(lc_pipe *pipe) {
    if (!pipe) return LC_OK;
    (void)hipSetDevice(pipe->ctx->device);
    if (pipe->deferred >= 0 && pipe->st_lines) { enqueue_lines(pipe, pipe->slots[pipe->deferred], nullptr); pipe->deferred = -1; }
    if (pipe->st_lines) for (size_t g = 0; g < pipe->groups.size(); g++) flush_group(pipe, (int)g);
    if (pipe->st_up) (void)hipStreamSynchronize(pipe->st_up);
    if (pipe->st_edge) (void)hipStreamSynchronize(pipe->st_edge);
    if (pipe->st_lines) (void)hipStreamSynchronize(pipe->st_lines);
    for (hipStream_t st : pipe->st_lines_more) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (pipe->st_post && pipe->st_post != pipe->st_edge) { (void)hipStreamSynchronize(pipe->st_post); (void)hipStreamDestroy(pipe->st_post); }
    for (Slot &s : pipe->slots) {
        for (auto &ev : s.ev) if (ev) (void)hipEventDestroy(ev);
        if (s.h_res) (void)hipHostFree(s.h_res);
    }
    for (lc_pipe::Group &G : pipe->groups) {
        if (G.h_res) (void)hipHostFree(G.h_res);
    }
    if (pipe->st_lines && pipe->st_lines != pipe->st_edge) (void)hipStreamDestroy(pipe->st_lines);
    if (pipe->st_edge) (void)hipStreamDestroy(pipe->st_edge);
    if (pipe->st_up) (void)hipStreamDestroy(pipe->st_up);
    delete pipe;
    return LC_OK;
}

int lc_pipe_submit_host(lc_pipe *pipe, int slot, const uint8_t *host_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!host_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "submit: bad frames argument");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    s->nframes = nframes;
    s->input_bytes = (size_t)nframes * pipe->src_frame_bytes;
    uint8_t *own = pipe->resizes ? s->w.src_in.p : s->w.bgr.p;
    s->input = own;
    if ((rc = upload_stream(pipe))) return rc;
    HIP_TRY(hipEventRecord(s->ev[EV_START], pipe->st_up));
    HIP_TRY(hipMemcpyAsync(own, host_bgr, s->input_bytes, hipMemcpyHostToDevice, pipe->st_up));
    return enqueue_all(pipe, *s, true);
}

int lc_pipe_submit_device(lc_pipe *pipe, int slot, const void *dev_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!dev_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "submit: bad frames argument");
    if (((uintptr_t)dev_bgr & 15) != 0) return fail(LC_EINVAL, "submit: device frames must be 16-byte aligned");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    s->nframes = nframes;
    s->input_bytes = (size_t)nframes * pipe->src_frame_bytes;
    s->input = (const uint8_t *)dev_bgr;
    return enqueue_all(pipe, *s, false);
}

int lc_pipe_wait(lc_pipe *pipe, int slot, int32_t *seg_xyxy, int32_t *seg_count) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!s->busy) return fail(LC_EINVAL, "slot %d has nothing in flight", slot);
    if (pipe->deferred == slot) { enqueue_lines(pipe, *s, nullptr); pipe->deferred = -1; }
    if (s->lines_pending) flush_group(pipe, s->group);
    // The wait polls the batch's last event and gives up after LANECV_WAIT_TIMEOUT_S seconds (default 120; 0 = block
    // in hipEventSynchronize for ever) with a report of which of the pipeline's events have fired, per slot, in the
    // order of the EV_ enumeration: a pipeline whose streams starve each other (more streams in the process than
    // hardware queues, see lc_pipe_create) then fails loudly instead of hanging its caller.
    static const double wait_limit = getenv("LANECV_WAIT_TIMEOUT_S") ? atof(getenv("LANECV_WAIT_TIMEOUT_S")) : 120.0;
    if (wait_limit > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
            const hipError_t q = hipEventQuery(s->ev[s->end_ev]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIP_TRY(q);
            if ((spins & 63) != 63) continue;                     // the clock is read every 64th query
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > wait_limit) {
                std::string rep;
                for (size_t k = 0; k < pipe->slots.size(); k++) {
                    Slot &o = pipe->slots[k];
                    rep += " slot" + std::to_string(k) + (o.busy ? "*:" : ":");
                    for (auto &ev : o.ev) rep += hipEventQuery(ev) == hipSuccess ? '1' : '0';
                }
                (void)hipGetLastError();
                return fail(LC_EHIP, "wait: slot %d not finished after %.0f s; events fired (START UPLOADED FRONT HYST EDGE POINTS VOTE "
                            "LINES END ESTART LSTART VSTART):%s", slot, wait_limit, rep.c_str());
            }
            if (waited > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));   // a batch takes about a millisecond
        }
    } else {
        HIP_TRY(hipEventSynchronize(s->ev[s->end_ev]));
    }
    s->busy = false;
    const lc_pipe_params &p = pipe->p;
    if (p.want_segments && (s->h_status[2] | s->h_gstatus[2])) pipe->ppht_full_for = 64;   // a frame outgrew half a CU's LDS (see lc_pipe)
    if (s->level < pipe->nlevels && s->h_status[3]) {                   // frames that needed the full pass after all
        int redone = 0;
        for (int i = 0; i < s->nframes; i++) redone += (s->h_rounds[i] >> 30) & 1;
        const char *ob = getenv("LANECV_SKIP_OFF_BATCHES");         // diagnostics / tests; 0 = never move a level out
        const int off_batches = ob ? atoi(ob) : 256;
        if (16 * redone > s->nframes && off_batches > 0 && pipe->level <= s->level) {
            pipe->level = s->level + 1;
            pipe->level_hold = off_batches;
        }
    }
    if (p.want_segments || p.want_accumulator || p.want_points) {
        if (s->h_status[0] | s->h_gstatus[0])
            return fail(LC_ECAP, "a frame holds more edge points than the pipeline keeps per frame (%d): its point list, accumulator and "
                        "segments are incomplete", s->w.nz_cap);
        if (s->h_status[1] | s->h_gstatus[1])
            return fail(LC_EINVAL, "HoughLinesP: a vote fell outside the accumulator (the angle table does not fit the image size)");
    }
    if (p.want_segments) {
        if (seg_count) memcpy(seg_count, s->h_segcount, sizeof(int) * s->nframes);
        if (seg_xyxy) memcpy(seg_xyxy, s->h_seg, sizeof(int32_t) * 4 * (size_t)s->nframes * s->w.seg_cap);
    }
    return LC_OK;
}

int lc_pipe_read(lc_pipe *pipe, int slot, int what, void *host_dst, size_t bytes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (!host_dst) return fail(LC_EINVAL, "null destination");
    Work &w = s->w;
    const void *src = nullptr;
    size_t need = 0;
    const size_t img = (size_t)s->nframes * w.npix;
    switch (what) {
        case LC_BUF_GRAY: src = w.inter ? w.gray.p : nullptr; need = img; break;
        case LC_BUF_BLUR: src = (w.inter && pipe->p.blur) ? w.blur.p : nullptr; need = img; break;
        case LC_BUF_EDGES: src = w.inter ? w.edges.p : nullptr; need = img; break;
        case LC_BUF_MASKED: src = w.masked.p; need = img; break;
        case LC_BUF_ACC: src = pipe->p.want_accumulator ? w.acc_full.p : nullptr;
            need = sizeof(int32_t) * (size_t)s->nframes * w.hg.numangle * w.hg.numrho; break;
        case LC_BUF_NPOINTS: src = (pipe->p.want_accumulator || pipe->p.want_segments || pipe->p.want_points) ? w.npoints.p : nullptr;
            need = sizeof(int) * s->nframes; break;
        case LC_BUF_RESIZED:
            if (pipe->fused_dec3())
                return fail(LC_EINVAL, "the resized frames are not materialised when the reduction by three is fused into the edge "
                            "kernel; configure keep_intermediates to read them");
            src = pipe->resizes ? w.bgr.p : nullptr; need = 3 * img; break;
        default: return fail(LC_EINVAL, "unknown buffer id %d", what);
    }
    if (!src) return fail(LC_EINVAL, "buffer %d is not produced by this pipeline's configuration", what);
    if (bytes < need) return fail(LC_ECAP, "destination holds %zu bytes, %zu needed", bytes, need);
    HIP_TRY(hipMemcpy(host_dst, src, need, hipMemcpyDeviceToHost));
    return LC_OK;
}

int lc_pipe_read_points(lc_pipe *pipe, int slot, int frame, uint32_t *xy, int capacity, int *n) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d has not been waited for", slot);
    if (!n) return fail(LC_EINVAL, "null count pointer");
    if (!(pipe->p.want_accumulator || pipe->p.want_segments || pipe->p.want_points))
        return fail(LC_EINVAL, "this pipeline's configuration does not produce the point list");
    if (frame < 0 || frame >= s->nframes) return fail(LC_EINVAL, "frame %d out of range [0, %d)", frame, s->nframes);
    Work &w = s->w;
    int np = 0;
    HIP_TRY(hipMemcpy(&np, w.npoints.p + frame, sizeof(int), hipMemcpyDeviceToHost));
    *n = np;
    if (np > w.nz_cap) return fail(LC_ECAP, "frame %d has %d edge points, the slot keeps %d", frame, np, w.nz_cap);
    if (np > capacity) return fail(LC_ECAP, "frame %d has %d edge points, destination holds %d", frame, np, capacity);
    if (np > 0) {
        if (!xy) return fail(LC_EINVAL, "null destination");
        HIP_TRY(hipMemcpy(xy, w.nz.p + (size_t)frame * w.nz_cap, sizeof(uint32_t) * np, hipMemcpyDeviceToHost));
    }
    return LC_OK;
}

int lc_pipe_stage_ms(lc_pipe *pipe, int slot, int stage, float *ms) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!ms) return fail(LC_EINVAL, "null output");
    if (s->busy || !s->timed) return fail(LC_EINVAL, "slot %d has no complete timing", slot);
    int a, b;
    switch (stage) {
        case LC_STAGE_UPLOAD: a = EV_START; b = EV_UPLOADED; break;
        case LC_STAGE_EDGE: a = EV_ESTART; b = EV_EDGE; break;
        case LC_STAGE_POINTS: a = EV_EDGE; b = EV_POINTS; break;
        case LC_STAGE_VOTE:   // with segments wanted the accumulator comes out of the HoughLinesP kernel: counted under LINES
            if (pipe->p.want_accumulator && pipe->p.want_segments) { *ms = 0.f; return LC_OK; }
            a = EV_POINTS; b = EV_VOTE;
            break;
        case LC_STAGE_LINES: a = EV_LSTART; b = EV_LINES; break;
        case LC_STAGE_TOTAL: a = EV_START; b = s->end_ev; break;
        case LC_STAGE_FRONT: a = EV_ESTART; b = EV_FRONT; break;
        case LC_STAGE_HYST: a = EV_FRONT; b = EV_HYST; break;
        case LC_STAGE_FINAL: a = EV_HYST; b = EV_EDGE; break;
        default: return fail(LC_EINVAL, "unknown stage %d", stage);
    }
    HIP_TRY(hipEventElapsedTime(ms, s->ev[a], s->ev[b]));
    return LC_OK;
}

int lc_pipe_hyst_rounds(lc_pipe *pipe, int slot, int *rounds_max, int *rounds_total) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    int mx = 0, tot = 0;
    for (int i = 0; i < s->nframes; i++) { const int r = s->h_rounds[i] & 0x3FFFFFFF; mx = std::max(mx, r); tot += r; }
    if (rounds_max) *rounds_max = mx;
    if (rounds_total) *rounds_total = tot;
    return LC_OK;
}

int lc_pipe_roi_skip(lc_pipe *pipe, int slot, int *first_row, long long *pixels_classified, int *frames_redone) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight", slot);
    if (first_row) *first_row = pipe->row_split;
    if (pixels_classified) *pixels_classified = s->level < pipe->nlevels ? pipe->levels[s->level].pixels : pipe->frame_pixels;
    if (frames_redone) {
        *frames_redone = 0;
        if (pipe->nlevels > 0 && s->level >= pipe->nlevels) *frames_redone = s->nframes;     // classified in full from the start
        else if (pipe->nlevels > 0 && s->nframes > 0) {
            std::vector<int> redo(s->nframes);
            HIP_TRY(hipMemcpy(redo.data(), s->w.redo.p, sizeof(int) * s->nframes, hipMemcpyDeviceToHost));
            for (int v : redo) *frames_redone += v != 0;
        }
    }
    return LC_OK;
}

int lc_pipe_shared_queues(lc_pipe *pipe, int *count) {
    if (!pipe || !count) return fail(LC_EINVAL, "null argument");
    *count = pipe->shared_queues;
    return LC_OK;
}

int lc_pipe_upload(lc_pipe *pipe, int slot, const uint8_t *host_bgr, int nframes) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!host_bgr || nframes <= 0 || nframes > pipe->p.max_batch) return fail(LC_EINVAL, "upload: bad frames argument");
    if (s->busy) return fail(LC_EINVAL, "slot %d is still in flight; call lc_pipe_wait first", slot);
    if ((rc = upload_stream(pipe))) return rc;
    HIP_TRY(hipMemcpyAsync(pipe->resizes ? s->w.src_in.p : s->w.bgr.p, host_bgr, (size_t)nframes * pipe->src_frame_bytes,
                           hipMemcpyHostToDevice, pipe->st_up));
    HIP_TRY(hipStreamSynchronize(pipe->st_up));
    return LC_OK;
}

int lc_pipe_input_device_ptr(lc_pipe *pipe, int slot, void **dev_ptr) {
    Slot *s;
    int rc;
    if ((rc = get_slot(pipe, slot, &s))) return rc;
    if (!dev_ptr) return fail(LC_EINVAL, "null output");
    *dev_ptr = pipe->resizes ? s->w.src_in.p : s->w.bgr.p;
    return LC_OK;
}

}  // extern "C"
