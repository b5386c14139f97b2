// C ABI of liblanecv.so (declared in include/lanecv.h): contexts and the per-op cv2-shaped entry points.
// Host logic only; the kernels live in kernels_*.hip, the drawing calls in api_draw.hip, the batched pipeline in
// pipe_create.hip and pipe_run.hip.
#include "host.h"

#include <mutex>

using namespace lanecv;

namespace lanecv {

static thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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

}  // namespace lanecv

// ---------------------------------------------------------------------------
// Working memory of lc_canny_noalloc: a fixed device array instead of hipMalloc'ed buffers, so that the call
// allocates nothing.  Every GPU the library runs on holds its own copy at its own address until the process ends.
// `scratch` itself is only a host-side handle of the symbol: the device address comes from hipGetSymbolAddress
// (scratch_address, per device) and is what host code passes to kernels and memsets.
// Layout for a w x h image (TX = ceil(w / 64), TY = ceil(h / 64)):
//   [planeS TX * h u64][planeW TX * h u64][wake 2 * TX * TY u32][rounds 1 i32]
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

namespace lanecv {

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

// Images of the stand-alone calls come from ordinary (pageable) host memory.  A 2-D copy from or to such memory goes row
// by row (4.9 ms for a 960 x 540 BGR image, 3.7 ms for its gray image back, against 0.06 and 0.03 ms for the same bytes
// as one linear copy), so an image whose rows lie back to back is copied linearly, and one with padded rows is packed on
// the host first.
int up2d(void *dst, const void *src, size_t row_bytes, int h, size_t stride, hipStream_t st) {
    if (stride == row_bytes) {
        HIP_TRY(hipMemcpyAsync(dst, src, row_bytes * (size_t)h, hipMemcpyHostToDevice, st));
        return LC_OK;
    }
    static thread_local std::vector<uint8_t> packed;
    packed.resize(row_bytes * (size_t)h);
    for (int y = 0; y < h; y++) memcpy(packed.data() + (size_t)y * row_bytes, (const uint8_t *)src + (size_t)y * stride, row_bytes);
    // (on the stream, then waited for: the buffer is reused by the next call, and the stream does not wait for the null stream)
    HIP_TRY(hipMemcpyAsync(dst, packed.data(), packed.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LC_OK;
}
int down2d(void *dst, size_t stride, const void *src, size_t row_bytes, int h, hipStream_t st) {
    HIP_TRY(hipGetLastError());   // a kernel launch of the call that failed (bad configuration, missing code object)
    if (stride == row_bytes) {
        HIP_TRY(hipMemcpyAsync(dst, src, row_bytes * (size_t)h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return LC_OK;
    }
    static thread_local std::vector<uint8_t> packed;
    packed.resize(row_bytes * (size_t)h);
    HIP_TRY(hipMemcpyAsync(packed.data(), src, packed.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int y = 0; y < h; y++) memcpy((uint8_t *)dst + (size_t)y * stride, packed.data() + (size_t)y * row_bytes, row_bytes);
    return LC_OK;
}

}  // namespace lanecv

namespace {

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
    HIP_TRY(hipMemsetAsync(wk.status.p, 0, 8 * sizeof(int), ctx->stream));
    const int line_length = (int)lrint(min_line_length), line_gap = (int)lrint(max_line_gap);
    const bool fast = launch_ppht_fast(wk.g, 1, wk.nz.p, wk.nz_cap, wk.npoints.p, wk.rowplane.p, wk.tab.p, na, nr, threshold, line_length,
                                       line_gap, wk.seg.p, cap, wk.segcount.p, wk.status.p, wk.fallback.p, wk.rng_out.p, wk.rng_len,
                                       nullptr, ctx->stream);
    // (a frame whose points are too scattered for its mask words to fit, or too many: the mask, then the points as well, from
    // device memory, before the general kernel)
    for (int state = 1; fast && state <= 2; state++)
        (void)launch_ppht_fast(wk.g, 1, wk.nz.p, wk.nz_cap, wk.npoints.p, wk.rowplane.p, wk.tab.p, na, nr, threshold, line_length,
                               line_gap, wk.seg.p, cap, wk.segcount.p, wk.status.p, wk.fallback.p, wk.rng_out.p, wk.rng_len, nullptr,
                               ctx->stream, 0, true, 8, 0, nullptr, state);
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
