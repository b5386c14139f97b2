// Host-to-device copy of the rows from 517 down of 256 pinned 1080p frames (one 2-D copy) against nine column windows
// per frame (3-D copies), as the level-0 units of the triangle ROI would need them.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
int main() {
    const int W = 1920, H = 1080, N = 256;
    const size_t fb = (size_t)3 * W * H;
    uint8_t *h, *d;
    CK(hipHostMalloc((void **)&h, fb * N, hipHostMallocDefault));
    memset(h, 7, fb * N);
    CK(hipMalloc((void **)&d, fb * N));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Win { int r0, r1, c0, c1; };
    // bands of 62 rows from 525, strips smin..smax as the level-0 list of the 1080p triangle holds them
    const int smin[9] = {3, 2, 2, 2, 1, 1, 0, 0, 0}, smax[9] = {4, 4, 5, 5, 6, 6, 7, 7, 7};
    Win win[9]; size_t wbytes = 0;
    for (int b = 0; b < 9; b++) {
        const int y0 = 525 + 62 * b, y1 = y0 + 62 > H ? H : y0 + 62;
        win[b] = {y0 - 4, y1 + 4 > H ? H : y1 + 4, 248 * smin[b] - 4 < 0 ? 0 : 248 * smin[b] - 4, 248 * smax[b] + 252 > W ? W : 248 * smax[b] + 252};
        wbytes += (size_t)(win[b].r1 - win[b].r0) * (win[b].c1 - win[b].c0) * 3;
    }
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        const size_t off = (size_t)517 * 3 * W, bytes = (size_t)(H - 517) * 3 * W;
        CK(hipMemcpy2DAsync(d + off, fb, h + off, fb, bytes, N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        for (int b = 0; b < 9; b++) {
            hipMemcpy3DParms p = {};
            p.srcPtr = make_hipPitchedPtr(h, (size_t)3 * W, (size_t)3 * W, H);
            p.dstPtr = make_hipPitchedPtr(d, (size_t)3 * W, (size_t)3 * W, H);
            p.srcPos = make_hipPos((size_t)3 * win[b].c0, win[b].r0, 0);
            p.dstPos = p.srcPos;
            p.extent = make_hipExtent((size_t)3 * (win[b].c1 - win[b].c0), win[b].r1 - win[b].r0, N);
            p.kind = hipMemcpyHostToDevice;
            CK(hipMemcpy3DAsync(&p, st));
        }
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipMemcpyAsync(d, h, fb * N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t3 = now();
        printf("rows: %.2f MB/frame %.2f ms %.1f GB/s | windows: %.2f MB/frame %.2f ms %.1f GB/s | whole: %.2f ms %.1f GB/s\n", bytes / 1e6,
               (t1 - t0) * 1e3, bytes * N / (t1 - t0) / 1e9, wbytes / 1e6, (t2 - t1) * 1e3, wbytes * N / (t2 - t1) / 1e9, (t3 - t2) * 1e3,
               fb * N / (t3 - t2) / 1e9);
    }
    return 0;
}
