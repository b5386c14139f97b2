// Do chains of short kernels on several HIP streams run side by side?  Each of K streams gets M kernels back to back
// (one small workgroup spinning for about DUR microseconds); the wall time for all of them is printed per K, and per
// way of waiting for the end (one event per stream / stream synchronise).  Ideal: the same time for every K.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_scaling_probe.hip -o tools/bin/stream_scaling_probe && tools/bin/stream_scaling_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void k_busy(long long ticks, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) *sink = 1;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 200;
    const double dur_us = argc > 2 ? atof(argv[2]) : 50.0;
    const int blocks = argc > 3 ? atoi(argv[3]) : 32;
    const int threads = argc > 4 ? atoi(argv[4]) : 1024;
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)(dur_us * 1e-6 * rate_khz * 1e3);
    int *sink = nullptr;
    CK(hipMalloc((void **)&sink, 4));
    std::vector<hipStream_t> st(6);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("wall clock %d kHz; %d kernels of %.0f us (%d x %d threads) per stream\n", rate_khz, M, dur_us, blocks, threads);
    for (int K = 1; K <= 6; K++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < M; i++)
                for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(threads), 0, st[k], ticks, sink);
            const auto t1 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; k++) CK(hipStreamSynchronize(st[k]));
            const auto t2 = std::chrono::steady_clock::now();
            if (rep == 1)
                printf("K=%d  submit %.1f us/kernel  total %.2f ms  = %.1f us per kernel per stream (ideal %.0f)\n", K,
                       std::chrono::duration<double>(t1 - t0).count() * 1e6 / (M * K), std::chrono::duration<double>(t2 - t0).count() * 1e3,
                       std::chrono::duration<double>(t2 - t0).count() * 1e6 / M, dur_us);
        }
    }
    return 0;
}
