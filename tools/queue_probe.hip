// Which HIP streams of a process share a hardware queue?  Creates N streams (plus the null stream), and for every
// ordered pair (a, b) launches a kernel on a that spins until the host releases it and a kernel on b that sets a
// flag: if b's flag arrives while a still spins, a and b sit on different hardware queues.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/bin/queue_probe && tools/bin/queue_probe 8
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void k_spin(volatile int *release, volatile int *started, long long max_cycles) {
    *started = 1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (*release == 0 && __builtin_amdgcn_s_memtime() - t0 < max_cycles) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_flag(volatile int *flag) { *flag = 1; }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6;
    std::vector<hipStream_t> st(n + 1, nullptr);     // st[0] = null stream
    for (int i = 1; i <= n; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    int *h = nullptr;
    CK(hipHostMalloc((void **)&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    volatile int *release = h, *started = h + 1, *flag = h + 2;
    int *d = nullptr;
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    printf("GPU_MAX_HW_QUEUES=%s; row a (spinning), column b: 1 = b ran beside a\n    ", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    for (int b = 0; b <= n; b++) printf("%2d ", b);
    printf("\n");
    for (int a = 0; a <= n; a++) {
        printf("%2d: ", a);
        for (int b = 0; b <= n; b++) {
            if (a == b) { printf(" . "); continue; }
            *release = 0; *started = 0; *flag = 0;
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, st[a], d, d + 1, 400000000ll);   // bounded: ~4 s of 100 MHz ticks at most
            const auto t0 = std::chrono::steady_clock::now();
            while (!*started && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) {}
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, st[b], d + 2);
            const auto t1 = std::chrono::steady_clock::now();
            while (!*flag && std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < 0.02) {}
            const int beside = *flag;
            *release = 1;
            CK(hipStreamSynchronize(st[a]));
            CK(hipStreamSynchronize(st[b]));
            printf("%2d ", beside);
        }
        printf("\n");
    }
    return 0;
}
