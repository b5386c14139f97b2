// Does a kernel launched with hipExtAnyOrderLaunch start beside the kernel queued before it on the SAME stream, and does
// it still wait for a hipStreamWaitEvent queued between the two?
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/bin/anyorder_probe && tools/bin/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void k_spin(volatile int *release, volatile int *started, long long max_cycles) {
    *started = 1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (*release == 0 && __builtin_amdgcn_s_memtime() - t0 < max_cycles) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_flag(volatile int *flag) { *flag = 1; }

static bool wait_for(volatile int *p, double s) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!*p && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < s) {}
    return *p != 0;
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void **)&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&d, h, 0));
    volatile int *v = h;
    const long long limit = 2000000000ll;
    for (int flags = 0; flags <= 1; flags++) {
        // 1. spin on s1, then the flag kernel on s1
        for (int i = 0; i < 16; i++) h[i] = 0;
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, s1, d, d + 1, limit);
        wait_for(v + 1, 1.0);
        hipExtLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s1, nullptr, nullptr, flags, d + 2);
        const bool beside = wait_for(v + 2, 0.05);
        h[0] = 1;
        CK(hipStreamSynchronize(s1));
        printf("flags=%d: second kernel of the stream ran beside the first: %d (finished after release: %d)\n", flags, beside, v[2]);
        // 2. spin A on s1; spin B on s2 + event; s1 waits for the event; flag kernel on s1
        for (int i = 0; i < 16; i++) h[i] = 0;
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, s1, d, d + 1, limit);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, s2, d + 4, d + 5, limit);
        CK(hipEventRecord(ev, s2));
        wait_for(v + 1, 1.0);
        wait_for(v + 5, 1.0);
        CK(hipStreamWaitEvent(s1, ev, 0));
        hipExtLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s1, nullptr, nullptr, flags, d + 2);
        const bool early = wait_for(v + 2, 0.05);          // must be 0: B still spins
        h[4] = 1;                                          // release B: the event fires, A still spins
        const bool after_b = wait_for(v + 2, 0.05);
        h[0] = 1;
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        printf("flags=%d: with an event wait between: before the event fired %d (must be 0), after it while the first kernel still ran %d, at the end %d\n",
               flags, early, after_b, v[2]);
        // 3. an event recorded behind the any-order kernel must not fire before the earlier kernel has finished
        for (int i = 0; i < 16; i++) h[i] = 0;
        hipEvent_t e2;
        CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, s1, d, d + 1, limit);
        wait_for(v + 1, 1.0);
        hipExtLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s1, nullptr, nullptr, flags, d + 2);
        CK(hipEventRecord(e2, s1));
        wait_for(v + 2, 0.05);
        const bool fired_early = hipEventQuery(e2) == hipSuccess;
        h[0] = 1;
        CK(hipStreamSynchronize(s1));
        printf("flags=%d: event behind both fired while the first kernel ran: %d (must be 0)\n", flags, fired_early);
        CK(hipEventDestroy(e2));
    }
    return 0;
}
