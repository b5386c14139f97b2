// LDS atomic-add rate probe: the ceiling the Hough vote kernels are measured against.
//
// Every workgroup of 256 threads owns 2048 int32 bins in LDS and every lane issues `iters` x 8 atomic adds to
// them; the addresses of the 8 adds of an unrolled step differ by a compile-time offset (folded into the DS
// instruction), so the loop body holds no vector ALU work beside the atomics.  The lane -> address pattern is what
// the probe varies:
//   run = 1     each lane of a wavefront its own consecutive word (no two lanes share a bank or an address)
//   run = r     r neighbouring lanes share one word (sorted edge points voting for one angle: rho moves by
//               less than one bin per pixel, so neighbours meet in a bin)
//   run = 64    the whole wavefront on one word
//   run = 0     pseudo-random words of a 1024-word row (points in random order, as HoughLinesP picks them)
// RET: the returning form (ds_add_rtn_u32, what HoughLinesP's vote uses to see the new count) instead of the
// fire-and-forget form (ds_add_u32, what the full-vote accumulator uses).
#include "common.h"
#include <chrono>
#include <cstdlib>

namespace lanecv {
namespace {

constexpr int PROBE_BINS = 2048, PROBE_UNROLL = 8;

template <bool RET>
__global__ __launch_bounds__(256) void k_lds_atomic_probe(int iters, int run, int *__restrict__ sink) {
    __shared__ int bins[PROBE_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < PROBE_BINS; i += 256) bins[i] = 0;
    __syncthreads();
    int off[PROBE_UNROLL];
    if (run > 0) {
        // a wavefront works in its own quarter of the bins: lane / run + 64 * k, k = 0..7
        for (int k = 0; k < PROBE_UNROLL; k++) off[k] = wave * 512 + lane / run + 64 * k;
    } else {
        uint32_t s = (uint32_t)(blockIdx.x * 256 + tid) * 2654435761u + 12345u;
        for (int k = 0; k < PROBE_UNROLL; k++) {
            s = s * 1664525u + 1013904223u;
            off[k] = (int)((s >> 12) & 1023u) + (wave & 1) * 1024;
        }
    }
    int got = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < PROBE_UNROLL; k++) {
            if (RET) got += __hip_atomic_fetch_add(&bins[off[k]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else (void)__hip_atomic_fetch_add(&bins[off[k]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (RET) atomicAdd(&bins[PROBE_BINS - 1], got);   // the returned counts must stay live
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = bins[0] + bins[PROBE_BINS - 1];
}

}  // namespace

// Runs the probe on `st` and reports the duration of the best of three launches and the number of lane-level
// atomic adds one launch performs.  sink: device memory for `blocks` ints.
int lds_atomic_probe(int run, bool returning, int blocks, int iters, int *sink, hipStream_t st, float *ms, double *adds) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    float best = 0.f;
    for (int rep = 0; rep < 4; rep++) {   // the first launch warms up
        (void)hipEventRecord(e0, st);
        if (returning) hipLaunchKernelGGL(k_lds_atomic_probe<true>, dim3(blocks), dim3(256), 0, st, iters, run, sink);
        else hipLaunchKernelGGL(k_lds_atomic_probe<false>, dim3(blocks), dim3(256), 0, st, iters, run, sink);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return -1; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        if (rep > 0 && (best == 0.f || t < best)) best = t;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = best;
    *adds = (double)blocks * 256.0 * PROBE_UNROLL * iters;
    return 0;
}

// ---- which streams share a hardware queue --------------------------------------------------------------------
// HIP folds a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and kernels of two
// streams that share one run one after the other.  StreamPicker finds out by experiment: a kernel on stream a spins
// until the host releases it (bounded by a clock, so it always ends), a kernel on stream b raises a flag; if the
// flag arrives while a is still spinning, the two streams sit on different queues.
namespace {
// flags (mapped pinned host memory): [0] release, [1] a started, [2] a finished, [3] b ran
__global__ void k_queue_spin(volatile int *flags, long long max_ticks) {
    flags[1] = 1;
    const long long t0 = wall_clock64();                       // constant 100 MHz clock
    while (flags[0] == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(16);
    flags[2] = 1;
}
__global__ void k_queue_flag(volatile int *flags) { flags[3] = 1; }

double seconds_since(const std::chrono::steady_clock::time_point &t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

StreamPicker::~StreamPicker() {
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
    if (flags_host) (void)hipHostFree(flags_host);
}

bool StreamPicker::side_by_side(hipStream_t a, hipStream_t b) {
    if (!flags_host) {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return true; }
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess || !dp) { (void)hipHostFree(hp); (void)hipGetLastError(); return true; }
        flags_host = static_cast<int *>(hp);
        flags_dev = static_cast<int *>(dp);
    }
    volatile int *f = flags_host;
    f[0] = f[1] = f[2] = f[3] = 0;
    hipLaunchKernelGGL(k_queue_spin, dim3(1), dim3(1), 0, a, flags_dev, 400000ll);     // at most 4 ms
    auto t0 = std::chrono::steady_clock::now();
    while (!f[1] && seconds_since(t0) < 3e-3) {}
    bool beside = false;
    if (f[1]) {
        hipLaunchKernelGGL(k_queue_flag, dim3(1), dim3(1), 0, b, flags_dev);
        t0 = std::chrono::steady_clock::now();
        while (!f[3] && !f[2] && seconds_since(t0) < 1e-3) {}
        beside = f[3] && !f[2];
    }
    f[0] = 1;
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    (void)hipGetLastError();
    return beside;
}

hipStream_t StreamPicker::pick(const std::vector<hipStream_t> &others, bool *own_queue) {
    static const char *off = getenv("LANECV_QUEUE_PROBE");     // 0: take the first stream HIP hands out, as it comes
    const bool probe = !(off && off[0] == '0');
    hipStream_t last = nullptr;
    for (int tries = 0; tries < 12; tries++) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        bool ok = true;
        if (probe)
            for (hipStream_t o : others)
                if (o && !side_by_side(o, st)) { ok = false; break; }
        if (ok) {
            if (last) rejected.push_back(last);
            if (own_queue) *own_queue = true;
            return st;
        }
        if (last) rejected.push_back(last);     // kept until the picker goes: destroyed now, HIP would hand its queue out again
        last = st;
    }
    if (own_queue) *own_queue = false;
    return last;                                 // shares a queue with one of `others` (or null: no stream at all)
}

}  // namespace lanecv
