// Streams that really run next to each other.
//
// HIP maps streams onto a small pool of hardware queues (GPU_MAX_HW_QUEUES, 4 by default)
// by least use, and two streams that land on the same queue execute strictly in order: a
// "side" stream for collectives or weight gradients then overlaps nothing, and every event
// hand-off between the two becomes an in-queue barrier.  Which queue a new stream gets
// depends on what RCCL, torch and the application created before it, so it is measured:
// a 2 ms spin kernel occupies the busy stream while an empty kernel is timed on the
// candidate; the candidate is kept only if the empty kernel did not have to wait.
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "dml_kernels.h"

namespace dml {

namespace {

__global__ void spin_kernel(long long ticks) {
    // wall_clock64 counts a constant 100 MHz clock; every wave reaches the exit condition
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

__global__ void empty_kernel() {}

// What a stream that waits for the host looks like to the kernels around it: one lane per block
// polling a flag word (system-scope loads), asleep in between, until the flag is raised or the
// time is up.
__global__ void poll_kernel(const unsigned* flag, long long ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > ticks) break;
    }
}

void check(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP ") + what + " failed: " + hipGetErrorString(e));
}

}  // namespace

void launch_pollers(int blocks, long long ticks, const unsigned* flag, hipStream_t stream) {
    if (blocks <= 0 || blocks > 4096 || ticks <= 0 || ticks > 1000000000LL)
        throw std::invalid_argument("launch_pollers: bad block count or duration");
    hipLaunchKernelGGL(poll_kernel, dim3(blocks), dim3(64), 0, stream, flag, ticks);
}

bool streams_run_concurrently(hipStream_t busy, hipStream_t candidate) {
    check(hipStreamSynchronize(busy), "hipStreamSynchronize");
    check(hipStreamSynchronize(candidate), "hipStreamSynchronize");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, busy, 200000LL);     // ~2 ms
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, candidate);
    const auto t0 = std::chrono::steady_clock::now();
    check(hipStreamSynchronize(candidate), "hipStreamSynchronize");
    const double ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    check(hipStreamSynchronize(busy), "hipStreamSynchronize");
    return ms < 1.0;
}

hipStream_t create_concurrent_stream(const std::vector<hipStream_t>& avoid, int max_tries) {
    std::vector<hipStream_t> rejected;      // kept alive so the next candidate lands elsewhere
    hipStream_t found = nullptr;
    for (int t = 0; t < max_tries && !found; ++t) {
        hipStream_t s = nullptr;
        // (plain streams on purpose: a low- or high-priority side stream measured no different,
        // and a CU-masked one (hipExtStreamCreateWithCUMask) more than doubled the step time)
        check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
        bool ok = true;
        for (hipStream_t a : avoid)
            if (!streams_run_concurrently(a, s)) { ok = false; break; }
        if (ok) found = s;
        else rejected.push_back(s);
    }
    if (!found) {
        // every candidate shares a hardware queue with a busy stream: still correct, no overlap
        found = rejected.back();
        rejected.pop_back();
        fprintf(stderr, "[dml] warning: no stream on a hardware queue of its own after %d tries "
                        "(GPU_MAX_HW_QUEUES too small?); work on it will not overlap\n", max_tries);
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    return found;
}

}  // namespace dml
