// Stream-side half of the loopback TEST transport (see LoopbackComm in dml_comm.h): two
// one-wave kernels that tie a HIP stream to a flag word in coherent host memory.
//   signal: "everything enqueued on this stream before me has finished" -> host
//   wait  : hold the stream until the host has written the flag (bounded by a wall-clock budget;
//           on expiry a time-out word is set and the stream is released, so a lost helper thread
//           can never leave waves spinning)
#include <hip/hip_runtime.h>

#include "dml_comm.h"

namespace dml {

namespace {

__global__ void loopback_signal_kernel(unsigned* flag, unsigned value) {
    if (threadIdx.x == 0)
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void loopback_wait_kernel(const unsigned* flag, unsigned value, unsigned* timed_out,
                                     long long max_ticks) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    // flag values only grow (operation sequence numbers), compared wrap-safe
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > max_ticks) {
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

}  // namespace

void loopback_enqueue_signal(unsigned* flag, unsigned value, hipStream_t stream) {
    hipLaunchKernelGGL(loopback_signal_kernel, dim3(1), dim3(64), 0, stream, flag, value);
}

// The same one-wave kernel serves the bucket reducer's departures (flag word in device memory,
// consumer = a wait-value packet on another queue of this GPU).
void enqueue_flag_raise(unsigned* flag, unsigned value, hipStream_t stream) {
    hipLaunchKernelGGL(loopback_signal_kernel, dim3(1), dim3(64), 0, stream, flag, value);
}

void loopback_enqueue_wait(const unsigned* flag, unsigned value, unsigned* timed_out,
                           long long max_ticks, hipStream_t stream) {
    hipLaunchKernelGGL(loopback_wait_kernel, dim3(1), dim3(64), 0, stream, flag, value, timed_out,
                       max_ticks);
}

}  // namespace dml
