// Probe: cost of a cross-stream hand-off on the producing stream.
//   (1) hipEventRecord on A + hipStreamWaitEvent on B          (what the engines do)
//   (2) flag written by the NEXT kernel on A + hipStreamWaitValue32 on B
// Prints, per variant, the gap between the producer's end and the next kernel's start on A and
// the delay until the consumer on B starts.  Times are wall_clock64 ticks (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void spin(long long ticks, long long* t_start, long long* t_end, unsigned* flag, unsigned val) {
    long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (t_start) *t_start = t0;
        if (flag) __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0 && t_end) *t_end = wall_clock64();
}

int main(int argc, char** argv) {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue = %d\n", can);
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    long long* t; CK(hipHostMalloc(&t, 64 * sizeof(long long)));
    unsigned* flag = nullptr;
    const char* mem = argc > 1 ? argv[1] : "signal";
    if (!strcmp(mem, "signal")) CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    else if (!strcmp(mem, "host")) CK(hipHostMalloc((void**)&flag, 8));
    else CK(hipMalloc((void**)&flag, 8));
    CK(hipMemset(flag, 0, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 6; ++rep) {
        memset(t, 0, 64 * sizeof(long long));
        // (1) events
        spin<<<256, 64, 0, A>>>(20000, &t[0], &t[1], nullptr, 0);          // producer, 200 us
        CK(hipEventRecord(ev, A));
        spin<<<256, 64, 0, A>>>(5000, &t[2], &t[3], nullptr, 0);           // next on A
        CK(hipStreamWaitEvent(B, ev, 0));
        spin<<<64, 64, 0, B>>>(1000, &t[4], &t[5], nullptr, 0);            // consumer on B
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        printf("events   : A gap %6.2f us, B start after producer end %6.2f us\n",
               (t[2] - t[1]) / 100.0, (t[4] - t[1]) / 100.0);
        if (!can) continue;
        // (2) wait-value
        unsigned val = rep + 1;
        spin<<<256, 64, 0, A>>>(20000, &t[8], &t[9], nullptr, 0);
        spin<<<256, 64, 0, A>>>(5000, &t[10], &t[11], flag, val);
        CK(hipStreamWaitValue32(B, flag, val, hipStreamWaitValueGte, 0xffffffffu));
        spin<<<64, 64, 0, B>>>(1000, &t[12], &t[13], nullptr, 0);
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        printf("waitvalue: A gap %6.2f us, B start after producer end %6.2f us\n",
               (t[10] - t[9]) / 100.0, (t[12] - t[9]) / 100.0);
        // (3) no hand-off at all: the floor for the gap on A
        spin<<<256, 64, 0, A>>>(20000, &t[16], &t[17], nullptr, 0);
        spin<<<256, 64, 0, A>>>(5000, &t[18], &t[19], nullptr, 0);
        CK(hipStreamSynchronize(A));
        printf("plain    : A gap %6.2f us\n", (t[18] - t[17]) / 100.0);
    }
    return 0;
}
