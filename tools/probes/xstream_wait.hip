// How fast does a stream resume behind a mark set on another stream?  Three mechanisms:
//   event : hipEventRecord on A / hipStreamWaitEvent on B                 (what torch and the engine used)
//   value : hipStreamWriteValue32 on A / hipStreamWaitValue32 on B        (command-processor memory semaphore)
//   flag  : a one-lane kernel on A stores a flag / a one-wave kernel on B polls it (bounded)
// Stream A: a1 (short), MARK, a2 (long).  Stream B: WAIT, b1.  Kernels stamp wall_clock64 (100 MHz) at start and end.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/xstream_wait.hip -o gpurun_out/xstream_wait && gpurun_out/xstream_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy(long long ticks, long long* stamp) {      // every block spins for `ticks` of the 100 MHz clock
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}
__global__ void set_flag(unsigned* flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void wait_flag(const unsigned* flag, unsigned v, unsigned* timed_out) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < (1 << 22); ++i) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= v) return;
        __builtin_amdgcn_s_sleep(16);
    }
    *timed_out = 1;
}

int main() {
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    long long* st; CK(hipHostMalloc((void**)&st, 64 * sizeof(long long)));
    unsigned* flag; CK(hipMalloc((void**)&flag, 256)); CK(hipMemset(flag, 0, 256));
    unsigned* sig = nullptr;
    bool have_sig = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) == hipSuccess;
    if (have_sig) CK(hipMemset(sig, 0, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const long long SHORT = 2000, LONG = 200000;      // 20 us, 2 ms
    const char* kinds[3] = {"event", "value", "flag"};
    unsigned epoch = 0;
    for (int trail = 0; trail <= 40; trail += 20)
    for (int third = 0; third < 3; ++third)
        for (int k = 0; k < 3; ++k) {
            if (k == 1 && !have_sig) { printf("value: no signal memory\n"); continue; }
            printf("%-5s trail %2d third stream %s:", kinds[k], trail, third == 0 ? "idle" : third == 1 ? "busy (4 long)" : "busy (80 short)");
            for (int rep = 0; rep < 5; ++rep) {
                ++epoch;
                memset(st, 0, 64 * sizeof(long long));
                CK(hipDeviceSynchronize());
                if (third == 1) for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, C, LONG / 2, (long long*)nullptr);
                if (third == 2) for (int i = 0; i < 80; ++i) hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, C, LONG / 40, (long long*)nullptr);
                hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, A, SHORT, st + 0);              // a1
                if (k == 0) CK(hipEventRecord(ev, A));
                else if (k == 1) CK(hipStreamWriteValue32(A, sig, epoch, 0));
                else hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, A, flag, epoch);
                if (trail == 0) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, A, LONG, st + 2);               // a2
                else for (int i = 0; i < trail; ++i) hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, A, LONG / trail, i == 0 ? st + 2 : (long long*)nullptr);
                if (k == 0) CK(hipStreamWaitEvent(B, ev, 0));
                else if (k == 1) CK(hipStreamWaitValue32(B, sig, epoch, hipStreamWaitValueGte, 0xffffffffu));
                else hipLaunchKernelGGL(wait_flag, dim3(1), dim3(64), 0, B, flag, epoch, flag + 16);
                hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, B, SHORT, st + 4);              // b1
                CK(hipDeviceSynchronize());
                printf("  %7.1f", (double)(st[4] - st[1]) / 100.0);       // b1 start - a1 end, us
            }
            printf("  us from a1's end to b1's start (a2 lasts %.0f us)\n", (double)(st[3] - st[2]) / 100.0);
        }
    unsigned to = 0; CK(hipMemcpy(&to, flag + 16, 4, hipMemcpyDeviceToHost));
    printf("flag waits timed out: %u\n", to);
    return 0;
}
