// When does a stream that waits for an event start, if the RECORDING stream has many more launches queued behind the record?
// s1: wait(start) k k rec(e) then `tail` more kernels (with or without further records between them);  s0: k0 rec(start) wait(e) kmain.
// Prints kmain.start - (end of the kernel in front of the record), device clock.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/record_chain.hip -o gpurun_out/record_chain && gpurun_out/record_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void stamp_kernel(long long ticks, long long* stamp) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
}
int main() {
    hipStream_t s0, s1, s2;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    long long* st; CK(hipHostMalloc((void**)&st, 256 * sizeof(long long)));
    hipEvent_t start, e, more[64];
    CK(hipEventCreateWithFlags(&start, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& m : more) CK(hipEventCreateWithFlags(&m, hipEventDisableTiming));
    const long long us = 100;                       // wall_clock64 ticks at 100 MHz
    for (int tail : {16})
        for (int recs : {0, 1})
            for (int third : {0, 1}) {
                printf("tail %2d kernels behind the record%s%s: wait us:", tail, recs ? ", a record after each" : "",
                       third ? ", a third stream busy" : "");
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipDeviceSynchronize());
                    hipLaunchKernelGGL(stamp_kernel, dim3(64), dim3(256), 0, s0, 400 * us, st + 0);      // holds everything back while the host queues
                    CK(hipEventRecord(start, s0));
                    CK(hipStreamWaitEvent(s1, start, 0));
                    if (third) {
                        CK(hipStreamWaitEvent(s2, start, 0));
                        for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(stamp_kernel, dim3(32), dim3(256), 0, s2, 10 * us, st + 200);
                    }
                    hipLaunchKernelGGL(stamp_kernel, dim3(48), dim3(256), 0, s1, 30 * us, st + 2);
                    hipLaunchKernelGGL(stamp_kernel, dim3(48), dim3(256), 0, s1, 10 * us, st + 4);
                    CK(hipEventRecord(e, s1));
                    for (int i = 0; i < tail; ++i) {
                        hipLaunchKernelGGL(stamp_kernel, dim3(120), dim3(256), 0, s1, 10 * us, st + 10 + 2 * i);
                        if (recs) CK(hipEventRecord(more[i], s1));
                    }
                    CK(hipStreamWaitEvent(s0, e, 0));
                    hipLaunchKernelGGL(stamp_kernel, dim3(32), dim3(256), 0, s0, 5 * us, st + 6);
                    CK(hipDeviceSynchronize());
                    printf(" %.1f", (st[6] - st[5]) / 100.0);
                }
                printf("\n");
            }
    // ---- how long the waiting stream has been blocked when the event fires
    for (int blocked_us : {20, 50, 100, 200, 400, 800, 1600})
        for (int third : {0, 1}) {
            printf("waiting stream blocked for %4d us%s: resume latency us:", blocked_us, third ? ", a third stream busy" : "");
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipDeviceSynchronize());
                hipLaunchKernelGGL(stamp_kernel, dim3(64), dim3(256), 0, s0, 300 * us, st + 0);
                CK(hipEventRecord(start, s0));
                CK(hipStreamWaitEvent(s1, start, 0));
                if (third) {
                    CK(hipStreamWaitEvent(s2, start, 0));
                    for (int i = 0; i < blocked_us / 10 + 10; ++i) hipLaunchKernelGGL(stamp_kernel, dim3(32), dim3(256), 0, s2, 10 * us, st + 200);
                }
                hipLaunchKernelGGL(stamp_kernel, dim3(48), dim3(256), 0, s1, (long long)blocked_us * us, st + 4);
                CK(hipEventRecord(e, s1));
                for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(stamp_kernel, dim3(120), dim3(256), 0, s1, 10 * us, st + 10 + 2 * i);
                CK(hipStreamWaitEvent(s0, e, 0));
                hipLaunchKernelGGL(stamp_kernel, dim3(32), dim3(256), 0, s0, 5 * us, st + 6);
                CK(hipDeviceSynchronize());
                printf(" %.1f", (st[6] - st[5]) / 100.0);
            }
            printf("\n");
        }
    return 0;
}
