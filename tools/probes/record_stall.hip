// Does an event record between two kernels of a stream delay the second one while ANOTHER stream streams stores?
// Stream A: back-to-back kernels that write 256 MB each (or spin, or nothing).  Stream B: k1, [hipEventRecord], k2, ...
// Prints the gap k1.end -> k2.start on B.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/record_stall.hip -o gpurun_out/record_stall && gpurun_out/record_stall
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void stamp_kernel(long long ticks, long long* stamp) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void writer(float4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void spinner(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int main() {
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    long long* st; CK(hipHostMalloc((void**)&st, 64 * sizeof(long long)));
    const size_t bytes = 256ull << 20;
    float4 *buf, *buf2; CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&buf2, bytes));
    hipEvent_t ev, ev2; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    const char* loads[3] = {"A idle", "A spins (2048 blocks)", "A writes 256 MB per kernel"};
    const char* mids[4] = {"nothing", "hipEventRecord", "record + waited for by C", "hipStreamWaitEvent on a finished event"};
    for (int load = 0; load < 3; ++load)
        for (int mid = 0; mid < 4; ++mid) {
            printf("%-28s between k1 and k2: %-40s gap us:", loads[load], mids[mid]);
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(ev2, C));
                CK(hipDeviceSynchronize());
                for (int i = 0; i < 12; ++i) {
                    if (load == 1) hipLaunchKernelGGL(spinner, dim3(2048), dim3(256), 0, A, 8000);
                    if (load == 2) hipLaunchKernelGGL(writer, dim3(4096), dim3(256), 0, A, i & 1 ? buf : buf2, bytes / 16, (float)i);
                }
                hipLaunchKernelGGL(stamp_kernel, dim3(8), dim3(256), 0, B, 3000, st + 0);      // k1: 30 us
                if (mid == 1 || mid == 2) CK(hipEventRecord(ev, B));
                if (mid == 2) { CK(hipStreamWaitEvent(C, ev, 0)); hipLaunchKernelGGL(spinner, dim3(1), dim3(64), 0, C, 100); }
                if (mid == 3) CK(hipStreamWaitEvent(B, ev2, 0));
                hipLaunchKernelGGL(stamp_kernel, dim3(8), dim3(256), 0, B, 3000, st + 2);      // k2
                CK(hipDeviceSynchronize());
                printf(" %7.1f", (double)(st[2] - st[1]) / 100.0);
            }
            printf("\n");
        }
    return 0;
}
