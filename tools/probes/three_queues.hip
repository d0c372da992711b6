// Do N streams with independent back-to-back kernels all make progress at once?  Each stream launches K kernels of T us
// (64 blocks: nowhere near the chip's capacity); every kernel stamps its start.  Prints, per stream, when its kernels ran.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/three_queues.hip -o gpurun_out/three_queues && gpurun_out/three_queues [streams] [priority of stream 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void busy(long long ticks, long long* stamp) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
}
int main(int argc, char** argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 3, K = 20;
    const int prio0 = argc > 2 ? atoi(argv[2]) : 0;
    hipStream_t s[8];
    for (int i = 0; i < NS; ++i) {
        if (i == 0 && prio0) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, prio0));
        else CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    }
    long long* st; CK(hipHostMalloc((void**)&st, NS * K * 2 * sizeof(long long)));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        // stream 0 starts LAST (like the conv chain that comes back from a pause while the two others are already busy)
        for (int k = 0; k < K; ++k)
            for (int i = NS - 1; i >= 0; --i) {
                if (i == 0 && k < 4) continue;                      // stream 0 joins 4 kernels late
                hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, s[i], 5000, st + (i * K + k) * 2);
            }
        CK(hipDeviceSynchronize());
    }
    long long t0 = st[(1 * K + 0) * 2];
    for (int i = 0; i < NS; ++i) {
        printf("stream %d starts (us):", i);
        for (int k = (i == 0 ? 4 : 0); k < K; ++k) printf(" %6.0f", (double)(st[(i * K + k) * 2] - t0) / 100.0);
        printf("\n");
    }
    return 0;
}
