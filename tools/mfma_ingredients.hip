// Which ingredient of a real GEMM main loop costs fp32-MFMA throughput on gfx950?  Each variant adds one.
//  V0: MFMA only.  V1: + 4 ds_read_b128 per 16 MFMAs (fragments really come from LDS).
//  V4: V3 with sched_barrier fences pinning [loads | MFMAs | wait+LDS stores].
//  V2: V1 + one __syncthreads per 64 MFMAs.  V3: V2 + 8 global_load_dwordx4 per 64 MFMAs written to LDS after.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256 * 36; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* as = lds + ((wave >> 1) * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
    const float* bs = lds + 128 * 36 + ((wave & 1) * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
    const float* gp = in + (size_t)blockIdx.x * 65536 + tid * 4;
    float4 stage[8];
    for (int it = 0; it < iters; ++it) {
        if (V >= 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) stage[u] = *reinterpret_cast<const float4*>(gp + ((it * 8 + u) & 15) * 1024);
            if (V >= 4) __builtin_amdgcn_sched_barrier(0);      // loads stay ABOVE the MFMA phase
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a[2], b[2];
            if (V >= 1) {
                a[0] = *reinterpret_cast<const float4*>(as + 8 * g);
                a[1] = *reinterpret_cast<const float4*>(as + 32 * 36 + 8 * g);
                b[0] = *reinterpret_cast<const float4*>(bs + 8 * g);
                b[1] = *reinterpret_cast<const float4*>(bs + 32 * 36 + 8 * g);
            } else {
                a[0] = a[1] = b[0] = b[1] = make_float4(1.f + lane, 2.f, 3.f, 4.f + g);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i * 2 + j], 0, 0, 0);
                }
        }
        if (V >= 4) __builtin_amdgcn_sched_barrier(0);          // wait + LDS stores stay BELOW the MFMA phase
        if (V >= 3) {
            float* dst = lds + ((it & 1) ? 0 : 256 * 36) + (tid >> 3) * 36 + 4 * (tid & 7);
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(dst + 32 * u * 36) = stage[u];
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// V5: staging by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write; unpadded 128-B rows with
// the 16-B chunk index XOR-swizzled by (row>>1)&7 on the SOURCE address and on the fragment read.
__global__ __launch_bounds__(256, 2) void k5(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][256 rows][32 floats]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256 * 32; i += 256) lds[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int ra = (wave >> 1) * 64 + (lane & 31), rb = 128 + (wave & 1) * 64 + (lane & 31);
    const float* gp = in + (size_t)blockIdx.x * 65536;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        // stage next tile: wave w fills rows [64w, 64w+64): 8 instructions of 8 rows x 128 B
        {
            float* dstbase = lds + (cur ^ 1) * 256 * 32 + wave * 64 * 32;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = wave * 64 + u * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const float* src = gp + ((it * 8 + u) & 15) * 1024 + (lane >> 3) * 32 + chunk * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dstbase + u * 8 * 32), 16, 0, 0);
            }
        }
        const float* base = lds + cur * 256 * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a[2], b[2];
            const int c = 2 * g + (lane >> 5);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r0 = ra + 32 * i, r1 = rb + 32 * i;
                a[i] = *reinterpret_cast<const float4*>(base + r0 * 32 + ((c ^ ((r0 >> 1) & 7)) << 2));
                b[i] = *reinterpret_cast<const float4*>(base + r1 * 32 + ((c ^ ((r1 >> 1) & 7)) << 2));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i * 2 + j], 0, 0, 0);
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i * 2 + j], 0, 0, 0);
                }
        }
        __syncthreads();
        cur ^= 1;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}
void run5(float* out, float* in, int blocks, int iters) {
    const size_t ldsb = 2 * 256 * 32 * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k5, dim3(blocks), dim3(256), ldsb, 0, out, in, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 64 * 4096.0;
        if (rep == 2) printf("V5 (LDS-DMA) blocks %d: %.2f ms  %.1f TFLOP/s\n", blocks, ms, flop / ms / 1e9);
    }
}
template <int V>
void run(float* out, float* in, int blocks, int iters) {
    const size_t ldsb = 2 * 256 * 36 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), ldsb, 0, out, in, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 64 * 4096.0;
        if (rep == 2) printf("V%d blocks %d: %.2f ms  %.1f TFLOP/s\n", V, blocks, ms, flop / ms / 1e9);
    }
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&in, (size_t)4096 * 65536 * 4 + 65536);
    (void)hipMemset(in, 0, (size_t)4096 * 65536 * 4 + 65536);
    for (int blocks : {256, 512, 900}) {
        run<0>(out, in, blocks, 2000); run<1>(out, in, blocks, 2000); run<2>(out, in, blocks, 2000); run<3>(out, in, blocks, 2000); run5(out, in, blocks, 2000);
    }
    return 0;
}
