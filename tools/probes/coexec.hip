// Does an fp32 MFMA co-execute with vector ALU work of the SAME SIMD on gfx950?  
//
// 512-thread workgroups, one per CU: waves w and w + 4 share SIMD w (MI355X_MICROARCH.md "Two waves per SIMD").  Per mode the
// in-kernel clock (s_memtime) of every wave and the wall time of the launch:
//   mfma        waves 0-3: N independent v_mfma_f32_16x16x4_f32 (four accumulators), waves 4-7 leave at once
//   valu        waves 4-7: R * N v_pk_fma_f32 (eight independent chains),            waves 0-3 leave at once
//   both        the two together: max(mfma, valu) if the pipes overlap across the waves of a SIMD, the sum if not
//   both.prio   both, the VALU waves at s_setprio 3;   both.swap: both with the roles of the older and the younger waves swapped
//   inwave      ONE wave per SIMD (waves 0-3) issuing  1 MFMA + R pk_fma  per step in program order: the same question for
//               the instruction stream of a single wave (what a software-pipelined fold inside the product kernel relies on)
//   inwave2     the same on both waves of every SIMD
// R (packed FMAs per MFMA) sweeps 0 ... 8: a 16x16x4 fp32 MFMA holds the matrix pipe for 32 cycles, a pk_fma the VALU for 8 (measured
// below), so up to R = 4 fit beside an MFMA if the pipes overlap.
// Counters: run each mode under  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES  in a pass
// of its own (tools/coexec.sh); the kernels are templates so that every mode has its own name in the counter csv.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA_F32(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// positive control: a bf16 MFMA (its multipliers are not the fp32 FMA lanes of the VALU)
#define MFMA_BF16(acc, a4, b4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a4), "v"(b4))
#define MFMA(acc, a, b) do { if (BF) MFMA_BF16(acc, a4, b4); else MFMA_F32(acc, a, b); } while (0)
#define PKFMA(x, m, c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))

enum { M_MFMA = 0, M_VALU = 1, M_BOTH = 2, M_INWAVE = 3, M_INWAVE2 = 4, M_BOTH_PRIO = 5, M_BOTH_SWAP = 6 };

template <int MODE, int R, bool BF = false>
__global__ __launch_bounds__(512) void coexec_kernel(float* out, int n, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6;
    // M_BOTH_PRIO: as M_BOTH with the VALU waves at s_setprio 3 (is the sum in M_BOTH the arbiter always serving the older wave?);
    // M_BOTH_SWAP: the roles swapped -- the OLDER waves 0-3 do the vector work, waves 4-7 the MFMAs
    const bool do_mfma = (MODE == M_MFMA || MODE == M_BOTH || MODE == M_BOTH_PRIO) ? wave < 4
                       : MODE == M_BOTH_SWAP ? wave >= 4 : (MODE == M_INWAVE ? wave < 4 : MODE == M_INWAVE2);
    const bool do_valu = (MODE == M_VALU || MODE == M_BOTH || MODE == M_BOTH_PRIO) ? wave >= 4 : (MODE == M_BOTH_SWAP ? wave < 4 : false);
    if (MODE == M_BOTH_PRIO && do_valu) __builtin_amdgcn_s_setprio(3);
    f32x4 acc[4];
    f32x2 x[8];
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + blockIdx.x * 1e-4f;
    const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    const f32x4 a4 = {a, b, a, b}, b4 = {b, a, b, a};      // (bit patterns read as 8 bf16 each: the values do not matter)
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) x[i] = f32x2{a + i, b - i};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == M_INWAVE || MODE == M_INWAVE2) {
        if (do_mfma) {
            for (int it = 0; it < n; it += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    MFMA(acc[u], a, b);
#pragma unroll
                    for (int r = 0; r < R; ++r) PKFMA(x[(u * R + r) & 7], m, c);
                }
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < n; it += 4) {
            MFMA(acc[0], a, b); MFMA(acc[1], a, b); MFMA(acc[2], a, b); MFMA(acc[3], a, b);
        }
    } else if (do_valu) {
        for (int it = 0; it < n; it += 4) {
#pragma unroll
            for (int r = 0; r < 4 * R; ++r) PKFMA(x[r & 7], m, c);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_out; static unsigned long long* g_clk;
static const int BLOCKS = 256, N = 1 << 16;

template <int MODE, int R, bool BF = false>
static void run(const char* name, double clk_per_memtime) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((coexec_kernel<MODE, R, BF>), dim3(BLOCKS), dim3(512), 0, 0, g_out, N, g_clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(BLOCKS * 8);
    hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;      // mean over blocks of the waves 0-3 resp. 4-7
    for (int b = 0; b < BLOCKS; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
    lo /= BLOCKS * 4; hi /= BLOCKS * 4;
    printf("%-12s R=%d  wall %8.3f ms   waves0-3 %7.1f cyc/step   waves4-7 %7.1f cyc/step\n", name, R, best,
           lo * clk_per_memtime / N, hi * clk_per_memtime / N);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int R>
static void sweep(double cpm) {
    run<M_VALU, R>("valu", cpm);
    run<M_BOTH, R>("both", cpm);
    if (R == 2 || R == 4 || R == 8) { run<M_BOTH_PRIO, R>("both.prio", cpm); run<M_BOTH_SWAP, R>("both.swap", cpm); }
    run<M_INWAVE, R>("inwave", cpm);
    run<M_INWAVE2, R>("inwave2", cpm);
}

int main(int argc, char** argv) {
    hipMalloc(&g_out, (size_t)BLOCKS * 512 * 4); hipMalloc(&g_clk, (size_t)BLOCKS * 8 * 8);
    // core clocks per s_memtime tick from an MFMA-only run: a v_mfma_f32_16x16x4_f32 occupies the pipe for 32 core cycles.
    double cpm = 1.0;
    {
        hipLaunchKernelGGL((coexec_kernel<M_MFMA, 0>), dim3(BLOCKS), dim3(512), 0, 0, g_out, N, g_clk);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(BLOCKS * 8);
        hipMemcpy(h.data(), g_clk, h.size() * 8, hipMemcpyDeviceToHost);
        double lo = 0; for (int b = 0; b < BLOCKS; ++b) for (int w = 0; w < 4; ++w) lo += (double)h[b * 8 + w];
        lo /= BLOCKS * 4;
        cpm = 32.0 * N / lo;
        printf("# calibration: MFMA-only wave, %d MFMAs in %.0f s_memtime ticks -> %.2f core cycles per tick (taking 32 cycles per MFMA)\n", N, lo, cpm);
    }
    printf("# 'cyc/step': core cycles per loop step of a wave = per (1 MFMA [+ R pk_fma]) resp. per R pk_fma; both = waves 0-3 MFMA, 4-7 VALU\n");
    run<M_MFMA, 0>("mfma", cpm);
    sweep<1>(cpm); sweep<2>(cpm); sweep<3>(cpm); sweep<4>(cpm); sweep<6>(cpm); sweep<8>(cpm);
    printf("# positive control: the same with v_mfma_f32_16x16x32_bf16 in place of the fp32 MFMA\n");
    run<M_MFMA, 0, true>("mfma.bf", cpm);
    run<M_BOTH, 2, true>("both.bf", cpm); run<M_INWAVE, 2, true>("inwave.bf", cpm);
    run<M_BOTH, 4, true>("both.bf", cpm); run<M_INWAVE, 4, true>("inwave.bf", cpm);
    run<M_VALU, 4, true>("valu.bf", cpm); run<M_BOTH_PRIO, 4, true>("both.prio.bf", cpm); run<M_BOTH_SWAP, 4, true>("both.swap.bf", cpm);
    run<M_BOTH, 8, true>("both.bf", cpm); run<M_INWAVE, 8, true>("inwave.bf", cpm);
    return 0;
}
