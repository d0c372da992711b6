// Main-loop laboratory for the fp32-MFMA NT GEMM (C[M][N] = A[M][K] * B[N][K]^T) on gfx950: the production loop of
// wesup_amd/csrc/gemm.hip rebuilt with the block shape, the wave tile, the LDS ring depth and the barrier placement
// as template parameters, on real operands with a spot check against the host, so that a structure that wins here
// can be moved into the library as it is.  Not part of the product; built and run by hand:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_lab.hip -o tools/gemm_lab && tools/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const float* src, unsigned lds_byte_addr_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte_addr_uniform)
                 : "memory");
}
// buffer form of the LDS-DMA (VAR bit 11): SRD in SGPRs, one 32-bit offset VGPR per lane, the K-step offset in an SGPR:
// no vector address arithmetic per instruction
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bglds16(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_byte_addr_uniform) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 ::"v"(voff), "s"(srd), "s"(soff), "s"(lds_byte_addr_uniform) : "memory");
}
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
    const unsigned long a = (unsigned long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// timing experiments on the M0 handling (results wrong): no M0 write at all / write without save+restore
__device__ __forceinline__ void glds16_nom0(const float* src) {
    asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(src) : "memory");
}
__device__ __forceinline__ void glds16_norestore(const float* src, unsigned lds_byte_addr_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr_uniform) : "memory");
}
// register staging (VAR bit 9): plain loads issued through asm (outside the compiler's wait bookkeeping), the wait
// statement carries the registers so that no use can move in front of it
__device__ __forceinline__ void gload16(f32x4& dst, const float* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define vm_wait if (!(VAR & 32)) vm_wait_
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(unsigned long)(lptr_t)p; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// NW waves as WAVES_M x WAVES_N, each wave WM x WN sub-tiles of 32x32.  STAGES = LDS ring depth (2: wait for the
// DMA of step k+1 at the end of step k; 3: the DMA of step k+2 is issued in step k and stays in flight across the
// barrier).  VAR bit 0: pin fragment reads of group g+1 in front of the MFMAs of group g (sched_barrier).
// VAR bit 1: barrier in front of the last MFMA group of a K-step (the first reads of the next step hide under it).
// Ablations (results are then wrong on purpose): bit 2: no DMA in the loop; bit 3: every K-step stages K-step 0 again
// (sources stay cache-hot); bit 4: fragments are not read from LDS; bit 5: no wait for the DMA.
// VAR bit 10: after every barrier wave w sleeps ~64*w cycles.
// VAR bit 6: the DMA instructions of a K-step are spread over its MFMAs (16 slots) instead of issued in one burst.
__device__ unsigned long long g_clk[2 * 8192];   // per block: shader cycles and 100 MHz ticks of the main loop
template <int NW, int WAVES_M, int WM, int WN, int STAGES, int MINB, int VAR, int BK = 32>
__global__ __launch_bounds__(NW * 64, MINB) void lab_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K, int tiles_n,
                                                            int ntiles) {
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int CPR = BK / 4;                     // 16-byte chunks per row (8 at BK = 32, 4 at BK = 16)
    constexpr int PR = NW * 64 / CPR;               // rows per staging pass
    constexpr int SWS = (CPR == 8) ? 1 : 2;         // swizzle: chunk ^= (row >> SWS) & (CPR - 1)
    constexpr int RA = BM / PR, RB = BN / PR;
    constexpr int NDMA = RA + RB;
    static_assert(BM % PR == 0 && BN % PR == 0, "staging passes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE_FLOATS = (BM + BN) * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int srow = tid / CPR;
    const int schunk = (tid % CPR) ^ ((srow >> SWS) & (CPR - 1));
    const int wm0 = (wave / WAVES_N) * 32 * WM, wn0 = (wave % WAVES_N) * 32 * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int swa = ((wm0 + l31) >> SWS) & (CPR - 1), swb = ((wn0 + l31) >> SWS) & (CPR - 1);
    const int nk = K / BK;
    const int lt = xcd_remap(blockIdx.x, ntiles);
    const int tile_n = lt % tiles_n, tile_m = lt / tiles_n;
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;

    const float* a_src[RA];
    const float* b_src[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) a_src[i] = A + (long)(m_blk + srow + PR * i) * K + 4 * schunk;
#pragma unroll
    for (int j = 0; j < RB; ++j) b_src[j] = B + (long)(n_blk + srow + PR * j) * K + 4 * schunk;

    // DMAs [lo, hi) of the staging of K-step kk (A passes first, then B passes)
    f32x4 stg[(VAR & 512) ? NDMA : 1];
    unsigned a_voff[RA], b_voff[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) a_voff[i] = (unsigned)(((long)(m_blk + srow + PR * i) * K + 4 * schunk) * 4);
#pragma unroll
    for (int j = 0; j < RB; ++j) b_voff[j] = (unsigned)(((long)(n_blk + srow + PR * j) * K + 4 * schunk) * 4);
    const i32x4 srdA = make_srd(A, 0xffffffffu), srdB = make_srd(B, 0xffffffffu);
    auto stage_range = [&](int kk, int buf, int lo, int hi) {
        if (VAR & 2048) {
            const unsigned adst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STAGE_FLOATS + wave * 256));
            const unsigned bdst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STAGE_FLOATS + BM * BK + wave * 256));
            const unsigned soff = (unsigned)kk * BK * 4;
#pragma unroll
            for (int i = 0; i < RA; ++i)
                if (i >= lo && i < hi) bglds16(a_voff[i], srdA, soff, adst + i * PR * BK * 4);
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (RA + j >= lo && RA + j < hi) bglds16(b_voff[j], srdB, soff, bdst + j * PR * BK * 4);
            return;
        }
        if (VAR & 512) {
#pragma unroll
            for (int i = 0; i < RA; ++i)
                if (i >= lo && i < hi) gload16(stg[i], a_src[i] + kk * BK);
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (RA + j >= lo && RA + j < hi) gload16(stg[RA + j], b_src[j] + kk * BK);
            return;
        }
        const unsigned adst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STAGE_FLOATS + wave * 256));
        const unsigned bdst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STAGE_FLOATS + BM * BK + wave * 256));
#pragma unroll
        for (int i = 0; i < RA; ++i)
            if (i >= lo && i < hi) {
                if (VAR & 128) glds16_nom0(a_src[i] + kk * BK);
                else if (VAR & 256) glds16_norestore(a_src[i] + kk * BK, adst + i * PR * BK * 4);
                else glds16(a_src[i] + kk * BK, adst + i * PR * BK * 4);
            }
#pragma unroll
        for (int j = 0; j < RB; ++j)
            if (RA + j >= lo && RA + j < hi) {
                if (VAR & 128) glds16_nom0(b_src[j] + kk * BK);
                else if (VAR & 256) glds16_norestore(b_src[j] + kk * BK, bdst + j * PR * BK * 4);
                else glds16(b_src[j] + kk * BK, bdst + j * PR * BK * 4);
            }
    };

    auto stage = [&](int kk, int buf) {
        if (VAR & (128 | 256 | 512 | 2048)) { stage_range(kk, buf, 0, NDMA); return; }
        const unsigned adst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STAGE_FLOATS + wave * 256));
        const unsigned bdst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STAGE_FLOATS + BM * BK + wave * 256));
#pragma unroll
        for (int i = 0; i < RA; ++i) glds16(a_src[i] + kk * BK, adst + i * PR * BK * 4);
#pragma unroll
        for (int j = 0; j < RB; ++j) glds16(b_src[j] + kk * BK, bdst + j * PR * BK * 4);
    };

    auto stg_write = [&](int buf) {
        if (!(VAR & 512)) return;
#pragma unroll
        for (int d = 0; d < NDMA; ++d) asm volatile("" : "+v"(stg[d]));
        float* ad = smem + buf * STAGE_FLOATS + tid * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(ad + i * PR * BK) = stg[i];
#pragma unroll
        for (int j = 0; j < RB; ++j) *reinterpret_cast<f32x4*>(ad + BM * BK + j * PR * BK) = stg[RA + j];
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 fa[2][WM], fb[2][WN];
    auto load_frag = [&](const float* as, const float* bs, int g, int sl) {
        if (VAR & 16) {
#pragma unroll
            for (int i = 0; i < WM; ++i) fa[sl][i] = make_float4(1.f + lane, 2.f + g, 3.f, 4.f + i);
#pragma unroll
            for (int j = 0; j < WN; ++j) fb[sl][j] = make_float4(1.f + lane, 2.f + g, 3.f, 4.f + j);
            asm volatile("" : "+v"(fa[sl][0].x), "+v"(fb[sl][0].x));
            return;
        }
        const int ca = ((2 * g + lhi) ^ swa) << 2, cb = ((2 * g + lhi) ^ swb) << 2;
#pragma unroll
        for (int i = 0; i < WM; ++i) fa[sl][i] = ld4(as + 32 * i * BK + ca);
#pragma unroll
        for (int j = 0; j < WN; ++j) fb[sl][j] = ld4(bs + 32 * j * BK + cb);
    };
    auto mfma_group = [&](int sl, int g = 0, int kn = -1, int nbuf = 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((VAR & 64) && kn >= 0) {
                // spread staging: slot 4g+t of 16 carries its share of the NDMA instructions, pinned between the MFMAs
                constexpr int NSLOT = BK / 2;
                const int slot = 4 * g + t;
                stage_range(kn, nbuf, (slot * NDMA + NSLOT - 1) / NSLOT, ((slot + 1) * NDMA + NSLOT - 1) / NSLOT);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const float av = t == 0 ? fa[sl][i].x : t == 1 ? fa[sl][i].y : t == 2 ? fa[sl][i].z : fa[sl][i].w;
                    const float bv = t == 0 ? fb[sl][j].x : t == 1 ? fb[sl][j].y : t == 2 ? fb[sl][j].z : fb[sl][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
            if (VAR & 64) __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (STAGES == 2) {
        stage(0, 0);
        vm_wait<0>();
        stg_write(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        if (nk > 1) vm_wait<NDMA>(); else vm_wait<0>();
        __builtin_amdgcn_s_barrier();
    }
    int cur = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (VAR & 2) {
        const float* as0 = smem + (wm0 + l31) * BK;
        load_frag(as0, as0 + (BM - wm0 + wn0) * BK, 0, 0);
    }
    for (int kk = 0; kk < nk; ++kk) {
        const float* as = smem + cur * STAGE_FLOATS + (wm0 + l31) * BK;
        const float* bs = smem + cur * STAGE_FLOATS + BM * BK + (wn0 + l31) * BK;
        if (!(VAR & 2)) load_frag(as, bs, 0, 0);
        int kn = -1, nbuf = 0;
        if (VAR & 64) {
            if (STAGES == 2) { if (kk + 1 < nk) { kn = kk + 1; nbuf = cur ^ 1; } }
            else { if (kk + 2 < nk) { kn = kk + 2; nbuf = cur + 2 >= 3 ? cur - 1 : cur + 2; } }
        } else if (!(VAR & 4)) {
            if (STAGES == 2) {
                if (kk + 1 < nk) stage((VAR & 8) ? 0 : kk + 1, cur ^ 1);
            } else {
                int nb = cur + 2; if (nb >= 3) nb -= 3;
                if (kk + 2 < nk) stage((VAR & 8) ? 0 : kk + 2, nb);
            }
        }
        if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int sl = g & 1;
            if (g + 1 < BK / 8) {
                load_frag(as, bs, g + 1, sl ^ 1);
                if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
            } else if (VAR & 2) {
                // end-of-step synchronisation in front of the last MFMA group; first reads of the next buffer under it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (STAGES == 2) vm_wait<0>();
                else { if (kk + 2 < nk) vm_wait<NDMA>(); else vm_wait<0>(); }
                __builtin_amdgcn_s_barrier();
                int nx = cur + 1; if (nx >= STAGES) nx = 0;
                const float* as2 = smem + nx * STAGE_FLOATS + (wm0 + l31) * BK;
                const float* bs2 = smem + nx * STAGE_FLOATS + BM * BK + (wn0 + l31) * BK;
                if (kk + 1 < nk) load_frag(as2, bs2, 0, sl ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_group(sl, g, kn, nbuf);
            if (VAR & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (!(VAR & 2)) {
            if (STAGES == 2) vm_wait<0>();
            else { if (kk + 2 < nk) vm_wait<NDMA>(); else vm_wait<0>(); }
            if (kk + 1 < nk) stg_write(cur ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (VAR & 1024) {          // stagger the waves of the block so that their DMA slots do not meet in the TA
            if (wave & 1) __builtin_amdgcn_s_sleep(1);
            if (wave & 2) __builtin_amdgcn_s_sleep(2);
            if (wave & 4) __builtin_amdgcn_s_sleep(4);
        }
        ++cur; if (cur >= STAGES) cur = 0;
    }
    if (tid == 0) {
        g_clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        g_clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // plain epilogue (not what is being studied)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int n = n_blk + wn0 + 32 * j + l31;
                C[(long)m * N + n] = acc[i][j][r];
            }
}

static std::vector<float> hA, hB;
static float *dA, *dB, *dC;

template <int NW, int WAVES_M, int WM, int WN, int STAGES, int MINB, int VAR, int BK = 32>
static void run(const char* name, int M, int N, int K) {
    constexpr int WAVES_N = NW / WAVES_M;
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    if (M % BM || N % BN) { printf("%-34s M=%d N=%d: shape not a multiple of %dx%d\n", name, M, N, BM, BN); return; }
    const int tiles_n = N / BN, ntiles = (M / BM) * tiles_n;
    const size_t lds = (size_t)STAGES * (BM + BN) * BK * sizeof(float);
    auto kern = lab_kernel<NW, WAVES_M, WM, WN, STAGES, MINB, VAR, BK>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        printf("%-34s: cannot get %zu B of LDS\n", name, lds); return;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(dC, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NW * 64), lds, 0, dA, dB, dC, M, N, K, tiles_n, ntiles);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-34s: launch failed\n", name); exit(1); }
    // spot check
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), dC, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 64; ++s) {
        const int m = (int)((s * 2654435761u) % (unsigned)M), n = (int)((s * 40503u + 17) % (unsigned)N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
        worst = fmax(worst, fabs(ref - hC[(size_t)m * N + n]) / (fabs(ref) + 1.0));
    }
    const int reps = 10;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NW * 64), lds, 0, dA, dB, dC, M, N, K, tiles_n, ntiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    static unsigned long long hc[2 * 8192];
    (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc));
    double cyc = 0, ticks = 0;
    for (int b = 0; b < ntiles; ++b) { cyc += (double)hc[2 * b]; ticks += (double)hc[2 * b + 1]; }
    const double mhz = cyc / ticks * 100.0;
    // MFMA cycles a SIMD needs for one block-tile (x2 when two blocks share the CU)
    const double ideal = (double)(K / BK) * (WM * WN * (BK / 2)) * 64.0 * (NW * MINB / 4);
    const double pipe = ideal / (cyc / ntiles);
    printf("%-34s %dx%d tile, M=%6d N=%4d K=%5d tiles=%5d: %8.1f us %6.1f TFLOP/s  %4.0f MHz (peak %5.1f) loop-pipe %4.1f%%  err %.1e%s\n", name, BM, BN, M, N, K, ntiles,
           ms * 1e3, 2.0 * M * N * K / ms / 1e9, mhz, 157.3 * mhz / 2400.0, 100.0 * pipe, worst, worst > 1e-4 ? "  WRONG" : "");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int Mmax = 131072, Nmax = 512;
    const int K = argc > 1 ? atoi(argv[1]) : 2304;
    hA.resize((size_t)Mmax * K); hB.resize((size_t)Nmax * K);
    unsigned s = 12345;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dC, (size_t)Mmax * Nmax * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const int N = 256;
    for (int M : {57600, 131072}) {
        //   NW WAVES_M WM WN STAGES MINB VAR [BK]
        run<4, 2, 2, 2, 2, 2, 2048>("base 4w 2blk/CU (buffer DMA)", M, N, K);
        run<4, 2, 2, 2, 2, 3, 2048, 16>("128x128 BK16 3blk/CU", M, N, K);
        run<4, 2, 4, 2, 2, 2, 2048, 16>("4w 256x128 BK16 2blk/CU", M, N, K);
        run<4, 2, 2, 4, 2, 2, 2048, 16>("4w 128x256 BK16 2blk/CU", M, N, K);
        run<4, 2, 4, 2, 3, 2, 2048, 16>("4w 256x128 BK16 ring3 2blk/CU", M, N, K);
        run<8, 2, 4, 2, 2, 1, 2048>("8w 256x256 (128x64/wave) 1blk", M, N, K);
        printf("\n");
    }
    return 0;
}
