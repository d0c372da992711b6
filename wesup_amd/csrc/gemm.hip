// fp32 MFMA GEMM family for gfx950: implicit-GEMM 3x3 convolution (fwd + dgrad), plain NT GEMM
// (1x1 side convs, fc layers), TN GEMM with deterministic split-K (all weight gradients), column sums.
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak; 155.8 measured
// with operands in registers, tools/mfma_peak.hip).  One MFMA takes 64 cycles on its SIMD, so the kernels are
// paced by MFMA issue and everything else has to hide under it.
//
// Operand staging is LDS-DMA (buffer_load_dwordx4 ... lds; global_load_lds_dwordx4 for the image layer): a
// wave-instruction copies 64 x 16 B from per-lane addresses straight into 1 KiB of LDS, no staging registers and
// no ds_write.  round-2 probe mfma_ingredients.hip, tools/README.md
// measured why: with register staging hipcc either waits for the loads in front of the MFMA phase or sinks them
// behind it (100-124 TFLOP/s for the bare loop); with LDS-DMA the only wait is the vmcnt(0) that __syncthreads()
// emits, exactly at the end of the K-step (128-137 TFLOP/s).
//   * the LDS destination of a wave-instruction is linear (base + lane*16), so tiles are unpadded; bank conflicts
//     of the NT kernel's ds_read_b128 fragment reads are removed by XOR-swizzling the 16-B chunk index with
//     (row>>1)&7 -- on the per-lane SOURCE address when staging and on the read address (same involution);
//   * a lane whose element is outside the image / matrix is given an out-of-range buffer offset and the DMA writes
//     zeros for it (implicit zero padding; the 64-bit global form of the image layer reads a zeroed page instead);
//   * ReLU of the previous layer is applied to the fragments after the ds_read (compile-time kernel variant).
// What the loops cost and why they look the way they do: DESIGN.md 3.1 and 6 (round-2 probe gemm_lab.hip, tools/README.md).
#include "common.hpp"
#include "ldsdma.hpp"
#include <cstdlib>
#include <cstdio>

#define BK 32

__device__ float4 g_zero_page[16];      // 256 B of zeros, source of masked lanes
// Debug instrumentation, compiled ONLY into the debug library (make debug -> libwesup_hip_debug.so, loaded through
// WESUP_HIP_LIB by tools/): the shipped kernels carry no clock read and no trace store (tests/test_isa_cpu.py).
//   clock probe: block 0 of the last NT launch leaves {shader cycles, 100 MHz ticks} of its main loop; wesup_debug_clock()
//   per-block trace: when set, every block stores {start, loop start, loop end, end} in 100 MHz ticks
#ifdef WESUP_GEMM_DEBUG
__device__ unsigned long long g_clock_probe[2];
__device__ unsigned long long* g_trace = nullptr;
#define WESUP_DBG(...) __VA_ARGS__
#else
#define WESUP_DBG(...)
#endif

#define NT_FLAG_STREAM 0x100      // internal bit of NtParams::flags: the output is large and read once later -> streaming stores
struct NtParams {
    const float* A;
    const float* Bw;
    const float* bias;
    float* C;
    float* C2;           // optional second output: max(C, 0) (the ReLU'd copy the next layer reads), same layout as C
    const float* mask;
    int M, N, K;
    int lda, ldb, ldc, ldmask;
    int H, W, Cin, cin_shift;
    int pad0_;           // (explicit padding, zero: by-value kernel parameters carry no indeterminate bytes, launch.hpp)
    FastDiv dW, dH;      // pixel index -> (image, row, column) without integer division (conv modes)
    int flags;
    int tiles_m, tiles_n;
    int full_tiles;      // blocks [0, full_tiles) own one tile each
    int sk_parts;        // stream-K blocks behind them (0: none) ...
    int sk_steps;        // ... sharing this many K-steps of the remaining tiles
    float* sk_ws;        // [sk_parts][2][BM*BN] partial tiles
    // batched plain GEMMs (MODE 0, no stream-K): nbatch products of one shape in one launch; tile ids run batch-major
    int nbatch;              // 0 / 1: a single product
    int pad1_;
    FastDiv dTiles;          // tiles per product
    long batchA, batchB, batchC;   // element strides between the products' operands
    long batchBias;                // ... and between their bias vectors (0: one bias for all)
    // fused 1x1 side conv (SIDE instantiations, N == BN, no stream-K): side_out[m][0..N/2) = C[m][:] . side_w^T + side_bias
    const float* side_w;     // [N/2][N] row-major
    const float* side_bias;  // [N/2] or NULL
    float* side_out;
    int ld_side;
    int pad2_;
};

// ---------------------------------------------------------------------------------------------
// NT kernel:  C[M][N] = A[M][K] * Bw[N][K]^T   with A either a plain row-major matrix or the implicit
// im2col view of an NHWC tensor under a 3x3/pad-1 window (K order = (32-channel chunk, tap, channel in chunk)).
// MODE 0: plain; 1: conv3x3 with Cin % 32 == 0; 2: conv3x3 with Cin == 4 (image layer, K padded to 64); 3: plain, batched.
// LDS image of both operands: [row][32 floats], chunk c of row r stored at chunk position c ^ ((r>>1)&7).
// A lane's fragments for 4 consecutive MFMA k-steps are ONE ds_read_b128; MFMA t of a group uses element t of the
// A and the B fragment: lanes 0-31 then carry k = 8g+t, lanes 32-63 k = 8g+4+t (any consistent k order is fine).
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void nt_wait_newest() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }      // all but the newest stage
template <int NW, int BM, int BN, int WM, int WN, int MODE, int MINB, bool RELU, bool SIDE = false>
__global__ __launch_bounds__(NW * 64, MINB) void gemm_nt_kernel(const NtParams p) {
    constexpr int NT = NW * 64;                        // threads per block
    constexpr int PR = NW * 8;                         // rows per staging pass (8 threads fetch one 128-B row)
    constexpr int RA = BM / PR, RB = BN / PR;          // staging passes
    constexpr int WAVES_N = BN / (32 * WN);
    constexpr int LDC = BN + 4;
    static_assert((BM / (32 * WM)) * WAVES_N == NW, "wave grid covers the tile");
    static_assert(BM % PR == 0 && BN % PR == 0, "whole staging passes");
    // The 64 x 64 tile with MINB = 3 is the three-stage form: a K-step of this tile is 0.4 us of MFMA time, and beside the other
    // streams' kernels a staging DMA takes several times that -- two K-steps of lead instead of one (48 KiB: three blocks per CU).
    constexpr int STAGES = (BM == 64 && BN == 64 && MINB == 3) ? 3 : 2;
    constexpr int LDS_FLOATS = STAGES * (BM + BN) * BK;
    constexpr int EP = (BM * LDC + LDS_FLOATS - 1) / LDS_FLOATS;     // epilogue passes (C tile staged in row slabs)
    constexpr int HR = BM / EP;
    static_assert(HR % (32 * WM) == 0 && HR * LDC <= LDS_FLOATS, "epilogue slab fits in the staging buffers");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [STAGES][BM][BK]
    float* Bs = smem + STAGES * BM * BK;    // [STAGES][BN][BK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WESUP_DBG(const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();)
    const float* zero = reinterpret_cast<const float*>(g_zero_page);
    // ---- staging role of this lane: row (tid>>3) of each PR-row pass, chunk position tid&7; the logical chunk it
    // fetches is position ^ swizzle(row) (the swizzle does not depend on the pass: PR*i >> 1 == 0 mod 8)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const int wm0 = (wave / WAVES_N) * 32 * WM, wn0 = (wave % WAVES_N) * 32 * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int swa = ((wm0 + l31) >> 1) & 7, swb = ((wn0 + l31) >> 1) & 7;   // read-side swizzle (same for every i / j)
    const int nk = p.K / BK;
    WESUP_DBG(unsigned long long clk0 = 0, rt0 = 0, tr2 = 0;)

    // ---- work of this block.  Blocks [0, full_tiles) compute one whole tile each.  The tiles that would form a partial
    // last round (tile count not a multiple of the resident block slots) are cut stream-K style instead: their K-steps,
    // laid end to end, are divided evenly over sk_parts blocks, so a block works on the tail of one tile and the head of
    // the next, leaves both partial accumulator tiles in the workspace, and nt_fixup_kernel adds the pieces of each
    // tile in K order and applies the epilogue.  Every slot of the chip then finishes together.
    const bool sk = (int)blockIdx.x >= p.full_tiles;
    int seg_g = 0, seg_end = nk, lt_full = 0, part = 0, slot = 0;
    if (!sk) {
        lt_full = xcd_remap(blockIdx.x, p.full_tiles);
    } else {
        part = xcd_remap(blockIdx.x - p.full_tiles, p.sk_parts);       // full_tiles % 8 == 0
        seg_g = (int)((long)part * p.sk_steps / p.sk_parts);
        seg_end = (int)((long)(part + 1) * p.sk_steps / p.sk_parts);
    }

    while (seg_g < seg_end) {
    int lt = lt_full, ks = 0, ke = nk;
    if (sk) {
        const int t = seg_g / nk;
        lt = p.full_tiles + t;
        ks = seg_g - t * nk;
        ke = min(nk, ks + seg_end - seg_g);
    }
    // n fastest: the N-tiles of one pixel tile run together and share the activation rows through their XCD's L2
    // (the m-fastest order, which keeps a weight slab in L2 instead, measured within 1 %)
    long zA = 0, zB = 0, zC = 0, zBias = 0;
    if (MODE == 3) {      // batched plain GEMMs: an instantiation of its own (and a kernel name of its own in profiles)
        const int z = fast_div(lt, p.dTiles);
        lt -= z * p.tiles_m * p.tiles_n;
        zA = z * p.batchA; zB = z * p.batchB; zC = z * p.batchC; zBias = z * p.batchBias;
    }
    const int tile_n = lt % p.tiles_n, tile_m = lt / p.tiles_n;
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;

    // Staging addresses.  MODE 0/1 use the buffer form: descriptor base = first row of the tile (MODE 1: moved down by
    // one image row + one pixel so that the shifted taps have non-negative scalar offsets), per-lane byte offset inside
    // the tile, masked lanes at WESUP_OOB.  MODE 2 (image layer, per-lane taps) keeps the 64-bit global form.
    long a_off[RA];
    unsigned a_msk[RA], a_vo[RA], b_vo[RB];
    const int margin = (MODE == 1) ? (p.W + 1) * p.Cin : 0;
    constexpr bool PLAIN = (MODE == 0 || MODE == 3);
    const float* a_base = PLAIN ? p.A + zA + (long)m_blk * p.lda : p.A + ((long)m_blk * p.Cin - margin);
    const i32x4 srdA = make_srd(a_base), srdB = make_srd(p.Bw + zB + (long)n_blk * p.ldb);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m_blk + srow + PR * i;
        if (PLAIN) {
            a_vo[i] = (m < p.M) ? (unsigned)(((srow + PR * i) * p.lda + 4 * schunk) * 4) : WESUP_OOB;
            a_msk[i] = 0; a_off[i] = 0;
        } else {
            // (b*H + h, w) by one multiply-high each: a generic integer division costs ~30 vector instructions, and
            // the tile set-up is paid per tile by waves that share their SIMD with MFMA-bound neighbours
            const int t = fast_div(m, p.dW);
            const int w = m - t * p.W;
            const int h = t - fast_div(t, p.dH) * p.H;
            // bit t = tap t (row t/3 - 1, column t%3 - 1) lies inside the image: the outer product of two 3-bit masks
            const unsigned wm = (w > 0 ? 1u : 0u) | 2u | (w < p.W - 1 ? 4u : 0u);
            unsigned msk = (h > 0 ? wm : 0u) | (wm << 3) | (h < p.H - 1 ? wm << 6 : 0u);
            a_msk[i] = (m < p.M) ? msk : 0u;
            a_off[i] = (long)m * p.Cin;
            a_vo[i] = (unsigned)(((srow + PR * i) * p.Cin + 4 * schunk) * 4);
        }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int n = n_blk + srow + PR * j;
        b_vo[j] = (n < p.N) ? (unsigned)(((srow + PR * j) * p.ldb + 4 * schunk) * 4) : WESUP_OOB;
    }

    // Staging of tile kk into buffer buf; part q issues the q-th PR-row pass of A and of B (kept separable so that
    // the issue can be spread over the MFMA groups -- tried, no gain, see the main loop).
    constexpr int NPART = (RA > RB) ? RA : RB;
    auto stage_part = [&](int kk, int buf, int q) {
        // wave-uniform LDS byte addresses: rows 8*wave.. of pass q
        const unsigned adst = __builtin_amdgcn_readfirstlane(lds_addr(As + buf * BM * BK + wave * 256)) + q * PR * BK * 4;
        const unsigned bdst = __builtin_amdgcn_readfirstlane(lds_addr(Bs + buf * BN * BK + wave * 256)) + q * PR * BK * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            if (i != q) continue;
            if (PLAIN) {
                bglds16(a_vo[i], srdA, (unsigned)(kk * BK * 4), adst);
            } else if (MODE == 1) {
                // K order = (32-channel chunk, tap, channel in chunk): the 9 taps of one chunk are consecutive K-steps,
                // so the shifted re-reads of the same 128 B per pixel hit L2 (with the tap outermost every tap streamed
                // the whole activation tensor again: FETCH_SIZE 9x the tensor)
                const int chunk = kk / 9;
                const int tap = kk - 9 * chunk;
                const int toff = ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * p.Cin + chunk * BK + margin;   // >= 0
                bglds16(((a_msk[i] >> tap) & 1u) ? a_vo[i] : WESUP_OOB, srdA, (unsigned)(toff * 4), adst);
            } else {
                const int tap = kk * 8 + schunk;                     // logical chunk = tap (4 channels each)
                const int toff = ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * 4;
                glds16((tap < 9 && ((a_msk[i] >> tap) & 1u)) ? p.A + a_off[i] + toff : zero, adst);
            }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (j != q) continue;
            bglds16(b_vo[j], srdB, (unsigned)(kk * BK * 4), bdst);
        }
    };
    auto stage = [&](int kk, int buf) {
#pragma unroll
        for (int q = 0; q < NPART; ++q) stage_part(kk, buf, q);
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(ks, 0);
    if constexpr (STAGES == 3) {
        if (ks + 1 < ke) { stage(ks + 1, 1); nt_wait_newest<RA + RB>(); } else glds_wait();
    } else {
        glds_wait();
    }
    __syncthreads();
    WESUP_DBG(if (slot == 0) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); })
    // (staggering every second resident block by 0.5-4 K-steps; s_setprio around either phase, alternating between the two
    // co-resident blocks per K-step, or fixed per block: no effect on throughput)
    int cur = 0;
    for (int kk = ks; kk < ke; ++kk) {
        const float* as = As + cur * BM * BK + (wm0 + l31) * BK;
        const float* bs = Bs + cur * BN * BK + (wn0 + l31) * BK;
        // Fragment double buffering: the ds_reads of group g+1 are issued BEFORE the 16 MFMAs of group g, so their
        // LDS latency hides under ~1000 MFMA cycles; only the first group of a K-step waits right after its reads.
        // The DMA for the next tile is issued after the first group's reads so that those are not queued behind it.
        float4 fa[2][WM], fb[2][WN];
        auto load_frag = [&](int g, int sl) {
            const int ca = ((2 * g + lhi) ^ swa) << 2, cb = ((2 * g + lhi) ^ swb) << 2;
#pragma unroll
            for (int i = 0; i < WM; ++i) fa[sl][i] = ld4(as + 32 * i * BK + ca);
#pragma unroll
            for (int j = 0; j < WN; ++j) fb[sl][j] = ld4(bs + 32 * j * BK + cb);
        };
        load_frag(0, 0);
        // (issuing one staging part per MFMA group instead was measured 3-4 % slower: round-2 probe gemm_trace.py, tools/README.md)
        if constexpr (STAGES == 3) {
            if (kk + 2 < ke) stage(kk + 2, cur >= 1 ? cur - 1 : 2);       // (cur + 2) % 3: the buffer of step kk - 1, free since its barrier
        } else {
            if (kk + 1 < ke) stage(kk + 1, cur ^ 1);
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int sl = g & 1;
            if (g + 1 < BK / 8) load_frag(g + 1, sl ^ 1);
            // ReLU-on-load, a compile-time variant of the kernel (applied unconditionally as max(x, -inf) it cost the
            // layers that do not need it 5 %)
            if constexpr (RELU) {
                // kept together in front of the group's MFMAs (fenced): spread between the MFMAs by the scheduler the
                // same 8 instructions cost 1 % of the kernel
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    fa[sl][i].x = vmax1(fa[sl][i].x); fa[sl][i].y = vmax1(fa[sl][i].y);
                    fa[sl][i].z = vmax1(fa[sl][i].z); fa[sl][i].w = vmax1(fa[sl][i].w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // element t of the fragments = k-step t of the group; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const float av = t == 0 ? fa[sl][i].x : t == 1 ? fa[sl][i].y : t == 2 ? fa[sl][i].z : fa[sl][i].w;
                        const float bv = t == 0 ? fb[sl][j].x : t == 1 ? fb[sl][j].y : t == 2 ? fb[sl][j].z : fb[sl][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        if constexpr (STAGES == 3) {
            if (kk + 2 < ke) nt_wait_newest<RA + RB>(); else glds_wait();     // step kk + 1's tile has landed, kk + 2's may still fly
            __syncthreads();
            cur = cur == 2 ? 0 : cur + 1;
        } else {
        glds_wait();                // this wave's DMA for the next tile has landed ...
        __syncthreads();            // ... and so has everybody's; every wave is done reading buf[cur]
        // (timing-only diagnostic without this barrier: main loop only 5 % shorter -- barrier coupling is not the limiter)
        cur ^= 1;
        }
    }

#ifdef WESUP_GEMM_DEBUG
    if (slot == 0) {
        tr2 = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0 && tid == 0) {
            g_clock_probe[0] = __builtin_amdgcn_s_memtime() - clk0;
            g_clock_probe[1] = tr2 - rt0;
        }
    }
#endif
    // ---- epilogue through LDS: the accumulator tile (lane holds D[(r&3)+8*(r>>2)+4*lhi][l31] of each 32x32
    // sub-tile) is written to a [rows][BN+4] image, then every thread handles 16-byte pieces of full rows so that
    // bias / ReLU mask / accumulate / store all move 16 B per lane on contiguous row segments.  A stream-K block
    // stores the raw tile to its workspace slot instead.
    float* Cs = smem;
    const bool relu_out = p.flags & WESUP_RELU_OUT, accum = p.flags & WESUP_ACCUM, use_mask = p.flags & WESUP_MASK;
    const int nt = (p.flags & NT_FLAG_STREAM) ? 1 : 0;      // outputs written with streaming stores (common.hpp)
    constexpr int QN = BN / 4;                 // float4 pieces per row
    constexpr int ROWS_PER_PASS = NT / QN;
    const int cq = tid % QN, r0 = tid / QN;
    const int n = n_blk + 4 * cq;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bv = zero4;
    if (!SIDE && !sk && p.bias && n < p.N) bv = ld4(p.bias + zBias + n);
    float* raw = sk ? p.sk_ws + ((long)part * 2 + slot) * (BM * BN) : nullptr;
    // Fused side conv: the LDS image of the output rows (bias already added) is the A operand of a second, small GEMM
    // against the 1x1 side weights: wave (sr0, sc0) of an (HR/32) x (BN/64) grid computes 32 rows x 32 side channels
    // over K = BN.  Its B fragments (side_w[sc0 + l31][8g + 4*lhi ..+3], the same k order as the A reads below) come
    // straight from global memory (the whole matrix is <= 32 KiB and L2-resident) into the registers the main loop's
    // fragments no longer need, before the first barrier of the epilogue.
    constexpr int SWN = BN / 64;
    static_assert(!SIDE || ((HR / 32) * SWN == NW && BN % 64 == 0), "side-conv wave grid covers the epilogue slab");
    const int sr0 = (wave / (SWN > 0 ? SWN : 1)) * 32, sc0 = (wave % (SWN > 0 ? SWN : 1)) * 32;
    float4 sb[SIDE ? BN / 8 : 1];
    float bcol[WN];
    float sbias = 0.f;
    if constexpr (SIDE) {
#pragma unroll
        for (int g = 0; g < BN / 8; ++g) sb[g] = ld4(p.side_w + (long)(sc0 + l31) * BN + 8 * g + 4 * lhi);
#pragma unroll
        for (int j = 0; j < WN; ++j) bcol[j] = p.bias ? p.bias[wn0 + 32 * j + l31] : 0.f;
        if (p.side_bias) sbias = p.side_bias[sc0 + l31];
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        if (e > 0) __syncthreads();
        if (wm0 >= e * HR && wm0 < (e + 1) * HR) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(wm0 - e * HR + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDC + wn0 + 32 * j + l31] =
                            SIDE ? acc[i][j][r] + bcol[j] : acc[i][j][r];
        }
        __syncthreads();
        if constexpr (SIDE) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const float* arow = Cs + (sr0 + l31) * LDC + 4 * lhi;
#pragma unroll
            for (int g = 0; g < BN / 8; ++g) {
                const float4 a4 = ld4(arow + 8 * g);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, sb[g].x, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, sb[g].y, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, sb[g].z, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, sb[g].w, sacc, 0, 0, 0);
            }
            float* so = p.side_out + (long)(m_blk + e * HR + sr0 + 4 * lhi) * p.ld_side + sc0 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                if (m_blk + e * HR + sr0 + 4 * lhi + row < p.M) so[(long)row * p.ld_side] = sacc[r] + sbias;
            }
        }
        if (sk) {
#pragma unroll 4
            for (int rr = r0; rr < HR; rr += ROWS_PER_PASS)
                st4(raw + (long)(e * HR + rr) * BN + 4 * cq, ld4(Cs + rr * LDC + 4 * cq));
        } else if (n < p.N) {                       // N % 4 == 0
#pragma unroll 4
            for (int rr = r0; rr < HR; rr += ROWS_PER_PASS) {
                const int m = m_blk + e * HR + rr;
                if (m >= p.M) break;
                float4 v = ld4(Cs + rr * LDC + 4 * cq);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (relu_out) v = relu4(v);
                if (use_mask) {
                    const float4 mk = ld4(p.mask + (long)m * p.ldmask + n);
                    v.x = mk.x > 0.f ? v.x : 0.f;
                    v.y = mk.y > 0.f ? v.y : 0.f;
                    v.z = mk.z > 0.f ? v.z : 0.f;
                    v.w = mk.w > 0.f ? v.w : 0.f;
                }
                float* c = p.C + zC + (long)m * p.ldc + n;
                if (accum) {
                    const float4 o = ld4(c);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                st4s(c, v, nt);
                if (p.C2) st4s(p.C2 + (long)m * p.ldc + n, relu4(v), nt);
            }
        }
    }
    seg_g += ke - ks;
    ++slot;
    if (seg_g < seg_end) __syncthreads();          // the staging buffers double as the epilogue image
    }   // segments

#ifdef WESUP_GEMM_DEBUG
    if (g_trace && tid == 0) {
        unsigned long long* t = g_trace + 6 * (long)blockIdx.x;
        t[0] = tr0; t[1] = rt0; t[2] = tr2; t[3] = __builtin_amdgcn_s_memrealtime();
        t[4] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        t[5] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    }
#endif
}

// Stream-K fix-up: C tile = epilogue(sum of the partial tiles of the parts that covered it, in K order).
// One float4 per thread; the part boundaries are the same integer formula the GEMM blocks used.
template <int BM, int BN>
__global__ __launch_bounds__(256) void nt_fixup_kernel(const NtParams p) {
    constexpr int PER_TILE = BM * BN / 4 / 256;
    const int t = blockIdx.x / PER_TILE;
    const int piece = (blockIdx.x % PER_TILE) * 256 + threadIdx.x;
    const int rr = piece / (BN / 4), cq = piece % (BN / 4);
    const int nk = p.K / BK;
    const long W = p.sk_steps;
    const int P = p.sk_parts;
    auto first_step = [&](int q) { return (long)q * W / P; };
    auto part_of = [&](long s) {
        int q = (int)(s * P / W);
        while (q + 1 < P && first_step(q + 1) <= s) ++q;
        return q;
    };
    const int q0 = part_of((long)t * nk), q1 = part_of((long)t * nk + nk - 1);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = q0; q <= q1; ++q) {
        const int slot = ((int)(first_step(q) / nk) == t) ? 0 : 1;
        const float4 w = ld4(p.sk_ws + ((long)q * 2 + slot) * (BM * BN) + rr * BN + 4 * cq);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const int lt = p.full_tiles + t;
    const int tile_n = lt % p.tiles_n, tile_m = lt / p.tiles_n;
    const int m = tile_m * BM + rr, n = tile_n * BN + 4 * cq;
    if (m >= p.M || n >= p.N) return;
    if (p.bias) {
        const float4 bv = ld4(p.bias + n);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (p.flags & WESUP_RELU_OUT) v = relu4(v);
    if (p.flags & WESUP_MASK) {
        const float4 mk = ld4(p.mask + (long)m * p.ldmask + n);
        v.x = mk.x > 0.f ? v.x : 0.f;
        v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f;
        v.w = mk.w > 0.f ? v.w : 0.f;
    }
    float* c = p.C + (long)m * p.ldc + n;
    if (p.flags & WESUP_ACCUM) {
        const float4 o = ld4(c);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    st4(c, v);
    if (p.C2) st4(p.C2 + (long)m * p.ldc + n, relu4(v));
}

// Block shapes of the NT family.  BIG: 256x128, 8 waves (the same 64x64 wave tile), ONE block per CU: the eight waves
// run in lock step behind one barrier, so every tile of a round ends at the same time and the stream-K tail starts
// level on all CUs (with two independent 128x128 blocks per CU the older block of a pair wins the MFMA arbitration and
// finishes a 72-step tile ~100 us before its partner, which leaves the end of the launch ragged); a quarter less
// operand traffic per FLOP comes with it.  round-2 probe gemm_lab.hip, tools/README.md: +2-4 % over the 2-blocks-per-CU form on whole rounds.
// STD: 128x128, 4 waves, two blocks per CU.
enum NtShape { NT_BIG = 0, NT_STD = 1, NT_N64 = 2, NT_SMALL = 3 };

// Stream-K plan of an (M, N, K) problem under bm x bn tiles and `slots` resident blocks: how many tiles run whole,
// how many blocks share the K-steps of the rest.  parts == 0: plain tiling.
struct SkPlan {
    int full, parts, steps;
    size_t ws_bytes;
};
static SkPlan plan_streamk(int M, int N, int K, int bm, int bn, int slots) {
    SkPlan pl = {0, 0, 0, 0};
    const long tiles = (long)ceil_div(M, bm) * ceil_div(N, bn);
    const int nk = K / BK;
    pl.full = (int)tiles;
    const int R = (int)(tiles % slots);
    // worth it when the last round is visibly short of full and a part still gets >= 8 K-steps; a part must be shorter
    // than a tile (so that it touches at most two tiles: two workspace slots per part)
    if (N <= 64 || nk < 16 || R == 0 || R * 10 > slots * 9) return pl;
    // a short-K problem that does not even fill one round would go through the workspace and the fix-up as a whole
    // (side convs at 60x60 / 30x30, K = 512: 55 / 30 us): plain smaller tiles are faster there (38 / 14 us)
    if (tiles < slots && nk < 32) return pl;
    const long W = (long)R * nk;
    const int P = (int)(W / 8 < slots ? W / 8 : slots);
    if (P <= R) return pl;
    pl.full = (int)tiles - R;
    pl.parts = P;
    pl.steps = (int)W;
    pl.ws_bytes = (size_t)P * 2 * bm * bn * sizeof(float);
    return pl;
}

// Shape choice (shared by the workspace queries and the dispatch).  The step uses STD: alone BIG is 1-3 % faster on the
// 120x120 / 60x60 layers (conv forward 5.27 -> 5.25, dgrad 5.12 -> 5.03 ms per step), but a block that takes 96 KiB of
// LDS and all wave slots of its CU leaves no room for the kernels of the other two streams, and the 3-stream step
// goes from 20.0 to 20.8 ms.  WESUP_NT_SHAPE=big selects it for measurements.
struct NtChoice {
    NtShape shape;
    SkPlan sk;       // with_ws: the stream-K plan the dispatch will use if the caller passes the workspace
};
static bool nt_force_std() {
    static const int v = [] { const char* e = getenv("WESUP_NT_SHAPE"); return (e && e[0] == 'b') ? 0 : 1; }();
    return v != 0;
}
static NtChoice choose_nt(int M, int N, int K, bool with_ws) {
    NtChoice c;
    c.sk = SkPlan{0, 0, 0, 0};
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    const long t256 = (long)ceil_div(M, 256) * ceil_div(N, 128);
    if (N > 64 && !nt_force_std() && M >= 2048) {
        if (with_ws) {
            const SkPlan sk = plan_streamk(M, N, K, 256, 128, 256);
            if (sk.parts > 0) { c.shape = NT_BIG; c.sk = sk; return c; }
        }
        if (t256 >= 192) { c.shape = NT_BIG; return c; }
    }
    if (N > 64) {
        if (with_ws) {
            const SkPlan sk = plan_streamk(M, N, K, 128, 128, 512);
            if (sk.parts > 0) { c.shape = NT_STD; c.sk = sk; return c; }
        }
        if (t128 >= 384) { c.shape = NT_STD; return c; }
    }
    if (N <= 64 && (long)ceil_div(M, 128) >= 384) { c.shape = NT_N64; return c; }
    c.shape = NT_SMALL;
    return c;
}

// 64 x 64 tiles: a grid that is resident all at once (<= 512 blocks: three per CU leave room) takes the three-stage form -- it
// has no second round of blocks to cover a late DMA with, and beside the other streams' kernels DMAs are late (batch 1 at 480^2:
// 3.56 -> 3.50 ms; nothing at batch 4, whose small grids are the side GEMMs off the chain; larger grids lose occupancy: 8.28 ->
// 8.34 ms with every 64 x 64 launch on three stages).
static bool nt_three_stages(long blocks) { return blocks <= 512; }
template <int NW, int BM, int BN, int WM, int WN, int MODE, int MINB, bool RELU, bool SIDE = false>
static int launch_nt(NtParams p, hipStream_t st, const SkPlan* sk = nullptr, void* ws = nullptr) {
    p.tiles_m = ceil_div(p.M, BM);
    p.tiles_n = ceil_div(p.N, BN);
    p.full_tiles = p.tiles_m * p.tiles_n * (p.nbatch > 1 ? p.nbatch : 1);
    p.dTiles = make_fastdiv(p.tiles_m * p.tiles_n);
    p.sk_parts = 0; p.sk_steps = 0; p.sk_ws = nullptr;
    if (sk && sk->parts > 0) {
        p.full_tiles = sk->full; p.sk_parts = sk->parts; p.sk_steps = sk->steps; p.sk_ws = (float*)ws;
    }
    const size_t lds = (size_t)((BM == 64 && BN == 64 && MINB == 3) ? 3 : 2) * (BM + BN) * BK * sizeof(float);
    auto kern = gemm_nt_kernel<NW, BM, BN, WM, WN, MODE, MINB, RELU, SIDE>;
    if (lds > 64 * 1024) {           // more than the default dynamic LDS limit: raise it once per instantiation
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return WESUP_ERR_LAUNCH;
    }
    WESUP_LAUNCH(kern, dim3(p.full_tiles + p.sk_parts), dim3(NW * 64), lds, st, p);
    WESUP_CHECK_LAUNCH();
    if (p.sk_parts > 0) {
        const int rem_tiles = p.tiles_m * p.tiles_n - p.full_tiles;
        WESUP_LAUNCH((nt_fixup_kernel<BM, BN>), dim3(rem_tiles * (BM * BN / 1024)), dim3(256), 0, st, p);
        WESUP_CHECK_LAUNCH();
    }
    return WESUP_OK;
}

template <int MODE, bool RELU>
static int dispatch_nt_r(NtParams p, hipStream_t st, void* ws, size_t ws_bytes) {
    const bool ws_ok = ws && !((uintptr_t)ws & 15);
    NtChoice c = choose_nt(p.M, p.N, p.K, ws_ok);
    if (c.sk.parts > 0 && ws_bytes < c.sk.ws_bytes) c = choose_nt(p.M, p.N, p.K, false);   // workspace too small: plain tiling
    const SkPlan* sk = c.sk.parts > 0 ? &c.sk : nullptr;
    switch (c.shape) {
        case NT_BIG: return launch_nt<8, 256, 128, 2, 2, MODE, 1, RELU>(p, st, sk, ws);
        case NT_STD: return launch_nt<4, 128, 128, 2, 2, MODE, 2, RELU>(p, st, sk, ws);
        case NT_N64: return launch_nt<4, 128, 64, 2, 1, MODE, 2, RELU>(p, st);
        default:
            if (nt_three_stages((long)ceil_div(p.M, 64) * ceil_div(p.N, 64) * (p.nbatch > 1 ? p.nbatch : 1)) && (MODE == 0 || MODE == 3)) return launch_nt<4, 64, 64, 1, 1, MODE, 3, RELU>(p, st);
            return launch_nt<4, 64, 64, 1, 1, MODE, 2, RELU>(p, st);
    }
}
// conv forward with the layer's 1x1 side conv fused into the epilogue: one N-tile holds all output channels
template <int MODE, bool RELU>
static int dispatch_nt_side(NtParams p, hipStream_t st) {
    if (p.N == 64) return launch_nt<4, 128, 64, 2, 1, MODE, 2, RELU, true>(p, st);
    if constexpr (MODE == 1) {
        if (p.N == 128) return launch_nt<4, 128, 128, 2, 2, MODE, 2, RELU, true>(p, st);
    }
    return WESUP_ERR_INVALID;
}
template <int MODE>
static int dispatch_nt(NtParams p, hipStream_t st, void* ws, size_t ws_bytes) {
    if (p.side_out) {
        if constexpr (MODE == 0) return WESUP_ERR_INVALID;
        else return (p.flags & WESUP_RELU_IN) ? dispatch_nt_side<MODE, true>(p, st) : dispatch_nt_side<MODE, false>(p, st);
    }
    return (p.flags & WESUP_RELU_IN) ? dispatch_nt_r<MODE, true>(p, st, ws, ws_bytes)
                                     : dispatch_nt_r<MODE, false>(p, st, ws, ws_bytes);
}

static int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

extern "C" size_t wesup_gemm_nt_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK)) return 0;
    return choose_nt(M, N, K, true).sk.ws_bytes;
}

extern "C" int wesup_gemm_nt(const float* A, int lda, const float* B, int ldb, const float* bias, float* C,
                             int ldc, const float* mask, int ldmask, int M, int N, int K, int flags,
                             void* ws, size_t ws_bytes, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || (K % BK) || (lda % 4) || (ldb % 4) || (N % 4) || (ldc % 4) ||
        (ldmask % 4) || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)mask | (uintptr_t)bias) & 15))
        return WESUP_ERR_INVALID;
    if ((flags & WESUP_MASK) && !mask) return WESUP_ERR_INVALID;
    NtParams p = {};
    p.A = A; p.Bw = B; p.bias = bias; p.C = C; p.mask = mask;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldmask = ldmask;
    p.flags = flags;
    return dispatch_nt<0>(p, (hipStream_t)stream, ws, ws_bytes);
}

// host-synchronous debug helpers (NOT part of the hot path; they only do something in the debug library, the shipped
// one answers WESUP_ERR_INVALID): in-kernel clock of the last NT GEMM launch in MHz
extern "C" int wesup_debug_clock(double* mhz_out) {
#ifndef WESUP_GEMM_DEBUG
    (void)mhz_out;
    return WESUP_ERR_INVALID;
#else
    unsigned long long h[2] = {0, 0};
    if (!mhz_out) return WESUP_ERR_INVALID;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clock_probe), sizeof(h)) != hipSuccess) return WESUP_ERR_LAUNCH;
    *mhz_out = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    return WESUP_OK;
#endif
}

extern "C" int wesup_debug_set_trace(void* device_buf /* >= 48 B per block of the next launches, or NULL */) {
#ifndef WESUP_GEMM_DEBUG
    (void)device_buf;
    return WESUP_ERR_INVALID;
#else
    unsigned long long* ptr = (unsigned long long*)device_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &ptr, sizeof(ptr)) == hipSuccess ? WESUP_OK : WESUP_ERR_LAUNCH;
#endif
}

extern "C" int wesup_conv3x3_kpad(int Ci) {
    const int cip = Ci < 4 ? 4 : Ci;
    return (9 * cip + BK - 1) / BK * BK;
}

struct SideConv {
    const float* w;      // [Cout/2][Cout]
    const float* bias;   // [Cout/2] or NULL
    float* out;          // [B*H*W] rows of ld floats
    int ld;
};

static int conv_common(const float* x, const float* w, const float* bias, float* y, const float* mask, int B,
                       int H, int W, int Cin, int Cout, int flags, void* ws, size_t ws_bytes, hipStream_t st,
                       float* y_relu = nullptr, const SideConv* side = nullptr) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0) return WESUP_ERR_INVALID;
    const bool small = (Cin == 4);
    if (!small && (Cin < 32 || (Cin & (Cin - 1)))) return WESUP_ERR_INVALID;
    if ((long)B * H * W * (long)(Cin > Cout ? Cin : Cout) >= (1l << 31)) return WESUP_ERR_INVALID;
    NtParams p = {};
    if (y_relu && ((uintptr_t)y_relu & 15)) return WESUP_ERR_INVALID;
    p.A = x; p.Bw = w; p.bias = bias; p.C = y; p.C2 = y_relu; p.mask = mask;
    p.M = B * H * W; p.N = Cout; p.K = wesup_conv3x3_kpad(Cin);
    p.lda = Cin; p.ldb = p.K; p.ldc = Cout; p.ldmask = Cout;
    p.H = H; p.W = W; p.Cin = Cin; p.cin_shift = ilog2(Cin);
    p.dW = make_fastdiv(W); p.dH = make_fastdiv(H);
    p.flags = flags | ((!(flags & WESUP_ACCUM) && wino_nt_stores(4.0 * p.M * Cout * (y_relu ? 2 : 1))) ? NT_FLAG_STREAM : 0);
    if (side) {
        if (!side->w || !side->out || (Cout != 64 && Cout != 128) || side->ld < Cout / 2 ||
            (((uintptr_t)side->w | (uintptr_t)side->out | (uintptr_t)side->bias) & 15))
            return WESUP_ERR_INVALID;
        p.side_w = side->w; p.side_bias = side->bias; p.side_out = side->out; p.ld_side = side->ld;
    }
    return small ? dispatch_nt<2>(p, st, nullptr, 0) : dispatch_nt<1>(p, st, ws, ws_bytes);
}

// workspace of the forward implicit GEMM (stream-K partial tiles) for (Cin -> Cout); for dgrad call it with the
// channel counts swapped (the input of that GEMM is dy)
extern "C" size_t wesup_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin < 32 || Cout <= 0) return 0;
    return choose_nt(B * H * W, Cout, 9 * Cin, true).sk.ws_bytes;
}

extern "C" int wesup_conv3x3_fwd(const float* x, const float* w_fwd, const float* bias, float* y, float* y_relu, int B,
                                 int H, int W, int Cin, int Cout, int relu_in, void* ws, size_t ws_bytes, void* stream) {
    return conv_common(x, w_fwd, bias, y, nullptr, B, H, W, Cin, Cout, relu_in ? WESUP_RELU_IN : 0, ws, ws_bytes,
                       (hipStream_t)stream, y_relu);
}

// conv3x3 forward of a layer with Cout in {64, 128} together with its 1x1 side conv (models/wesup.py:246-266: the
// hook's tap is the conv output, the side conv maps it to Cout/2 channels): side_out[pixel][0..Cout/2) =
// y[pixel][:] . side_w^T + side_bias, computed from the output tile while it is still in LDS, so y is not read again.
extern "C" int wesup_conv3x3_fwd_side(const float* x, const float* w_fwd, const float* bias, float* y, float* y_relu,
                                      const float* side_w, const float* side_bias, float* side_out, int ld_side, int B,
                                      int H, int W, int Cin, int Cout, int relu_in, void* stream) {
    const SideConv side = {side_w, side_bias, side_out, ld_side};
    return conv_common(x, w_fwd, bias, y, nullptr, B, H, W, Cin, Cout, relu_in ? WESUP_RELU_IN : 0, nullptr, 0,
                       (hipStream_t)stream, y_relu, &side);
}

extern "C" int wesup_conv3x3_dgrad(const float* dy, const float* w_dgrad, const float* mask_src, float* dx, int B,
                                   int H, int W, int Cin, int Cout, int accumulate, void* ws, size_t ws_bytes,
                                   void* stream) {
    // the same implicit GEMM with the roles of the channel counts swapped: input dy has Cout channels
    int flags = (mask_src ? WESUP_MASK : 0) | (accumulate ? WESUP_ACCUM : 0);
    return conv_common(dy, w_dgrad, nullptr, dx, mask_src, B, H, W, Cout, Cin, flags, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// TN kernel (weight gradients):  C[M][N] = sum_k A[k][M] * B[k][N], k = pixel / row index.
// MODE 0: plain matrices.  MODE 1: conv3x3 wgrad, grid.x also enumerates the 9 taps; B row k is the
// input pixel shifted by the tap.  MODE 2: conv3x3 wgrad for the 4-channel image: N = 9 taps x 4.
// Both operands are m-contiguous in memory ([pixel][channel]), so the LDS image is k-major [32 k][BM] and the MFMA
// fragments of a lane for the WM sub-tiles of its wave are one conflict-free ds_read of 4*WM bytes (lane l reads
// T[2*kp + (l>>5)][m0 + WM*(l&31) ...]: the sub-tiles interleave the rows of the wave tile).  Rows are unpadded:
// a wave-instruction of the LDS-DMA fills 1 KiB = 256 consecutive floats of the image.
// Split-K over grid.y; every split writes its own slab (deterministic), reduced by a second kernel.
// Bias gradients ride along: out[m] = sum_k A[k][m] is accumulated from the staged A tile by the blocks of the first
// N-tile (and first tap) and stored behind the slab, so the activations' gradient is not read a second time.
// ---------------------------------------------------------------------------------------------
struct TnParams {
    const float* A;
    const float* Bx;
    float* slab;     // [S][slab_stride]: M x Nslab partial products, then M column sums of A (bias gradient)
    long slab_stride;
    int want_colsum;
    int colsum_batch;                  // want_colsum: only the blocks of this batch entry sum A's columns (-1: every entry)
    long batchA, batchB, batch_slab;   // grid.z = batch index: element strides of A, B and the slab stack per batch entry
    int M, N, K;     // N: columns per tap (MODE 1) or total
    int lda, ldb;
    int Nslab;       // slab row length
    int H, W;
    FastDiv dW, dH;
    int relu_b;
    int tiles_m, tiles_n, taps;
    int k_per_split;  // multiple of BK
    int w_old, w_young;   // > 0: weighted split of a single-round grid (see the kernel); 0: equal splits
    int xcd_group;        // walk the grid XCD by XCD in tile-fastest order (see the kernel)
};

template <int BM, int BN, int WM, int WN, int MODE_, bool RELU, bool TINY>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
    constexpr int MODE = (MODE_ == 3) ? 0 : MODE_;     // 3 = plain like 0: the batched Winograd-domain products, under a kernel name of their own
    constexpr int QA = BM / 4, RPA = 256 / QA, NA = BK / RPA;
    constexpr int QB = BN / 4, RPB = 256 / QB, NB = BK / RPB;
    constexpr int WAVES_N = BN / (32 * WN);
    static_assert((BM / (32 * WM)) * WAVES_N == 4, "4 waves per block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;         // [2][BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    WESUP_DBG(const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();)
    // Block -> (tile, split, batch entry).  xcd_group: the grid is walked so that each XCD gets a contiguous range of the
    // tile-fastest order -- the tiles of one (batch entry, split), which read the same K range of both operands, then sit
    // behind ONE L2 instead of being dealt round-robin over eight (blocks b and b + 8 share an XCD).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_group) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int L = xcd_remap(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        const int q = L / gx;
        bx = L - q * gx;
        bz = q / gy;
        by = q - bz * gy;
    }
    int lt = bx;
    const int tile_n = lt % p.tiles_n;
    lt /= p.tiles_n;
    const int tile_m = lt % p.tiles_m;
    const int tap = lt / p.tiles_m;          // 0 for MODE 0/2
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    // K range of this split.  A grid of at most 512 blocks is dispatched as "blocks 0..255 one per CU, blocks 256.. into
    // the second slot of each CU", and of the two blocks that share a CU the one that arrived first wins the MFMA
    // arbitration (round-2 probe tn_trace.py, tools/README.md: with equal ranges the first blocks end at 500 us, their partners at 657 us, the
    // last quarter of the launch running one block per CU at two thirds of the pipe).  So the ranges are weighted:
    // a block of the first 256 gets w_old K-steps for every w_young of a later one, every tile's splits still
    // partition [0, K) exactly, and the pairs end together.  Static, so the summation order stays fixed.
    int k_begin, k_end;
    if (p.w_old > 0) {
        const int gx = gridDim.x, S = gridDim.y, y = by;                  // (weighted splits: never with xcd_group)
        const int T = (p.K + BK - 1) / BK;
        const int n_o = ((int)blockIdx.x < 256) ? min(S, (255 - (int)blockIdx.x) / gx + 1) : 0;   // splits of this tile among the first 256 blocks
        const long wtot = (long)p.w_old * n_o + (long)p.w_young * (S - n_o);
        const long w0 = (long)p.w_old * min(y, n_o) + (long)p.w_young * max(0, y - n_o);
        const long w1 = (long)p.w_old * min(y + 1, n_o) + (long)p.w_young * max(0, y + 1 - n_o);
        k_begin = (int)(T * w0 / wtot) * BK;
        k_end = min(p.K, (int)(T * w1 / wtot) * BK);
    } else {
        k_begin = by * p.k_per_split;
        k_end = min(p.K, k_begin + p.k_per_split);
    }
    const float* Ab = p.A + (long)bz * p.batchA;
    const float* Bb = p.Bx + (long)bz * p.batchB;

    const int qa = tid % QA, ra_row = tid / QA;
    const int qb = tid % QB, rb_row = tid / QB;
    const bool a_col_ok = (m_blk + 4 * qa) < p.M;      // M, N multiples of 4
    const bool b_col_ok = (n_blk + 4 * qb) < p.N;
    const int dh = (MODE == 1) ? tap / 3 - 1 : 0, dw = (MODE == 1) ? tap % 3 - 1 : 0;
    // MODE 2: column quad qb is tap qb (4 channels each), only 9 of the 16 quads are real
    const int dh2 = qb / 3 - 1, dw2 = qb % 3 - 1;

    // Staging by the buffer form of the LDS-DMA.  Descriptor bases are the first row of this block's K range (B: shifted
    // by the tap), the per-lane offsets inside that range never change, a K-step adds a scalar offset.  The A operand
    // and the plain B operand need no per-step vector work at all: rows >= k_end fall behind num_records and columns
    // outside the matrix carry WESUP_OOB, and the DMA writes zeros for both.  The shifted B rows of the convolution
    // modes need the border test of their pixel: mask_b() turns a K-step into the per-lane offset-or-OOB (integer VALU
    // work only: pixel -> (b, h, w), border tests), computed one K-step ahead in slices between the MFMA groups so that
    // it stays in their shadow; issue() fires the DMA.
    const int rows = k_end - k_begin;
    unsigned a_vo[NA], b_vo[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_vo[i] = a_col_ok ? (unsigned)(((ra_row + RPA * i) * p.lda + 4 * qa) * 4) : WESUP_OOB;
    const i32x4 srdA = make_srd(Ab + (long)k_begin * p.lda + m_blk, (unsigned)rows * (unsigned)p.lda * 4u);
    const float* b_base;
    unsigned b_nr = WESUP_OOB;
    if (MODE == 0) {
        b_base = Bb + (long)k_begin * p.ldb + n_blk;
        b_nr = (unsigned)rows * (unsigned)p.ldb * 4u;
    } else if (MODE == 1) {
        b_base = Bb + ((long)k_begin + dh * p.W + dw) * p.ldb + n_blk;
    } else {
        b_base = Bb + ((long)k_begin - (p.W + 1)) * 4;          // per-lane tap shifts are >= -(W+1) pixels
    }
    const i32x4 srdB = make_srd(b_base, b_nr);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (MODE == 2) b_vo[i] = (unsigned)((rb_row + RPB * i + dh2 * p.W + dw2 + p.W + 1) * 16);
        else b_vo[i] = b_col_ok ? (unsigned)(((rb_row + RPB * i) * p.ldb + 4 * qb) * 4) : WESUP_OOB;
    }
    // The 32 pixels of a K-step are consecutive, so their image coordinates follow from those of the first one by a
    // carry or two: the two divisions of "pixel -> (b, h, w)" are done ONCE per K-step on the scalar unit (k0 is
    // wave-uniform) and each lane only adds its row, wraps w (twice when an image row is shorter than the 32-pixel
    // step) and h, and compares -- ~10 vector instructions per staged row instead of ~25.
    auto mask_b = [&](int k0, int i) -> unsigned {
        if (MODE == 0) return b_vo[i];
        if constexpr (TINY) {                  // images with W < 16 or H < 2 (a kernel variant of their own, so that the
                                               // loop below stays free of branches and fully unrolled): more carries
                                               // than the fast path handles; divide per lane
            const int k = k0 + rb_row + RPB * i;
            const int t = fast_div(k, p.dW);
            const int w = k - t * p.W;
            const int h = t - fast_div(t, p.dH) * p.H;
            const int hh = h + (MODE == 1 ? dh : dh2), ww = w + (MODE == 1 ? dw : dw2);
            const bool ok = (MODE == 1 ? b_col_ok : (qb < 9)) & (k < k_end) & (hh >= 0) & (hh < p.H) & (ww >= 0) & (ww < p.W);
            return ok ? b_vo[i] : WESUP_OOB;
        }
        const int k0u = __builtin_amdgcn_readfirstlane(k0);
        const int t0 = fast_div(k0u, p.dW);    // b*H + h of the first pixel of the step (scalar)
        const int w0 = k0u - t0 * p.W;
        const int h0 = t0 - fast_div(t0, p.dH) * p.H;
        const int r = rb_row + RPB * i;        // this lane's pixel inside the step, < 32
        int w = w0 + r, h = h0;
        const bool c1 = w >= p.W;
        w = c1 ? w - p.W : w; h = c1 ? h + 1 : h;
        const bool c2 = w >= p.W;              // only possible when W < 32 (W >= 16 here: two carries suffice)
        w = c2 ? w - p.W : w; h = c2 ? h + 1 : h;
        h = h >= p.H ? h - p.H : h;            // next image (two carries cannot pass a whole image: H >= 2)
        const int hh = h + (MODE == 1 ? dh : dh2), ww = w + (MODE == 1 ? dw : dw2);
        const bool ok = (MODE == 1 ? b_col_ok : (qb < 9)) & (k0 + r < k_end) & ((unsigned)hh < (unsigned)p.H) &
                        ((unsigned)ww < (unsigned)p.W);
        return ok ? b_vo[i] : WESUP_OOB;
    };
    // MODE 1 (every lane of the block has the same tap): whether pixel k0 + j is inside the image under the tap does
    // not depend on the lane's column, so ONE evaluation per K-step -- lane j & 31 tests pixel k0 + j -- is balloted into
    // a scalar 32-bit mask and a staged row only tests its own bit (3 vector instructions instead of ~25 per row: the
    // wgrad loop carried 140 vector instructions per 64 MFMAs against 52 in the forward kernel, and vector
    // instructions do not overlap with MFMAs on a SIMD).
    auto step_mask = [&](int k0) -> unsigned {
        const int k0u = __builtin_amdgcn_readfirstlane(k0);
        const int t0 = fast_div(k0u, p.dW);
        const int w0 = k0u - t0 * p.W;
        const int h0 = t0 - fast_div(t0, p.dH) * p.H;
        const int r = lane & 31;
        int w = w0 + r, h = h0;
        const bool c1 = w >= p.W;
        w = c1 ? w - p.W : w; h = c1 ? h + 1 : h;
        const bool c2 = w >= p.W;
        w = c2 ? w - p.W : w; h = c2 ? h + 1 : h;
        h = h >= p.H ? h - p.H : h;
        const bool ok = (k0u + r < k_end) & ((unsigned)(h + dh) < (unsigned)p.H) & ((unsigned)(w + dw) < (unsigned)p.W);
        return (unsigned)__builtin_amdgcn_ballot_w64(ok);          // lanes 0..31 (32..63 repeat them)
    };
    unsigned row_bit[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) row_bit[i] = 1u << (rb_row + RPB * i);
    const unsigned step_a = (unsigned)(BK * p.lda * 4), step_b = (unsigned)(BK * (MODE == 2 ? 4 : p.ldb) * 4);
    auto issue = [&](int step, const unsigned (&vb)[NB], int buf) {
        // the image is linear in tid (4*tid floats per pass): wave-uniform LDS byte addresses
        const unsigned adst = __builtin_amdgcn_readfirstlane(lds_addr(As + buf * BK * BM + wave * 256));
        const unsigned bdst = __builtin_amdgcn_readfirstlane(lds_addr(Bs + buf * BK * BN + wave * 256));
#pragma unroll
        for (int i = 0; i < NA; ++i) bglds16(a_vo[i], srdA, (unsigned)step * step_a, adst + i * 4096);
#pragma unroll
        for (int i = 0; i < NB; ++i) bglds16(vb[i], srdB, (unsigned)step * step_b, bdst + i * 4096);
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave / WAVES_N) * 32 * WM, wn0 = (wave % WAVES_N) * 32 * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nk = (k_end - k_begin + BK - 1) / BK;

    const bool do_cs = p.want_colsum && (p.colsum_batch < 0 || bz == p.colsum_batch) && tile_n == 0 &&
                       tap == 0 && tid < BM;     // wave-uniform (BM % 64 == 0)
    float csum = 0.f;
    unsigned vb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) vb[i] = mask_b(k_begin, i);
    if (nk > 0) issue(0, vb, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i) vb[i] = mask_b(k_begin + BK, i);      // offsets of K-step 1 (all out of range beyond k_end)
    glds_wait();
    __syncthreads();
    WESUP_DBG(const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();)
    int cur = 0;
    unsigned smask = 0;
    for (int kk = 0; kk < nk; ++kk) {
        issue(kk + 1, vb, cur ^ 1);                // unconditional (past k_end every lane is out of range: zeros)
        const int k2 = k_begin + (kk + 2) * BK;    // its offsets are consumed by the next iteration's issue()
        if (do_cs) {
            const float* col = As + cur * BK * BM + tid;
#pragma unroll
            for (int kr = 0; kr < BK; ++kr) csum += col[kr * BM];
        }
        // Fragment mapping: MFMA sub-tile i of the wave takes rows wm0 + WM*a + i (a = A-lane), so the WM operands of a
        // lane are adjacent in the k-major image and come with ONE ds_read of 4*WM bytes (ds_read_b64 for the 64x64
        // wave tile: twice the LDS rate of ds_read_b32 and half the instructions); same for the columns.
        const float* as = As + cur * BK * BM + lhi * BM + wm0 + WM * l31;
        const float* bs = Bs + cur * BK * BN + lhi * BN + wn0 + WN * l31;
        // fragment double buffering in groups of two k-pairs: the reads of group g+1 are in flight while the 2*WM*WN
        // MFMAs of group g issue
        constexpr int GK = 2, NG = BK / 2 / GK;
        typedef float fragA __attribute__((ext_vector_type(WM)));
        typedef float fragB __attribute__((ext_vector_type(WN)));
        fragA a[2][GK];
        fragB b[2][GK];
        auto load_group = [&](int g, int sl) {
#pragma unroll
            for (int u = 0; u < GK; ++u) {
                a[sl][u] = *reinterpret_cast<const fragA*>(as + 2 * (GK * g + u) * BM);
                b[sl][u] = *reinterpret_cast<const fragB*>(bs + 2 * (GK * g + u) * BN);
            }
        };
        load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int sl = g & 1;
            if (g + 1 < NG) load_group(g + 1, sl ^ 1);
#pragma unroll
            for (int u = 0; u < GK; ++u) {
                if constexpr (RELU) {          // ReLU of the layer input
#pragma unroll
                    for (int j = 0; j < WN; ++j) b[sl][u][j] = vmax1(b[sl][u][j]);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][u][i], b[sl][u][j], acc[i][j], 0, 0, 0);
                // one slice of the border arithmetic per k-pair, fenced so that it stays in this group's shadow
                const int kp = GK * g + u;
                if constexpr (MODE == 1 && !TINY) {
                    if (kp == 0) smask = step_mask(k2);
                    else if (kp <= NB) vb[kp - 1] = (smask & row_bit[kp - 1]) ? b_vo[kp - 1] : WESUP_OOB;
                } else if (MODE != 0 && kp < NB) {
                    vb[kp] = mask_b(k2, kp);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        glds_wait();
        __syncthreads();
        cur ^= 1;
    }

    WESUP_DBG(const unsigned long long tr2 = __builtin_amdgcn_s_memrealtime();)
    float* slab = p.slab + (long)bz * p.batch_slab + (long)by * p.slab_stride;
    if (do_cs && m_blk + tid < p.M) slab[(long)p.M * p.Nslab + m_blk + tid] = csum;
    const int ncol0 = (MODE == 1) ? tap * p.N : 0;
    // accumulator register r of sub-tile (i, j): row wm0 + WM*q + i with q = (r&3) + 8*(r>>2) + 4*lhi, column
    // wn0 + WN*l31 + j -- the WN columns of a lane are adjacent: one 4*WN-byte store
    const int n0 = n_blk + wn0 + WN * l31;
    if (n0 < p.N) {                      // N % 4 == 0 and WN | 4: n0 < N implies the whole piece is inside
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm0 + WM * ((r & 3) + 8 * (r >> 2) + 4 * lhi) + i;
                if (m >= p.M) continue;
                float* d = slab + (long)m * p.Nslab + ncol0 + n0;
                if constexpr (WN == 2) {
                    *reinterpret_cast<float2*>(d) = make_float2(acc[i][0][r], acc[i][1][r]);
                } else {
#pragma unroll
                    for (int j = 0; j < WN; ++j) d[j] = acc[i][j][r];
                }
            }
        }
    }
#ifdef WESUP_GEMM_DEBUG
    if (g_trace && tid == 0) {         // per-block timeline, as in the NT kernel
        const long bid = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
        unsigned long long* t = g_trace + 6 * bid;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memrealtime();
        t[4] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        t[5] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    }
#endif
}

// C[m][n] = sum_s slab[s][m][n]     (plain)            -- or, for conv weights --
// dW[co][ci][t] = sum_s slab[s][co][t*Cs + ci]          (torch (Co,Ci,3,3) layout; Cs = slab channels per tap)
__global__ void tn_reduce_kernel(const float* slab, long stride, float* C, int ldc, int M, int N, int Nslab, int S,
                                 float* colsum_out, long batch_slab, long batchC) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    slab += (long)blockIdx.y * batch_slab;             // grid.y = batch index
    C += (long)blockIdx.y * batchC;
    if (idx >= (long)M * N) {
        const long m = idx - (long)M * N;              // the column sums of A sit behind each slab
        if (colsum_out && m < M) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += slab[(long)k * stride + (long)M * Nslab + m];
            colsum_out[m] = s;
        }
        return;
    }
    const int m = idx / N, n = idx - (long)m * N;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += slab[(long)k * stride + (long)m * Nslab + n];
    C[(long)m * ldc + n] = s;
}
__global__ void wgrad_reduce_kernel(const float* slab, long stride, float* dw, int Co, int Ci, int Cs, int Nslab, int S,
                                    float* db) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Co * Ci) {
        const long m = idx - (long)Co * Ci;
        if (db && m < Co) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += slab[(long)k * stride + (long)Co * Nslab + m];
            db[m] = s;
        }
        return;
    }
    const int co = idx / Ci, ci = idx - (long)co * Ci;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int k = 0; k < S; ++k) {
        const float* s = slab + (long)k * stride + (long)co * Nslab + ci;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += s[t * Cs];
    }
    float* d = dw + ((long)co * Ci + ci) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) d[t] = acc[t];
}

// Many slabs (tiny outputs under a long K, e.g. the 32x64 side-conv gradients over 921600 pixels: S > 1000): a
// thread per output element walking all S slabs serially took 100-500 us.  The slab stack is a [S][M*Nslab] matrix
// and its reduction a column sum, so a first pass of colsum_stage1 folds groups of slabs into <= 16 partial slabs
// (same layout) and the final kernel only walks those.  Fixed order, deterministic.
__global__ void colsum_stage1(const float* __restrict__ A, int lda, float* __restrict__ part, int M, int N, int rows_per);
static int slab_fold_chunks(int S) { return S > 32 ? 16 : 0; }          // 0: reduce the slabs directly
static size_t slab_fold_bytes(int S, size_t slab_elems) {
    return slab_fold_chunks(S) ? (size_t)slab_fold_chunks(S) * slab_elems * sizeof(float) : 0;
}
// returns the slab stack the final reduce kernel should read, and its depth
static const float* slab_fold(const float* slab, int& S, long slab_elems, float* part, hipStream_t st) {
    const int chunks = slab_fold_chunks(S);
    if (!chunks) return slab;
    const int rows_per = ceil_div(S, chunks);
    WESUP_LAUNCH(colsum_stage1, dim3((unsigned)ceil_div(slab_elems, 64l), chunks), dim3(256), 0, st, slab,
                       (int)slab_elems, part, S, (int)slab_elems, rows_per);
    S = ceil_div(S, rows_per);
    return part;
}

struct TnPlan {
    int bm, bn;       // 128x128 or 64x64
    int tiles_m, tiles_n, taps, S, k_per_split, Nslab;
};

static TnPlan plan_tn(int M, int N, int K, int taps, int nbatch = 1) {
    TnPlan pl;
    bool big = (M >= 128 && N >= 128);
    // Few output tiles under a long K (the deep side convs' weight gradients: 256 x 512 outputs over 14 400 / 3 600 pixels; the
    // commuted ones: 128 x 256 over B*Kmax rows): eight 128 x 128 tiles needed 50 splits to fill the chip -- more slabs than the
    // reduce kernel walks, so a fold launch (colsum_stage1: 130 - 200 us in the step for 26 MB) came on top.  64 x 64 tiles give
    // four times the tiles: <= 32 splits fill the slots, no fold, four waves per SIMD hide the staging latency (round 5;
    // WESUP_TN_SMALL_RULE=0: the old plan).
    static const int small_rule = [] { const char* e = getenv("WESUP_TN_SMALL_RULE"); return e ? atoi(e) : 1; }();
    if (big && small_rule) {
        const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128) * taps * nbatch;
        const int ks = ceil_div(K, BK), ms = ks / 8 > 32 ? 32 : (ks / 8 > 0 ? ks / 8 : 1);
        if (t128 * ms < 384) big = false;
    }
    pl.bm = big ? 128 : 64;
    pl.bn = big ? 128 : 64;
    pl.tiles_m = ceil_div(M, pl.bm);
    pl.tiles_n = ceil_div(N, pl.bn);
    pl.taps = taps;
    const int tiles = pl.tiles_m * pl.tiles_n * taps * nbatch;
    const int ksteps = ceil_div(K, BK);
    // Split-K factor: the grid (tiles x S blocks, all of equal cost) should fill the resident block slots of the chip
    // a whole number of times -- 2 blocks/CU for the 128x128 tile (64 KiB LDS each), 4 for 64x64 -- so that no
    // partial round is left at the end.  Among 1..3 rounds pick the fullest (starting from 2-4 rounds instead measured
    // 0-4 % slower); every split keeps >= 8 K-steps.
    const int slots = 256 * (big ? 2 : 4);
    const int maxS = ksteps / 8 > 0 ? ksteps / 8 : 1;
    int S = 1;
    double best = -1.0;
    for (int rounds = 1; rounds <= 3; ++rounds) {
        int cand = (slots * rounds) / tiles;
        if (cand < 1) cand = 1;
        if (cand > maxS) cand = maxS;
        const int blocks = tiles * cand;
        const int r = ceil_div(blocks, slots);
        const double fill = (double)blocks / ((double)r * slots);
        // prefer fuller rounds; among equal fills prefer fewer splits (less slab traffic)
        if (fill > best + 0.02) { best = fill; S = cand; }
    }
    // A grid that fills the chip's slots without splitting is not split at all (round 5): a fuller last round did not pay for
    // S slabs written and read again (696 tiles of Wm^T g at 60 x 60: S = 2 bought 0.91 instead of 0.68 of three rounds and
    // cost 176 MB and a reduce launch of its own on the way to the first input gradient); with S = 1 and no column sums the
    // plain entries store straight into C.  WESUP_TN_S1_MIN_TILES moves the bound for the A/B (0: never).
    static const int s1_min = [] { const char* e = getenv("WESUP_TN_S1_MIN_TILES"); return e ? atoi(e) : 600; }();
    if (s1_min > 0 && tiles >= s1_min) S = 1;
    if (small_rule && S > 32 && (long)tiles * 32 >= 192) S = 32;        // no fold launch where 32 splits keep most CUs busy
    const int steps_per = ceil_div(ksteps, S);
    pl.k_per_split = steps_per * BK;
    pl.S = ceil_div(K, pl.k_per_split);
    pl.Nslab = N * taps;
    return pl;
}

static long grid_blocks(const TnPlan& pl) { return (long)pl.tiles_m * pl.tiles_n * pl.taps * pl.S; }
template <int MODE>
static int launch_tn(TnParams p, const TnPlan& pl, hipStream_t st, int nbatch = 1) {
    p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n; p.taps = pl.taps; p.k_per_split = pl.k_per_split;
    p.Nslab = pl.Nslab;
    // buffer-form staging: the byte offsets inside one split's K range must stay below the 2 GiB out-of-range marker
    const long ldmax = p.lda > p.ldb ? p.lda : p.ldb;
    // weighted splits (single round of 128x128 blocks, one product): default 3 : 2, WESUP_TN_WEIGHTS="a:b" overrides
    // (measured at the bench shape: 1:1 5.38, 5:4 5.33, 3:2 5.29, 25:16 5.30, 2:1 5.32 ms of wgrad per step)
    p.w_old = p.w_young = 0;
    if (pl.bm == 128 && nbatch == 1 && pl.S >= 2 && (long)grid_blocks(pl) <= 512) {
        static const int wts = [] {
            const char* e = getenv("WESUP_TN_WEIGHTS");
            int a = 3, b = 2;
            if (e && sscanf(e, "%d:%d", &a, &b) == 2 && a >= 0 && b > 0 && a < 1024 && b < 1024) return a * 1024 + b;
            return 3 * 1024 + 2;
        }();
        p.w_old = wts / 1024; p.w_young = wts % 1024;
    }
    // XCD grouping where several tiles share a K range and the weighted split (which reasons about dispatch order) is off;
    // WESUP_TN_XCD=0 switches it off for the A/B
    static const int xcd_on = [] { const char* e = getenv("WESUP_TN_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
    p.xcd_group = (xcd_on && p.w_old == 0 && (long)pl.tiles_m * pl.tiles_n * pl.taps > 1 &&
                   (long)grid_blocks(pl) * nbatch < (1l << 30)) ? 1 : 0;
    const long kmax = p.w_old > 0 ? 2l * pl.k_per_split : pl.k_per_split;      // a weighted range stays below twice the mean
    if ((kmax + 2 * BK) * ldmax * 4 + (1l << 22) >= (1l << 31)) return WESUP_ERR_INVALID;
    dim3 grid(pl.tiles_m * pl.tiles_n * pl.taps, pl.S, nbatch);
    const bool tiny = (MODE == 1 || MODE == 2) && (p.W < 16 || p.H < 2);
#define WESUP_TN_LAUNCH(BM_, WM_, RELU_, TINY_)                                                                        \
    WESUP_LAUNCH((gemm_tn_kernel<BM_, BM_, WM_, WM_, MODE, RELU_, TINY_>), grid, dim3(256),                     \
                       (size_t)2 * BK * (BM_ + BM_) * sizeof(float), st, p)
#define WESUP_TN_PICK(BM_, WM_)                                                                                        \
    do {                                                                                                               \
        if (tiny) { if (p.relu_b) WESUP_TN_LAUNCH(BM_, WM_, true, (MODE == 1 || MODE == 2)); else WESUP_TN_LAUNCH(BM_, WM_, false, (MODE == 1 || MODE == 2)); } \
        else { if (p.relu_b) WESUP_TN_LAUNCH(BM_, WM_, true, false); else WESUP_TN_LAUNCH(BM_, WM_, false, false); }  \
    } while (0)
    if (pl.bm == 128) WESUP_TN_PICK(128, 2);
    else WESUP_TN_PICK(64, 1);
#undef WESUP_TN_PICK
#undef WESUP_TN_LAUNCH
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

static size_t tn_slab_stride(int M, int Nslab) { return (size_t)M * Nslab + M; }    // products + column sums of A
// the epilogue's 8-byte stores can go straight into C: one split, nothing summed behind the slabs, rows of C 8-byte aligned
static bool tn_direct(const TnPlan& pl, const float* colsum_a, const float* C, int ldc) {
    return pl.S == 1 && !colsum_a && (ldc % 2) == 0 && (((uintptr_t)C) & 7) == 0;
}

extern "C" size_t wesup_gemm_tn_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan pl = plan_tn(M, N, K, 1);
    const size_t elems = tn_slab_stride(M, pl.Nslab);
    return align_up((size_t)pl.S * elems * sizeof(float), 256) + slab_fold_bytes(pl.S, elems);
}

extern "C" int wesup_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* colsum_a,
                             int M, int N, int K, int relu_b, void* ws, size_t ws_bytes, void* stream) {
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0 || (M % 4) || (N % 4) || (lda % 4) || (ldb % 4) ||
        (((uintptr_t)A | (uintptr_t)B) & 15))
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_gemm_tn_workspace_bytes(M, N, K)) return WESUP_ERR_WORKSPACE;
    const TnPlan pl = plan_tn(M, N, K, 1);
    TnParams p = {};
    p.A = A; p.Bx = B; p.slab = (float*)ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.relu_b = relu_b; p.H = 1; p.W = 1; p.dW = make_fastdiv(1); p.dH = make_fastdiv(1);
    p.slab_stride = (long)tn_slab_stride(M, pl.Nslab); p.want_colsum = colsum_a != nullptr; p.colsum_batch = -1;
    hipStream_t st = (hipStream_t)stream;
    if (tn_direct(pl, colsum_a, C, ldc)) {          // one split, no column sums: the tiles are the result
        p.slab = C; p.Nslab = ldc;
        TnPlan pd = pl; pd.Nslab = ldc;
        return launch_tn<0>(p, pd, st);
    }
    int rc = launch_tn<0>(p, pl, st);
    if (rc) return rc;
    const long tot = (long)M * N;
    int S = pl.S;
    float* part = (float*)((char*)ws + align_up((size_t)pl.S * p.slab_stride * sizeof(float), 256));
    const float* src = slab_fold((const float*)ws, S, p.slab_stride, part, st);
    WESUP_LAUNCH(tn_reduce_kernel, dim3((unsigned)((tot + M + 255) / 256)), dim3(256), 0, st, src, p.slab_stride, C,
                       ldc, M, N, pl.Nslab, S, colsum_a, 0l, 0l);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// nbatch independent products of one shape in one launch (grid.z) + one reduce: C_b = A_b^T . B_b with element strides
// between the batch entries.  Used for the per-image interpolation-pooling GEMMs (4 small products per resolution).
extern "C" size_t wesup_gemm_tn_batched_workspace_bytes(int nbatch, int M, int N, int K) {
    if (nbatch <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan pl = plan_tn(M, N, K, 1, nbatch);
    return (size_t)nbatch * pl.S * tn_slab_stride(M, pl.Nslab) * sizeof(float);
}
extern "C" int wesup_gemm_tn_batched(const float* A, int lda, long strideA, const float* B, int ldb, long strideB, float* C,
                                     int ldc, long strideC, int nbatch, int M, int N, int K, int relu_b, void* ws,
                                     size_t ws_bytes, void* stream) {
    if (!A || !B || !C || !ws || nbatch <= 0 || nbatch > 65535 || M <= 0 || N <= 0 || K <= 0 || (M % 4) || (N % 4) ||
        (lda % 4) || (ldb % 4) || (strideA % 4) || (strideB % 4) || (((uintptr_t)A | (uintptr_t)B) & 15))
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_gemm_tn_batched_workspace_bytes(nbatch, M, N, K)) return WESUP_ERR_WORKSPACE;
    const TnPlan pl = plan_tn(M, N, K, 1, nbatch);
    TnParams p = {};
    p.A = A; p.Bx = B; p.slab = (float*)ws; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    p.relu_b = relu_b; p.H = 1; p.W = 1; p.dW = make_fastdiv(1); p.dH = make_fastdiv(1);
    p.slab_stride = (long)tn_slab_stride(M, pl.Nslab); p.want_colsum = 0; p.colsum_batch = -1;
    p.batchA = strideA; p.batchB = strideB; p.batch_slab = (long)pl.S * p.slab_stride;
    hipStream_t st = (hipStream_t)stream;
    if (tn_direct(pl, nullptr, C, ldc) && strideC % 2 == 0) {
        p.slab = C; p.batch_slab = strideC;
        TnPlan pd = pl; pd.Nslab = ldc;
        return launch_tn<0>(p, pd, st, nbatch);
    }
    int rc = launch_tn<0>(p, pl, st, nbatch);
    if (rc) return rc;
    const long tot = (long)M * N;
    WESUP_LAUNCH(tn_reduce_kernel, dim3((unsigned)((tot + 255) / 256), nbatch), dim3(256), 0, st, (const float*)ws,
                       p.slab_stride, C, ldc, M, N, pl.Nslab, pl.S, (float*)nullptr, p.batch_slab, strideC);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ---- column sums (bias gradients): two passes of one kernel, fixed order, 16 B per lane
// block = 16 row-groups x 16 column quads (64 columns); each thread sums its rows of the chunk, the 16 row-groups
// are combined through LDS in a fixed order.  Pass 2 runs the same kernel over the [chunks][N] partial matrix.
__global__ void colsum_stage1(const float* __restrict__ A, int lda, float* __restrict__ part, int M, int N,
                              int rows_per) {
    __shared__ float4 sh[16][16];
    const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int n = blockIdx.x * 64 + 4 * q;
    const int m0 = blockIdx.y * rows_per;
    const int m1 = min(M, m0 + rows_per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        for (int m = m0 + rg; m < m1; m += 16) {
            const float4 v = ld4(A + (long)m * lda + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[rg][q] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            const float4 v = sh[r][q];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        st4(part + (long)blockIdx.y * N + n, s);
    }
}
static int colsum_chunks(int M, int N) {
    int chunks = 2048 / ceil_div(N, 64);
    const int maxc = ceil_div(M, 64);
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    return chunks;
}
extern "C" size_t wesup_colsum_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)colsum_chunks(M, N) * N * sizeof(float);
}
extern "C" int wesup_colsum(const float* A, int lda, float* out, int M, int N, void* ws, size_t ws_bytes,
                            void* stream) {
    if (!A || !out || !ws || M <= 0 || N <= 0 || (N % 4) || (lda % 4) || ((uintptr_t)A & 15)) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_colsum_workspace_bytes(M, N)) return WESUP_ERR_WORKSPACE;
    const int chunks = colsum_chunks(M, N);
    const int rows_per = ceil_div(M, chunks);
    hipStream_t st = (hipStream_t)stream;
    WESUP_LAUNCH(colsum_stage1, dim3(ceil_div(N, 64), chunks), dim3(256), 0, st, A, lda, (float*)ws, M, N,
                       rows_per);
    WESUP_LAUNCH(colsum_stage1, dim3(ceil_div(N, 64), 1), dim3(256), 0, st, (const float*)ws, N, out, chunks, N,
                       chunks);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

// ---- conv3x3 wgrad entry
static TnPlan plan_wgrad(int B, int H, int W, int Ci, int Cout) {
    const int K = B * H * W;
    if (Ci < 32) {                     // image layer: tensor has 4 channels, N = 64 covers 9 taps x 4
        TnPlan pl = plan_tn(Cout, 64, K, 1);
        return pl;
    }
    return plan_tn(Cout, Ci, K, 9);
}
extern "C" size_t wesup_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Cout <= 0) return 0;
    const TnPlan pl = plan_wgrad(B, H, W, Ci, Cout);
    const size_t elems = tn_slab_stride(Cout, pl.Nslab);
    return align_up((size_t)pl.S * elems * sizeof(float), 256) + align_up(slab_fold_bytes(pl.S, elems), 256);
}
extern "C" int wesup_conv3x3_wgrad(const float* x, const float* dy, float* dw_kcrs, float* db, int B, int H, int W,
                                   int Ci, int Cout, int relu_in, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw_kcrs || !ws || B <= 0 || H <= 0 || W <= 0) return WESUP_ERR_INVALID;
    const bool small = Ci < 32;
    if (small && Ci != 3) return WESUP_ERR_INVALID;
    if (!small && (Ci & (Ci - 1))) return WESUP_ERR_INVALID;
    if (Cout % 32) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_conv3x3_wgrad_workspace_bytes(B, H, W, Ci, Cout)) return WESUP_ERR_WORKSPACE;
    const TnPlan pl = plan_wgrad(B, H, W, Ci, Cout);
    hipStream_t st = (hipStream_t)stream;
    TnParams p = {};
    p.A = dy; p.Bx = x; p.slab = (float*)ws; p.M = Cout; p.K = B * H * W; p.lda = Cout;
    p.relu_b = relu_in; p.H = H; p.W = W; p.dW = make_fastdiv(W); p.dH = make_fastdiv(H);
    p.slab_stride = (long)tn_slab_stride(Cout, pl.Nslab); p.want_colsum = db != nullptr; p.colsum_batch = -1;
    int rc;
    if (small) {
        p.N = 64; p.ldb = 4;
        rc = launch_tn<2>(p, pl, st);
    } else {
        p.N = Ci; p.ldb = Ci;
        rc = launch_tn<1>(p, pl, st);
    }
    if (rc) return rc;
    const long tot = (long)Cout * Ci;
    // slab rows: [co][tap*Cs + ci] with Cs = 4 for the image layer (Nslab = 64), Ci otherwise; db rides behind them
    const size_t slab_b = align_up((size_t)pl.S * p.slab_stride * sizeof(float), 256);
    int S = pl.S;
    const float* src = slab_fold((const float*)ws, S, p.slab_stride, (float*)((char*)ws + slab_b), st);
    WESUP_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((tot + Cout + 255) / 256)), dim3(256), 0, st, src,
                       p.slab_stride, dw_kcrs, Cout, Ci, small ? 4 : Ci, pl.Nslab, S, db);
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}


// ---------------------------------------------------------------------------------------------
// Winograd F(m x m, 3x3)-domain convolutions (m = 2 or 4): the GEMM side.  The transforms between activations / filters and
// the [P positions][tiles][C] operands (P = (m+2)^2) are in winograd.hip (entries wesup_winograd_*); here are the batched
// products and the three conv passes that chain transforms and products.  Per m x m output tile
//     Y = A^T [ (G g G^T) o (B^T d B) ] A ,
// so with V = B^T d B and U = G g G^T the sum over input channels is, for each of the P positions p, ONE matrix product
//     forward:          M_p[tiles][Cout] = V_p[tiles][Cin] . U_p[Cout][Cin]^T           (NT, K = Cin)
//     input gradient:   the same over dy with the rotated filter, channel roles swapped   (NT, K = Cout)
//     weight gradient:  dU_p[Cout][Cin]  = dM_p[tiles][Cout]^T . V_p[tiles][Cin]         (TN, K = tiles, split-K)
// i.e. P x (pixels / m^2) multiply-adds per (co, ci) pair instead of 9 x pixels: 2.25x less MFMA work for 4x the operand
// bytes with m = 2, 4x less for 2.25x the bytes with m = 4, and two memory-bound transform passes.  It pays where the direct
// kernel is MFMA-bound and the channel counts make the products' arithmetic intensity high: from 128 input channels up
// (tools/wino_table.py).
// ---------------------------------------------------------------------------------------------
#include "winograd.hpp"

// workspace layout: [V: P T Ci][dM: P T Cout][slabs: P S (Cout Ci + Cout)]
extern "C" size_t wesup_conv3x3_wgrad_winograd_workspace_bytes(int B, int H, int W, int Ci, int Cout, int m) {
    if (!wino_shape_ok(B, H, W, Ci, Cout, m)) return 0;
    const long T = wino_tiles(B, H, W, m);
    const int P = wino_positions(m);
    const TnPlan pl = plan_tn(Cout, Ci, (int)T, 1, P);
    return align_up((size_t)P * T * Ci * sizeof(float), 256) + align_up((size_t)P * T * Cout * sizeof(float), 256) +
           align_up((size_t)P * pl.S * tn_slab_stride(Cout, pl.Nslab) * sizeof(float), 256) +
           wesup_winograd_outgrad_workspace_bytes(B, H, W, Cout, m);
}
// The same dW / db as wesup_conv3x3_wgrad (torch autograd of Conv2d(k=3, pad=1), models/wesup.py:199; the scheme of the
// non-fused Winograd backward-filter algorithms of vendor conv libraries).  The bias gradient: m = 2, the column sum of dM
// at position (1,1) (A dY A^T there = the sum of a tile's gradients), taken from the staged A tiles of that batch entry;
// m = 4 (no such position in its point set), summed per block inside the outgrad transform from the values it loads.
extern "C" int wesup_conv3x3_wgrad_winograd(const float* x, const float* v_pre, const float* dy, float* dw_kcrs, float* db,
                                            int B, int H, int W, int Ci, int Cout, int relu_in, int m, void* ws,
                                            size_t ws_bytes, void* stream) {
    if ((!x && !v_pre) || !dy || !dw_kcrs || !ws || !wino_shape_ok(B, H, W, Ci, Cout, m) ||
        (((uintptr_t)x | (uintptr_t)v_pre | (uintptr_t)dy | (uintptr_t)ws) & 15))
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_conv3x3_wgrad_winograd_workspace_bytes(B, H, W, Ci, Cout, m)) return WESUP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long T = wino_tiles(B, H, W, m);
    const int P = wino_positions(m);
    float* V = (float*)ws;
    float* dM = (float*)((char*)ws + align_up((size_t)P * T * Ci * sizeof(float), 256));
    float* slab = (float*)((char*)dM + align_up((size_t)P * T * Cout * sizeof(float), 256));
    int rc;
    if (v_pre) {             // the transformed input the Winograd forward of this layer kept
        V = const_cast<float*>(v_pre);
    } else if ((rc = wesup_winograd_input_transform(x, V, 0, B, H, W, Ci, relu_in, m, stream))) {
        return rc;
    }
    const TnPlan pl = plan_tn(Cout, Ci, (int)T, 1, P);
    // bias gradient: m = 2 from the TN GEMM (column sums of position (1,1)); m = 4 as per-block rows of the outgrad transform
    // that the filter-gradient reduce folds (or, for channel counts the block sum does not cover, a column sum of dy)
    float* ows = (float*)((char*)slab + align_up((size_t)P * pl.S * tn_slab_stride(Cout, pl.Nslab) * sizeof(float), 256));
    const int bias_rows = (m == 4 && db) ? (int)wino4_bias_rows(B, H, W, Cout) : 0;
    if ((rc = wino_outgrad_launch(dy, dM, bias_rows ? ows : nullptr, B, H, W, Cout, m, stream))) return rc;
    if (m == 4 && db && !bias_rows &&
        (rc = wesup_colsum(dy, Cout, db, B * H * W, Cout, ows, wesup_colsum_workspace_bytes(B * H * W, Cout), stream)))
        return rc;
    TnParams p = {};
    p.A = dM; p.Bx = V; p.slab = slab; p.M = Cout; p.N = Ci; p.K = (int)T; p.lda = Cout; p.ldb = Ci;
    p.relu_b = 0; p.H = 1; p.W = 1; p.dW = make_fastdiv(1); p.dH = make_fastdiv(1);
    p.slab_stride = (long)tn_slab_stride(Cout, pl.Nslab); p.want_colsum = (db != nullptr && m == 2); p.colsum_batch = 5;
    p.batchA = T * Cout; p.batchB = T * Ci; p.batch_slab = (long)pl.S * p.slab_stride;
    if ((rc = launch_tn<3>(p, pl, st, P))) return rc;
    return wino_filter_grad_launch(slab, p.slab_stride, p.batch_slab, pl.S, dw_kcrs, (m == 2 || bias_rows) ? db : nullptr, Cout,
                                   Ci, m, bias_rows ? ows : nullptr, bias_rows, stream);
}

// The F(4x4) weight gradient from operands that exist already: v_pre [36][tiles][Ci] (the forward's kept transformed input)
// and dm_pre [36][tiles][Cout] with its bias rows (wesup_winograd_dual_transform, which the layer's input gradient shares):
// the batched TN products and the filter-gradient reduce only.  db needs bias_part (bias_rows = wesup_winograd_bias_rows).
// Workspace: wesup_conv3x3_wgrad_winograd_workspace_bytes (the slabs; its V / dM regions stay unused).
extern "C" int wesup_conv3x3_wgrad_winograd_pre(const float* v_pre, const float* dm_pre, const float* bias_part, int bias_rows,
                                                float* dw_kcrs, float* db, int B, int H, int W, int Ci, int Cout, void* ws,
                                                size_t ws_bytes, void* stream) {
    const int m = 4;
    if (!v_pre || !dm_pre || !dw_kcrs || !ws || !wino_shape_ok(B, H, W, Ci, Cout, m) || (db && (!bias_part || bias_rows <= 0)) ||
        (bias_part && bias_rows != (int)wino4_bias_rows(B, H, W, Cout)) ||
        (((uintptr_t)v_pre | (uintptr_t)dm_pre | (uintptr_t)bias_part | (uintptr_t)ws) & 15))
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_conv3x3_wgrad_winograd_workspace_bytes(B, H, W, Ci, Cout, m)) return WESUP_ERR_WORKSPACE;
    const long T = wino_tiles(B, H, W, m);
    const int P = wino_positions(m);
    float* slab = (float*)((char*)ws + align_up((size_t)P * T * Ci * sizeof(float), 256) + align_up((size_t)P * T * Cout * sizeof(float), 256));
    const TnPlan pl = plan_tn(Cout, Ci, (int)T, 1, P);
    TnParams p = {};
    p.A = dm_pre; p.Bx = v_pre; p.slab = slab; p.M = Cout; p.N = Ci; p.K = (int)T; p.lda = Cout; p.ldb = Ci;
    p.relu_b = 0; p.H = 1; p.W = 1; p.dW = make_fastdiv(1); p.dH = make_fastdiv(1);
    p.slab_stride = (long)tn_slab_stride(Cout, pl.Nslab); p.want_colsum = false; p.colsum_batch = 5;
    p.batchA = T * Cout; p.batchB = T * Ci; p.batch_slab = (long)pl.S * p.slab_stride;
    int rc;
    if ((rc = launch_tn<3>(p, pl, (hipStream_t)stream, P))) return rc;
    return wino_filter_grad_launch(slab, p.slab_stride, p.batch_slab, pl.S, dw_kcrs, db, Cout, Ci, m, db ? bias_part : nullptr,
                                   db ? bias_rows : 0, stream);
}

// workspace of one forward / dgrad call: [V: P T Cin][M: P T Cout] (for dgrad ask with the channel counts swapped)
extern "C" size_t wesup_conv3x3_winograd_workspace_bytes(int B, int H, int W, int Cin, int Cout, int m) {
    if (!wino_shape_ok(B, H, W, Cin, Cout, m) || (Cin % 32)) return 0;
    const long T = wino_tiles(B, H, W, m);
    const int P = wino_positions(m);
    return align_up((size_t)P * T * Cin * sizeof(float), 256) + align_up((size_t)P * T * Cout * sizeof(float), 256);
}

// nbatch products C_b[M][N] = A_b[M][K] . B_b[N][K]^T of one shape in one launch (element strides between the entries)
extern "C" int wesup_gemm_nt_batched(const float* A, int lda, long strideA, const float* Bw, int ldb, long strideB, float* C,
                                     int ldc, long strideC, int nbatch, int M, int N, int K, void* stream) {
    if (!A || !Bw || !C || nbatch <= 0 || M <= 0 || N <= 0 || K <= 0 || (K % BK) || (lda % 4) || (ldb % 4) || (N % 4) ||
        (ldc % 4) || (strideA % 4) || (strideB % 4) || (strideC % 4) || (((uintptr_t)A | (uintptr_t)Bw | (uintptr_t)C) & 15))
        return WESUP_ERR_INVALID;
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128) * nbatch, t64 = (long)ceil_div(M, 64) * ceil_div(N, 64) * nbatch;
    if (t64 >= (1l << 31) / 2) return WESUP_ERR_INVALID;
    NtParams p = {};
    p.A = A; p.Bw = Bw; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldmask = ldc;
    p.nbatch = nbatch; p.batchA = strideA; p.batchB = strideB; p.batchC = strideC;
    hipStream_t st = (hipStream_t)stream;
    if (N > 64 && t128 >= 384) {
        // 128 x 64 tiles three to a CU instead of 128 x 128 two to a CU (round 5): the 36 products of conv4_x at 4 x 480^2
        // (900 x 512 x 512 each) are 1152 tiles of the second kind -- 4.5 per CU: half the CUs run a fifth block alone -- and
        // 2304 = 9 x 256 of the first; alone 186 -> 177 us (900 x 256 x 512: 107 -> 94), and no slower where the grid is many
        // rounds deep (36 x 8192 x 512 x 512: 1179 -> 1168 us, 132 TF; tools/ntb_micro.py).  The transposed shape, 64 x 128, where
        // it pads the problem less (M = 900: 960 rows of tiles instead of 1024): 177 -> 176 us alone, 7.92 -> 7.87 ms in the
        // step (three alternating pairs).  WESUP_NTB_SHAPE=0: 128 x 128, 1: 128 x 64 always, 3: 64 x 128 always.
        static const int alt = [] { const char* e = getenv("WESUP_NTB_SHAPE"); return e ? atoi(e) : 2; }();
        if (alt == 3 || (alt == 2 && (long)ceil_div(M, 64) * 64 * ceil_div(N, 128) * 128 < (long)ceil_div(M, 128) * 128 * ceil_div(N, 64) * 64))
            return launch_nt<4, 64, 128, 1, 2, 3, 3, false>(p, st);
        if (alt) return launch_nt<4, 128, 64, 2, 1, 3, 3, false>(p, st);
        return launch_nt<4, 128, 128, 2, 2, 3, 2, false>(p, st);
    }
    if (nt_three_stages(t64)) return launch_nt<4, 64, 64, 1, 1, 3, 3, false>(p, st);
    return launch_nt<4, 64, 64, 1, 1, 3, 2, false>(p, st);
}

// The same with a bias per product (element stride between the bias vectors; bias may be NULL): the 1x1 side convs of the
// layers that share a resolution (models/wesup.py:208-209,253) as ONE launch -- three products of 14 400 (3 600) pixels each
// fill 226 (58) of the 512 block slots one at a time.
extern "C" int wesup_gemm_nt_batched_bias(const float* A, int lda, long strideA, const float* Bw, int ldb, long strideB,
                                          const float* bias, long strideBias, float* C, int ldc, long strideC, int nbatch,
                                          int M, int N, int K, void* stream) {
    if (!A || !Bw || !C || nbatch <= 0 || M <= 0 || N <= 0 || K <= 0 || (K % BK) || (lda % 4) || (ldb % 4) || (N % 4) ||
        (ldc % 4) || (strideA % 4) || (strideB % 4) || (strideC % 4) || (strideBias % 4) ||
        (((uintptr_t)A | (uintptr_t)Bw | (uintptr_t)C | (uintptr_t)bias) & 15))
        return WESUP_ERR_INVALID;
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128) * nbatch, t64 = (long)ceil_div(M, 64) * ceil_div(N, 64) * nbatch;
    if (t64 >= (1l << 31) / 2) return WESUP_ERR_INVALID;
    NtParams p = {};
    p.A = A; p.Bw = Bw; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldmask = ldc;
    p.nbatch = nbatch; p.batchA = strideA; p.batchB = strideB; p.batchC = strideC; p.batchBias = strideBias;
    hipStream_t st = (hipStream_t)stream;
    if (N > 64 && t128 >= 384) return launch_nt<4, 128, 128, 2, 2, 3, 2, false>(p, st);
    if (nt_three_stages(t64)) return launch_nt<4, 64, 64, 1, 1, 3, 3, false>(p, st);
    return launch_nt<4, 64, 64, 1, 1, 3, 2, false>(p, st);
}

// in (B,H,W,Cin) --Winograd conv with u [P][Cout][Cin]--> out (B,H,W,Cout) with the conv epilogue.
// v_keep (optional): the transformed input is written there instead of the workspace (P T Cin floats).
static int wino_conv(const float* in, const float* u, const float* bias, const float* mask, float* out, float* out_relu,
                     float* out_pool, int pool_relu, float* v_keep, int B, int H, int W, int Cin, int Cout, int relu_in,
                     int accum, int m, void* ws, size_t ws_bytes, void* st, const float* unpool_src = nullptr,
                     float* unpool_dst = nullptr, int Hu = 0, int Wu = 0) {
    if (!in || !u || (!out && !unpool_src) || !ws || !wino_shape_ok(B, H, W, Cin, Cout, m) || (Cin % 32) ||
        (((uintptr_t)u | (uintptr_t)v_keep | (uintptr_t)ws) & 15))
        return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_conv3x3_winograd_workspace_bytes(B, H, W, Cin, Cout, m)) return WESUP_ERR_WORKSPACE;
    const long T = wino_tiles(B, H, W, m);
    const int P = wino_positions(m);
    float* V = v_keep ? v_keep : (float*)ws;
    float* Mt = (float*)((char*)ws + align_up((size_t)P * T * Cin * sizeof(float), 256));
    int rc = wesup_winograd_input_transform(in, V, 0, B, H, W, Cin, relu_in, m, st);
    if (rc) return rc;
    // short products (64 / 128 channels): products + output transform in one kernel, no transformed output in between
    const int fused = out_relu ? 0 : wino_fused_route(Cin, Cout, m, T);
    if (fused == 2 || (fused == 1 && !mask && !accum && !unpool_src))
        return wesup_winograd_gemm_output_transform(V, 0, u, bias, mask, out, out_pool, pool_relu, unpool_src, unpool_dst, Hu, Wu,
                                                    B, H, W, Cin, Cout, accum, st);
    rc = wesup_gemm_nt_batched(V, Cin, T * Cin, u, Cin, (long)Cout * Cin, Mt, Cout, T * Cout, P, (int)T, Cout, Cin, st);
    if (rc) return rc;
    if (unpool_src)
        return wesup_winograd_output_transform_unpool(Mt, 0, bias, mask, unpool_src, unpool_dst, B, H, W, Hu, Wu, Cout, m, st);
    return wesup_winograd_output_transform(Mt, 0, bias, mask, out, out_relu, out_pool, pool_relu, B, H, W, Cout, accum, m, st);
}

extern "C" int wesup_conv3x3_fwd_winograd(const float* x, const float* u_fwd, const float* bias, float* y, float* y_relu,
                                          float* y_pool, int pool_relu, float* v_keep, int B, int H, int W, int Cin,
                                          int Cout, int relu_in, int m, void* ws, size_t ws_bytes, void* stream) {
    return wino_conv(x, u_fwd, bias, nullptr, y, y_relu, y_pool, pool_relu, v_keep, B, H, W, Cin, Cout, relu_in, 0, m, ws,
                     ws_bytes, stream);
}

// dx = conv_transpose(dy) through the same pipeline: input dy (Cout channels), filter u_dgrad [P][Cin][Cout]
extern "C" int wesup_conv3x3_dgrad_winograd(const float* dy, const float* u_dgrad, const float* mask_src, float* dx, int B,
                                            int H, int W, int Cin, int Cout, int accumulate, int m, void* ws, size_t ws_bytes,
                                            void* stream) {
    return wino_conv(dy, u_dgrad, nullptr, mask_src, dx, nullptr, nullptr, 0, nullptr, B, H, W, Cout, Cin, 0, accumulate, m,
                     ws, ws_bytes, stream);
}

// wesup_conv3x3_dgrad_winograd (accumulate form) / _unpool for a destination whose old content is the side-branch gradient of
// a native-resolution layer with the side conv commuted behind the superpixel mean: that gradient is a gather
// (side[b][new_row[b][pixel]] / area), taken in the epilogue -- dx is written, never read, and nobody materialises the
// gather.  m = 4, product shapes of the fused kernel only (wesup_winograd_fused_supported(Cout, Cin, 4) == 2).
extern "C" int wesup_conv3x3_dgrad_winograd_gather(const float* dy, const float* u_dgrad, const float* mask_src,
                                                   const float* unpool_src, float* dx, const float* side,
                                                   const int32_t* new_row, const int32_t* area_new, int Kmax, int B, int H, int W,
                                                   int Hu, int Wu, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !u_dgrad || !dx || !ws || !wino_shape_ok(B, H, W, Cout, Cin, 4) || wino_fused_supported(Cout, Cin, 4) != 2 ||
        (((uintptr_t)u_dgrad | (uintptr_t)ws) & 15))
        return WESUP_ERR_INVALID;
    if (unpool_src && (Hu / 2 != H || Wu / 2 != W)) return WESUP_ERR_INVALID;
    if (ws_bytes < wesup_conv3x3_winograd_workspace_bytes(B, H, W, Cout, Cin, 4)) return WESUP_ERR_WORKSPACE;
    float* V = (float*)ws;
    const int rc = wesup_winograd_input_transform(dy, V, 0, B, H, W, Cout, 0, 4, stream);
    if (rc) return rc;
    return wesup_winograd_gemm_output_transform_gather(V, 0, u_dgrad, mask_src, dx, unpool_src, Hu, Wu, side, new_row, area_new,
                                                       Kmax, B, H, W, Cout, Cin, stream);
}

// The input gradient of a layer that follows a max-pool, taken straight through the pooling's backward (m = 4): nothing is
// written at pooled resolution; every value is added to unpool_dst (B,Hu,Wu,Cin) -- the gradient w.r.t. the pre-pool
// activations unpool_src, which already holds the side-branch gradient -- at the first positive maximum of its window.
extern "C" int wesup_conv3x3_dgrad_winograd_unpool(const float* dy, const float* u_dgrad, const float* unpool_src,
                                                   float* unpool_dst, int B, int H, int W, int Hu, int Wu, int Cin, int Cout,
                                                   int m, void* ws, size_t ws_bytes, void* stream) {
    if (!unpool_src || !unpool_dst) return WESUP_ERR_INVALID;
    return wino_conv(dy, u_dgrad, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, B, H, W, Cout, Cin, 0, 0, m, ws,
                     ws_bytes, stream, unpool_src, unpool_dst, Hu, Wu);
}
