// Winograd F(4x4,3x3): the batched products and the output transform in ONE kernel, for the layers whose products are
// short (K = 64 or 128 input channels of the product).  DESIGN.md 3.1.1.
//
// The unfused pipeline writes the transformed output Mt [36][tiles][N] (2.25x the activation) from the batched GEMM and
// reads it back in the output transform; at 64 / 128 channels those products are HBM-bound (conv1_2: 531 MB in and 531 MB
// out for 17 GFLOP), so the two passes over Mt are most of the layer's time.  Here a block owns 32 tiles x 64 output
// channels, walks the 36 positions itself -- per position one small product M_p[32][64] = V_p[32][K] . U_p[64][K]^T on
// the fp32 MFMA (v_mfma_f32_16x16x4_f32: a wave holds 16 tiles x 32 channels) -- and folds A^T M A into 16 output
// accumulators per (tile, channel) as it goes (separably: Z[jj] += A^T[jj][nu] M over a row of positions, then
// Y[i][jj] += A^T[i][xi] Z[jj]).  Mt never exists; V is read once, y written once, and the filter planes (16 / 32 KB per
// position and block) come from L2.  Operand staging is the LDS-DMA of gemm.hip (buffer form, XOR-swizzled 16-byte
// chunks, double-buffered over the positions).  The epilogue is the output transform's: bias, ReLU mask of the layer
// below, accumulation, the 2x2 max-pool behind the layer, or the max-pool backward (wesup_conv3x3_dgrad_winograd_unpool).
#include "winograd.hpp"
#include "ldsdma.hpp"
#include <cstdlib>
#include <cstdio>
#include <type_traits>

struct FusedParams {
    const float* V;          // [36][plane_v] transformed input, rows of K floats
    long plane_v;            // elements between two position planes of V
    const float* U;          // [36][N][K]
    const float* bias;       // [N] or NULL
    const float* mask;       // (B,H,W,N) or NULL: result kept where mask > 0
    float* y;                // (B,H,W,N), or NULL in unpool mode
    float* y_pool;           // (B,H/2,W/2,N) or NULL
    int pool_relu, accum;
    const float* up_src;     // unpool mode: pre-pool activations (B,Hu,Wu,N)
    float* up_dst;           // ... and the gradient they receive (accumulated into)
    int Hu, Wu;
    int nt, pad_;            // nt: y / the unpooled gradient written with streaming stores (common.hpp: wino_nt_stores)
    WinoGather gat;          // side-branch gradient gathered per pixel in place of reading y / up_dst (src NULL: off)
    const unsigned char* mask_bits;     // the mask as sign bits [B][H][W][N/4] (wesup_winograd_input_transform_bits) instead of mask
    unsigned short* pool_code;          // forward: the max-pool's decisions out, [B][H/2][W/2][N/4] (winograd.hpp)
    const unsigned short* up_code;      // unpool mode: ... and in, instead of reading up_src
    int H, W, Th, Tw;
    long T;                  // tiles
    int N;
    int n_fastest;           // block order: 1 = the channel blocks of one tile set run together on one XCD (see the kernel)
    int tile_blocks, n_blocks;
    FastDiv dTw, dTh, dNb;
};

// rows of A^T for the points 0, +-3/4, +-3/2, inf (oracle/winograd_oracle.py _AT[4])
__device__ __constant__ float c_wino4_at[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                                                  {0.f, 0.75f, -0.75f, 1.5f, -1.5f, 0.f},
                                                  {0.f, 0.5625f, 0.5625f, 2.25f, 2.25f, 0.f},
                                                  {0.f, 27.f / 64.f, -27.f / 64.f, 27.f / 8.f, -27.f / 8.f, 1.f}};
template <int JJ, int NU>
struct WinoAT {      // the same entries at compile time (zero weights drop out of the unrolled accumulation)
    static constexpr float v = JJ == 0 ? (NU < 5 ? 1.f : 0.f)
                             : NU == 0 ? 0.f
                             : NU == 5 ? (JJ == 3 ? 1.f : 0.f)
                             : JJ == 1 ? (NU == 1 ? 0.75f : NU == 2 ? -0.75f : NU == 3 ? 1.5f : -1.5f)
                             : JJ == 2 ? (NU <= 2 ? 0.5625f : 2.25f)
                             : (NU == 1 ? 27.f / 64.f : NU == 2 ? -27.f / 64.f : NU == 3 ? 27.f / 8.f : -27.f / 8.f);
};

__device__ __forceinline__ f32x4 fma4(float s, f32x4 a, f32x4 b) {
    f32x4 r;
    r[0] = fmaf(s, a[0], b[0]); r[1] = fmaf(s, a[1], b[1]); r[2] = fmaf(s, a[2], b[2]); r[3] = fmaf(s, a[3], b[3]);
    return r;
}

// WESUP_FUSED_TRACE (tools/probes/fused_phases.hip includes this file with it defined; never set for the library): every block
// leaves {start, end of the products, end} in 100 MHz ticks (s_memrealtime: one clock for all XCDs) and its HW_REG_HW_ID / XCC_ID behind.
#ifdef WESUP_FUSED_TRACE
__device__ unsigned long long* g_fused_trace = nullptr;
#define FUSED_TRACE(...) __VA_ARGS__
#else
#define FUSED_TRACE(...)
#endif

template <int N>
__device__ __forceinline__ void glds_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KC: K = 64 * KC channels of the product.  TW: 16-tile groups per block -- 2 (32 tiles, 4 waves, two blocks per CU) or 4 (64 tiles,
// 8 waves, one block per CU: a filter plane is staged once per 64 tiles instead of once per 32, a third less staging traffic
// per CU).  RING: stages in the LDS ring (RING - 1 in flight while one is read).
template <int KC, int TW, int RING>
__global__ __launch_bounds__(128 * TW, 2) void wino4_gemm_out_kernel(const FusedParams p) {
    constexpr int K = 64 * KC;
    constexpr int S = 36 * KC;                               // stages: (position, 64-channel chunk)
    constexpr int NW = 2 * TW, NT = 64 * NW, TILES = 16 * TW;
    constexpr int STG = (TILES + 64) * 64;                   // floats per stage: A [TILES tiles][64], B [64 channels][64]
    constexpr int AQ = 2, BQ = 16 / NW;                      // DMA instructions (1 KiB each) per wave and stage
    constexpr int INFL = (RING - 2) * (AQ + BQ);             // instructions of the stages behind the one awaited
    constexpr int ES = 8 * 64 + 4;                           // epilogue image: floats per tile (2 rows x 4 columns x 64 channels, padded)
    extern __shared__ __attribute__((aligned(16))) float smem[];      // ring of RING stages; the epilogue image reuses it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    FUSED_TRACE(const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();)
    // Block order.  The channel blocks of one tile set read the same rows of V, the tile sets of one channel block the
    // same filter planes.  The grid is walked XCD by XCD with the channel block fastest: the N / 64 readers of a tile set sit
    // behind one L2 at the same time and V comes from HBM once (tile-set fastest, the hardware order, re-read it N / 64
    // times: fetch past L2 per launch 293 -> 215 MiB at K = 128, 296 -> 159 MiB at K = 256, although the filter planes of
    // all channel blocks -- up to 9.4 MB -- then share that L2; the step time is the same within noise).
    // WESUP_WINO_FUSED_NFAST_KB: filter sets above this size keep the tile set fastest (default: none do).
    int tb, nb;
    if (p.n_fastest) {
        const int L = xcd_remap(blockIdx.x, p.tile_blocks * p.n_blocks);
        tb = fast_div(L, p.dNb);
        nb = L - tb * p.n_blocks;
    } else {
        nb = blockIdx.x / p.tile_blocks;
        tb = blockIdx.x - nb * p.tile_blocks;
    }
    const int t0 = tb * TILES, n0 = nb * 64;
    const int wa = wave % TW, wb = wave / TW;                // this wave: tiles 16 wa .. +15, channels 32 wb .. +31
    const int l15 = lane & 15, kq = lane >> 4;

    // ---- staging roles.  A wave-instruction fills 1 KiB = 4 rows of 64 floats; lane i supplies chunk position i & 15 of
    // row i >> 4 and fetches the logical chunk position ^ (row & 15) (the same involution on the read side).  A stage is
    // AQ + BQ DMA instructions per wave; the ring is at least three stages deep (a position's 32 MFMAs last ~0.5 us, less
    // than the latency of a DMA under load: with one stage in flight the kernel waited on every position).
    const int rows_valid = (int)min((long)TILES, p.T - t0);
    unsigned a_vo[AQ], b_vo[BQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const int row = 4 * (AQ * wave + q) + (lane >> 4);
        a_vo[q] = (unsigned)((row * K + 4 * ((lane & 15) ^ (row & 15))) * 4);      // rows >= rows_valid fall behind num_records
    }
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        const int row = 4 * (BQ * wave + q) + (lane >> 4);
        b_vo[q] = (unsigned)((row * K + 4 * ((lane & 15) ^ (row & 15))) * 4);
    }
    auto stage = [&](int s, int buf) {
        const int pos = s / KC, c = s - pos * KC;
        const i32x4 srdA = make_srd(p.V + (long)pos * p.plane_v + (long)t0 * K, (unsigned)rows_valid * (unsigned)K * 4u);
        const i32x4 srdB = make_srd(p.U + ((long)pos * p.N + n0) * K, 64u * (unsigned)K * 4u);
        // wave-uniform LDS byte addresses (the DMA takes them through M0): this wave's AQ resp. BQ KiB of the stage
        const unsigned adst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STG + wave * AQ * 256));
        const unsigned bdst = __builtin_amdgcn_readfirstlane(lds_addr(smem + buf * STG + TILES * 64 + wave * BQ * 256));
#pragma unroll
        for (int q = 0; q < AQ; ++q) bglds16(a_vo[q], srdA, (unsigned)(c * 256), adst + q * 1024);
#pragma unroll
        for (int q = 0; q < BQ; ++q) bglds16(b_vo[q], srdB, (unsigned)(c * 256), bdst + q * 1024);
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 Y[4][4][2];        // [output row i][output column jj][MFMA block j]: component r = tile 4 kq + r of the wave
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) Y[i][jj][0] = Y[i][jj][1] = zero4;

#pragma unroll
    for (int i = 0; i < RING - 1; ++i) stage(i, i);
    glds_wait_n<INFL>();        // stage 0 has landed (stages 1 .. RING - 2 in flight)
    __syncthreads();
    int cur = 0, s = 0;
    const int arow = (16 * wa + l15) * 64, brow = TILES * 64 + (32 * wb + l15) * 64;
    // Fragment double buffering ACROSS the stages: the reads of group g + 1 fly under the 8 MFMAs of group g, and the reads of the
    // next stage's first group under those of this stage's last -- the end-of-stage barrier sits in front of the last group's MFMAs
    // (every read of buf[cur] has landed by then).  A wave that has its SIMD's matrix pipe to itself -- its neighbour block in the
    // epilogue: half of the time at 4 x 480 x 480, nearly always once a grid has many rounds (tools/probes/fused_phases.hip) -- is
    // otherwise exposed to an LDS round trip per group, ~500 of the 1024 cycles of a stage.  (Left alone the compiler sinks the
    // reads of a group behind the MFMAs of the one before -- it reuses the fragment registers --: the sched_barriers pin them.)
    float4 fa[2], fb0[2], fb1[2];
    auto load_frag = [&](int buf, int g, int sl) {        // lane (row, kq) holds k = 16 g + 4 kq + t, t = step within the group
        const int ch = ((4 * g + kq) ^ l15) << 2;
        const float* as = smem + buf * STG + arow;
        const float* bs = smem + buf * STG + brow;
        fa[sl] = ld4(as + ch);
        fb0[sl] = ld4(bs + ch);
        fb1[sl] = ld4(bs + 16 * 64 + ch);
    };
    load_frag(0, 0, 0);
    for (int xi = 0; xi < 6; ++xi) {
        f32x4 Z[4][2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) Z[jj][0] = Z[jj][1] = zero4;
        auto position = [&](auto NU) {
            constexpr int nu = decltype(NU)::value;
            f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
            for (int c = 0; c < KC; ++c, ++s) {
                const int nxt = cur == 0 ? RING - 1 : cur - 1;      // (cur + RING - 1) % RING: everybody left that buffer before the last barrier
                if (s + RING - 1 < S) stage(s + RING - 1, nxt);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int sl = g & 1;
                    if (g + 1 < 4) {
                        load_frag(cur, g + 1, sl ^ 1);
                    } else {
                        if (s + RING - 1 < S) glds_wait_n<INFL>();      // stage s + 1 has landed, the stages behind it may still fly
                        else glds_wait();
                        __syncthreads();                // ... for everybody; every wave has read all of buf[cur]
                        cur = cur == RING - 1 ? 0 : cur + 1;
                        if (s + 1 < S) load_frag(cur, 0, sl ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].x, fb0[sl].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].x, fb1[sl].x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].y, fb0[sl].y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].y, fb1[sl].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].z, fb0[sl].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].z, fb1[sl].z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].w, fb0[sl].w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl].w, fb1[sl].w, acc1, 0, 0, 0);
                }
            }
            // (.) A over the row of positions: Z[jj] += A^T[jj][nu] M_(xi, nu)
            if constexpr (WinoAT<0, nu>::v != 0.f) { Z[0][0] = fma4(WinoAT<0, nu>::v, acc0, Z[0][0]); Z[0][1] = fma4(WinoAT<0, nu>::v, acc1, Z[0][1]); }
            if constexpr (WinoAT<1, nu>::v != 0.f) { Z[1][0] = fma4(WinoAT<1, nu>::v, acc0, Z[1][0]); Z[1][1] = fma4(WinoAT<1, nu>::v, acc1, Z[1][1]); }
            if constexpr (WinoAT<2, nu>::v != 0.f) { Z[2][0] = fma4(WinoAT<2, nu>::v, acc0, Z[2][0]); Z[2][1] = fma4(WinoAT<2, nu>::v, acc1, Z[2][1]); }
            if constexpr (WinoAT<3, nu>::v != 0.f) { Z[3][0] = fma4(WinoAT<3, nu>::v, acc0, Z[3][0]); Z[3][1] = fma4(WinoAT<3, nu>::v, acc1, Z[3][1]); }
        };
        position(std::integral_constant<int, 0>{});
        position(std::integral_constant<int, 1>{});
        position(std::integral_constant<int, 2>{});
        position(std::integral_constant<int, 3>{});
        position(std::integral_constant<int, 4>{});
        position(std::integral_constant<int, 5>{});
        // A^T (.) down the rows of positions: Y[i][jj] += A^T[i][xi] Z[jj]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float wi = c_wino4_at[i][xi];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                Y[i][jj][0] = fma4(wi, Z[jj][0], Y[i][jj][0]);
                Y[i][jj][1] = fma4(wi, Z[jj][1], Y[i][jj][1]);
            }
        }
    }

    FUSED_TRACE(const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime(); unsigned long long trE[4] = {0, 0, 0, 0};)
    // (the thread's indices again, from a value the compiler cannot tie to the ones the products used: kept alive across the main
    // loop for the epilogue's sake they cost the K = 256 instantiation a spilled register)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6, wa_e = wave_e % TW, wb_e = wave_e / TW, l15_e = lane_e & 15, kq_e = lane_e >> 4;
    // ---- epilogue through LDS, two output rows of every tile per pass.  D of a 16x16 block: lane_e (l15_e, kq_e), component r =
    // row 4 kq_e + r, column l15_e -- this lane_e owns tiles 16 wa_e + 4 kq_e + r at channels 32 wb_e + 16 j + l15_e.  The image
    // E[tile][row & 1][column][channel] (tile stride padded by 4 floats: the four kq_e groups land 16 banks apart) is read
    // back as 16-byte channel quads, a thread per (pixel, quad), so that bias / mask / accumulate / pooling / unpooling and
    // the stores move 16 B per lane_e on 256-byte channel segments -- the output transform's own epilogue.
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    const WinoUnpool up = {p.up_src, p.up_dst, p.Hu, p.Wu};
    const int q4 = tid_e & 15, slot = (tid_e >> 4) & 7, tsel = tid_e >> 7;          // reader: channel quad, (row & 1, column), tile index mod NW / 2
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = ld4(p.bias + n0 + 4 * q4);
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) {
        // (the main loop ended with a barrier: the ring is free)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        smem[(16 * wa_e + 4 * kq_e + r) * ES + (4 * i2 + jj) * 64 + 32 * wb_e + 16 * j + l15_e] = Y[2 * ip + i2][jj][j][r];
        __syncthreads();
        FUSED_TRACE(trE[2 * ip] = __builtin_amdgcn_s_memrealtime();)          // the image of this row pair is in LDS
        const int i = 2 * ip + (slot >> 2), jj = slot & 3;
        // The y forms that READ something per pixel (sign bits, the old gradient, a gathered side-gradient row): four pixels at a
        // time, every load of a level issued before the first use -- out-of-image pixels are clamped and predicated instead of
        // skipped, so that nothing divergent stands between the loads.
        const bool batched = !up.dst && !p.mask && !(p.gat.src && p.gat.area) && (p.mask_bits || p.accum || p.gat.src);
        // Every form below: a wave_e walks ITS tiles (TILES / NW of the block's), its lanes = (column of the tile row, channel quad).
        // The tile's decomposition, the row's pixel base and every 64-bit address part are wave_e-uniform -- scalar instructions --
        // and a lane_e adds one constant 32-bit offset.  (Round 5: beside a block that is in its products the epilogue's VECTOR
        // instructions wait for gaps in the neighbour's MFMA stream, ~40 cycles each -- tools/probes/coexec.hip, fused_phases.hip:
        // the epilogue lasted as long as the products --; with a thread per (pixel, quad) of sixteen different tiles every store
        // cost ~45 vector instructions of index arithmetic.)
        constexpr int TPW = TILES / NW;
        const int wv = __builtin_amdgcn_readfirstlane(wave_e);
        const int cc = lane_e >> 4;                                  // column of the tile row this lane_e handles
        const unsigned vq = (unsigned)(cc * p.N + 4 * q4);        // ... as an element offset from the row's first pixel
        // item j of a wave_e's pass: tile wv TPW + (j >> 1), row 2 ip + (j & 1) of it; wave_e-uniform
        struct Item { long pix0; int b, h; bool ok; };
        auto item = [&](int j) {
            Item it;
            long tile = (long)t0 + wv * TPW + (j >> 1);
            it.ok = tile < p.T;
            tile = it.ok ? tile : p.T - 1;
            const int bi = fast_div((int)tile, p.dTw);
            const int tj = (int)tile - bi * p.Tw;
            it.b = fast_div(bi, p.dTh);
            const int ti = bi - it.b * p.Th;
            it.h = 4 * ti + 2 * ip + (j & 1);
            it.ok = it.ok && it.h < p.H;
            it.h = min(it.h, p.H - 1);
            it.pix0 = ((long)it.b * p.H + it.h) * p.W + 4 * tj;        // the row's first pixel of the tile, batch-wide index
            return it;
        };
        // The y forms that READ something per pixel (sign bits, the old gradient, a gathered side-gradient row): four items at a
        // time, every load of a level issued before the first use -- rows and columns outside the image are clamped and
        // predicated instead of skipped, so that nothing divergent stands between the loads.
        if (batched) {
            const int wmax = p.W - 1;
#pragma unroll 1
            for (int j0 = 0; j0 < 2 * TPW; j0 += 4) {
                bool ok[4]; unsigned mb[4]; float4 old[4]; int row[4]; long off0[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Item it = item(j0 + k);
                    const int w0 = (int)(it.pix0 - ((long)it.b * p.H + it.h) * p.W);      // 4 tj
                    const int dw = min(cc, wmax - w0);                                      // column clamped into the image
                    ok[k] = it.ok && w0 + cc <= wmax;
                    off0[k] = it.pix0 * p.N + n0;
                    const unsigned vqc = (unsigned)(dw * p.N + 4 * q4);
                    mb[k] = p.mask_bits ? (p.mask_bits + it.pix0 * (p.N >> 2) + (n0 >> 2))[dw * (p.N >> 2) + q4] : 15u;
                    old[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    row[k] = 0;
                    if (p.gat.src) row[k] = (p.gat.row + it.pix0)[dw] + it.b * p.gat.Kmax;       // (pix = b * HW + the pixel of image b)
                    else if (p.accum) old[k] = ld4(p.y + off0[k] + vqc);
                }
                if (p.gat.src) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) old[k] = ld4(p.gat.src + (long)row[k] * p.N + n0 + 4 * q4);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int tl = wv * TPW + ((j0 + k) >> 1);
                    const float4 e = ld4(smem + tl * ES + (4 * ((j0 + k) & 1) + cc) * 64 + 4 * q4);
                    float4 v = make_float4(e.x + bv.x, e.y + bv.y, e.z + bv.z, e.w + bv.w);
                    v.x = (mb[k] & 1) ? v.x : 0.f; v.y = (mb[k] & 2) ? v.y : 0.f;
                    v.z = (mb[k] & 4) ? v.z : 0.f; v.w = (mb[k] & 8) ? v.w : 0.f;
                    v = make_float4(v.x + old[k].x, v.y + old[k].y, v.z + old[k].z, v.w + old[k].w);
                    if (ok[k]) st4s(p.y + off0[k] + vq, v, p.nt);
                }
            }
        }
        // The unpooling forms with the pooling's decisions as codes, two windows at a time in the same manner: codes (and the
        // four row indices of a gathered window) first, then the destination's old values at the chosen positions resp. the
        // gathered rows, then the stores.
        const bool ubatched = up.dst && p.up_code && !p.mask && !p.mask_bits && !(p.gat.src && p.gat.area);
        if (ubatched) {
            const long rs = (long)p.Wu * p.N;
            const int wmax = p.W - 1;
#pragma unroll 1
            for (int j0 = 0; j0 < 2 * TPW; j0 += 2) {
                long o00[2]; bool ok[2]; unsigned code[2]; int row[2][4]; float4 val[2][4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const Item it = item(j0 + k);
                    const int w0 = (int)(it.pix0 - ((long)it.b * p.H + it.h) * p.W);
                    const int dw = min(cc, wmax - w0);
                    ok[k] = it.ok && w0 + cc <= wmax;
                    code[k] = (p.up_code + it.pix0 * (p.N >> 2) + (n0 >> 2))[dw * (p.N >> 2) + q4];
                    const long pix00 = ((long)it.b * p.Hu + 2 * it.h) * p.Wu + 2 * w0;          // the row's first window, batch-wide index (uniform)
                    o00[k] = pix00 * p.N + n0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        row[k][q] = p.gat.src ? (p.gat.row + pix00 + (q >> 1) * p.Wu + (q & 1))[2 * dw] + it.b * p.gat.Kmax : 0;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const WinoPicks pk = wino_code_picks(code[k]);
                    const unsigned vqu = (unsigned)(2 * cc * p.N + 4 * q4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        val[k][q] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.gat.src) val[k][q] = ld4(p.gat.src + (long)row[k][q] * p.N + n0 + 4 * q4);
                        else if (ok[k] && (pk.kx == q || pk.ky == q || pk.kz == q || pk.kw == q))
                            val[k][q] = ld4(up.dst + o00[k] + (q >> 1) * rs + (q & 1) * p.N + vqu);
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int tl = wv * TPW + ((j0 + k) >> 1);
                    const float4 e = ld4(smem + tl * ES + (4 * ((j0 + k) & 1) + cc) * 64 + 4 * q4);
                    const float4 v = make_float4(e.x + bv.x, e.y + bv.y, e.z + bv.z, e.w + bv.w);
                    const WinoPicks pk = wino_code_picks(code[k]);
                    const unsigned vqu = (unsigned)(2 * cc * p.N + 4 * q4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool hit = pk.kx == q || pk.ky == q || pk.kz == q || pk.kw == q;
                        if (!ok[k] || (!p.gat.src && !hit)) continue;
                        float4 o = val[k][q];
                        o.x += pk.kx == q ? v.x : 0.f; o.y += pk.ky == q ? v.y : 0.f; o.z += pk.kz == q ? v.z : 0.f; o.w += pk.kw == q ? v.w : 0.f;
                        float* dptr = up.dst + o00[k] + (q >> 1) * rs + (q & 1) * p.N + vqu;
                        if (p.gat.src) st4s(dptr, o, p.nt); else st4(dptr, o);
                    }
                }
            }
        }
        // The plain form (forward: y = result + bias): eight rows of four pixels at a time -- their LDS reads in flight together,
        // one wait, then the adds and the stores (one row at a time it was a chain of read -> wait -> add -> store per row, each
        // link stretched by the neighbour block's MFMA stream)
        const bool plain = !batched && !ubatched && !p.mask && !up.dst && !p.gat.src && !p.accum;
        if (plain) {
#pragma unroll 1
            for (int j0 = 0; j0 < 2 * TPW; j0 += 8) {
                float4 e[8]; long off0[8]; bool ok[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const Item it = item(j0 + k);
                    const int w0 = (int)(it.pix0 - ((long)it.b * p.H + it.h) * p.W);
                    ok[k] = it.ok && w0 + cc < p.W;
                    off0[k] = it.pix0 * p.N + n0;
                    e[k] = ld4(smem + (wv * TPW + ((j0 + k) >> 1)) * ES + (4 * ((j0 + k) & 1) + cc) * 64 + 4 * q4);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (ok[k]) st4s(p.y + off0[k] + vq, make_float4(e[k].x + bv.x, e[k].y + bv.y, e[k].z + bv.z, e[k].w + bv.w), p.nt);
            }
        }
        // The general form (dense float masks, the unpooling forms without codes, gathered rows that still need their areas)
        if (!(batched || ubatched || plain)) {
            for (int k = 0; k < TPW; ++k) {
                const int tl = wv * TPW + k;
                const long tile = (long)t0 + tl;
                if (tile >= p.T) break;
                const int bi = fast_div((int)tile, p.dTw);
                const int tj = (int)tile - bi * p.Tw;
                const int b = fast_div(bi, p.dTh);
                const int ti = bi - b * p.Th;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int h = 4 * ti + 2 * ip + r, w = 4 * tj + cc;
                    if (h >= p.H) continue;                            // (wave_e-uniform)
                    const float4 e = ld4(smem + tl * ES + (4 * r + cc) * 64 + 4 * q4);
                    if (w >= p.W) continue;
                    float4 v = make_float4(e.x + bv.x, e.y + bv.y, e.z + bv.z, e.w + bv.w);
                    const long pix0 = ((long)b * p.H + h) * p.W + 4 * tj;                 // the row's first pixel (uniform)
                    const long off0 = pix0 * p.N + n0;
                    if (p.mask_bits) {
                        const unsigned mb = (p.mask_bits + pix0 * (p.N >> 2) + (n0 >> 2))[cc * (p.N >> 2) + q4];
                        v.x = (mb & 1) ? v.x : 0.f; v.y = (mb & 2) ? v.y : 0.f;
                        v.z = (mb & 4) ? v.z : 0.f; v.w = (mb & 8) ? v.w : 0.f;
                    } else if (p.mask) {
                        const float4 mk = ld4(p.mask + off0 + vq);
                        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                    }
                    if (up.dst) {            // input gradient at pooled resolution: straight through the max-pool backward
                        if (p.gat.src) wino_unpool_gather(up, p.up_code, p.gat, b, h, w, p.N, n0 + 4 * q4, v, p.nt);
                        else wino_unpool_add(up, p.up_code, b, h, w, p.N, n0 + 4 * q4, v);
                        continue;
                    }
                    if (p.gat.src) {
                        const float4 old = wino_gather(p.gat, b, (long)h * p.W + w, p.N, n0 + 4 * q4);
                        v = make_float4(v.x + old.x, v.y + old.y, v.z + old.z, v.w + old.w);
                    } else if (p.accum) {
                        const float4 old = ld4(p.y + off0 + vq);
                        v = make_float4(v.x + old.x, v.y + old.y, v.z + old.z, v.w + old.w);
                    }
                    st4s(p.y + off0 + vq, v, p.nt);
                }
            }
        }
        FUSED_TRACE(trE[2 * ip + 1] = __builtin_amdgcn_s_memrealtime();)      // ... and its stores are issued
        if (p.y_pool) {              // forward only (v = Y + bias): the two pooled rows 2 ti + ip of every tile
            // thread = (tile, window column k, quad): TILES x 2 x 16 = 4 NT items
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int item = it * NT + tid_e;
                const int q = item & 15, k = (item >> 4) & 1, tl = item >> 5;
                const long tile = (long)t0 + tl;
                if (tile >= p.T) continue;
                const int bi = fast_div((int)tile, p.dTw);
                const int tj = (int)tile - bi * p.Tw;
                const int b = fast_div(bi, p.dTh);
                const int ti = bi - b * p.Th;
                const int ph = 2 * ti + ip, pw = 2 * tj + k;
                if (ph >= Hp || pw >= Wp) continue;
                const float* e = smem + tl * ES + 4 * q;
                const float4 b4 = bv;      // (q == q4: NT is a multiple of 16 -- the bias quad this thread loaded once)
                const float4 e0 = ld4(e + (2 * k) * 64), e1 = ld4(e + (2 * k + 1) * 64), e2 = ld4(e + (4 + 2 * k) * 64), e3 = ld4(e + (5 + 2 * k) * 64);
                if (p.pool_code) {       // the window's decisions on the stored values y = e + bias, for the backward
                    auto plus = [&](float4 a) { return make_float4(a.x + b4.x, a.y + b4.y, a.z + b4.z, a.w + b4.w); };
                    p.pool_code[(((long)b * Hp + ph) * Wp + pw) * (p.N >> 2) + (n0 >> 2) + q] =
                        wino_picks_code(wino_picks(plus(e0), plus(e1), plus(e2), plus(e3)));
                }
                float4 m;
                m.x = fmaxf(fmaxf(e0.x, e1.x), fmaxf(e2.x, e3.x)) + b4.x;
                m.y = fmaxf(fmaxf(e0.y, e1.y), fmaxf(e2.y, e3.y)) + b4.y;
                m.z = fmaxf(fmaxf(e0.z, e1.z), fmaxf(e2.z, e3.z)) + b4.z;
                m.w = fmaxf(fmaxf(e0.w, e1.w), fmaxf(e2.w, e3.w)) + b4.w;
                st4(p.y_pool + (((long)b * Hp + ph) * Wp + pw) * p.N + n0 + 4 * q, p.pool_relu ? relu4(m) : m);
            }
        }
        if (ip == 0) __syncthreads();      // everybody has read the image before the second row pair overwrites it
    }
    FUSED_TRACE(if (g_fused_trace && tid == 0) {
        unsigned long long* t = g_fused_trace + 9 * (long)blockIdx.x;
        t[0] = tr0; t[1] = tr1; t[2] = __builtin_amdgcn_s_memrealtime();
        t[5] = trE[0]; t[6] = trE[1]; t[7] = trE[2]; t[8] = trE[3];
        t[3] = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        t[4] = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    })
}

// WESUP_WINO_FUSED: 0 = never, 1 = forward epilogues only (bias, pooled output), 2 = every epilogue (default)
static int fused_env_level() {
    static const int v = [] { const char* e = getenv("WESUP_WINO_FUSED"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }();
    return v;
}
// internal (winograd.hpp): 0 = the fused products + output transform do not cover this product shape; 1 = they do for the
// forward epilogue; 2 = for every epilogue
static int fused_max_k() {      // WESUP_WINO_FUSED_MAXK (experiment): the longest product that takes this route (default 256)
    static const int v = [] { const char* e = getenv("WESUP_WINO_FUSED_MAXK"); const int k = e ? atoi(e) : 256; return (k == 64 || k == 128 || k == 256) ? k : 256; }();
    return v;
}
int wino_fused_supported(int K, int N, int m) {
    return (m == 4 && (K == 64 || K == 128 || K == 256) && K <= fused_max_k() && N >= 64 && (N % 64) == 0) ? fused_env_level() : 0;
}

extern "C" int wesup_winograd_fused_supported(int K, int N, int m) { return wino_fused_supported(K, N, m); }
// ... and whether a problem of `tiles` tiles should take it: the one-kernel route walks 36 K/64 stages per block whatever the
// grid, so below ~200 blocks (32 tiles x 64 channels each: conv3_x / conv4_1 at batch 1, 480 x 480) the batched GEMM + output
// transform -- 36 x more blocks -- is faster (tools/fused_micro.py --unfused: 103 -> 72 us at 116 blocks, 100 -> 39 us at 64;
// 106 vs 113 us at 228 blocks).  0 = two kernels; else wesup_winograd_fused_supported's answer.
static int g_fused_min_blocks = 200;
int wino_fused_route(int K, int N, int m, long tiles) {
    const int cap = wino_fused_supported(K, N, m);
    if (!cap || tiles <= 0) return cap;
    return ((tiles + 31) / 32) * (N / 64) >= g_fused_min_blocks ? cap : 0;
}
// tuning knob of that rule (process-wide; returns the previous value): 0 = every supported shape takes the one-kernel route
extern "C" int wesup_winograd_set_fused_min_blocks(int blocks) {
    const int was = g_fused_min_blocks;
    if (blocks >= 0) g_fused_min_blocks = blocks;
    return was;
}
extern "C" int wesup_winograd_fused_route(int K, int N, int m, long tiles) { return wino_fused_route(K, N, m, tiles); }

// V [36][tiles][K] (plane stride plane_elems, 0 = tiles * K) x U [36][N][K] -> y = A^T (V_p . U_p^T) A + the output
// transform's epilogue (wesup_winograd_output_transform / _unpool), without the transformed output in between.
struct FusedBits {       // the compact forms of mask_src / unpool_src (in) and of the pooling decisions (out); all optional
    const unsigned char* mask_bits;
    unsigned short* pool_code;
    const unsigned short* up_code;
};
// Block shape of the one-kernel route: TW = 2 tile groups (32 tiles, 4 waves, two blocks per CU), a ring of 3 stages.
// Round 4 measured the alternatives (profiles/r04_fused_shapes.txt, tools/fused_micro.py): 64-tile blocks of 8 waves (TW = 4, one
// block per CU) are 3 ... 5 % faster alone where the grid still fills the chip (conv1_2, conv2_2, conv3_2 at 480 x 480) and 50 %
// slower where it does not; as a rule ("64-tile blocks from 200 blocks up") neutral in the 480 x 480 step (9.10 vs 9.10 ms) and
// at 1024 x 1024, 2.5 % slower at 800 x 800 (21.7 vs 21.1 ms) -- not kept.  Rings of 4 and 5 stages (possible at one block per
// CU) change nothing: the kernel does not wait for the latency of its staging DMA.
struct FusedShape { int tw, ring; };
static FusedShape fused_shape(long, int) { return FusedShape{2, 3}; }
template <int KC, int TW, int RING>
static int fused_go(const FusedParams& p, dim3 grid, void* stream) {
    constexpr size_t ring = (size_t)RING * (16 * TW + 64) * 64 * sizeof(float);
    constexpr size_t image = (size_t)16 * TW * (8 * 64 + 4) * sizeof(float);      // the epilogue's [tile][2 rows x 4 columns x 64] image
    constexpr size_t lds = ring > image ? ring : image;      // 72 KiB (TW 2, RING 3) ... 160 KiB: above the default dynamic limit
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_gemm_out_kernel<KC, TW, RING>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return WESUP_ERR_LAUNCH;
    WESUP_LAUNCH((wino4_gemm_out_kernel<KC, TW, RING>), grid, dim3(128 * TW), lds, (hipStream_t)stream, p);
    return WESUP_OK;
}
static int fused_launch(const float* V, long plane_elems, const float* U, const float* bias, const float* mask_src, float* y,
                        float* y_pool, int pool_relu, const float* unpool_src, float* unpool_dst, int Hu, int Wu, int B, int H,
                        int W, int K, int N, int accumulate, const WinoGather& gat, const FusedBits& bits, void* stream) {
    if (bits.pool_code && !y_pool) return WESUP_ERR_INVALID;
    const bool unpool = unpool_src || bits.up_code;
    if (!V || !U || (!y && !unpool) || !wino_shape_ok(B, H, W, K, N, 4) || !(K == 64 || K == 128 || K == 256) || N < 64 || (N % 64) ||
        (((uintptr_t)V | (uintptr_t)U) & 15) || (plane_elems % 4))
        return WESUP_ERR_INVALID;
    const long T = wino_tiles(B, H, W, 4);
    if (plane_elems > 0 && plane_elems < T * K) return WESUP_ERR_INVALID;
    if (unpool && (!unpool_dst || y_pool || accumulate || Hu / 2 != H || Wu / 2 != W)) return WESUP_ERR_INVALID;
    if (T > (1l << 31) / 64 || (long)32 * K * 4 >= (1l << 31)) return WESUP_ERR_INVALID;
    FusedParams p = {};
    p.V = V; p.plane_v = plane_elems > 0 ? plane_elems : T * K; p.U = U; p.bias = bias; p.mask = mask_src; p.y = y;
    p.y_pool = y_pool; p.pool_relu = pool_relu; p.accum = accumulate; p.up_src = unpool_src; p.up_dst = unpool_dst;
    p.gat = gat;
    p.mask_bits = bits.mask_bits; p.pool_code = bits.pool_code; p.up_code = bits.up_code;
    p.nt = wino_nt_stores(4.0 * B * (unpool ? 4.0 : 1.0) * H * W * N);
    p.Hu = Hu; p.Wu = Wu; p.H = H; p.W = W; p.Th = (H + 3) / 4; p.Tw = (W + 3) / 4; p.T = T; p.N = N;
    p.dTw = make_fastdiv(p.Tw); p.dTh = make_fastdiv(p.Th);
    const FusedShape fs = fused_shape(T, N);
    p.tile_blocks = (int)((T + 16 * fs.tw - 1) / (16 * fs.tw)); p.n_blocks = N / 64; p.dNb = make_fastdiv(p.n_blocks);
    static const long u_limit = [] { const char* e = getenv("WESUP_WINO_FUSED_NFAST_KB"); return (e ? atol(e) : 65536l) * 1024l; }();
    p.n_fastest = (p.n_blocks > 1 && (long)36 * N * K * 4 <= u_limit) ? 1 : 0;
    if ((long)p.tile_blocks * p.n_blocks >= (1l << 24)) return WESUP_ERR_INVALID;
    const dim3 grid((unsigned)(p.tile_blocks * p.n_blocks));
    int rc = WESUP_ERR_INVALID;
#define FUSED_CASE(TW_, RING_)                                                                        \
    if (fs.tw == TW_ && fs.ring == RING_)                                                             \
        rc = K == 64 ? fused_go<1, TW_, RING_>(p, grid, stream)                                       \
           : K == 128 ? fused_go<2, TW_, RING_>(p, grid, stream) : fused_go<4, TW_, RING_>(p, grid, stream);
    FUSED_CASE(2, 3)
#undef FUSED_CASE
    if (rc != WESUP_OK) return rc;
    WESUP_CHECK_LAUNCH();
    return WESUP_OK;
}

extern "C" int wesup_winograd_gemm_output_transform(const float* V, long plane_elems, const float* U, const float* bias,
                                                    const float* mask_src, float* y, float* y_pool, int pool_relu,
                                                    const float* unpool_src, float* unpool_dst, int Hu, int Wu, int B, int H,
                                                    int W, int K, int N, int accumulate, void* stream) {
    return fused_launch(V, plane_elems, U, bias, mask_src, y, y_pool, pool_relu, unpool_src, unpool_dst, Hu, Wu, B, H, W, K, N,
                        accumulate, WinoGather{nullptr, nullptr, nullptr, 0, 0, 0}, FusedBits{nullptr, nullptr, nullptr}, stream);
}

// Every option of the one-kernel route in one entry.  Beyond wesup_winograd_gemm_output_transform[_gather]:
//   mask_bits   the ReLU mask as sign bits [B][H][W][N/4] (wesup_winograd_input_transform_bits) instead of mask_src;
//   pool_code   (with y_pool) OUT: the decisions of the 2x2 max-pool per window and channel, [B][H/2][W/2][N/4] uint16, 3 bits
//               per channel of the quad: 0 = maximum not positive, k + 1 = first maximum at window position k (row-major);
//   unpool_code the same codes IN, instead of reading unpool_src (which may then be NULL): [B][H][W][N/4] at the resolution
//               of this product's output (the pooled resolution).
// side != NULL selects the gather forms (y / unpool_dst written, never read).
extern "C" int wesup_winograd_gemm_output_transform_ex(const float* V, long plane_elems, const float* U, const float* bias,
                                                       const float* mask_src, const unsigned char* mask_bits, float* y,
                                                       float* y_pool, int pool_relu, unsigned short* pool_code,
                                                       const float* unpool_src, const unsigned short* unpool_code,
                                                       float* unpool_dst, int Hu, int Wu, const float* side,
                                                       const int32_t* new_row, const int32_t* area_new, int Kmax, int B, int H,
                                                       int W, int K, int N, int accumulate, void* stream) {
    const bool unpool = unpool_src || unpool_code;
    if (unpool && (!unpool_dst || Hu / 2 != H || Wu / 2 != W)) return WESUP_ERR_INVALID;
    WinoGather gat = {nullptr, nullptr, nullptr, 0, 0, 0};
    if (side) {
        if (!new_row || Kmax <= 0 || accumulate || bias || y_pool || (((uintptr_t)side) & 15)) return WESUP_ERR_INVALID;
        if (unpool && ((Hu & 1) || (Wu & 1))) return WESUP_ERR_INVALID;
        gat = WinoGather{side, new_row, area_new, Kmax, 0, unpool ? (long)Hu * Wu : (long)H * W};
    }
    return fused_launch(V, plane_elems, U, bias, mask_src, unpool ? nullptr : y, y_pool, pool_relu, unpool_src,
                        unpool ? unpool_dst : nullptr, Hu, Wu, B, H, W, K, N, accumulate, gat,
                        FusedBits{mask_bits, pool_code, unpool_code}, stream);
}

// The same with the old content of the destination replaced by a per-pixel gather (WinoGather, winograd.hpp): the destination
// is written, never read.  y form: y = gather(pixel) + masked result; unpool form (unpool_src given, Hu, Wu even): every
// position of a window = its gather, the first positive maximum gets the window's result on top.
extern "C" int wesup_winograd_gemm_output_transform_gather(const float* V, long plane_elems, const float* U, const float* mask_src,
                                                           float* y, const float* unpool_src, int Hu, int Wu, const float* side,
                                                           const int32_t* new_row, const int32_t* area_new, int Kmax, int B,
                                                           int H, int W, int K, int N, void* stream) {
    if (!side || !new_row || Kmax <= 0 || !y || (((uintptr_t)side) & 15)) return WESUP_ERR_INVALID;      // (area_new NULL: rows pre-scaled)
    if (unpool_src && ((Hu & 1) || (Wu & 1))) return WESUP_ERR_INVALID;
    const WinoGather gat = {side, new_row, area_new, Kmax, 0, unpool_src ? (long)Hu * Wu : (long)H * W};
    const FusedBits nobits = {nullptr, nullptr, nullptr};
    if (unpool_src)
        return fused_launch(V, plane_elems, U, nullptr, mask_src, nullptr, nullptr, 0, unpool_src, y, Hu, Wu, B, H, W, K, N, 0, gat,
                            nobits, stream);
    return fused_launch(V, plane_elems, U, nullptr, mask_src, y, nullptr, 0, nullptr, nullptr, 0, 0, B, H, W, K, N, 0, gat, nobits,
                        stream);
}
